"""ORACLE TOOLING: the list of golden-vector cases (shared by `oracle/make_golden.py`, which writes
them from the imported reference, and by the tests, which regenerate inputs/weights from the same
recipes).  Every case is fully described by seeds + sizes; no tensor data lives here.

model: E=num_model, H=num_head (head dim 64 like the reference's 512/8, except the `*_h32` / `*_h128` cases),
       FF=num_feedforward, enc/dec = layer counts, L=num_lines, seq_len = max_face_length (parallel)
       or label_seq_length (seq2seq).
"""

SMALL = dict(E=128, H=2, FF=256, enc=2, dec=2)
FULL = dict(E=512, H=8, FF=1024, enc=6, dec=6)
SMALL_H32 = dict(E=128, H=4, FF=256, enc=2, dec=2)     # num_model / num_head = 32-wide heads (transformer.py:131,191-192 take any)
SMALL_H128 = dict(E=128, H=1, FF=256, enc=2, dec=2)    # ... one 128-wide head


def _m(base, L, seq_len):
    d = dict(base)
    d.update(L=L, seq_len=seq_len)
    return d


def _mc(base, L, seq_len, **ctor):
    """... with constructor arguments no reference config sets (model.py:14-18: normalize_before, activation)."""
    d = _m(base, L, seq_len)
    d.update(ctor)
    return d


CASES = [
    # --- SurfaceFormer_Parallel -------------------------------------------------------------------
    dict(name="par_small_default", kind="parallel", model=_m(SMALL, 24, 9), recipe="default",
         wseed=0, n_edges=[20, 13], seeds=[1, 2]),
    dict(name="par_small_gain4", kind="parallel", model=_m(SMALL, 24, 9), recipe="gain4",
         wseed=0, n_edges=[20, 13], seeds=[1, 2]),
    # ragged batch incl. a 1-edge wireframe and a full one; exercises padding anchors (token 3)
    dict(name="par_small_ragged", kind="parallel", model=_m(SMALL, 40, 7), recipe="gain4",
         wseed=3, n_edges=[40, 1, 17, 33], seeds=[5, 6, 7, 8]),
    # all sequences hit a special token in one step -> early break + zero padding branch
    dict(name="par_small_earlybreak", kind="parallel", model=_m(SMALL, 16, 10), recipe="gain4",
         wseed=10, n_edges=[12, 9], seeds=[3, 4]),
    dict(name="par_small_break1", kind="parallel", model=_m(SMALL, 16, 10), recipe="bias05",
         wseed=0, n_edges=[12, 9], seeds=[3, 4]),
    # full-size model (weights regenerated from the seed), short wireframe
    dict(name="par_full_n40_gain4", kind="parallel", model=_m(FULL, 48, 12), recipe="gain4",
         wseed=0, n_edges=[40], seeds=[11]),
    dict(name="par_full_n40_default", kind="parallel", model=_m(FULL, 48, 12), recipe="default",
         wseed=0, n_edges=[40], seeds=[11]),
    # BASELINE config B itself: ours.yml sizes with num_lines=256, one 256-edge wireframe, 36 steps
    dict(name="par_full_B256_default", kind="parallel", model=_m(FULL, 256, 37), recipe="default",
         wseed=0, n_edges=[256], seeds=[0], keep_logit_rows=[0, 1, 5, 64, 128, 200, 255], slow=True),
    dict(name="par_full_B256_gain4", kind="parallel", model=_m(FULL, 256, 37), recipe="gain4",
         wseed=0, n_edges=[256], seeds=[0], keep_logit_rows=[0, 1, 5, 64, 128, 200, 255], slow=True),
    # --- SurfaceFormer (seq2seq) --------------------------------------------------------------------
    dict(name="seq_small_default", kind="seq2seq", model=_m(SMALL, 24, 30), recipe="default",
         wseed=0, n_edges=[20, 13], seeds=[1, 2]),
    dict(name="seq_small_gain4", kind="seq2seq", model=_m(SMALL, 24, 30), recipe="gain4",
         wseed=0, n_edges=[20, 13], seeds=[1, 2]),
    dict(name="seq_small_eos", kind="seq2seq", model=_m(SMALL, 16, 20), recipe="gain4",
         wseed=6, n_edges=[12], seeds=[3]),
    # round 5: the seq2seq BATCH stop rule (cumulative EOS count == batch size, model.py:191,207-210) firing before every sample has
    # produced its own EOS: wireframe 0 emits EOS at steps 5, 6 and 7, wireframe 1 its first at step 9 -> the reference stops after
    # step 6 (searched with the oracle over weight seeds; tests/test_parity_golden.py builds its stop_each_eos test on the same pair)
    dict(name="seq_small_repeat_eos", kind="seq2seq", model=_m(SMALL, 16, 20), recipe="gain4",
         wseed=9, n_edges=[7, 12], seeds=[33, 30]),
    # config E style: ragged batch with up to 300 edges (S = 304 > 288: attention key chunking, pointer
    # tail), small model dims so the reference finishes in seconds
    dict(name="par_small_ragged300", kind="parallel", model=_m(SMALL, 300, 6), recipe="gain4",
         wseed=5, n_edges=[300, 64, 129], seeds=[21, 22, 23], keep_logit_rows=[0, 7, 299, 300, 363, 600, 728, 899]),
    # config D style: an extra pointer mask OR-ed into the padding mask (the reference has no such
    # operand: the golden comes from the reference with the mask argument of select_next widened)
    dict(name="seq_small_extramask", kind="seq2seq", model=_m(SMALL, 24, 30), recipe="gain4",
         wseed=2, n_edges=[20, 13], seeds=[1, 2], extra_mask_seed=9),
    dict(name="par_small_extramask", kind="parallel", model=_m(SMALL, 24, 9), recipe="gain4",
         wseed=0, n_edges=[20, 13], seeds=[1, 2], extra_mask_seed=4),
    # --- round 6: the constructor arguments the reference hands to its layers and no config sets (model.py:14-18,33-45) ---------
    # post-norm layers (transformer.py:148-162, 211-233; encoder.norm is None then, model.py:36) and gelu feed-forward blocks:
    # the package decodes these through its HIP sub-modules (models/common.py: _forward_eval_modules), not the engine
    dict(name="par_small_postnorm_gelu", kind="parallel", model=_mc(SMALL, 24, 9, normalize_before=False, activation="gelu"),
         recipe="gain4", wseed=0, n_edges=[20, 13], seeds=[1, 2]),
    dict(name="par_small_prenorm_gelu", kind="parallel", model=_mc(SMALL, 24, 9, activation="gelu"),
         recipe="gain4", wseed=1, n_edges=[20, 13], seeds=[1, 2]),
    dict(name="seq_small_postnorm", kind="seq2seq", model=_mc(SMALL, 24, 30, normalize_before=False),
         recipe="gain4", wseed=0, n_edges=[20, 13], seeds=[1, 2]),
    # head widths other than 64 (num_model / num_head of the constructor: the reference's layers take any, every config gives 64):
    # ff_attention_general under the sub-module loop
    dict(name="par_small_h32", kind="parallel", model=_m(SMALL_H32, 24, 9), recipe="gain4", wseed=2, n_edges=[20, 13], seeds=[1, 2]),
    dict(name="seq_small_h128", kind="seq2seq", model=_m(SMALL_H128, 24, 30), recipe="gain4", wseed=17, n_edges=[20, 13], seeds=[1, 2]),
    # configs/seq2seq.yml sizes (config A): L=110, T=259, one 64-edge wireframe
    dict(name="seq_full_A64_gain4", kind="seq2seq", model=_m(FULL, 110, 259), recipe="gain4",
         wseed=0, n_edges=[64], seeds=[3], slow=True),
    # round 5: a BATCH of the single-sequence model at config A's sizes (four wireframes of different edge counts in one forward):
    # pins the sequence-wise micro-batching of SurfaceFormer.forward_eval and the cumulative EOS rule at the full model size
    dict(name="seq_full_A4_gain4", kind="seq2seq", model=_m(FULL, 110, 259), recipe="gain4",
         wseed=0, n_edges=[64, 40, 90, 110], seeds=[3, 103, 104, 105], slow=True),
    # --- round 2: BASELINE configs C, D, E at the FULL model size ---------------------------------------
    # config C (per-GPU batch of 256-edge wireframes, ours.yml sizes): four distinct wireframes in ONE
    # reference batch; the GPU property test decodes a 16-wireframe batch whose first four must
    # reproduce these rows under every micro-batching
    dict(name="par_full_C4x256_gain4", kind="parallel", model=_m(FULL, 256, 37), recipe="gain4",
         wseed=0, n_edges=[256, 256, 256, 256], seeds=[0, 1, 2, 3],
         keep_logit_rows=[0, 1, 255, 256, 300, 511, 512, 700, 1000, 1023], slow=True),
    # config D (seq2seq+coedge.yml sizes: L=216, T=259) with the extra pointer mask operand
    dict(name="seq_full_D216_extramask", kind="seq2seq", model=_m(FULL, 216, 259), recipe="gain4",
         wseed=1, n_edges=[216], seeds=[5], extra_mask_seed=11, slow=True),
    # config E (ours-perspective.yml with num_lines=1024, ragged 64..1024 edges): S = 1028 keys, F = 1024
    # anchor sequences per wireframe of which 960 / 724 are padding anchors in the short wireframes;
    # max_face_length 4 keeps the imported reference (which replicates memory F times) within minutes
    dict(name="par_full_E1024_gain4", kind="parallel", model=_m(FULL, 1024, 4), recipe="gain4",
         wseed=0, n_edges=[1024, 64, 300], seeds=[31, 32, 33],
         keep_logit_rows=[0, 1, 500, 1023, 1024, 1060, 1087, 1088, 2047, 2048, 2200, 2347, 2348, 3071], slow=True),
    # --- round 3: a LONG prefix at config-E key counts -------------------------------------------------------
    # ours-perspective.yml sizes (max_face_length 38) with num_lines=512: a 512-edge and a 64-edge wireframe, all 37
    # decode steps -- the block-shared cross-attention (S = 516 > 288 keys) and the padding-anchor de-duplication
    # (448 padding anchors in the short wireframe) with t*F query rows for every t up to 37
    dict(name="par_full_E512_T38_gain4", kind="parallel", model=_m(FULL, 512, 38), recipe="gain4",
         wseed=0, n_edges=[512, 64], seeds=[41, 42],
         keep_logit_rows=[0, 1, 300, 511, 512, 540, 575, 576, 800, 1023], slow=True),
    # --- round 4: the LONGEST key set with a long prefix --------------------------------------------------------
    # ours-perspective.yml sizes with num_lines=1024: one 1024-edge wireframe, all 37 decode steps -- S = 1028 keys x
    # t*F query rows for every t up to 37 (until now covered at op level and by the 3-step par_full_E1024_gain4 only)
    dict(name="par_full_E1024_T38_gain4", kind="parallel", model=_m(FULL, 1024, 38), recipe="gain4",
         wseed=0, n_edges=[1024], seeds=[51],
         keep_logit_rows=[0, 1, 2, 300, 511, 512, 800, 1023], slow=True),
]
