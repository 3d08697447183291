"""GPU: the product path (faceformer_amd models on libfaceformer_hip.so) against the golden vectors
captured from the reference, and against the oracle on fresh seeded inputs.

Bars (BASELINE.json north_star): greedy edge-index sequences bit-exact; logits within 1e-3 in fp32.
The 1e-3 is absolute at the default-init logit scale (|logit| ~ 40); for the 'gain4' parity weights
(|logit| up to ~370) the same bound is applied relative to the step's logit scale:
    |logit_hip - logit_ref| <= 1e-3 * max(1, max|logit_ref| / 40).
Token equality is REQUIRED wherever the reference's own top-2 margin exceeds that tolerance (a
different fp32 summation order may legitimately flip an exact or near tie); once a sequence has taken
a different (tied) branch its later tokens are no longer comparable and are skipped.
"""
import os

import numpy as np
import pytest
import torch

from conftest import batch_to, build_model, case_weights_and_batch, golden_names, load_golden, token_ns

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3
LOGIT_SCALE = 40.0


def _tol(ref_logits_step):
    fill = np.finfo(np.float32).min
    live = ref_logits_step[ref_logits_step > fill]
    scale = float(np.abs(live).max()) if live.size else 1.0
    return LOGIT_TOL * max(1.0, scale / LOGIT_SCALE)


def run_traced(model, case, batch):
    """Run encode+decode through the engine with tracing; returns dict of numpy arrays."""
    from faceformer_amd.hip import lib as L
    kind = case["kind"]
    eng, memory, mask, kv_len = model._encode(batch)
    T = case["model"]["seq_len"]
    extra = model._extra_mask(batch)
    if kind == "parallel":
        ni = [int(n) for n in batch["num_input"]]
        out = eng.decode(memory, mask, kv_len, L.FF_PARALLEL, T=T, F=max(ni), num_input=ni, trace=True,
                         extra_mask=extra, sync_every=model.sync_every, flags=model.decode_flags,
                         x3_min_rows=model.x3_min_rows,
                         chunk_wireframes=model.chunk_wireframes, chunk_seqs=model.chunk_seqs,
                         chunk_max_seqs=model.chunk_max_seqs, num_streams=model.num_streams,
                         ln_fuse_max_rows=getattr(model, "ln_fuse_max_rows", 0))
    else:
        out = eng.decode(memory, mask, kv_len, L.FF_SEQ2SEQ, T=T, F=1, trace=True, sync_every=1,
                         extra_mask=extra, flags=model.decode_flags, return_pointer=True,
                         x3_min_rows=model.x3_min_rows,
                         chunk_wireframes=model.chunk_wireframes, chunk_max_seqs=getattr(model, "chunk_max_seqs", 0))
    out["memory"] = memory
    return out


def compare_with_golden(case, z, out):
    steps = int(z["steps"])
    assert out["steps"] == steps, "number of executed decode steps differs"
    kind = case["kind"]
    T = case["model"]["seq_len"]
    pred = out["predict"].cpu().numpy().reshape(-1, T)
    gold = z["predict"].reshape(-1, T)
    B = gold.shape[0]
    margin, rows = z["margin"], z["logit_rows"]
    logits = out["logits"].cpu().numpy()
    # encoder memory
    mem = out["memory"].cpu().numpy()
    if "memory" in z:
        assert np.abs(mem - z["memory"]).max() < 2e-4 * max(1.0, np.abs(z["memory"]).max())
    else:
        assert np.abs(mem[:, :8] - z["memory_head"]).max() < 2e-4 * max(1.0, np.abs(z["memory_head"]).max())
    alive = np.ones(B, dtype=bool)      # sequences whose prefix still equals the reference's
    n_cmp = n_skip = 0
    worst = 0.0
    worst_rel, worst_at = 0.0, (0, 0)   # |dlogit| / max|logit_ref| of the step (SURVEY 7.3-1's relative reading), and where
    for s in range(steps):
        tol = _tol(z["logits"][s])
        scale = tol / LOGIT_TOL * LOGIT_SCALE          # = max(LOGIT_SCALE, max|logit_ref| of the step)
        # logits of the stored rows (only while the prefix is identical)
        for ri, b in enumerate(rows):
            if alive[b]:
                d = np.abs(logits[s, b] - z["logits"][s, ri]).max()
                worst = max(worst, d / tol)
                if d / scale > worst_rel:
                    worst_rel, worst_at = d / scale, (s, int(b))
                assert d <= tol, "step %d seq %d: |dlogit|=%g > tol %g" % (s, b, d, tol)
        same = pred[:, s + 1] == gold[:, s + 1]
        must = alive & (margin[s] > 2 * tol)
        assert same[must].all(), "step %d: token mismatch at a decisive margin (seqs %s)" % (
            s, np.where(must & ~same)[0][:8])
        n_cmp += int(must.sum())
        n_skip += int((alive & ~must).sum())
        alive &= same
    # zero padding after the stop step
    assert (pred[:, steps + 1:] == 0).all()
    assert np.array_equal(pred[:, 0], gold[:, 0])
    if kind == "parallel":
        # the benchmark configs must be decisive almost everywhere, otherwise the test is vacuous
        assert n_cmp >= 0.9 * (n_cmp + n_skip)
    return dict(worst_logit_over_tol=worst, compared=n_cmp, skipped=n_skip, identical=float(alive.mean()),
                worst_rel=worst_rel, worst_rel_at=worst_at)


# ---- the bar, frozen (round 6) -------------------------------------------------------------------------------------------------
# Besides the bar above (|dlogit| <= 1e-3 max(1, max|logit| / 40), i.e. 2.5e-5 of the logit scale for the gain-4 weights), every
# FULL-SIZE golden (E = 512, the BASELINE configurations) must
#   (1) stay below FROZEN_FRACTION of that bar in every arithmetic form (today's worst full-size line is 0.63), and
#   (2) with the gain-4 parity weights, meet SURVEY 7.3-1's tighter reading -- |dlogit| <= 1e-5 x the step's logit scale --
#       except for the goldens listed in REL_1E5_EXCEPTIONS with the bound they are held to instead (measured value, rounded up;
#       where it comes from is in profiles/r06/parity_margins.txt: `worst_rel ... at step/sequence`).
# More folding / re-association that pushes a line past these fails the suite instead of creeping towards the bar.
FROZEN_FRACTION = 0.70
REL_1E5 = 1.0e-5
REL_1E5_EXCEPTIONS = {
    # name: relative bound it is held to.  Long key sets (516 / 1028 keys x 37 steps) and the 258-step single-sequence decode
    # accumulate the most; all of them are below 0.64 of the 2.5e-5 bar.
    "par_full_E512_T38_gain4": 1.7e-5,
    "seq_full_A64_gain4": 1.2e-5,
    "par_full_E1024_gain4": 1.1e-5,
    "par_full_n40_gain4": 1.6e-5,
}


def check_frozen_bar(name, stats):
    if "_full_" not in name:
        return
    assert stats["worst_logit_over_tol"] <= FROZEN_FRACTION, (
        "%s: worst |dlogit| / tol = %.3f exceeds the frozen fraction %.2f of the bar" % (name, stats["worst_logit_over_tol"], FROZEN_FRACTION))
    if "gain4" in name or "extramask" in name:
        bound = REL_1E5_EXCEPTIONS.get(name, REL_1E5)
        assert stats["worst_rel"] <= bound, (
            "%s: |dlogit| = %.3g of the logit scale at step %d, sequence %d (bound %.3g)"
            % (name, stats["worst_rel"], stats["worst_rel_at"][0], stats["worst_rel_at"][1], bound))


@pytest.mark.parametrize("name", golden_names())
def test_golden_parity(hip_lib, name):
    case, z = load_golden(name)
    sd, batch = case_weights_and_batch(case)
    model = build_model(case, sd, "cuda")
    out = run_traced(model, case, batch_to(batch, "cuda"))
    stats = compare_with_golden(case, z, out)
    print(name, stats)
    _record_margin(name, "package default", stats)


@pytest.mark.parametrize("name", golden_names(module_path=True))
def test_golden_parity_of_postnorm_and_gelu_models_through_the_submodule_loop(hip_lib, name):
    """Round 6: the reference hands `normalize_before` and `activation` to every layer (model.py:14-18,33-45); no config sets
    them and the native engine implements pre-norm + relu only.  A model built with the other values decodes through
    `_forward_eval_modules` -- the reference's loop over this package's HIP sub-modules -- and must meet the same bars against
    goldens captured from the imported reference built with those arguments (post-norm + gelu, pre-norm + gelu, post-norm)."""
    case, z = load_golden(name)
    sd, batch = case_weights_and_batch(case)
    model = build_model(case, sd, "cuda")
    assert not model.engine_supported()
    model._module_trace = []
    with torch.no_grad():
        res = model(batch_to(batch, "cuda"))
    T = case["model"]["seq_len"]
    out = dict(steps=len(model._module_trace), predict=res["predict"].reshape(-1, T), logits=torch.stack(model._module_trace),
               memory=model._module_memory)
    stats = compare_with_golden(case, z, out)
    print(name, stats)
    _record_margin(name, "HIP sub-module loop", stats)
    if case["kind"] == "seq2seq":       # the single-sequence model also returns `embedding` and `pointer` (model.py:216-217)
        assert tuple(res["embedding"].shape) == tuple(model._module_memory.shape)
        ptr = res["pointer"][:, -1].cpu().numpy()
        assert np.abs(ptr - z["pointer_last"]).max() < 2e-4 * max(1.0, np.abs(z["pointer_last"]).max())


def _record_margin(name, form, stats):
    """FF_PARITY_MARGINS=<file>: one line per (golden, arithmetic form) with the worst |dlogit| / tol of the run -- kept
    under profiles/<round>/parity_margins.txt so that drift towards the bar is visible between rounds."""
    path = os.environ.get("FF_PARITY_MARGINS")
    if path:   # (the frozen-bar check below runs either way)
        with open(path, "a") as f:
            f.write("%-28s %-34s worst_logit_over_tol %.3f  decisive_selections_equal %d  skipped_near_ties %d  "
                    "sequences_identical %.4f  worst_rel %.2e at step %d seq %d\n" % (
                        name, form, stats["worst_logit_over_tol"], stats["compared"], stats["skipped"], stats["identical"],
                        stats["worst_rel"], stats["worst_rel_at"][0], stats["worst_rel_at"][1]))
    check_frozen_bar(name, stats)


@pytest.mark.parametrize("name", ["par_full_B256_gain4", "par_full_B256_default", "par_full_n40_gain4", "par_small_ragged300"])
def test_golden_parity_on_the_f32_matrix_cores_only(hip_lib, name):
    """The package default sends large decoder projections through the 3 x bf16 split kernel and folds the
    LayerNorms into the projections of the small steps; this is the plain form (f32 MFMA everywhere, standalone
    LayerNorm launches) -- what bench.py's headline measures for the GEMMs -- and the always-fused form."""
    case, z = load_golden(name)
    sd, batch = case_weights_and_batch(case)
    model = build_model(case, sd, "cuda")
    model.x3_min_rows = 0
    out = run_traced(model, case, batch_to(batch, "cuda"))
    st = compare_with_golden(case, z, out)
    print(name, "f32", st)
    _record_margin(name, "f32 MFMA only, LN folded <= 12288", st)
    model.ln_fuse_max_rows = 4096      # the mixed form: steps above the limit run standalone LayerNorm launches
    st = compare_with_golden(case, z, run_traced(model, case, batch_to(batch, "cuda")))
    _record_margin(name, "f32 MFMA only, LN folded <= 4096", st)
    model.ln_fuse_max_rows = 0
    from faceformer_amd.hip import lib as L
    model.decode_flags = model.decode_flags & ~L.FF_FUSE_LAYERNORM
    out = run_traced(model, case, batch_to(batch, "cuda"))
    st = compare_with_golden(case, z, out)
    print(name, "f32, unfused", st)
    _record_margin(name, "f32 MFMA only, standalone LN", st)
    model.decode_flags = model.decode_flags | L.FF_FUSE_LAYERNORM
    model.ln_fuse_max_rows = 1 << 30
    out = run_traced(model, case, batch_to(batch, "cuda"))
    st = compare_with_golden(case, z, out)
    print(name, "f32, LayerNorm fused at every size", st)
    _record_margin(name, "f32 MFMA only, LN folded always", st)


@pytest.mark.parametrize("kind", ["bf16x3", "fp16x2"])
@pytest.mark.parametrize("min_rows", [1, 300])
@pytest.mark.parametrize("name", golden_names())
def test_golden_parity_with_bf16_split_projections(hip_lib, name, min_rows, kind):
    """The same bars with the decoder projections evaluated as split products on the 16-bit matrix cores -- three bf16 terms / six
    products, and (round 6) two fp16 terms / three products -- on every step (min_rows=1) or only the longer prefixes (300):
    with the LayerNorms folded into them (ff_gemm_x3_ln / ff_gemm_x2h_ln: the E = 512 goldens) and, second and third pass, with
    the rows normalised before the product and with standalone LayerNorm launches in front of the plain split products."""
    case, z = load_golden(name)
    sd, batch = case_weights_and_batch(case)
    model = build_model(case, sd, "cuda")
    model.x3_min_rows = min_rows
    model.split_kind = kind
    assert model.x3_ln_in_epilogue is None and model._ln_in_epilogue() == (kind == "bf16x3")   # the default form goes by the kind
    model.x3_ln_in_epilogue = True      # first pass: LayerNorm in the epilogue of the split product
    out = run_traced(model, case, batch_to(batch, "cuda"))
    assert model.engine().split_kind == kind
    stats = compare_with_golden(case, z, out)
    print(name, min_rows, stats)
    _record_margin(name, "%s from %d rows" % (kind, min_rows), stats)
    model.x3_ln_in_epilogue = False     # rows normalised before the product (the engine re-binds)
    stats = compare_with_golden(case, z, run_traced(model, case, batch_to(batch, "cuda")))
    _record_margin(name, "%s from %d rows, LN before product" % (kind, min_rows), stats)
    model.x3_ln_in_epilogue = True
    from faceformer_amd.hip import lib as L
    model.decode_flags = model.decode_flags & ~L.FF_FUSE_LAYERNORM
    stats = compare_with_golden(case, z, run_traced(model, case, batch_to(batch, "cuda")))
    _record_margin(name, "%s from %d rows, standalone LN" % (kind, min_rows), stats)


@pytest.mark.parametrize("name", ["par_small_gain4", "par_small_ragged", "par_small_earlybreak",
                                  "par_full_n40_gain4", "seq_small_gain4", "seq_small_eos",
                                  "par_small_extramask", "par_small_ragged300"])
@pytest.mark.parametrize("flags,chunk,sync,cseq,nstr", [(0, 0, 1, 0, 1), (3, 0, 0, 0, 1), (3, 1, 3, 0, 2),
                                                        (1, 2, 1, 0, 3), (2, 1, 0, 0, 1), (3, 0, 2, 5, 4),
                                                        (3, 0, 0, 7, 2), (0, 0, 1, 3, 8),
                                                        # 16 = FF_DEDUP_PAD_ANCHORS: one padding-anchor sequence
                                                        # per wireframe, width-bucketed micro-batches
                                                        (19, 0, 1, 0, 1), (19, 2, 0, 0, 2), (16, 0, 2, 5, 3),
                                                        (19, 3, 4, 0, 1),
                                                        # 32 = FF_FUSE_LAYERNORM (the default): LayerNorm folded into
                                                        # the projections; the rows above run the unfused kernels
                                                        (32, 0, 1, 0, 1), (35, 1, 0, 0, 2), (51, 0, 2, 5, 3),
                                                        (33, 2, 1, 0, 1), (34, 0, 0, 0, 1)])
def test_engine_options_do_not_change_results(hip_lib, name, flags, chunk, sync, cseq, nstr):
    """Pruning flags, micro-batching (by wireframe or by sequence group), concurrent streams and the
    host sync period are pure scheduling choices."""
    case, z = load_golden(name)
    sd, batch = case_weights_and_batch(case)
    model = build_model(case, sd, "cuda")
    model.decode_flags, model.chunk_wireframes, model.sync_every = flags, chunk, sync
    model.chunk_seqs, model.num_streams = cseq, nstr
    out = run_traced(model, case, batch_to(batch, "cuda"))
    compare_with_golden(case, z, out)


@pytest.mark.parametrize("name", ["par_small_gain4", "par_full_n40_default", "seq_small_default"])
def test_model_forward_dict_contract(hip_lib, name):
    """forward(inputs) mutates and returns the same dict with the reference's keys/shapes/dtypes."""
    case, z = load_golden(name)
    sd, batch = case_weights_and_batch(case)
    model = build_model(case, sd, "cuda")
    b = batch_to(batch, "cuda")
    with torch.no_grad():
        out = model(b)
    assert out is b
    assert out["predict"].dtype == torch.int64 and out["predict"].is_cuda
    assert tuple(out["predict"].shape) == tuple(z["predict"].shape)
    T = case["model"]["seq_len"]
    margin_ok = (z["margin"] > 1e-2).all()
    if margin_ok:
        assert np.array_equal(out["predict"].cpu().numpy(), z["predict"])
    if case["kind"] == "seq2seq":
        N, S = z["predict"].shape[0], case["model"]["L"] + 4
        assert tuple(out["embedding"].shape) == (N, S, case["model"]["E"])
        assert tuple(out["pointer"].shape) == (N, int(z["steps"]), case["model"]["E"])
        got = out["pointer"][:, -1].cpu().numpy()
        assert np.abs(got - z["pointer_last"]).max() < 1e-3 * max(1.0, np.abs(z["pointer_last"]).max())
    with pytest.raises(NotImplementedError):
        model.train()(b)


@pytest.mark.parametrize("name", ["par_small_ragged", "par_small_ragged300", "par_full_E1024_gain4"])
def test_padding_anchor_dedup_and_sorting_are_exact(hip_lib, name):
    """BASELINE config E machinery: (i) with FF_DEDUP_PAD_ANCHORS the engine decodes n_w + 1 sequences per
    wireframe instead of F = max(num_input) and copies the one padding-anchor sequence into every row
    f >= n_w -- `predict` must equal the golden in EVERY row, so the copies are checked against the
    reference's individually decoded padding rows; (ii) the model decodes a ragged batch sorted by edge count
    and un-sorts the result; (iii) far fewer sequences are decoded."""
    from faceformer_amd.hip import lib as L
    case, z = load_golden(name)
    sd, batch = case_weights_and_batch(case)
    model = build_model(case, sd, "cuda")
    ni = [int(n) for n in batch["num_input"]]
    F, T = max(ni), case["model"]["seq_len"]
    gold = z["predict"]
    decisive = (z["margin"] > 4 * 1e-3 * max(1.0, float(np.abs(z["best"]).max()) / 40.0)).all()
    b = batch_to(batch, "cuda")
    eng, memory, mask, kv_len = model._encode(b)
    outs = {}
    for dedup in (0, L.FF_DEDUP_PAD_ANCHORS):
        outs[dedup] = eng.decode(memory, mask, kv_len, L.FF_PARALLEL, T=T, F=F, num_input=ni, sync_every=0,
                                 flags=3 | dedup, trace=True)
    full, dd = outs[0], outs[L.FF_DEDUP_PAD_ANCHORS]
    assert full["decoded_seqs"] == len(ni) * F
    want = sum(min(F, n + 1) for n in ni)
    assert want <= dd["decoded_seqs"] <= int(1.34 * want) + 1      # surplus only from 25 % width buckets
    assert dd["steps"] == full["steps"] == int(z["steps"])
    if decisive:
        assert np.array_equal(dd["predict"].cpu().numpy().reshape(gold.shape), gold)
        assert np.array_equal(full["predict"].cpu().numpy().reshape(gold.shape), gold)
    compare_with_golden(case, z, dict(dd, memory=memory))
    # rows f >= n_w of a wireframe all come from ONE decoded sequence
    rows = dd["seq_of_row"].view(len(ni), F).cpu().numpy()
    for w, n in enumerate(ni):
        if n < F:
            assert (rows[w, n:] == rows[w, n]).all() and len(set(rows[w, : n + 1])) == n + 1
    # model-level path: sorted by edge count, default flags
    with torch.no_grad():
        pred = model(batch_to(batch, "cuda"))["predict"].cpu().numpy()
    if decisive:
        assert np.array_equal(pred, gold)
    assert model.last_decode_stats["decoded_seqs"] == want


def test_config_c_batch_of_256_edge_wireframes_full_size(hip_lib):
    """BASELINE config C's per-GPU workload at the full model size: a batch of 16 distinct 256-edge
    wireframes (seeds 0..15).  Wireframes 0..3 must reproduce the reference's rows of the 4-wireframe golden
    `par_full_C4x256_gain4` (tokens at decisive margins, logits within tolerance), and the result must not
    depend on the micro-batching (1, 8, 16 wireframes per chunk, or the whole batch)."""
    from faceformer_amd.synth import make_wireframes
    case, z = load_golden("par_full_C4x256_gain4")
    sd, _ = case_weights_and_batch(case)
    m = case["model"]
    model = build_model(case, sd, "cuda")
    N = 16
    batch = batch_to(make_wireframes(256, m["L"], m["seq_len"], "parallel", seeds=list(range(N))), "cuda")
    T, F = m["seq_len"], 256
    gold = z["predict"]                                   # [4, 256, 37]
    steps = int(z["steps"])
    tol_scale = 1e-3 * max(1.0, float(np.abs(z["best"]).max()) / LOGIT_SCALE)
    base = None
    for chunk in (16, 1, 8, 0):
        model.chunk_wireframes = chunk
        case16 = dict(case, n_edges=[256] * N, seeds=list(range(N)))
        out = run_traced(model, case16, batch)
        assert out["steps"] == steps                      # gain-4 weights never stop early: 36 steps
        pred = out["predict"].cpu().numpy().reshape(N, F, T)
        best = out["best"].cpu().numpy()[:steps].reshape(steps, N, F)
        # first four wireframes vs the reference
        alive = np.ones((4, F), dtype=bool)
        n_cmp = 0
        for s_ in range(steps):
            mg = z["margin"][s_].reshape(4, F)
            must = alive & (mg > 2 * tol_scale)
            same = pred[:4, :, s_ + 1] == gold[:, :, s_ + 1]
            assert same[must].all(), "chunk=%d step %d: token mismatch at a decisive margin" % (chunk, s_)
            d = np.abs(best[s_, :4] - z["best"][s_].reshape(4, F))[alive]
            assert d.max() <= tol_scale, "chunk=%d step %d: best logit off by %g" % (chunk, s_, d.max())
            n_cmp += int(must.sum())
            alive &= same
        assert n_cmp >= 0.9 * 4 * F * steps
        # stored logit rows (wireframes 0..3) at the first and last step
        logits = out["logits"].cpu().numpy()
        for ri, b_ in enumerate(z["logit_rows"]):
            for s_ in (0, steps - 1):
                if s_ == 0 or alive.reshape(-1)[b_]:
                    assert np.abs(logits[s_, b_] - z["logits"][s_, ri]).max() <= _tol(z["logits"][s_])
        # all 16 wireframes: invariant under the micro-batching
        if base is None:
            base = (pred, best)
        else:
            same = pred == base[0]
            frac = same.mean()
            assert frac > 0.995, "chunk=%d: only %.4f of the tokens equal the 16-per-chunk run" % (chunk, frac)
            assert np.abs(best[0] - base[1][0]).max() <= tol_scale     # step 0: identical prefixes


@pytest.mark.parametrize("name,N", [("seq_small_gain4", 40), ("seq_full_A64_gain4", 64)])
def test_seq2seq_batch_is_micro_batched_by_sequences_and_equals_one_wireframe_decodes(hip_lib, name, N):
    """SurfaceFormer.forward_eval of a batch (reference model.py:193-210: N sequences per step) is cut
    into micro-batches of up to chunk_max_seqs SEQUENCES.  With the stop rule off (so that every wireframe runs all T-1
    steps) a batch of N wireframes must give, wireframe by wireframe, what N one-wireframe decodes give wherever the
    one-wireframe decode's own top-2 margin is decisive -- for micro-batches of 1, 16 and all N sequences -- and the model's
    forward (cumulative EOS rule, reference model.py:191,207-210) must equal the no-stop tokens cut at its stop step."""
    from faceformer_amd.hip import lib as L
    from faceformer_amd.synth import make_wireframes
    case, z = load_golden(name)
    sd, _ = case_weights_and_batch(case)
    m = case["model"]
    model = build_model(case, sd, "cuda")
    model.x3_min_rows = 0
    T = m["seq_len"]
    rng = np.random.default_rng(7)
    n_edges = [int(v) for v in rng.integers(max(4, m["L"] // 3), m["L"] + 1, size=N)]
    n_edges[0] = case["n_edges"][0]
    seeds = [case["seeds"][0]] + list(range(100, 100 + N - 1))
    batch = batch_to(make_wireframes(n_edges, m["L"], T, "seq2seq", seeds=seeds), "cuda")
    eng, memory, mask, kv_len = model._encode(batch)

    def decode(mem, msk, kvl, seqs, trace=True):
        return eng.decode(mem, msk, kvl, L.FF_SEQ2SEQ, T=T, F=1, trace=trace, sync_every=1, flags=model.decode_flags,
                          chunk_wireframes=model.chunk_wireframes, chunk_max_seqs=seqs, no_stop=True,
                          tok_sos=model.token.SOS, tok_eos=model.token.EOS)
    singles = [decode(memory[i:i + 1], mask[i:i + 1], kv_len[i:i + 1], 1) for i in range(N)]
    one_pred = np.stack([o["predict"].cpu().numpy()[0] for o in singles])                       # [N, T]
    one_best = np.stack([o["best"].cpu().numpy()[:, 0] for o in singles], axis=1)               # [T-1, N]
    one_second = np.stack([o["second"].cpu().numpy()[:, 0] for o in singles], axis=1)
    # wireframe 0 is the golden's wireframe: the one-wireframe decode reproduces the reference's tokens up to its stop step
    gsteps = int(z["steps"])
    gold0 = z["predict"].reshape(-1, T)[0]
    if name == "seq_full_A64_gain4":
        assert np.array_equal(one_pred[0, :gsteps + 1], gold0[:gsteps + 1])
    scale = np.abs(one_best).max(axis=1)
    tol = LOGIT_TOL * np.maximum(1.0, scale / LOGIT_SCALE)                                       # [T-1]
    for seqs in (1, 16, N):
        out = decode(memory, mask, kv_len, seqs)
        pred = out["predict"].cpu().numpy()
        best = out["best"].cpu().numpy()
        alive = np.ones(N, dtype=bool)
        n_cmp = 0
        for s_ in range(T - 1):
            d = np.abs(best[s_] - one_best[s_])[alive]
            assert d.size == 0 or d.max() <= tol[s_], "chunk_max_seqs=%d step %d: best logit off by %g" % (seqs, s_, d.max())
            must = alive & ((one_best[s_] - one_second[s_]) > 2 * tol[s_])
            same = pred[:, s_ + 1] == one_pred[:, s_ + 1]
            assert same[must].all(), "chunk_max_seqs=%d step %d: token mismatch at a decisive margin" % (seqs, s_)
            n_cmp += int(must.sum())
            alive &= same
        assert n_cmp >= 0.9 * N * (T - 1) and alive.mean() > 0.9
        if seqs == 1:
            assert np.array_equal(pred, one_pred)     # micro-batches of one sequence ARE the one-wireframe launches
    # the model's forward: the same tokens, cut by the cumulative EOS rule of the reference
    model.chunk_max_seqs = N
    with torch.no_grad():
        fwd = model(dict(batch))["predict"].cpu().numpy()
    full = decode(memory, mask, kv_len, N, trace=False)["predict"].cpu().numpy()
    cum = np.cumsum((full[:, 1:] == model.token.EOS).sum(axis=0))
    hit = np.nonzero(cum == N)[0]                # (equality, like the reference: a count that jumps past N never stops)
    steps = int(hit[0]) + 1 if hit.size else T - 1
    assert np.array_equal(fwd[:, :steps + 1], full[:, :steps + 1]) and (fwd[:, steps + 1:] == 0).all()
    # inputs['pointer'] (project(decoder(...)) of every prefix row at the stop step, reference model.py:217) with SEVERAL
    # micro-batches: one strided copy per micro-batch into [steps, N, E] -- against the one-micro-batch forward
    with torch.no_grad():
        one_mb = model(dict(batch))
    model.chunk_max_seqs = 16
    with torch.no_grad():
        many_mb = model(dict(batch))
    assert torch.equal(one_mb["predict"], many_mb["predict"]) or (one_mb["predict"] == many_mb["predict"]).float().mean() > 0.97
    pa, pb = one_mb["pointer"], many_mb["pointer"]
    assert pa.shape == pb.shape == (N, steps, m["E"])
    same_prefix = (one_mb["predict"][:, :steps] == many_mb["predict"][:, :steps]).all(dim=1)    # rows fed the same tokens
    assert same_prefix.float().mean() > 0.9
    err = (pa[same_prefix] - pb[same_prefix]).abs().max() / pa.abs().max()
    assert err < 2e-4, err


def test_seq2seq_batch_with_a_repeated_eos_keeps_every_wireframes_own_tokens(hip_lib):
    """the reference's batch rule stops when the CUMULATIVE EOS count equals the batch size
    (model.py:207-210).  Small seq2seq model, weight seed 9: wireframe A (7 edges, seed 33) emits EOS at steps 5, 6 and 7,
    wireframe B (12 edges, seed 30) its first EOS at step 9 (searched with the oracle).  In the batch [A, B] the reference's
    loop -- and the engine's default rule -- stops after step 6, before B's EOS.  With `stop_each_eos` the decode runs until
    both have one, and each wireframe's tokens up to its own stop equal its one-wireframe decode: what main.py
    --batch-size and dist.decode_to_face_json parse their records from."""
    from faceformer_amd import faces as FZ
    from faceformer_amd.synth import make_state_dict, make_wireframes, state_dict_spec
    from oracle import refpath
    L_, T = 16, 20
    case = dict(kind="seq2seq", model=dict(E=128, H=2, FF=256, enc=2, dec=2, L=L_, seq_len=T))
    sd = make_state_dict(state_dict_spec("seq2seq", L_, T, 128, 256, 2, 2), "gain4", 9)
    model = build_model(case, sd, "cuda")
    tok = model.token
    pair = make_wireframes([7, 12], L_, T, "seq2seq", seeds=[33, 30])
    ref = refpath.seq2seq_forward_eval(sd, dict(pair), num_head=2)["predict"].numpy()
    with torch.no_grad():
        got = model(batch_to(pair, "cuda"))["predict"].cpu().numpy()
    assert np.array_equal(got, ref)                                  # the reference's rule, bit for bit
    assert (ref[0, 1:] == tok.EOS).sum() == 2 and (ref[1] == tok.EOS).sum() == 0 and (ref[:, 8:] == 0).all()
    singles = []
    for n, sd_ in ((7, 33), (12, 30)):
        one = make_wireframes([n], L_, T, "seq2seq", seeds=[sd_])
        with torch.no_grad():
            singles.append(model(batch_to(one, "cuda"))["predict"].cpu().numpy()[0])
        assert np.array_equal(singles[-1], refpath.seq2seq_forward_eval(sd, dict(one), num_head=2)["predict"].numpy()[0])
    assert (singles[1] == tok.EOS).sum() == 1 and int(np.nonzero(singles[1] == tok.EOS)[0][0]) == 10
    model.stop_each_eos = True
    for chunk in (256, 1):
        model.chunk_max_seqs = chunk
        with torch.no_grad():
            each = model(batch_to(pair, "cuda"))["predict"].cpu().numpy()
        assert (each[:, 11:] == 0).all() and each[1, 10] == tok.EOS      # ran through step 9, B's first EOS
        for i in range(2):
            own = FZ.apply_own_stop_rule(each[i], tok, False)
            assert np.array_equal(own, singles[i]), "wireframe %d: batch tokens differ from its one-wireframe decode" % i
    # the default rule of a one-wireframe batch is untouched by the flag
    with torch.no_grad():
        one_each = model(batch_to(make_wireframes([12], L_, T, "seq2seq", seeds=[30]), "cuda"))["predict"].cpu().numpy()[0]
    assert np.array_equal(one_each, singles[1])


def test_stop_rule_without_host_slots_drains_and_copies(hip_lib):
    """A decode with more (step, micro-batch) counters than host-mapped slots checks its stop rule the slow way (drain, copy,
    sum).  FF_PINNED_COUNTERS=8 in a child process forces that path on the early-stopping and EOS goldens: same tokens, same
    stop step as the golden (and as the slot path, which the other tests run)."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from conftest import batch_to, build_model, case_weights_and_batch, load_golden\n"
        "for name in ('par_small_earlybreak', 'par_small_break1', 'seq_small_eos', 'seq_small_repeat_eos', 'par_small_ragged'):\n"
        "    case, z = load_golden(name)\n"
        "    sd, batch = case_weights_and_batch(case)\n"
        "    model = build_model(case, sd, 'cuda')\n"
        "    model.chunk_wireframes = 1\n"
        "    with torch.no_grad():\n"
        "        pred = model(batch_to(batch, 'cuda'))['predict'].cpu().numpy()\n"
        "    assert np.array_equal(pred.reshape(-1), z['predict'].reshape(-1)), name\n"
        "print('ok')\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, FF_PINNED_COUNTERS="8"))
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stderr[-2000:]


def test_decode_steps_after_the_first_launch_no_layernorm(hip_lib):
    """Round 5: the pointer launch that appends a step's input rows also leaves their LayerNorm segment statistics, so the
    layer-0 q|k|v projection of the next step normalises them itself (ff_gemm_f32_ln) like every other projection of the
    decoder.  Counted with the library's profiling hooks on a golden: the standalone LayerNorm kernel runs in the encoder
    (2 per layer + the final norm) and in decode step 1 (rows from init_tokens_kernel) only.  The previous launch forms stay
    selectable -- FF_L0_FOLD=0 (that LayerNorm as its own launch), FF_POINTER_FOLD=0 (project and the pointer GEMM as two launches
    instead of logits = LN(x) (memory W')^T + memory b'), FF_LAST_QKV_ONE_LAUNCH_ROWS=0 (pruned last layer: k|v and q apart) -- and
    pass the same parity checks in a child process."""
    import ctypes
    import subprocess
    import sys
    case, z = load_golden("par_small_gain4")
    sd, batch = case_weights_and_batch(case)
    model = build_model(case, sd, "cuda")
    b = batch_to(batch, "cuda")
    with torch.no_grad():
        pred = model(dict(b))["predict"]
    assert np.array_equal(pred.cpu().numpy(), z["predict"])
    ncat = 7
    ms, work, cnt = (ctypes.c_double * ncat)(), (ctypes.c_double * ncat)(), (ctypes.c_longlong * ncat)()
    torch.cuda.synchronize()
    hip_lib.ff_profile_begin()
    with torch.no_grad():
        model(dict(b))
    assert hip_lib.ff_profile_end(ms, work, cnt, ncat) == 0
    n_enc = case["model"]["enc"]
    steps = int((z["predict"].reshape(-1, z["predict"].shape[-1])[:, 1:] != 0).any(axis=0).nonzero()[0].max()) + 1
    assert steps >= 3
    ln = int(cnt[2])                         # category 2 = layernorm_kernel (bench.py CAT_NAMES)
    # encoder (2 per layer + final norm) + step 1 of every micro-batch (wireframes of different compact width decode apart);
    # one more per step and micro-batch in the previous form
    assert 2 * n_enc + 1 < ln <= 2 * n_enc + 1 + len(case["n_edges"]) < 2 * n_enc + 1 + steps, (ln, n_enc, steps)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_parity_golden.py"), "-q", "-m", "gpu", "-x",
                        "-k", "test_golden_parity and (par_small_gain4 or seq_small_gain4 or par_small_ragged or seq_small_eos)"],
                       capture_output=True, text=True, timeout=900, cwd=root,
                       env={k: v for k, v in dict(os.environ, FF_L0_FOLD="0", FF_POINTER_FOLD="0", FF_LAST_QKV_ONE_LAUNCH_ROWS="0").items()
                            if k != "FF_PARITY_MARGINS"})
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-2000:] + p.stderr[-1000:]


@pytest.mark.parametrize("name", ["par_small_gain4", "seq_small_gain4", "par_small_ragged"])
def test_launch_forms_flipped_in_process_through_the_tuning_table(hip_lib, name):
    """Round 6: every A/B knob of the library is one int in ONE table (ff_set_tuning / ff_get_tuning), so a test can
    flip a launch form per call instead of per child process; the two knobs that change the workspace layout are also decode
    flags (FF_NO_L0_FOLD, FF_NO_POINTER_FOLD).  Each form must reproduce the golden; the LayerNorm launch count shows that the
    form really changed; afterwards the defaults are back."""
    import ctypes
    from faceformer_amd.hip import lib as L
    from faceformer_amd.hip import ops
    case, z = load_golden(name)
    sd, batch = case_weights_and_batch(case)
    model = build_model(case, sd, "cuda")
    b = batch_to(batch, "cuda")

    def layernorm_launches():
        ms, work, cnt = (ctypes.c_double * 7)(), (ctypes.c_double * 7)(), (ctypes.c_longlong * 7)()
        torch.cuda.synchronize()
        hip_lib.ff_profile_begin()
        with torch.no_grad():
            pred = model(dict(b))["predict"]
        assert hip_lib.ff_profile_end(ms, work, cnt, 7) == 0
        assert np.array_equal(pred.cpu().numpy(), z["predict"])
        return int(cnt[2])
    base = layernorm_launches()
    assert ops.get_tuning("FF_L0_FOLD") == 1 and ops.get_tuning("FF_POINTER_FOLD") == 1
    try:
        assert ops.set_tuning("FF_L0_FOLD", 0) == 1
        unfolded = layernorm_launches()
        assert unfolded > base                                  # one LayerNorm launch per step and micro-batch again
        ops.set_tuning("FF_L0_FOLD", 1)
        ops.set_tuning("FF_POINTER_FOLD", 0)
        ops.set_tuning("FF_LAST_QKV_ONE_LAUNCH_ROWS", 0)
        assert layernorm_launches() == base
        ops.reset_tuning()
        flags0 = model.decode_flags
        model.decode_flags = flags0 | L.FF_NO_L0_FOLD | L.FF_NO_POINTER_FOLD      # the same two forms, per call
        assert layernorm_launches() == unfolded
        model.decode_flags = flags0
        ops.set_tuning("FF_PINNED_COUNTERS", 8)                 # the stop rule through drain + copy
        assert layernorm_launches() == base
    finally:
        ops.reset_tuning()
    assert layernorm_launches() == base
    with pytest.raises(L.HipExtensionError):
        ops.set_tuning("FF_NO_SUCH_KNOB", 1)


def test_model_outside_fp16_range_falls_back_to_the_bf16_terms(hip_lib):
    """Round 6: the package default splits an fp32 operand into two fp16 terms, and fp16 has five exponent bits.  When the planes
    are bound the engine bounds every operand of those products (LayerNorm rows by sqrt(E), attention outputs and feed-forward
    hidden rows by sqrt(E) ||W'_n||_2 + |b_n|, the weights themselves); a model whose bounds do not fit -- here: linear1 of one
    layer scaled by 3e4 -- is bound with the bf16 terms instead, with a warning, and decodes exactly like a model that asked for
    them.  The goldens' own weights are inside the range (no warning: the other tests run with warnings as they come)."""
    import warnings
    case, z = load_golden("par_small_gain4")
    sd, batch = case_weights_and_batch(case)
    sd = dict(sd)
    sd["decoder.layers.0.linear1.weight"] = sd["decoder.layers.0.linear1.weight"] * 3.0e4
    preds = {}
    for kind in ("fp16x2", "bf16x3"):
        model = build_model(case, sd, "cuda")
        model.x3_min_rows, model.split_kind = 1, kind
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            with torch.no_grad():
                preds[kind] = model(batch_to(batch, "cuda"))["predict"].cpu().numpy()
        eng = model.engine()
        assert eng.split_kind == "bf16x3" and eng.requested_kind == kind
        assert (len([x for x in w if "fp16" in str(x.message)]) == 1) == (kind == "fp16x2")
        with torch.no_grad():                     # the fallback is bound once, not on every call
            model(batch_to(batch, "cuda"))
        assert model.engine() is eng
    assert np.array_equal(preds["fp16x2"], preds["bf16x3"])
    ok = build_model(case, case_weights_and_batch(case)[0], "cuda")
    ok.x3_min_rows, ok.split_kind = 1, "fp16x2"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with torch.no_grad():
            ok(batch_to(batch, "cuda"))
    assert ok.engine().split_kind == "fp16x2" and not [x for x in w if "fp16" in str(x.message)]
    assert max(ok.engine().fp16_operand_bounds.values()) < 6.0e4


def test_json_gather_over_rccl(hip_lib, tmp_path):
    """The north-star's 'RCCL all-gather of predicted face-loop JSON': decode_to_face_json on the nccl backend
    (world size 1 on this box; the gloo tests cover world sizes 2 and 3) incl. the co-edge post-processing
    branch and the extra-mask pass-through of decode_sharded."""
    import json
    import torch.distributed as dist
    from faceformer_amd import dist as ffd
    from faceformer_amd import faces as FZ
    from conftest import token_ns
    case, z = load_golden("par_small_gain4")
    sd, batch = case_weights_and_batch(case)
    model = build_model(case, sd, "cuda")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1)
        created = True
    try:
        recs = ffd.decode_to_face_json(model, batch_to(batch, "cuda"), dist)
        case_m, z_m = load_golden("par_small_extramask")
        sd_m, batch_m = case_weights_and_batch(case_m)
        model_m = build_model(case_m, sd_m, "cuda")
        out_m = ffd.decode_sharded(model_m, batch_to(batch_m, "cuda"), dist)
        out_l = ffd.decode_sharded(model, batch_to(batch, "cuda"), dist, local_shard=True)
        # the single-sequence model through the same calls: the batch whose reference stop rule fires before wireframe 1's own EOS
        # (golden seq_small_repeat_eos).  decode_sharded keeps the reference's rule; decode_to_face_json decodes until EVERY
        # wireframe has produced an EOS (stop_each_eos inside the call, restored afterwards): records of one-wireframe decodes
        case_s, z_s = load_golden("seq_small_repeat_eos")
        sd_s, batch_s = case_weights_and_batch(case_s)
        model_s = build_model(case_s, sd_s, "cuda")
        out_s = ffd.decode_sharded(model_s, batch_to(batch_s, "cuda"), dist)
        recs_s = ffd.decode_to_face_json(model_s, batch_to(batch_s, "cuda"), dist)
        assert model_s.stop_each_eos is False
        singles = []
        for i in range(2):
            one = {k: (v[i:i + 1] if torch.is_tensor(v) else v) for k, v in batch_s.items()}
            with torch.no_grad():
                singles.append(model_s(batch_to(one, "cuda"))["predict"].cpu().numpy()[0])
    finally:
        if created:
            dist.destroy_process_group()
    assert len(recs) == batch["input"].size(0)
    for i, r in enumerate(recs):
        pf, _ = FZ.parse_parallel_faces(z["predict"][i], batch["label"][i].numpy(), batch["num_input"][i], token_ns())
        want = [[t, list(f)] for t, f in FZ.unique_faces_with_majority_type(pf)]
        assert json.loads(r)["pred_faces"] == want
    assert np.array_equal(out_m["predict"].cpu().numpy(), z_m["predict"])
    assert np.array_equal(out_l["predict"].cpu().numpy(), z["predict"]) and out_l["shard_sizes"] == [2]
    assert np.array_equal(out_s["predict"].cpu().numpy(), z_s["predict"])
    for i, r in enumerate(recs_s):
        n_i = int((~batch_s["input_mask"][i]).sum())
        pf, _ = FZ.parse_faces(singles[i], batch_s["label"][i].numpy(), n_i, token_ns())
        m = FZ.face_metrics(pf, FZ.parse_faces(singles[i], batch_s["label"][i].numpy(), n_i, token_ns())[1])
        assert json.loads(r)["pred_faces"] == json.loads(FZ.dumps_record(FZ.faces_record([], [], m["predictions"], m["labels"])))["pred_faces"]


FRESH_CASES = {
    # name: (model dims, recipe, wireframe edge counts)
    "ragged3": (dict(E=128, H=2, FF=256, enc=2, dec=2, L=30, seq_len=8), "gain4", [30, 11, 24]),
    "single_edge": (dict(E=128, H=2, FF=256, enc=1, dec=1, L=6, seq_len=4), "gain4", [1]),
    "one_step": (dict(E=128, H=2, FF=256, enc=1, dec=2, L=12, seq_len=2), "gain4", [12, 5]),
    "very_ragged": (dict(E=64, H=1, FF=128, enc=1, dec=2, L=40, seq_len=6), "bias05", [1, 2, 40, 3, 17]),
    "long_prefix": (dict(E=128, H=2, FF=256, enc=1, dec=1, L=9, seq_len=70), "gain4", [9, 4]),
    "wide_head_count": (dict(E=512, H=8, FF=1024, enc=1, dec=1, L=20, seq_len=5), "default", [20, 7]),
    # num_feedforward that the bf16-split kernel (K % 32) and the LayerNorm-folded forms (FF % 64) cannot take: the engine
    # must bind (null planes for linear2, no folding) and decode on the plain f32 kernels
    "ff_not_multiple_of_16": (dict(E=128, H=2, FF=100, enc=1, dec=2, L=12, seq_len=6), "gain4", [12, 7]),
    "ff_200": (dict(E=128, H=2, FF=200, enc=1, dec=2, L=12, seq_len=6), "gain4", [9, 12]),
    # widths between the configured ones: K = 256 (the 256-wide form of the small-M panel kernel) and K = 384 / 768
    # (no small-M form, six heads: the general kernels and the folded forms' K tails)
    "e256_h4": (dict(E=256, H=4, FF=512, enc=1, dec=2, L=24, seq_len=7), "gain4", [24, 9, 17]),
    "e384_h6": (dict(E=384, H=6, FF=768, enc=1, dec=2, L=16, seq_len=6), "gain4", [16, 5]),
}


@pytest.mark.parametrize("x3_min_rows", [None, 1])
@pytest.mark.parametrize("name", sorted(FRESH_CASES))
def test_fresh_inputs_against_oracle(hip_lib, name, x3_min_rows):
    """Not only stored vectors: fresh seeded cases incl. degenerate shapes (one edge, one decode step,
    prefixes longer than two key tiles, 1..40-edge wireframes in one batch), HIP path vs the oracle run
    on the host."""
    from oracle import refpath
    dims, recipe, n_edges = FRESH_CASES[name]
    case = dict(kind="parallel", model=dims, recipe=recipe, wseed=77, n_edges=n_edges,
                seeds=[70 + i for i in range(len(n_edges))])
    T = dims["seq_len"]
    sd, batch = case_weights_and_batch(case)
    trace = {}
    ref = refpath.parallel_forward_eval(sd, {k: (v.clone() if torch.is_tensor(v) else list(v)) for k, v in batch.items()},
                                        num_head=dims["H"], trace=trace)
    model = build_model(case, sd, "cuda")
    if x3_min_rows is not None:      # every eligible projection on the bf16 matrix cores (None: the package default)
        model.x3_min_rows = x3_min_rows
    out = run_traced(model, case, batch_to(batch, "cuda"))
    steps = len(trace["logits"])
    assert out["steps"] == steps
    ref_logits = torch.stack(trace["logits"]).numpy()
    got = out["logits"].cpu().numpy()[:steps]
    pred, gold = out["predict"].cpu().numpy().reshape(-1, T), ref["predict"].numpy().reshape(-1, T)
    alive = np.ones(gold.shape[0], dtype=bool)
    for s in range(steps):
        tol = _tol(ref_logits[s])
        assert np.abs(got[s][alive] - ref_logits[s][alive]).max() <= tol
        if ref_logits.shape[2] > 1:
            srt = np.sort(ref_logits[s], axis=1)
            margin = srt[:, -1] - srt[:, -2]
        else:
            margin = np.full(gold.shape[0], np.inf)
        same = pred[:, s + 1] == gold[:, s + 1]
        assert same[alive & (margin > 2 * tol)].all()
        alive &= same
    assert (pred[:, steps + 1:] == 0).all()


@pytest.mark.parametrize("period", [1, 2, 4])
@pytest.mark.parametrize("name", ["par_small_earlybreak", "par_small_break1", "par_small_ragged", "seq_small_eos", "seq_small_gain4"])
def test_external_stop_rule_through_the_c_callback(hip_lib, name, period):
    """ff_decode_params.stop_fn: the engine hands the per-step counters to the caller's rule every sync_every steps (one period
    behind the enqueued steps) and ends the decode when it says so; `predict` keeps every executed step.  With the reference's
    own rule in the callback the result -- after the caller's zero-padding -- must be the golden tensor, the decode must end at
    most two periods after the reference's stop step, and an exception raised inside the callback must surface in Python."""
    from faceformer_amd.dist import apply_global_stop, check_points, stop_step
    from faceformer_amd.hip import lib as L
    case, z = load_golden(name)
    sd, batch = case_weights_and_batch(case)
    model = build_model(case, sd, "cuda")
    b = batch_to(batch, "cuda")
    parallel = case["kind"] == "parallel"
    variant = L.FF_PARALLEL if parallel else L.FF_SEQ2SEQ
    T, steps, N = case["model"]["seq_len"], int(z["steps"]), len(case["n_edges"])
    eng, memory, mask, kv_len = model._encode(b)
    seen = []

    def rule(counts):
        seen.append(len(counts))
        return stop_step(counts, N, variant) is not None
    kw = dict(T=T, extra_mask=model._extra_mask(b), flags=model.decode_flags, sync_every=period, stop_callback=rule)
    if parallel:
        ni = [int(n) for n in b["num_input"]]
        out = eng.decode(memory, mask, kv_len, variant, F=max(ni), num_input=ni, **kw)
    else:
        out = eng.decode(memory, mask, kv_len, variant, F=1, tok_sos=model.token.SOS, tok_eos=model.token.EOS, **kw)
    # the engine's check points ARE dist.check_points (idle ranks of a sharded decode replay them: one host collective each);
    # the decode ends with the check that saw the reference's stop step, i.e. at most two periods later -- also for periods
    # that do not divide T - 1 and for the sharded default, 2
    cps = check_points(T, period)
    assert seen == [n for _e, n in cps][: len(seen)] and len(out["step_counts"]) == out["steps"]
    if len(seen) < len(cps) or (seen and stop_step(out["step_counts"][: seen[-1]], N, variant) is not None):
        assert out["steps"] == cps[len(seen) - 1][0]       # stopped by the last check: executed = enqueued at that check
    assert steps <= out["steps"] <= min(T - 1, max(steps + 2 * period, 2 * period))
    pred, stop = apply_global_stop(out["predict"].clone(), out["step_counts"], N, variant)
    assert stop == steps
    assert np.array_equal(pred.cpu().numpy().reshape(z["predict"].shape), z["predict"])

    def broken(counts):
        raise RuntimeError("rule failed")
    if not cps:      # a period so long that this decode has no check point: the rule is never asked
        return
    with pytest.raises(RuntimeError, match="rule failed"):
        if parallel:
            eng.decode(memory, mask, kv_len, variant, F=max(ni), num_input=ni, **dict(kw, stop_callback=broken))
        else:
            eng.decode(memory, mask, kv_len, variant, F=1, tok_sos=model.token.SOS, tok_eos=model.token.EOS,
                       **dict(kw, stop_callback=broken))


def test_c_callback_keeps_its_cadence_on_the_drain_path(hip_lib):
    """with more (step, micro-batch) counters than host-mapped slots ff_decode drains and copies -- and a
    caller's stop_fn must still be asked at dist.check_points' cadence (the first enq - sync_every steps, from enq = 2
    sync_every on), because the peers and idle ranks of a sharded decode replay exactly that sequence of host collectives.
    FF_PINNED_COUNTERS=8 in a child process puts every case of the callback test on the drain path: the same assertions
    (`seen == check_points`, stop step, golden tokens, exceptions from the rule) must hold there."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_parity_golden.py"), "-m", "gpu", "-q", "-x",
                        "-k", "test_external_stop_rule_through_the_c_callback", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=900, cwd=os.path.dirname(here),
                       env=dict(os.environ, FF_PINNED_COUNTERS="8"))
    assert p.returncode == 0 and " passed" in p.stdout, (p.stdout[-1500:], p.stderr[-1500:])


def test_degenerate_inputs_behave_like_the_reference(hip_lib):
    """Empty and zero-edge inputs (checked against the imported reference in the build container, tools/edge_probe.py): a wireframe
    WITHOUT edges inside a batch decodes like the oracle (its anchors are all padding anchors, model_para.py:204-205); an empty
    batch of the single-sequence model returns empty tensors (the reference's loop stops after its first step: 0 EOS == batch
    size 0, model.py:207); an empty batch / a batch whose widest wireframe has no edges raises in the parallel model, as the
    reference does (`max()` of an empty list, model_para.py:200; a reshape to [-1, 0, T], model_para.py:238)."""
    from faceformer_amd.models import SurfaceFormer, SurfaceFormer_Parallel
    from oracle import refpath
    torch.manual_seed(0)
    kw = dict(num_model=128, num_head=2, num_feedforward=256, num_encoder_layers=1, num_decoder_layers=1, num_lines=8, token=token_ns())
    m = SurfaceFormer_Parallel(max_face_length=5, **kw).eval()
    s = SurfaceFormer(label_seq_length=6, **kw).eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m, s = m.cuda(), s.cuda()

    def batch(num_input, L=8, T=5):
        N = len(num_input)
        g = torch.Generator().manual_seed(N + 17)
        mask = torch.arange(L)[None, :] >= torch.tensor(num_input, dtype=torch.long)[:, None] if N else torch.zeros(0, L, dtype=torch.bool)
        return dict(input=torch.randn(N, L, 50, 2, generator=g), input_mask=mask, label=torch.zeros(N, L, T, dtype=torch.long),
                    num_input=list(num_input))

    for ni in ([3, 0], [0, 5, 0], [8, 1]):
        b = batch(ni)
        with torch.no_grad():
            got = m(batch_to(b, "cuda"))["predict"].cpu().numpy()
        ref = refpath.parallel_forward_eval(sd, {k: (v.clone() if torch.is_tensor(v) else list(v)) for k, v in b.items()}, num_head=2)
        assert np.array_equal(got, ref["predict"].numpy()), ni
    with torch.no_grad():
        out = s(dict(input=torch.randn(0, 8, 50, 2).cuda(), input_mask=torch.zeros(0, 8, dtype=torch.bool).cuda(),
                     label=torch.zeros(0, 6, dtype=torch.long).cuda()))
    assert tuple(out["embedding"].shape) == (0, 12, 128) and tuple(out["pointer"].shape) == (0, 1, 128)
    assert tuple(out["predict"].shape) == (0, 6) and out["predict"].dtype == torch.int64
    with pytest.raises(ValueError):
        m(batch_to(batch([]), "cuda"))
    with pytest.raises(Exception):
        m(batch_to(batch([0]), "cuda"))


@pytest.mark.parametrize("name", ["par_small_ragged", "par_small_earlybreak", "par_small_break1", "seq_small_gain4"])
def test_sharded_equals_single(hip_lib, name):
    """decode_sharded on the real engine (world_size 1 over RCCL): no-stop decode + counters +
    global stop rule + all-gather must reproduce the plain forward."""
    import torch.distributed as dist
    from faceformer_amd.dist import decode_sharded
    case, z = load_golden(name)
    sd, batch = case_weights_and_batch(case)
    model = build_model(case, sd, "cuda")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
        created = True
    try:
        out = decode_sharded(model, batch_to(batch, "cuda"), dist)
        # the host-side twin of an RCCL group that carries the periodic stop checks of a world > 1 decode (HOST tensors)
        from faceformer_amd.dist import _control_group
        ctrl = _control_group(dist, None)
        assert dist.get_backend(ctrl) == "gloo" and _control_group(dist, None) is ctrl
        t = torch.tensor([3, 0, 5], dtype=torch.int64)
        dist.all_reduce(t, group=ctrl)
        assert t.tolist() == [3, 0, 5] and t.device.type == "cpu"
    finally:
        if created:
            dist.destroy_process_group()
    assert np.array_equal(out["predict"].cpu().numpy(), z["predict"])


def test_cli_decode_writes_reference_json(hip_lib, tmp_path):
    """main.py test branch end to end on the GPU: synthetic Lightning checkpoint + wireframe JSONs ->
    per-sample JSON records whose faces equal the oracle's parsed faces."""
    import json
    import sys
    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
    import main as cli
    from faceformer_amd import faces as FZ
    from faceformer_amd.config import load_cfg
    from faceformer_amd.models import SurfaceFormer_Parallel
    from faceformer_amd.synth import make_state_dict, state_dict_spec
    from conftest import token_ns
    from oracle import refpath
    root = tmp_path / "data"
    (root / "json").mkdir(parents=True)
    rng = np.random.default_rng(5)
    names = []
    for i in range(2):
        n = 9 + i
        raw = {"edges": [rng.uniform(-1, 1, size=(2, 2)).tolist() for _ in range(n)],
               "faces_indices": [[0, [[0, 1, 2]]], [1, [[3, 4, 5, 6]]]], "pairings": {}, "dominant_directions": [[1, 0, 0]]}
        json.dump(raw, open(root / "json" / ("%08d.json" % i), "w"))
        names.append("json/%08d.json" % i)
    open(root / "test.txt", "w").write("\n".join(names) + "\n")
    cfg = load_cfg("configs/ours.yml", ["model.num_lines", "16", "model.max_face_length", "8", "model.num_model", "128",
                                         "model.num_head", "2", "model.num_feedforward", "256",
                                         "model.num_encoder_layers", "2", "model.num_decoder_layers", "2",
                                         "root_dir", str(root), "post_process.is_coedge", "False"])
    spec = state_dict_spec("parallel", 16, 8, 128, 256, 2, 2)
    sd = make_state_dict(spec, "gain4", 3)
    ckpt = tmp_path / "last.ckpt"
    torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}, "hyper_parameters": dict(cfg)}, ckpt)
    out_dir = cli.run_test(cfg, str(ckpt), out_dir=str(tmp_path / "out"))
    # --batch-size 2: both samples in ONE model(batch) call (batch-wide F and stop step) -> byte-identical records
    out_b2 = cli.run_test(cfg, str(ckpt), out_dir=str(tmp_path / "out_b2"), batch_size=2)
    for fn in sorted(os.listdir(out_dir)):
        assert open(os.path.join(out_dir, fn), "rb").read() == open(os.path.join(out_b2, fn), "rb").read()
    from faceformer_amd import datasets as D
    ds = D.ABCDataset_Parallel(str(root), ["test.txt"], cfg.model)
    for i in range(2):
        rec = json.load(open(os.path.join(out_dir, "%08d.json" % i)))
        assert set(rec) == {"edges", "dominant_directions", "pred_faces", "label_faces"}
        batch = D.collate([ds[i]])
        ref = refpath.parallel_forward_eval(sd, batch, num_head=2)
        pf, lf = FZ.parse_parallel_faces(ref["predict"][0].numpy(), ds[i]["label"], len(ds.raw_datas[i]["edges"]), token_ns())
        m = FZ.face_metrics(pf, lf)
        assert rec["pred_faces"] == [[t, list(f)] for t, f in m["predictions"]]
        assert sorted(map(tuple, map(lambda x: (x[0], tuple(x[1])), rec["label_faces"]))) == sorted(m["labels"])


def test_cli_decode_with_coedge_postprocessing_matches_reference(hip_lib, tmp_path):
    """SURVEY 8f row 2 end to end: main.py's test branch with post_process.is_coedge at its DEFAULT (True,
    reference config.py:52) on wireframes whose edges chain into closed loops and carry co-edge pairings.
    Expected records = what the imported reference produced for the same files and weights
    (oracle/make_golden_cli.py: dataset -> forward_eval -> face_accuracy, trainer.py:210-300)."""
    import json
    import sys
    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
    import main as cli
    from faceformer_amd.config import load_cfg
    from faceformer_amd.synth import make_state_dict, state_dict_spec
    from conftest import GOLDEN
    gold = json.load(open(os.path.join(GOLDEN, "cli_coedge_case.json")))
    m = gold["model"]
    root = tmp_path / "data"
    (root / "json").mkdir(parents=True)
    names = []
    for i, smp in enumerate(gold["samples"]):
        json.dump(smp["raw"], open(root / "json" / ("%08d.json" % i), "w"))
        names.append("json/%08d.json" % i)
    open(root / "test.txt", "w").write("\n".join(names) + "\n")
    cfg = load_cfg("configs/ours.yml", ["model.num_lines", str(m["num_lines"]), "model.max_face_length",
                                         str(m["max_face_length"]), "model.num_model", str(m["num_model"]),
                                         "model.num_head", str(m["num_head"]), "model.num_feedforward",
                                         str(m["num_feedforward"]), "model.num_encoder_layers",
                                         str(m["num_encoder_layers"]), "model.num_decoder_layers",
                                         str(m["num_decoder_layers"]), "root_dir", str(root)])
    assert cfg.post_process.is_coedge is True and cfg.post_process.enclosedness_tol == gold["tol"]
    spec = state_dict_spec("parallel", m["num_lines"], m["max_face_length"], m["num_model"], m["num_feedforward"],
                           m["num_encoder_layers"], m["num_decoder_layers"])
    sd = make_state_dict(spec, gold["recipe"], gold["wseed"])
    ckpt = tmp_path / "last.ckpt"
    torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}, "hyper_parameters": dict(cfg)}, ckpt)
    out_dir = cli.run_test(cfg, str(ckpt), out_dir=str(tmp_path / "out"))
    for i, smp in enumerate(gold["samples"]):
        rec = json.load(open(os.path.join(out_dir, "%08d.json" % i)))
        assert set(rec) == {"edges", "dominant_directions", "pred_faces", "label_faces"}
        assert rec["edges"] == smp["raw"]["edges"] and rec["dominant_directions"] == smp["raw"]["dominant_directions"]
        assert len(smp["pred_faces"]) > 0                       # the post-processing branch saw real faces
        assert rec["pred_faces"] == smp["pred_faces"]
        assert sorted(map(json.dumps, rec["label_faces"])) == sorted(map(json.dumps, smp["label_faces"]))


def test_cli_single_sequence_model_batched_run_writes_the_one_sample_files(hip_lib, tmp_path):
    """main.py's test branch with the single-sequence model (SurfaceFormer, reference model.py:169-219) and --batch-size 4.
    The reference's test loader is fixed at one sample; its batch stop rule (cumulative EOS count == N, model.py:211-212)
    would cut samples 1-3 of this batch short of their own EOS (weights picked for that: the oracle shows it below), so
    run_test switches the model to the per-sequence rule (FF_STOP_EACH_EOS) and the files equal the one-sample run's."""
    import json
    import sys
    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
    import main as cli
    from faceformer_amd import datasets as D
    from faceformer_amd import faces as FZ
    from faceformer_amd.config import load_cfg
    from faceformer_amd.synth import make_state_dict, state_dict_spec
    from oracle import refpath
    root = tmp_path / "data"
    (root / "json").mkdir(parents=True)
    rng = np.random.default_rng(21)
    names = []
    for i in range(4):
        n = 6 + 3 * i
        raw = {"edges": [rng.uniform(-1, 1, size=(2, 2)).tolist() for _ in range(n)],
               "faces_indices": [[[0, 1, 2]], [[3, 4, 5]]], "pairings": {}, "dominant_directions": [[1, 0, 0]]}
        json.dump(raw, open(root / "json" / ("%08d.json" % i), "w"))
        names.append("json/%08d.json" % i)
    open(root / "test.txt", "w").write("\n".join(names) + "\n")
    cfg = load_cfg("configs/seq2seq.yml", ["model.num_lines", "16", "model.label_seq_length", "24", "model.num_model", "128",
                                            "model.num_head", "2", "model.num_feedforward", "256",
                                            "model.num_encoder_layers", "2", "model.num_decoder_layers", "2",
                                            "root_dir", str(root), "post_process.is_coedge", "False"])
    assert cfg.model_class == "SurfaceFormer"
    sd = make_state_dict(state_dict_spec("seq2seq", 16, 24, 128, 256, 2, 2), "gain4", 51)
    ckpt = tmp_path / "last.ckpt"
    torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}, "hyper_parameters": dict(cfg)}, ckpt)
    one = cli.run_test(cfg, str(ckpt), out_dir=str(tmp_path / "b1"), batch_size=1)
    four = cli.run_test(cfg, str(ckpt), out_dir=str(tmp_path / "b4"), batch_size=4)
    files = sorted(os.listdir(one))
    assert len(files) == 4
    for fn in files:
        assert open(os.path.join(one, fn), "rb").read() == open(os.path.join(four, fn), "rb").read()
    # the case is a real one: under the reference's batch rule the four-sample call stops before samples 1-3 reach their EOS
    ds = D.ABCDataset(str(root), ["test.txt"], cfg.model)
    items = [ds[i] for i in range(4)]
    pb = refpath.seq2seq_forward_eval(sd, dict(D.collate(items)), num_head=2)["predict"].numpy()
    short = []
    for i in range(4):
        ps = refpath.seq2seq_forward_eval(sd, dict(D.collate([items[i]])), num_head=2)["predict"].numpy()[0]
        if not np.array_equal(FZ.apply_own_stop_rule(pb[i], cfg.model.token, False), ps):
            short.append(i)
        rec = json.load(open(os.path.join(one, "%08d.json" % i)))
        text, _ = cli.record_of(cfg, ds.raw_datas[i], items[i], ps, False)
        assert json.loads(text)["pred_faces"] == rec["pred_faces"]      # and the files carry the reference's one-sample faces
    assert short == [1, 2, 3]


@pytest.mark.parametrize("name", ["par_small_gain4", "par_full_n40_gain4", "par_full_n40_default"])
def test_error_against_fp64_truth_is_fp32_class(hip_lib, name):
    """Whose logits are closer to exact arithmetic?  The oracle restated in float64 is the truth; the
    HIP path's first-step logits must be no further from it than a small multiple of the distance of
    the reference's own fp32 CPU logits (both are fp32 evaluations with different summation orders)."""
    from oracle import refpath
    case, z = load_golden(name)
    sd, batch = case_weights_and_batch(case)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
    tr64 = {}
    refpath.parallel_forward_eval(sd64, b64, num_head=case["model"]["H"], trace=tr64)
    truth = tr64["logits"][0].numpy()                       # step 0: identical prefixes by construction
    rows = z["logit_rows"]
    ref32 = z["logits"][0]                                   # reference fp32 (golden), selected rows
    model = build_model(case, sd, "cuda")
    out = run_traced(model, case, batch_to(batch, "cuda"))
    hip = out["logits"][0].cpu().numpy()[rows]
    live = ref32 > np.finfo(np.float32).min
    err_ref = np.abs(ref32.astype(np.float64) - truth[rows])[live].max()
    err_hip = np.abs(hip.astype(np.float64) - truth[rows])[live].max()
    scale = np.abs(truth[rows][live]).max()
    print(name, "max |logit| %.1f  err(reference fp32) %.3g  err(HIP) %.3g" % (scale, err_ref, err_hip))
    assert err_hip <= 4.0 * err_ref + 1e-6 * scale
