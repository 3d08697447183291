"""Host side of the whole-path engine: binds a model's parameters (by reference, no repacking) into
the C `ff_model` struct and runs ff_encode / ff_decode on torch-owned device memory."""
import ctypes as C

import torch

from . import lib as _L
from .ops import SPLIT_KINDS, _dev, _p, _stream, split_weight

import operator

_VERSION_OF = operator.attrgetter("_version")

DEFAULT_FLAGS = (_L.FF_REUSE_LAYER0_QKV | _L.FF_LAST_LAYER_LAST_ROW | _L.FF_DEDUP_PAD_ANCHORS
                 | _L.FF_FUSE_LAYERNORM)


def _kv_len_from_mask(mask_u8):
    """1 + index of the last unmasked key per row (0 if every key is masked)."""
    S = mask_u8.size(1)
    idx = torch.arange(1, S + 1, device=mask_u8.device, dtype=torch.int32)
    return ((mask_u8 == 0).to(torch.int32) * idx).amax(dim=1).to(torch.int32).contiguous()


class PathEngine:
    """Encoder + greedy pointer decode of one model instance on one ROCm device.

    `tensors` is a mapping name -> fp32 CUDA tensor with the reference state_dict names
    (SURVEY.md Appendix B).  Tensors are referenced, not copied: in-place weight updates are seen.
    """

    def __init__(self, tensors, num_head, num_token=4, ln_eps=1e-5, bf16_split_planes=False, fold_layernorm=True,
                 ln_in_epilogue=True, split_kind="bf16x3"):
        self._lib = _L.load()
        if split_kind not in SPLIT_KINDS:
            raise ValueError("split_kind must be one of %s" % sorted(SPLIT_KINDS))
        self.split_kind = split_kind
        self._keep = {}
        self._bound_at = {}
        self._planes = {}
        self._want_planes = bool(bf16_split_planes)
        self.ln_in_epilogue = bool(ln_in_epilogue)
        self._folded = {}
        m = _L.Model()
        get = self._get
        self.tensors = tensors
        E = tensors["project.weight"].shape[0]
        m.E, m.H = E, num_head
        m.FF = tensors["decoder.layers.0.linear1.weight"].shape[0]
        n_enc = 1 + max([int(k.split(".")[2]) for k in tensors if k.startswith("encoder.layers.")] or [-1])
        n_dec = 1 + max(int(k.split(".")[2]) for k in tensors if k.startswith("decoder.layers."))
        if n_enc > _L.FF_MAX_LAYERS or n_dec > _L.FF_MAX_LAYERS:
            raise ValueError("at most %d layers are supported" % _L.FF_MAX_LAYERS)
        if E != num_head * _L.FF_HEAD_DIM:
            raise _L.HipExtensionError(
                "the HIP attention kernels need num_model == num_head * 64 (got %d, %d)" % (E, num_head))
        m.num_enc_layers, m.num_dec_layers = n_enc, n_dec
        m.in_dim = tensors["val_enc.embedding_value.0.weight"].shape[1]
        m.num_token = num_token
        m.pos_len = tensors["pos_enc.pos_embed.weight"].shape[0]
        m.qpos_len = tensors["query_pos_enc.pos_embed.weight"].shape[0]
        m.ln_eps = ln_eps
        m.tok_embed = get("val_enc.embedding_token.weight")
        m.emb_w1, m.emb_b1 = get("val_enc.embedding_value.0.weight"), get("val_enc.embedding_value.0.bias")
        m.emb_w2, m.emb_b2 = get("val_enc.embedding_value.2.weight"), get("val_enc.embedding_value.2.bias")
        m.pos_table, m.qpos_table = get("pos_enc.pos_embed.weight"), get("query_pos_enc.pos_embed.weight")

        def mha(dst, p):
            dst.in_proj_w, dst.in_proj_b = get(p + ".in_proj_weight"), get(p + ".in_proj_bias")
            dst.out_w, dst.out_b = get(p + ".out_proj.weight"), get(p + ".out_proj.bias")

        def layer(dst, p, decoder):
            mha(dst.self_attn, p + ".self_attn")
            if decoder:
                mha(dst.cross_attn, p + ".multihead_attn")
            dst.lin1_w, dst.lin1_b = get(p + ".linear1.weight"), get(p + ".linear1.bias")
            dst.lin2_w, dst.lin2_b = get(p + ".linear2.weight"), get(p + ".linear2.bias")
            dst.norm1_w, dst.norm1_b = get(p + ".norm1.weight"), get(p + ".norm1.bias")
            dst.norm2_w, dst.norm2_b = get(p + ".norm2.weight"), get(p + ".norm2.bias")
            if decoder:
                dst.norm3_w, dst.norm3_b = get(p + ".norm3.weight"), get(p + ".norm3.bias")

        for i in range(n_enc):
            layer(m.enc[i], "encoder.layers.%d" % i, False)
        m.enc_norm_w, m.enc_norm_b = get("encoder.norm.weight"), get("encoder.norm.bias")
        for i in range(n_dec):
            layer(m.dec[i], "decoder.layers.%d" % i, True)
        m.dec_norm_w, m.dec_norm_b = get("decoder.norm.weight"), get("decoder.norm.bias")
        m.proj_w, m.proj_b = get("project.weight"), get("project.bias")
        if fold_layernorm and E % 64 == 0 and E >= 128 and m.FF % 64 == 0 and m.FF >= 128:
            # FF_FUSE_LAYERNORM: gamma / beta of every decoder LayerNorm and the query-position table are folded
            # ONCE into the projection that consumes them (derived copies: re-made when a weight is updated in place)
            dev = tensors["project.weight"].device
            qpos = tensors["query_pos_enc.pos_embed.weight"]
            with torch.cuda.device(dev):
                for i in range(n_dec):
                    p = "decoder.layers.%d." % i
                    lw = m.dec[i]
                    lw.ln1_w, lw.ln1_b, lw.ln1_pos = self._fold(
                        (i, 1), tensors[p + "self_attn.in_proj_weight"], tensors[p + "self_attn.in_proj_bias"],
                        tensors[p + "norm1.weight"], tensors[p + "norm1.bias"], qpos, 2 * E)
                    lw.ln2_w, lw.ln2_b, lw.ln2_pos = self._fold(
                        (i, 2), tensors[p + "multihead_attn.in_proj_weight"][:E], tensors[p + "multihead_attn.in_proj_bias"][:E],
                        tensors[p + "norm2.weight"], tensors[p + "norm2.bias"], qpos, E)
                    lw.ln3_w, lw.ln3_b, _ = self._fold(
                        (i, 3), tensors[p + "linear1.weight"], tensors[p + "linear1.bias"],
                        tensors[p + "norm3.weight"], tensors[p + "norm3.bias"], None, 0)
                m.proj_fold_w, m.proj_fold_b, _ = self._fold(
                    ("proj",), tensors["project.weight"], tensors["project.bias"],
                    tensors["decoder.norm.weight"], tensors["decoder.norm.bias"], None, 0)
        # Split planes (x3_min_rows > 0): of the raw weights (steps that launch their LayerNorms) and of the LayerNorm-folded ones.
        # A model whose operand bounds leave fp16's range gets the bf16 terms instead of the fp16 ones (with a warning): the
        # range of bf16 is fp32's.  `requested_kind` is what the caller asked for, `split_kind` what was bound.
        self.requested_kind = split_kind
        if bf16_split_planes:
            self._make_planes(m, tensors, n_dec, E, split_kind, ln_in_epilogue)
            if split_kind == "fp16x2":
                bad = self._check_fp16_range(tensors, n_dec, E)
                if bad:
                    import warnings
                    warnings.warn("faceformer_amd: split_kind='fp16x2' needs every operand of the split products inside fp16's range; "
                                  "this model's bounds are not (%s) -- binding the bf16x3 planes instead" % bad)
                    self.split_kind = "bf16x3"
                    self._planes = {}
                    self._make_planes(m, tensors, n_dec, E, "bf16x3", ln_in_epilogue)
        m.split_kind = SPLIT_KINDS[self.split_kind] if self._planes else 0
        self.model = m
        self.E, self.H, self.num_token = E, num_head, num_token
        self.device = tensors["project.weight"].device
        self._ws = None
        # address and in-place version of every bound tensor AS CAPTURED WHEN IT WAS BOUND (_get): a parameter moved or updated
        # between that moment and the first forward is then seen as stale by pointers_current(), not recorded as current
        self._versions = {k: self._bound_at[k][1] for k in self._keep} if (self._planes or self._folded) else {}
        self._fast_check = ([self.tensors[k] for k in self._keep], tuple(self._bound_at[k][0] for k in self._keep),
                            [self.tensors[k] for k in self._versions], tuple(self._versions.values()))
        # the stream-K exchange buffer of the launch stream is allocated here, not inside the first decode
        with torch.cuda.device(self.device):
            _L.check(self._lib.ff_gemm_prepare_stream(_stream()), "ff_gemm_prepare_stream")

    def _make_planes(self, m, tensors, n_dec, E, kind, ln_in_epilogue):
        """Split planes of the decoder projections: three exact bf16 planes per weight (+1.5x their bytes) or two fp16 planes
        (+1x), split once here -- ff_decode then runs the large steps' projections on the 16-bit matrix cores (x3_min_rows);
        re-bound after in-place weight updates (pointers_current)."""
        for i in range(n_dec):
            for field, name in (("in_proj_planes", "self_attn.in_proj_weight"), ("lin1_planes", "linear1.weight"),
                                ("lin2_planes", "linear2.weight"), ("self_out_planes", "self_attn.out_proj.weight"),
                                ("cross_q_planes", "multihead_attn.in_proj_weight"),
                                ("cross_out_planes", "multihead_attn.out_proj.weight")):
                wt = tensors["decoder.layers.%d.%s" % (i, name)]
                if field == "cross_q_planes":
                    wt = wt[:E]       # q rows only: k|v of the cross attention are projected once per batch
                if wt.shape[1] % 32 or wt.shape[1] < 64:
                    setattr(m.dec[i], field, None)
                    continue          # the split kernel needs K % 32 == 0: this weight stays f32-only (null planes)
                pl = split_weight(wt, kind)
                self._planes[(i, field)] = pl
                setattr(m.dec[i], field, pl.data_ptr())
            if E % 32 == 0:
                # planes of the FOLDED weights: the steps that take the split projections keep the LayerNorm folding
                # (ff_gemm_x3_ln / ff_gemm_x2h_ln) instead of launching their LayerNorms
                for field, cfield, key in (("ln1_planes", "ln1_csum", (i, 1)), ("ln2_planes", "ln2_csum", (i, 2)),
                                           ("ln3_planes", "ln3_csum", (i, 3))):
                    if key not in self._folded:
                        continue
                    pl = split_weight(self._folded[key][0], kind)
                    self._planes[(i, field)] = pl
                    setattr(m.dec[i], field, pl.data_ptr())
                    if ln_in_epilogue:
                        # row sums of the folded weight (fp64 sum, rounded once): LN(x) W'^T = rstd (x W'^T - mean s)
                        cs = self._folded[key][0].double().sum(dim=1).float().contiguous()
                        self._planes[(i, cfield)] = cs
                        setattr(m.dec[i], cfield, cs.data_ptr())

    def _check_fp16_range(self, tensors, n_dec, E):
        """fp16 has five exponent bits: every operand of a "2 x fp16" product must stay below 65504 in magnitude.  Weights are
        checked directly.  The activations are bounded a priori: LayerNorm-normalised rows by sqrt(E); the raw rows of the
        epilogue form are fed at 2^-6 (< 4.2e6); the attention outputs by max |v| <= sqrt(E) ||Wv_n||_2 + |b_n| (a row of a
        softmax-weighted mean of value rows; v = LN(.) Wv'^T + b or memory Wv^T + b, memory being LayerNorm output as well),
        the feed-forward hidden rows by sqrt(E) ||W1'_n||_2 + |b1_n| (Cauchy-Schwarz, gamma / beta folded in).  A model
        whose bounds do not fit gets the bf16 terms instead (the caller warns).  Returns "" or the offending bounds."""
        lim = 6.0e4
        root = float(E) ** 0.5

        def bound(W, b):
            return float((W.detach().double().norm(dim=1) * root + b.detach().double().abs()).max())
        worst = {"weights": max(float(t.float().abs().max()) for (_i, f), t in self._planes.items() if t.dim() == 4)}
        for i in range(n_dec):
            p = "decoder.layers.%d." % i
            for name, key in (("self-attention values", (i, 1)), ("feed-forward hidden", (i, 3))):
                if key in self._folded:
                    Wf, bf, _ = self._folded[key]
                    Wv = Wf[2 * E:] if key[1] == 1 else Wf
                    bv = bf[2 * E:] if key[1] == 1 else bf
                    worst[name] = max(worst.get(name, 0.0), bound(Wv, bv))
            # memory = gamma * n + beta with ||n||_2 <= sqrt(E) (the encoder's final LayerNorm): v_n = n . (W_n * gamma) + W_n . beta + b_n
            Wc, bc = tensors[p + "multihead_attn.in_proj_weight"][2 * E:], tensors[p + "multihead_attn.in_proj_bias"][2 * E:]
            ge, be = tensors["encoder.norm.weight"], tensors["encoder.norm.bias"]
            with torch.no_grad():
                worst["cross-attention values"] = max(worst.get("cross-attention values", 0.0), bound(Wc * ge, Wc @ be + bc))
            for nm in ("norm1", "norm2", "norm3"):     # un-folded steps: y = gamma * n + beta goes through the planes of the raw weight
                g_, b_ = tensors[p + nm + ".weight"], tensors[p + nm + ".bias"]
                worst["LayerNorm outputs"] = max(worst.get("LayerNorm outputs", 0.0),
                                                 float(g_.detach().abs().max()) * root + float(b_.detach().abs().max()))
        self.fp16_operand_bounds = worst
        bad = {k: v for k, v in worst.items() if not v < lim}
        return ", ".join("%s <= %.3g" % kv for kv in sorted(bad.items()))

    def _fold(self, key, W, bias, gamma, beta, pos, pos_cols):
        """(Wf, bf, P) device pointers of ff_fold_layernorm_linear for one LayerNorm -> Linear pair."""
        N, K = W.shape
        Wf = torch.empty((N, K), device=W.device, dtype=torch.float32)
        bf = torch.empty((N,), device=W.device, dtype=torch.float32)
        P = torch.empty((pos.shape[0], pos_cols), device=W.device, dtype=torch.float32) if pos is not None else None
        _L.check(self._lib.ff_fold_layernorm_linear(
            _p(W), W.stride(0), N, K, _p(bias), _p(gamma), _p(beta), _p(pos), pos.stride(0) if pos is not None else 0,
            pos.shape[0] if pos is not None else 0, pos_cols, _p(Wf), _p(bf), _p(P), _stream()),
            "ff_fold_layernorm_linear")
        self._folded[key] = (Wf, bf, P)
        return Wf.data_ptr(), bf.data_ptr(), (P.data_ptr() if P is not None else None)

    def _get(self, name):
        t = self.tensors[name]
        _dev(t, name)
        if not t.is_contiguous():
            raise ValueError("parameter %s must be contiguous" % name)
        if t.data_ptr() % 16:
            raise ValueError("parameter %s is not 16-byte aligned" % name)
        self._keep[name] = t
        self._bound_at[name] = (t.data_ptr(), t._version)      # what the struct / the derived copies were made from
        return t.data_ptr()

    def pointers_current(self):
        """True while every bound tensor still lives at the address captured in the struct -- and, when
        derived copies of the weights exist (the bf16 planes), while no bound tensor was updated in place
        since they were made (load_state_dict / optimizer.step bump `_version`).  Runs in front of EVERY forward:
        two C-level passes over the ~200 tensors (33 us; the dict / generator form took 60 us of a 58.6 ms call)."""
        kt, ptrs, vt, vers = self._fast_check
        return tuple(map(torch.Tensor.data_ptr, kt)) == ptrs and tuple(map(_VERSION_OF, vt)) == vers

    @property
    def has_planes(self):
        """True when the engine was bound WITH the bf16 planes (weights whose K the split kernel cannot take have none)."""
        return self._want_planes

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(int(nbytes), device=self.device, dtype=torch.uint8)
        return self._ws

    # ---------------------------------------------------------------------------------------------
    def prepare_mask(self, input_mask):
        """input_mask [N, L] bool / uint8 (True = padding) -> (mask_u8 [N, S] with the never-masked special-token columns in
        front, kv_len [N] int32): process_masks + key lengths as ONE launch (ff_prepare_mask)."""
        if input_mask.dtype not in (torch.bool, torch.uint8) or input_mask.dim() != 2:
            raise ValueError("input_mask must be a 2-D bool / uint8 tensor")
        self._same_device(input_mask, "input_mask")
        m = input_mask.contiguous()
        N, L = m.shape
        S = L + self.num_token
        mask_u8 = torch.empty((N, S), device=self.device, dtype=torch.uint8)
        kv_len = torch.empty((N,), device=self.device, dtype=torch.int32)
        with torch.cuda.device(self.device):
            _L.check(self._lib.ff_prepare_mask(_p(m), N, L, self.num_token, _p(mask_u8), _p(kv_len), _stream()), "ff_prepare_mask")
        return mask_u8, kv_len

    def stage_num_input(self, num_input):
        """(host int array, device int32 tensor) of the per-wireframe edge counts for decode(): made BEFORE the encoder is
        enqueued -- the pageable host-to-device copy inside decode() waited for the encoder's kernels on the same stream and
        left the GPU idle until the host had caught up (~40 us per call)."""
        vals = [int(x) for x in num_input]
        return (C.c_int * len(vals))(*vals), torch.tensor(vals, dtype=torch.int32).to(self.device), vals

    def encode(self, inp, mask_u8, kv_len=None):
        """inp [N, L, in_dim] fp32, mask_u8 [N, S] uint8 (1 = padding) -> (memory [N,S,E], kv_len)."""
        _dev(inp, "input"), _dev(mask_u8, "mask", torch.uint8)
        N, L = inp.shape[0], inp.shape[1]
        S = L + self.num_token
        inp = inp.reshape(N, L, -1).contiguous()
        if inp.shape[2] != self.model.in_dim:
            raise ValueError("input has %d values per edge, model expects %d" % (inp.shape[2], self.model.in_dim))
        mask_u8 = mask_u8.contiguous()
        if kv_len is None:
            kv_len = _kv_len_from_mask(mask_u8)
        memory = torch.empty((N, S, self.E), device=self.device, dtype=torch.float32)
        self._same_device(inp, "input"), self._same_device(mask_u8, "mask")
        nbytes = self._lib.ff_encode_workspace_bytes(C.byref(self.model), N, L)
        ws = self._workspace(nbytes)
        with torch.cuda.device(self.device):   # the C side launches on the CURRENT device / stream
            _L.check(self._lib.ff_encode(C.byref(self.model), _p(inp), _p(mask_u8), _p(kv_len), N, L,
                                         _p(memory), _p(ws), ws.numel(), _stream()), "ff_encode")
        return memory, kv_len

    def _same_device(self, t, name):
        if t is not None and t.device != self.device:
            raise _L.HipExtensionError("%s is on %s but the engine's weights are on %s" % (name, t.device, self.device))

    def decode(self, memory, mask_u8, kv_len, variant, T, F=1, num_input=None, extra_mask=None,
               chunk_wireframes=0, chunk_seqs=0, num_streams=1, sync_every=4, flags=DEFAULT_FLAGS,
               tok_sos=1, tok_eos=3, x3_min_rows=0, chunk_max_seqs=0, ln_fuse_max_rows=0,
               trace=False, return_pointer=False, no_stop=False, stop_callback=None, staged_num_input=None):
        """Greedy decode. Returns dict(predict [N*F, T] int64, steps, decoded_seqs, [pointer], [trace
        tensors indexed like predict's rows])."""
        _dev(memory, "memory")
        self._same_device(memory, "memory"), self._same_device(mask_u8, "mask"), self._same_device(kv_len, "kv_len")
        N, S, E = memory.shape
        L = S - self.num_token
        prm = _L.DecodeParams()
        prm.variant, prm.N, prm.L, prm.F, prm.T = variant, N, L, F, T
        prm.chunk_wireframes, prm.sync_every = chunk_wireframes, sync_every
        prm.chunk_seqs, prm.num_streams = chunk_seqs, num_streams
        prm.chunk_max_seqs = int(chunk_max_seqs)
        prm.ln_fuse_max_rows = int(ln_fuse_max_rows)
        prm.flags = flags | (_L.FF_RETURN_POINTER if return_pointer else 0) | (_L.FF_NO_STOP if no_stop else 0)
        if return_pointer or extra_mask is not None:   # (every padding-anchor row has its own extra-mask row)
            prm.flags &= ~_L.FF_DEDUP_PAD_ANCHORS
        prm.tok_sos, prm.tok_eos = tok_sos, tok_eos
        prm.x3_min_rows = int(x3_min_rows) if self._planes else 0
        B = N * F
        dev = self.device
        predict = torch.empty((B, T), device=dev, dtype=torch.int64)
        ni = ni_host = None
        if staged_num_input is not None:
            ni_host, ni, vals = staged_num_input
            if len(vals) != N or (num_input is not None and [int(x) for x in num_input] != vals):
                raise ValueError("staged num_input does not belong to this batch")
        elif num_input is not None:
            vals = [int(x) for x in num_input]
            if len(vals) != N:
                raise ValueError("num_input has %d entries for %d wireframes" % (len(vals), N))
            ni_host = (C.c_int * N)(*vals)
            ni = torch.tensor(vals, dtype=torch.int32).to(dev)
        if extra_mask is not None:
            _dev(extra_mask, "extra_mask", torch.uint8)
            self._same_device(extra_mask, "extra_mask")
            extra_mask = extra_mask.contiguous()
        pointer = torch.zeros((max(T - 1, 1), B, E), device=dev, dtype=torch.float32) if return_pointer else None
        tl = tb = ts = rows = None
        if trace:
            tl = torch.full((max(T - 1, 1), B, S), float("nan"), device=dev, dtype=torch.float32)
            tb = torch.full((max(T - 1, 1), B), float("nan"), device=dev, dtype=torch.float32)
            ts = torch.full((max(T - 1, 1), B), float("nan"), device=dev, dtype=torch.float32)
        rows = torch.empty(B, device=dev, dtype=torch.int32)
        nbytes = self._lib.ff_decode_workspace_bytes(C.byref(self.model), C.byref(prm), ni_host)
        ws = self._workspace(nbytes)
        cb_error, cb = [], None
        if stop_callback is not None and not no_stop:
            # external stop rule (sharded decodes): `stop_callback(counts)` gets this call's per-step counters of the steps a
            # period behind the enqueued ones and returns True to end the decode; `predict` keeps every executed step
            def _stop(_user, cnt, n):
                try:
                    return 1 if stop_callback([cnt[i] for i in range(n)]) else 0
                except BaseException as e:   # never unwind through the C frames: stop, re-raise below
                    cb_error.append(e)
                    return 1
            cb = _L.STOP_FN(_stop)
            prm.stop_fn = C.cast(cb, C.c_void_p)
        steps = C.c_int(0)
        counts = (C.c_int * max(T - 1, 1))()
        with torch.cuda.device(dev):
            _L.check(self._lib.ff_decode(
                C.byref(self.model), C.byref(prm), _p(memory), _p(mask_u8), _p(kv_len), _p(ni), ni_host,
                _p(extra_mask), _p(predict), C.byref(steps), counts, _p(pointer), _p(tl), _p(tb), _p(ts),
                _p(rows), _p(ws), ws.numel(), _stream()), "ff_decode")
        if cb_error:
            raise cb_error[0]
        out = {"predict": predict, "steps": steps.value, "step_counts": list(counts)[: steps.value],
               "seq_of_row": rows}
        if return_pointer:
            out["pointer"] = pointer[: steps.value]
        if trace:
            # the C side indexes its traces by DECODED sequence (padding-anchor rows share one); expand to
            # one entry per row of `predict`
            idx = rows.long()
            out["decoded_seqs"] = int(idx.max().item()) + 1 if B else 0
            out["logits"], out["best"], out["second"] = tl[:, idx], tb[:, idx], ts[:, idx]
        return out
