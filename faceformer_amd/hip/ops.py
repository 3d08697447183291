"""Torch-tensor front end of the C-ABI kernels (device memory + stream plumbing only).

Every function takes fp32 CUDA(ROCm) tensors, hands raw pointers to libfaceformer_hip.so on the
current torch stream and returns torch tensors.  CPU tensors are rejected: there is no fallback.
"""
import ctypes as C

import torch

from . import lib as _L


import functools


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _on_tensor_device(fn):
    """Run `fn` with the CUDA device of its first device tensor argument made current: the C side launches
    on the current HIP device and on torch's current stream OF THAT DEVICE, so a tensor on cuda:1 must
    never be handed over while cuda:0 is current.  Tensors on different devices are rejected."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        devs = {a.device for a in list(args) + list(kwargs.values()) if torch.is_tensor(a) and a.is_cuda}
        if len(devs) > 1:
            raise _L.HipExtensionError("%s: tensors on different devices %s" % (fn.__name__, sorted(map(str, devs))))
        if not devs:
            return fn(*args, **kwargs)   # the op's own checks reject CPU tensors
        with torch.cuda.device(next(iter(devs))):
            return fn(*args, **kwargs)
    return wrapper


def _dev(t, name, dtype=torch.float32):
    if not torch.is_tensor(t):
        raise TypeError("%s must be a tensor" % name)
    if not t.is_cuda:
        raise _L.HipExtensionError(
            "%s is on %s: the faceformer_amd decode path only runs on a ROCm device "
            "(no CPU fallback)" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t


def _rows(t, name):
    """2-D row-major view [rows, cols] with unit inner stride; returns (tensor, ld)."""
    _dev(t, name)
    if t.dim() != 2:
        raise ValueError("%s must be 2-D" % name)
    if t.stride(1) != 1 and t.size(1) != 1:
        t = t.contiguous()
    return t, (t.stride(0) if t.size(0) > 1 else max(t.size(1), t.stride(0)))


def _p(t):
    return None if t is None else t.data_ptr()


@_on_tensor_device
def layernorm(x, gamma, beta, eps=1e-5, pos=None, pos_div=1, pos_mod=1, want_y=True):
    """(y, ypos): y = LN(x); ypos = y + pos[(row // pos_div) % pos_mod] (None when pos is None)."""
    x, ldx = _rows(x, "x")
    rows, E = x.shape
    y = torch.empty((rows, E), device=x.device, dtype=torch.float32) if want_y else None
    ypos = torch.empty((rows, E), device=x.device, dtype=torch.float32) if pos is not None else None
    if pos is not None:
        pos, ldpos = _rows(pos, "pos")
    else:
        ldpos = 0
    lib = _L.load()
    _L.check(lib.ff_layernorm(_p(x), ldx, _p(_dev(gamma, "gamma")), _p(_dev(beta, "beta")), eps,
                              _p(y), E, _p(ypos), E, _p(pos), ldpos, pos_div, pos_mod, rows, E,
                              _stream()), "ff_layernorm")
    return y, ypos


@_on_tensor_device
def add_pos(x, pos, pos_div=1, pos_mod=1):
    x, ldx = _rows(x, "x")
    pos, ldpos = _rows(pos, "pos")
    out = torch.empty_like(x, memory_format=torch.contiguous_format)
    _L.check(_L.load().ff_add_pos(_p(x), ldx, _p(pos), ldpos, pos_div, pos_mod, _p(out), x.size(1),
                                  x.size(0), x.size(1), _stream()), "ff_add_pos")
    return out


@_on_tensor_device
def gelu_(x):
    """In-place exact (erf) GELU of a 2-D row tensor; returns x."""
    x2, ldx = _rows(x, "x")
    if x2.data_ptr() != x.data_ptr():
        raise ValueError("gelu_ works in place: x needs a unit inner stride")
    _L.check(_L.load().ff_gelu(_p(x2), ldx, x2.size(0), x2.size(1), _stream()), "ff_gelu")
    return x


@_on_tensor_device
def linear(x, weight, bias=None, act=0, residual=None, x2=None, n_split=0, tile=0, out=None):
    """out = act(xsel @ weight.T + bias) + residual on the f32 matrix cores (F.linear layout)."""
    x, lda = _rows(x, "x")
    weight, ldw = _rows(weight, "weight")
    M, K = x.shape
    N = weight.size(0)
    if weight.size(1) != K:
        raise ValueError("linear: x is [%d,%d] but weight is [%d,%d]" % (M, K, N, weight.size(1)))
    if x2 is not None:
        x2, lda2 = _rows(x2, "x2")
        if x2.shape != x.shape or lda2 != lda:
            raise ValueError("linear: x2 must match x")
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=torch.float32)
    out, ldc = _rows(out, "out")
    ldr = 0
    if residual is not None:
        residual, ldr = _rows(residual, "residual")
    if bias is not None:
        _dev(bias, "bias")
    _L.check(_L.load().ff_gemm_f32(_p(x), lda, _p(x2), n_split, _p(weight), ldw, _p(bias),
                                   _p(residual), ldr, _p(out), ldc, M, N, K, act, tile, _stream()),
             "ff_gemm_f32")
    return out


@_on_tensor_device
def linear_ln(x, weight, bias=None, act=0, residual=None, stats_in=None, eps=1e-5, row_table=None, row_div=1,
              row_cols=0, want_stats=False, tile=0, out=None):
    """ff_gemm_f32_ln: out = act(z @ weight.T + bias [+ row_table[row // row_div]]) [+ residual], where z = x, or --
    with `stats_in` ([M, K/32, 2] segment statistics of x's rows) -- the row-normalised x.  `want_stats` also
    returns the [M, N/32, 2] (mean, M2) segment statistics of `out` for the next LayerNorm."""
    x, lda = _rows(x, "x")
    weight, ldw = _rows(weight, "weight")
    M, K = x.shape
    N = weight.size(0)
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=torch.float32)
    out, ldc = _rows(out, "out")
    d = _L.GemmLnDesc()
    d.A, d.lda, d.W, d.ldw, d.bias = _p(x), lda, _p(weight), ldw, _p(bias)
    if residual is not None:
        residual, ldr = _rows(residual, "residual")
        d.residual, d.ldr = _p(residual), ldr
    d.C, d.ldc, d.M, d.N, d.K, d.act, d.tile = _p(out), ldc, M, N, K, act, tile
    if stats_in is not None:
        _dev(stats_in, "stats_in")
        stats_in = stats_in.contiguous()
        d.ln_stats_in, d.ln_nseg, d.ln_eps = _p(stats_in), stats_in.size(1), eps
    if row_table is not None:
        row_table, ldt = _rows(row_table, "row_table")
        d.row_table, d.ld_row_table, d.row_div, d.row_cols = _p(row_table), ldt, row_div, row_cols or row_table.size(1)
    stats = None
    if want_stats:
        stats = torch.full((M, N // 32, 2), float("nan"), device=x.device, dtype=torch.float32)
        d.ln_stats_out = _p(stats)
    _L.check(_L.load().ff_gemm_f32_ln(C.byref(d), _stream()), "ff_gemm_f32_ln")
    return (out, stats) if want_stats else out


@_on_tensor_device
def linear_x3_ln(x, planes, bias=None, act=0, residual=None, stats_in=None, eps=1e-5, row_table=None, row_div=1,
                 row_cols=0, want_stats=False, out=None, row0=0, rows=0, colsum=None):
    """ff_gemm_x3_ln: linear_ln() on the bf16 matrix cores with fp32 accuracy; `planes` = split_weight(folded weight).
    row0 / rows: use only weight rows [row0, row0 + rows) of the planes (bias, table and output then have `rows` columns).
    colsum ([plane rows] row sums of the folded weight): apply the normalisation in the epilogue (plain K loop)."""
    _check_planes(planes, "planes")
    plane_rows, K = planes.size(2), planes.size(1) * 16
    N = rows or plane_rows
    x, lda = _rows(x, "x")
    M = x.size(0)
    if x.size(1) != K:
        raise ValueError("linear_x3_ln: x is [%d,%d] but the weight is [%d,%d]" % (M, x.size(1), N, K))
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=torch.float32)
    out, ldc = _rows(out, "out")
    d = _L.GemmLnDesc()
    d.A, d.lda, d.bias = _p(x), lda, _p(bias)
    if residual is not None:
        residual, ldr = _rows(residual, "residual")
        d.residual, d.ldr = _p(residual), ldr
    d.C, d.ldc, d.M, d.N, d.K, d.act = _p(out), ldc, M, N, K, act
    if stats_in is not None:
        _dev(stats_in, "stats_in")
        stats_in = stats_in.contiguous()
        d.ln_stats_in, d.ln_nseg, d.ln_eps = _p(stats_in), stats_in.size(1), eps
    if row_table is not None:
        row_table, ldt = _rows(row_table, "row_table")
        d.row_table, d.ld_row_table, d.row_div, d.row_cols = _p(row_table), ldt, row_div, row_cols or row_table.size(1)
    stats = None
    if want_stats:
        stats = torch.full((M, N // 32, 2), float("nan"), device=x.device, dtype=torch.float32)
        d.ln_stats_out = _p(stats)
    if colsum is not None:
        _dev(colsum, "colsum")
    lib = _L.load()
    fn, who = (lib.ff_gemm_x2h_ln, "ff_gemm_x2h_ln") if planes.dtype == torch.float16 else (lib.ff_gemm_x3_ln, "ff_gemm_x3_ln")
    _L.check(fn(C.byref(d), planes.data_ptr(), plane_rows, row0, _p(colsum), _stream()), who)
    return (out, stats) if want_stats else out


def set_x3_tuning(shape=0):
    """Launch shape of the 3 x bf16 kernel: 0 the default (whole tiles), 1 whole tiles, 2 equal K-unit ranges."""
    _L.check(_L.load().ff_set_x3_tuning(int(shape)), "ff_set_x3_tuning")


@_on_tensor_device
def fold_layernorm_linear(weight, bias, gamma, beta, pos=None, pos_cols=0):
    """(Wf, bf, P) of ff_fold_layernorm_linear: LN(x) @ weight.T + bias == z @ Wf.T + bf with z the normalised x;
    P = pos @ weight[:pos_cols].T."""
    weight, ldw = _rows(weight, "weight")
    N, K = weight.shape
    Wf = torch.empty((N, K), device=weight.device, dtype=torch.float32)
    bf = torch.empty((N,), device=weight.device, dtype=torch.float32)
    P, ldpos = None, 0
    if pos is not None:
        pos, ldpos = _rows(pos, "pos")
        P = torch.empty((pos.size(0), pos_cols), device=weight.device, dtype=torch.float32)
    _L.check(_L.load().ff_fold_layernorm_linear(_p(weight), ldw, N, K, _p(bias), _p(gamma), _p(beta), _p(pos), ldpos,
                                                pos.size(0) if pos is not None else 0, pos_cols, _p(Wf), _p(bf), _p(P),
                                                _stream()), "ff_fold_layernorm_linear")
    return Wf, bf, P


@_on_tensor_device
def split_kv(k, v, num_groups, num_heads, nk, k_group_stride, k_stride):
    """fp16 planes of K | V for the 2 x fp16 attention kernel (ff_attention_split_kv): a uint8 tensor of
    ff_attention_planes_bytes(num_groups, num_heads) bytes; 1 <= nk <= 288."""
    _dev(k, "k"), _dev(v, "v")
    lib = _L.load()
    planes = torch.empty(int(lib.ff_attention_planes_bytes(num_groups, num_heads)), device=k.device, dtype=torch.uint8)
    _L.check(lib.ff_attention_split_kv(_p(k), _p(v), k.stride(0), v.stride(0), num_groups, num_heads, nk, k_group_stride, k_stride,
                                       planes.data_ptr(), _stream()), "ff_attention_split_kv")
    return planes


_attn_algo = 0      # what set_attention_algo() last set: 4 makes attention() split K | V itself (tests of the 2 x fp16 kernel)


@_on_tensor_device
def attention(q, k, v, num_groups, num_heads, nq, nk, q_group_stride, q_inner, q_outer_stride,
              k_group_stride, k_stride, kv_len=None, key_mask=None, causal=False, scale=0.125,
              out=None, kv_planes=None):
    """Raw descriptor-level attention (see ff_attn_desc).  q/k/v/out are 2-D row tensors (views
    into wider buffers are fine: the leading dimension is taken from stride(0)).  kv_planes: split_kv(k, v, ...) of the same
    K | V -- eligible launches then run on the fp16 matrix cores."""
    _dev(q, "q"), _dev(k, "k"), _dev(v, "v")
    if kv_planes is None and _attn_algo == 4 and 0 < nk <= 288 and not causal:
        kv_planes = split_kv(k, v, num_groups, num_heads, nk, k_group_stride, k_stride)
    if out is None:
        out = torch.empty((q.size(0), num_heads * _L.FF_HEAD_DIM), device=q.device, dtype=torch.float32)
    d = _L.AttnDesc()
    d.q, d.k, d.v, d.o = _p(q), _p(k), _p(v), _p(out)
    d.ldq, d.ldk, d.ldv, d.ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    d.num_groups, d.num_heads, d.nq, d.nk = num_groups, num_heads, nq, nk
    d.q_group_stride, d.q_inner, d.q_outer_stride = q_group_stride, q_inner, q_outer_stride
    d.k_group_stride, d.k_stride = k_group_stride, k_stride
    if kv_len is not None:
        _dev(kv_len, "kv_len", torch.int32)
    if key_mask is not None:
        _dev(key_mask, "key_mask", torch.uint8)
        d.mask_stride = key_mask.stride(0)
    d.kv_len, d.key_mask = _p(kv_len), _p(key_mask)
    d.causal = 1 if causal else 0
    d.scale = scale
    d.kv_planes = _p(kv_planes)
    _L.check(_L.load().ff_attention(C.byref(d), _stream()), "ff_attention")
    return out


@_on_tensor_device
def attention_general(q, k, v, num_groups, num_heads, head_dim, nq, nk, q_group_stride, q_inner, q_outer_stride,
                      k_group_stride, k_stride, kv_len=None, key_mask=None, causal=False, attn_bias=None, attn_mask=None,
                      scale=None, out=None):
    """ff_attention_general: any head width, torch's `attn_mask` forms.  attn_bias: fp32 [nq, nk] or [num_groups*num_heads, nq, nk]
    (added to the scores); attn_mask: uint8 / bool of the same shapes (non-zero = key removed).  A query without keys -> NaN (torch)."""
    _dev(q, "q"), _dev(k, "k"), _dev(v, "v")
    if out is None:
        out = torch.empty((q.size(0), num_heads * head_dim), device=q.device, dtype=torch.float32)
    d = _L.AttnGeneralDesc()
    d.q, d.k, d.v, d.o = _p(q), _p(k), _p(v), _p(out)
    d.ldq, d.ldk, d.ldv, d.ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    d.num_groups, d.num_heads, d.head_dim, d.nq, d.nk = num_groups, num_heads, head_dim, nq, nk
    d.q_group_stride, d.q_inner, d.q_outer_stride = q_group_stride, q_inner, q_outer_stride
    d.k_group_stride, d.k_stride = k_group_stride, k_stride
    if kv_len is not None:
        _dev(kv_len, "kv_len", torch.int32)
    if key_mask is not None:
        _dev(key_mask, "key_mask", torch.uint8)
        d.mask_stride = key_mask.stride(0)
    d.kv_len, d.key_mask = _p(kv_len), _p(key_mask)
    d.causal = 1 if causal else 0
    keep = []
    for name, m, dt in (("attn_bias", attn_bias, torch.float32), ("attn_mask", attn_mask, torch.uint8)):
        if m is None:
            continue
        if m.dtype == torch.bool and dt == torch.uint8:
            m = m.to(torch.uint8)
        _dev(m, name, dt)
        if m.dim() == 2:
            want = (nq, nk)
        elif m.dim() == 3:
            want = (num_groups * num_heads, nq, nk)
        else:
            raise ValueError("%s must be 2-D [nq, nk] or 3-D [num_groups*num_heads, nq, nk]" % name)
        if tuple(m.shape) != want:
            raise ValueError("%s must have shape %s, got %s" % (name, want, tuple(m.shape)))
        m = m.contiguous()
        keep.append(m)
        ld, bs = nk, (nq * nk if m.dim() == 3 else 0)
        if (d.attn_bias or d.attn_mask) and (d.attn_ld != ld or d.attn_batch_stride != bs):
            # both given with different batching: expand the shared one (rare; keeps the descriptor to one addressing)
            raise ValueError("attn_bias and attn_mask must both be 2-D or both 3-D")
        d.attn_ld, d.attn_batch_stride = ld, bs
        setattr(d, name, m.data_ptr())
    d.scale = float(head_dim) ** -0.5 if scale is None else scale
    _L.check(_L.load().ff_attention_general(C.byref(d), _stream()), "ff_attention_general")
    del keep
    return out


@_on_tensor_device
def pointer_argmax(p, memory, mask=None, kv_len=None, extra_mask=None, seqs_per_group=1,
                   want_logits=False, want_rows=False, counters=None, ge_bound=0, eq_value=0):
    """select_next: returns dict(next, best, second, [logits], [rows])."""
    p, ldp = _rows(p, "p")
    _dev(memory, "memory")
    if memory.dim() != 3 or not memory.is_contiguous():
        raise ValueError("memory must be a contiguous [N, S, E] tensor")
    B, E = p.shape
    S = memory.size(1)
    dev = p.device
    nxt = torch.empty(B, device=dev, dtype=torch.int32)
    best = torch.empty(B, device=dev, dtype=torch.float32)
    second = torch.empty(B, device=dev, dtype=torch.float32)
    logits = torch.empty((B, S), device=dev, dtype=torch.float32) if want_logits else None
    rows = torch.empty((B, E), device=dev, dtype=torch.float32) if want_rows else None
    if mask is not None:
        _dev(mask, "mask", torch.uint8)
    if kv_len is not None:
        _dev(kv_len, "kv_len", torch.int32)
    ldextra = 0
    if extra_mask is not None:
        _dev(extra_mask, "extra_mask", torch.uint8)
        ldextra = extra_mask.stride(0)
    cge = ceq = None
    if counters is not None:
        _dev(counters, "counters", torch.int32)
        cge, ceq = counters.data_ptr(), counters.data_ptr() + 4
    _L.check(_L.load().ff_pointer_argmax(
        _p(p), ldp, _p(memory), S, E, _p(mask), _p(kv_len), _p(extra_mask), ldextra, B,
        seqs_per_group, _p(nxt), _p(best), _p(second), _p(logits), S, _p(rows), E,
        cge, ge_bound, ceq, eq_value, _stream()), "ff_pointer_argmax")
    out = {"next": nxt, "best": best, "second": second}
    if want_logits:
        out["logits"] = logits
    if want_rows:
        out["rows"] = rows
    return out


@_on_tensor_device
def gather_rows(memory, tok, seqs_per_group=1):
    _dev(memory, "memory"), _dev(tok, "tok", torch.int32)
    N, S, E = memory.shape
    out = torch.empty((tok.numel(), E), device=memory.device, dtype=torch.float32)
    _L.check(_L.load().ff_gather_rows(_p(memory.contiguous()), S, E, _p(tok), tok.numel(),
                                      seqs_per_group, _p(out), E, _stream()), "ff_gather_rows")
    return out


def set_attention_algo(algo):
    """0 automatic, 1 block-shared LDS staging, 2 wave-independent, 3 K/V-resident, 4 the 2 x fp16 kernel (attention() then splits
    K | V itself when the caller gives no planes); returns the previous value."""
    global _attn_algo
    _attn_algo = int(algo)
    return _L.load().ff_set_attention_algo(int(algo))


def set_tuning(name, value):
    """One tuning knob of the library (include/faceformer_hip.h: ff_set_tuning), e.g. set_tuning("FF_L0_FOLD", 0);
    returns the previous value.  Process-wide; a decode snapshots the knobs that shape it when it starts."""
    import ctypes
    lib, old = _L.load(), ctypes.c_int(0)
    _L.check(lib.ff_get_tuning(name.encode(), ctypes.byref(old)), "ff_get_tuning")
    _L.check(lib.ff_set_tuning(name.encode(), int(value)), "ff_set_tuning")
    return old.value


def get_tuning(name):
    import ctypes
    v = ctypes.c_int(0)
    _L.check(_L.load().ff_get_tuning(name.encode(), ctypes.byref(v)), "ff_get_tuning")
    return v.value


def reset_tuning():
    """Every knob back to its built-in default."""
    _L.check(_L.load().ff_reset_tuning(), "ff_reset_tuning")


def set_gemm_tuning(min_units=2, two_per_cu_units=2048, fix_tenths=25, small_max_rows=1024):
    """Launch shape of the stream-K projection kernel and row limit of the small-M kernel (see
    include/faceformer_hip.h); no arguments = the defaults."""
    _L.check(_L.load().ff_set_gemm_tuning(int(min_units), int(two_per_cu_units), int(fix_tenths),
                                          int(small_max_rows)), "ff_set_gemm_tuning")


SPLIT_KINDS = {"bf16x3": 0, "fp16x2": 1}     # ff_model.split_kind


@_on_tensor_device
def split_weight(weight, kind="bf16x3"):
    """[N, K] fp32 matrix -> its split planes in the K-blocked layout [terms, K/16, N, 16]: three bf16 planes ("bf16x3",
    planes_to_matrix(planes) == weight up to 2^-25 relative) or two fp16 planes ("fp16x2": w1 = fp16(w), w2' = fp16((w - w1) 2^11),
    22 mantissa bits; |w| must be < 65504)."""
    weight, ldw = _rows(weight, "weight")
    N, K = weight.shape
    if K % 16:
        raise ValueError("split_weight: K must be a multiple of 16")
    if kind == "fp16x2":
        planes = torch.empty((2, K // 16, N, 16), device=weight.device, dtype=torch.float16)
        _L.check(_L.load().ff_split_weight_fp16x2(_p(weight), ldw, N, K, planes.data_ptr(), _stream()), "ff_split_weight_fp16x2")
        return planes
    if kind != "bf16x3":
        raise ValueError("split_weight: kind must be 'bf16x3' or 'fp16x2'")
    planes = torch.empty((3, K // 16, N, 16), device=weight.device, dtype=torch.bfloat16)
    _L.check(_L.load().ff_split_weight_bf16x3(_p(weight), ldw, N, K, planes.data_ptr(), _stream()),
             "ff_split_weight_bf16x3")
    return planes


def planes_to_matrix(planes):
    """Inverse of split_weight (fp64 sum of the terms), [rows, K]."""
    kb, rows = planes.size(1), planes.size(2)
    p = planes.double()
    total = p[0] + p[1] / 2048.0 if planes.dtype == torch.float16 else p.sum(0)
    return total.permute(1, 0, 2).reshape(rows, kb * 16)


def _check_planes(p, what):
    ok = (p.dtype == torch.bfloat16 and p.size(0) == 3) or (p.dtype == torch.float16 and p.size(0) == 2)
    if not ok or p.dim() != 4 or p.size(3) != 16 or not p.is_contiguous():
        raise ValueError("linear_x3: %s must be a contiguous [3, K/16, rows, 16] bf16 or [2, K/16, rows, 16] fp16 tensor "
                         "(split_weight)" % what)


@_on_tensor_device
def linear_x3(x, planes, bias=None, act=0, residual=None, x2=None, n_split=0, out=None):
    """linear() on the bf16 matrix cores with fp32 accuracy; `planes` comes from split_weight()."""
    _check_planes(planes, "planes")
    N, K = planes.size(2), planes.size(1) * 16
    x, lda = _rows(x, "x")
    M = x.size(0)
    if x.size(1) != K:
        raise ValueError("linear_x3: x is [%d,%d] but the weight is [%d,%d]" % (M, x.size(1), N, K))
    if x2 is not None:
        x2, lda2 = _rows(x2, "x2")
        if x2.shape != x.shape or lda2 != lda:
            raise ValueError("linear_x3: x2 must match x")
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=torch.float32)
    out, ldc = _rows(out, "out")
    ldr = 0
    if residual is not None:
        residual, ldr = _rows(residual, "residual")
    if bias is not None:
        _dev(bias, "bias")
    lib = _L.load()
    fn, who = (lib.ff_gemm_x2h, "ff_gemm_x2h") if planes.dtype == torch.float16 else (lib.ff_gemm_x3, "ff_gemm_x3")
    _L.check(fn(_p(x), lda, _p(x2), n_split, planes.data_ptr(), _p(bias), _p(residual), ldr,
                _p(out), ldc, M, N, K, act, _stream()), who)
    return out
