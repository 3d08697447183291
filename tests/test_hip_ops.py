"""GPU: every C-ABI kernel against a plain PyTorch reference of the same op (computed in fp64 on the
CPU and compared at fp32 round-off class tolerances)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 1, 2, 3, 4], ids=["attn-auto", "attn-lds", "attn-wave", "attn-resident", "attn-x2h"])
def attn_algo(ops, request):
    old = ops.set_attention_algo(request.param)
    yield request.param
    ops.set_attention_algo(old)


@pytest.fixture(scope="module")
def ops(hip_lib):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from faceformer_amd.hip import ops as _ops
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


def rel_err(got, want):
    want = want.double()
    return float((got.double().cpu() - want).abs().max() / (want.abs().max() + 1e-30))


# ---- GELU (module surface: activation="gelu", reference transformer.py:276-284) ----------------------
@pytest.mark.parametrize("rows,E,ld", [(1, 256, 256), (37, 1024, 1024), (1000, 256, 320), (3, 4, 8)])
def test_gelu_rows_match_fp64_erf_form(hip_lib, ops, rows, E, ld):
    """x <- 0.5 x (1 + erf(x / sqrt 2)) in place on the first E columns of rows with leading dimension ld (torch F.gelu's
    default form) against fp64; columns beyond E stay untouched."""
    buf = rnd(rows, ld, seed=3, scale=2.5).cuda()
    before = buf.clone()
    view = buf[:, :E]
    out = ops.gelu_(view)
    assert out.data_ptr() == view.data_ptr()
    want = F.gelu(before[:, :E].double().cpu())
    assert float((buf[:, :E].double().cpu() - want).abs().max()) < 4e-7 * max(1.0, float(want.abs().max()))
    assert torch.equal(buf[:, E:], before[:, E:])


def test_head_width_other_than_64_runs_on_the_general_kernel(hip_lib, ops):
    """num_model / num_head != 64 (a constructor choice of the boundary, reference transformer.py:131-132): the layer runs on
    ff_attention_general (round 6; it was a clear error before) and matches torch's nn.MultiheadAttention arithmetic in fp64."""
    from faceformer_amd.transformer import MultiheadAttention
    for E, H in ((128, 4), (128, 1), (96, 2), (64, 8)):       # heads of 32 / 128 / 48 / 8
        mha = MultiheadAttention(E, H).eval()
        ref = torch.nn.MultiheadAttention(E, H).double().eval()
        ref.load_state_dict({k: v.double() for k, v in mha.state_dict().items()})
        q, kk = rnd(6, 3, E, seed=E + H), rnd(11, 3, E, seed=E + H + 1)
        kpm = torch.arange(11)[None, :] >= torch.tensor([11, 7, 3])[:, None]
        with torch.no_grad():
            want = ref(q.double(), kk.double(), kk.double(), key_padding_mask=kpm)[0]
            got = mha.cuda()(q.cuda(), kk.cuda(), kk.cuda(), key_padding_mask=kpm.cuda())[0]
        assert rel_err(got, want) < 5e-6, (E, H)


def test_float_key_padding_mask_and_mixed_mask_forms_match_torch(hip_lib, ops):
    """torch adds a floating-point key_padding_mask to the scores of its keys; combined with a boolean [L, S] attn_mask or an
    additive one per (batch, head) the module merges them like torch's multi_head_attention_forward does."""
    from faceformer_amd.transformer import MultiheadAttention
    E, H, L, S, B = 128, 2, 5, 9, 3
    mha = MultiheadAttention(E, H).eval()
    ref = torch.nn.MultiheadAttention(E, H).double().eval()
    ref.load_state_dict({k: v.double() for k, v in mha.state_dict().items()})
    mha = mha.cuda()
    q, kk = rnd(L, B, E, seed=1), rnd(S, B, E, seed=2)
    kpm_f = rnd(B, S, seed=3, scale=1.5)
    am_b = rnd(L, S, seed=4) > 0.7
    am_b[:, 0] = False
    am_f3 = rnd(B * H, L, S, seed=5, scale=1.5)
    for kpm, am in ((kpm_f, None), (kpm_f, am_f3), (kpm_f, am_b.float().masked_fill(am_b, float("-inf")))):
        with torch.no_grad():
            want = ref(q.double(), kk.double(), kk.double(), key_padding_mask=kpm.double(), attn_mask=None if am is None else am.double())[0]
            got = mha(q.cuda(), kk.cuda(), kk.cuda(), key_padding_mask=kpm.cuda(), attn_mask=None if am is None else am.cuda())[0]
        assert rel_err(got, want) < 5e-6


def _ref_attention_general(q, k, v, hd, kpm=None, causal=False, bias=None, amask=None):
    """q [G,H,nq,hd], k / v [G,H,nk,hd] fp64; kpm [G,nk] bool; bias / amask [nq,nk] or [G*H,nq,nk]."""
    G, H, nq, _ = q.shape
    s = (q * hd ** -0.5) @ k.transpose(-1, -2)
    if bias is not None:
        s = s + (bias.double().view(G, H, nq, -1) if bias.dim() == 3 else bias.double())
    if amask is not None:
        s = s.masked_fill(amask.view(G, H, nq, -1) if amask.dim() == 3 else amask, float("-inf"))
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :], float("-inf"))
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(nq, s.shape[-1], dtype=torch.bool), 1), float("-inf"))
    return torch.softmax(s, -1) @ v


@pytest.mark.parametrize("G,H,hd,nq,nk,form", [
    (2, 2, 64, 33, 33, "plain"), (3, 4, 32, 17, 70, "kpm"), (2, 1, 128, 9, 130, "kpm"), (2, 8, 16, 40, 40, "causal"),
    (1, 2, 48, 5, 19, "kpm"), (2, 3, 20, 7, 65, "plain"), (1, 16, 8, 6, 200, "kpm"), (1, 2, 4, 3, 9, "plain"), (1, 1, 256, 4, 31, "kpm"),
    (2, 2, 64, 12, 20, "bias2d"), (2, 2, 32, 12, 20, "bias3d"), (3, 2, 64, 10, 15, "mask2d"), (2, 4, 32, 10, 15, "mask3d"),
    (2, 2, 64, 8, 300, "bias2d+mask2d"), (2, 2, 64, 9, 9, "causal+bias2d"), (1, 1, 64, 3, 2000, "kpm"), (1, 3, 7, 5, 11, "mask2d")])
def test_attention_general_matches_fp64(hip_lib, ops, G, H, hd, nq, nk, form):
    """ff_attention_general (any head width; additive and boolean attn_mask, one matrix or one per (group, head); key padding,
    kv_len, causal) against the explicit fp64 softmax, in the position-major layout of the blocks (row = position * G + g)."""
    E = H * hd
    q, kv = rnd(nq * G, E, seed=1), rnd(nk * G, 2 * E + 8, seed=2)       # k | v packed in one buffer with a wider ld
    kpm = kv_len = bias = amask = None
    if "kpm" in form:
        keep = torch.tensor([max(1, nk - 3 * g - nk // 4) for g in range(G)])
        kpm = torch.arange(nk)[None, :] >= keep[:, None]
        if nk > 4:
            kpm[:, 1] = True
        kv_len = keep.to(torch.int32)
    if "bias2d" in form:
        bias = rnd(nq, nk, seed=3, scale=2.0)
        bias[0, min(3, nk - 1)] = float("-inf")       # torch allows -inf inside an additive mask
    if "bias3d" in form:
        bias = rnd(G * H, nq, nk, seed=4, scale=2.0)
    if "mask2d" in form:
        amask = rnd(nq, nk, seed=5) > 0.5
        amask[:, 0] = False
    if "mask3d" in form:
        amask = rnd(G * H, nq, nk, seed=6) > 0.5
        amask[..., 0] = False
    dkv = kv.cuda()
    out = ops.attention_general(q.cuda(), dkv[:, :E], dkv[:, E:2 * E], G, H, hd, nq, nk, q_group_stride=1, q_inner=1, q_outer_stride=G,
                                k_group_stride=1, k_stride=G, kv_len=None if kv_len is None else kv_len.cuda(),
                                key_mask=None if kpm is None else kpm.to(torch.uint8).cuda(), causal="causal" in form,
                                attn_bias=None if bias is None else bias.cuda(), attn_mask=None if amask is None else amask.cuda())
    qd = q.double().view(nq, G, H, hd).permute(1, 2, 0, 3)
    kd = kv[:, :E].double().view(nk, G, H, hd).permute(1, 2, 0, 3)
    vd = kv[:, E:2 * E].double().view(nk, G, H, hd).permute(1, 2, 0, 3)
    ref = _ref_attention_general(qd, kd, vd, hd, kpm, "causal" in form, bias, amask).permute(2, 0, 1, 3).reshape(nq * G, E)
    assert torch.isfinite(out).all()
    assert rel_err(out, ref) < 5e-6


def test_attention_general_random_shapes_and_mask_mixes(hip_lib, ops):
    """40 seeded random problems (head widths 1..160 incl. odd ones, 1..6 heads, 1..3 groups, up to 90 queries x 400 keys, random mixes
    of kv_len / key mask / causal / additive / boolean masks, group-major AND position-major row addressing, padded leading
    dimensions) against the explicit fp64 softmax."""
    rng = np.random.default_rng(2026)
    for case in range(40):
        G, H = int(rng.integers(1, 4)), int(rng.integers(1, 7))
        hd = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 24, 32, 40, 64, 72, 96, 128, 160]))
        nq, nk = int(rng.integers(1, 91)), int(rng.integers(1, 401))
        E, pad = H * hd, 4 * int(rng.integers(0, 3))
        group_major = bool(rng.integers(0, 2))
        q = rnd(nq * G, E + pad, seed=100 + case)
        k, v = rnd(nk * G, E + pad, seed=200 + case), rnd(nk * G, E + pad, seed=300 + case)
        use = rng.integers(0, 2, size=5).astype(bool)   # kv_len, key mask, causal, bias, bool mask
        kv_len = kpm = bias = amask = None
        causal = bool(use[2]) and nq == nk
        if use[0]:
            kv_len = torch.from_numpy(rng.integers(max(1, nk // 2), nk + 1, size=G).astype(np.int32))
        if use[1]:
            kpm = torch.from_numpy(rng.random((G, nk)) > 0.7)
            kpm[:, 0] = False
        if use[3]:
            shape = (G * H, nq, nk) if rng.integers(0, 2) else (nq, nk)
            bias = torch.from_numpy(rng.normal(size=shape).astype(np.float32) * 2)
        if use[4]:
            shape = tuple(bias.shape) if bias is not None else ((G * H, nq, nk) if rng.integers(0, 2) else (nq, nk))
            amask = torch.from_numpy(rng.random(shape) > 0.6)
            amask[..., 0] = False
        if group_major:      # row = g * n + i
            addr = dict(q_group_stride=nq, q_inner=nq, q_outer_stride=0, k_group_stride=nk, k_stride=1)
            qd = q[:, :E].double().view(G, nq, H, hd).permute(0, 2, 1, 3)
            kd = k[:, :E].double().view(G, nk, H, hd).permute(0, 2, 1, 3)
            vd = v[:, :E].double().view(G, nk, H, hd).permute(0, 2, 1, 3)
        else:                # row = i * G + g
            addr = dict(q_group_stride=1, q_inner=1, q_outer_stride=G, k_group_stride=1, k_stride=G)
            qd = q[:, :E].double().view(nq, G, H, hd).permute(1, 2, 0, 3)
            kd = k[:, :E].double().view(nk, G, H, hd).permute(1, 2, 0, 3)
            vd = v[:, :E].double().view(nk, G, H, hd).permute(1, 2, 0, 3)
        dq, dk, dv = q.cuda(), k.cuda(), v.cuda()
        out = ops.attention_general(dq[:, :E], dk[:, :E], dv[:, :E], G, H, hd, nq, nk, kv_len=None if kv_len is None else kv_len.cuda(),
                                    key_mask=None if kpm is None else kpm.to(torch.uint8).cuda(), causal=causal,
                                    attn_bias=None if bias is None else bias.cuda(), attn_mask=None if amask is None else amask.cuda(), **addr)
        full = kpm.clone() if kpm is not None else torch.zeros(G, nk, dtype=torch.bool)
        if kv_len is not None:
            full |= torch.arange(nk)[None, :] >= kv_len[:, None].long()
        ref = _ref_attention_general(qd, kd, vd, hd, full, causal, bias, amask)
        ref = ref.permute(0, 2, 1, 3).reshape(G * nq, E) if group_major else ref.permute(2, 0, 1, 3).reshape(nq * G, E)
        ok = ~torch.isnan(ref).any(dim=1)          # (a random mix may remove every key of a query: NaN on both sides)
        assert torch.isnan(out.cpu()[~ok]).any(dim=1).all(), case
        assert ok.sum() > 0 and rel_err(out[ok.cuda()], ref[ok]) < 1e-5, (case, G, H, hd, nq, nk, use.tolist())


def test_attention_general_query_without_keys_is_nan_like_torch(hip_lib, ops):
    """torch's softmax over a row of -inf is NaN and so is that query's output (nn.MultiheadAttention with a boolean mask that
    removes every key); the general kernel keeps that, other queries of the launch are unaffected."""
    G, H, hd, nq, nk = 1, 2, 32, 4, 6
    q, k, v = rnd(nq, H * hd, seed=1), rnd(nk, H * hd, seed=2), rnd(nk, H * hd, seed=3)
    amask = torch.zeros(nq, nk, dtype=torch.bool)
    amask[2] = True
    out = ops.attention_general(q.cuda(), k.cuda(), v.cuda(), G, H, hd, nq, nk, q_group_stride=1, q_inner=1, q_outer_stride=1,
                                k_group_stride=1, k_stride=1, attn_mask=amask.cuda()).cpu()
    assert torch.isnan(out[2]).all() and torch.isfinite(out[[0, 1, 3]]).all()
    with pytest.raises(ValueError):
        ops.attention_general(q.cuda(), k.cuda(), v.cuda(), G, H, hd, nq, nk, 1, 1, 1, 1, 1, attn_mask=amask[:, :3].cuda())
    from faceformer_amd.hip.lib import HipExtensionError
    with pytest.raises(HipExtensionError, match="LDS"):      # 4 rows of (head_dim + nk) floats must fit a CU's LDS
        big = torch.zeros(20000, 64, device="cuda")
        ops.attention_general(big[:1], big, big, 1, 1, 64, 1, 20000, 1, 1, 1, 1, 1)


# ---- LayerNorm -------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,E", [(1, 512), (37, 512), (1000, 512), (130, 128), (5, 64), (9, 1024), (3, 2048)])
def test_layernorm_pos(ops, rows, E):
    x = rnd(rows, E, seed=1, scale=3.0) + 0.5
    g, b = rnd(E, seed=2) + 1.0, rnd(E, seed=3)
    div, mod = 4, 7
    pos = rnd(mod, E, seed=4)
    y, yp = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-5, pos.cuda(), div, mod)
    ref = F.layer_norm(x.double(), (E,), g.double(), b.double(), 1e-5)
    idx = (torch.arange(rows) // div) % mod
    assert rel_err(y, ref) < 2e-6
    assert rel_err(yp, ref + pos.double()[idx]) < 2e-6
    y2, none = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), 1e-5)
    assert none is None and torch.equal(y2, y)


def test_add_pos_and_gather(ops):
    x, pos = rnd(50, 512, seed=1), rnd(10, 512, seed=2)
    out = ops.add_pos(x.cuda(), pos.cuda(), 5, 10)
    idx = (torch.arange(50) // 5) % 10
    assert torch.equal(out.cpu(), x + pos[idx])
    mem = rnd(3, 20, 512, seed=3)
    tok = torch.tensor([0, 19, 5, 7, 7, 1], dtype=torch.int32)
    rows = ops.gather_rows(mem.cuda(), tok.cuda(), seqs_per_group=2)
    want = torch.stack([mem[i // 2, int(t)] for i, t in enumerate(tok)])
    assert torch.equal(rows.cpu(), want)


# ---- GEMM -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 9, 10])
@pytest.mark.parametrize("M,N,K", [(256, 512, 512), (1, 512, 512), (77, 1536, 512), (333, 512, 1024),
                                   (520, 512, 100), (4100, 1024, 512), (64, 260, 512), (130, 96, 36),
                                   (7680, 512, 512), (8192, 1024, 512)])
def test_gemm_bias_act_residual(ops, M, N, K, tile):
    a, w, bias, res = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3), rnd(M, N, seed=4)
    # asymmetric operands: a transposed C-write or swapped operand cannot pass
    ref0 = a.double() @ w.double().t() + bias.double()
    out = ops.linear(a.cuda(), w.cuda(), bias.cuda(), tile=tile)
    assert rel_err(out, ref0) < 3e-6
    out = ops.linear(a.cuda(), w.cuda(), bias.cuda(), act=1, residual=res.cuda(), tile=tile)
    assert rel_err(out, torch.relu(ref0) + res.double()) < 3e-6
    out = ops.linear(a.cuda(), w.cuda(), None, tile=tile)
    assert rel_err(out, a.double() @ w.double().t()) < 3e-6


@pytest.mark.parametrize("M,N,K", [(256, 512, 512), (1, 512, 512), (77, 1536, 512), (333, 512, 1024), (4100, 1024, 512),
                                   (64, 260, 512), (130, 96, 64), (7680, 512, 512), (8448, 1536, 512), (9216, 512, 1024),
                                   (6400, 512, 512), (6400, 1536, 512), (6656, 512, 1024)])   # K-pieces beyond one per CU
@pytest.mark.parametrize("dma_tile", [11, 12])
def test_gemm_dma_kernel(ops, M, N, K, dma_tile):
    """Tile 11 / 12: the LDS-DMA kernel of the f32 family with 64 x 128 / 64 x 64 tiles (both operands by DMA, transposed accumulators,
    whole tiles + the hybrid remainder split): bias, ReLU, aliased residual, ragged edges in M and N, an output wider than N, determinism."""
    a, w, bias, res = rnd(M, K, seed=21), rnd(N, K, seed=22, scale=0.1), rnd(N, seed=23), rnd(M, N, seed=24)
    ref0 = a.double() @ w.double().t() + bias.double()
    out = ops.linear(a.cuda(), w.cuda(), bias.cuda(), tile=dma_tile)
    e11 = rel_err(out, ref0)
    assert e11 < 3e-6 and e11 < 2.0 * rel_err(ops.linear(a.cuda(), w.cuda(), bias.cuda(), tile=3), ref0) + 1e-7
    x = res.cuda()
    ops.linear(a.cuda(), w.cuda(), bias.cuda(), act=1, residual=x, out=x, tile=dma_tile)
    assert rel_err(x, torch.relu(ref0) + res.double()) < 3e-6
    y = res.cuda()
    ops.linear(a.cuda(), w.cuda(), bias.cuda(), act=1, residual=y, out=y, tile=dma_tile)
    assert torch.equal(x, y)
    wide = torch.full((M, N + 8), 7.0, device="cuda")
    ops.linear(a.cuda(), w.cuda(), None, out=wide[:, :N], tile=dma_tile)
    assert torch.equal(wide[:, N:], torch.full((M, 8), 7.0, device="cuda"))
    assert rel_err(wide[:, :N], a.double() @ w.double().t()) < 3e-6


@pytest.mark.parametrize("dma_tile", [11, 12])
def test_gemm_dma_kernel_split_a_and_identity(ops, dma_tile):
    E, M = 512, 5120
    yq, y, w, b = rnd(M, E, seed=1), rnd(M, E, seed=2), rnd(3 * E, E, seed=3, scale=0.05), rnd(3 * E, seed=4)
    out = ops.linear(yq.cuda(), w.cuda(), b.cuda(), x2=y.cuda(), n_split=2 * E, tile=dma_tile)
    ref = torch.cat([yq.double() @ w[: 2 * E].double().t(), y.double() @ w[2 * E:].double().t()], 1) + b.double()
    assert rel_err(out, ref) < 3e-6
    eye, wi = torch.eye(64), rnd(96, 64, seed=9)
    assert torch.equal(ops.linear(eye.cuda(), wi.cuda(), None, tile=dma_tile).cpu(), wi.t().contiguous())
    from faceformer_amd.hip import lib as L
    with pytest.raises(L.HipExtensionError):      # K = 100 is not a multiple of 32: not this kernel's
        ops.linear(rnd(64, 100).cuda(), rnd(32, 100).cuda(), None, tile=dma_tile)


def test_gemm_identity_layout(ops):
    """A = I against an asymmetric W: catches any row/col swap exactly (values are copied, not summed)."""
    K = 64
    eye = torch.eye(K)
    w = rnd(96, K, seed=9)
    for tile in (1, 2, 3, 4, 5, 6, 7, 9, 10):
        out = ops.linear(eye.cuda(), w.cuda(), None, tile=tile)
        assert torch.equal(out.cpu(), w.t().contiguous())


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 9, 10])
@pytest.mark.parametrize("M", [300, 5120])
def test_gemm_split_a_and_inplace_residual(ops, tile, M):
    E = 512
    yq, y, w, b = rnd(M, E, seed=1), rnd(M, E, seed=2), rnd(3 * E, E, seed=3, scale=0.05), rnd(3 * E, seed=4)
    out = ops.linear(yq.cuda(), w.cuda(), b.cuda(), x2=y.cuda(), n_split=2 * E, tile=tile)
    ref = torch.cat([yq.double() @ w[: 2 * E].double().t(), y.double() @ w[2 * E:].double().t()], 1) + b.double()
    assert rel_err(out, ref) < 3e-6
    # residual aliasing the output (x += o W^T + b), strided views as operands
    x = rnd(M, E, seed=5).cuda()
    x0 = x.clone()
    wo = rnd(E, E, seed=6, scale=0.05)
    ops.linear(out[:, :E], wo.cuda(), None, residual=x, out=x, tile=tile)
    ref2 = x0.double().cpu() + ref[:, :E] @ wo.double().t()
    assert rel_err(x, ref2) < 3e-6


@pytest.mark.parametrize("M", [2304, 2560, 3072, 3840, 2100, 3333, 4352, 5120, 6400, 6500])   # (from 4352: two / three whole tiles per CU)
@pytest.mark.parametrize("N,K", [(512, 512), (1024, 512), (1536, 512), (512, 1024), (260, 512)])
def test_gemm_hybrid_streamk_launch(ops, M, N, K):
    """Round 5: the hybrid launch of the 64x64 family (automatic choice, tile 0): whole tiles on the first 256 blocks, the units
    of the remaining tiles (less than one per CU) dealt to a second block per CU, cut tiles summed in block order.  The decode's
    own shapes at t = 9 ... 15 (256 t rows) and ragged ones; bias / ReLU / aliased residual; split A; twice the same launch
    bit-identical; the plain whole-tile kernel (tile 3) as a second reference for the summation order's error class."""
    a, w, bias, res = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3), rnd(M, N, seed=4)
    ref = torch.relu(a.double() @ w.double().t() + bias.double()) + res.double()
    x = res.cuda()
    ops.linear(a.cuda(), w.cuda(), bias.cuda(), act=1, residual=x, out=x)
    assert rel_err(x, ref) < 3e-6
    y = res.cuda()
    ops.linear(a.cuda(), w.cuda(), bias.cuda(), act=1, residual=y, out=y)
    assert torch.equal(x, y)
    z = ops.linear(a.cuda(), w.cuda(), bias.cuda(), act=1, residual=res.cuda(), tile=3)
    assert rel_err(z, ref) < 3e-6
    if N % 128 == 0:   # q | k from one operand, v from another (the decoder's q|k|v projection)
        a2 = rnd(M, K, seed=9)
        out = ops.linear(a.cuda(), w.cuda(), bias.cuda(), x2=a2.cuda(), n_split=N // 2)
        want = torch.cat([a.double() @ w.double()[: N // 2].t(), a2.double() @ w.double()[N // 2:].t()], dim=1) + bias.double()
        assert rel_err(out, want) < 3e-6


@pytest.mark.parametrize("min_units,two_per_cu", [(1, 1), (1, 1 << 30), (2, 2048), (3, 100), (7, 5000)])
@pytest.mark.parametrize("M,N,K,batch", [(256, 512, 512, 1), (64, 64, 512, 1), (2304, 512, 512, 1),
                                         (1100, 1536, 512, 1), (4400, 512, 1024, 1), (200, 260, 512, 3),
                                         (9216, 512, 128, 1)])
def test_gemm_streamk_splits(ops, M, N, K, batch, min_units, two_per_cu):
    """Stream-K launch shapes that cut tiles between 2..16 blocks (contribute / own+fix paths), with
    bias, ReLU and an aliased residual; twice the same launch must give bit-identical results."""
    ops.set_gemm_tuning(min_units, two_per_cu, 25, 1024)
    try:
        if batch == 1:
            a, w, bias, res = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3), rnd(M, N, seed=4)
            ref = torch.relu(a.double() @ w.double().t() + bias.double()) + res.double()
            x = res.cuda()
            ops.linear(a.cuda(), w.cuda(), bias.cuda(), act=1, residual=x, out=x, tile=6)
            assert rel_err(x, ref) < 3e-6
            y = res.cuda()
            ops.linear(a.cuda(), w.cuda(), bias.cuda(), act=1, residual=y, out=y, tile=6)
            assert torch.equal(x, y)
        else:
            a, w = rnd(batch, M, K, seed=5), rnd(batch, N, K, seed=6, scale=0.1)
            out = torch.empty(batch, M, N, device="cuda")
            L = ops._L
            ac, wc = a.cuda(), w.cuda()
            L.check(L.load().ff_gemm_f32_batched(ac.data_ptr(), K, None, 0, wc.data_ptr(), K, None, None, 0,
                                                 out.data_ptr(), N, M, N, K, 0, 6, batch, M * K, N * K, M * N,
                                                 torch.cuda.current_stream().cuda_stream), "ff_gemm_f32_batched")
            assert rel_err(out, a.double() @ w.double().transpose(1, 2)) < 3e-6
    finally:
        ops.set_gemm_tuning()


@pytest.mark.parametrize("M,N,K", [(256, 512, 512), (1, 512, 512), (77, 1536, 512), (333, 512, 1024), (64, 260, 512),
                                   (130, 96, 128), (1500, 1024, 512), (31, 33, 256)])
def test_gemm_small_rows_kernel(ops, M, N, K):
    """Unstaged split-K kernel for launches with few rows (tile 8; tile 0 picks it up to 1024 rows)."""
    a, w, bias, res = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3), rnd(M, N, seed=4)
    ref0 = a.double() @ w.double().t() + bias.double()
    for tile in (8, 0):  # tile 0 takes the small kernel for K >= 512 and at most 1024 rows
        out = ops.linear(a.cuda(), w.cuda(), bias.cuda(), tile=tile)
        assert rel_err(out, ref0) < 3e-6
        x = res.cuda()
        ops.linear(a.cuda(), w.cuda(), bias.cuda(), act=1, residual=x, out=x, tile=tile)
        assert rel_err(x, torch.relu(ref0) + res.double()) < 3e-6
    # split-A (q|k from one operand, v from another) as on the path
    if N % 96 == 0:
        a2 = rnd(M, K, seed=7)
        n_split = 2 * N // 3
        out = ops.linear(a.cuda(), w.cuda(), bias.cuda(), x2=a2.cuda(), n_split=n_split, tile=8)
        ref = torch.cat([a.double() @ w[:n_split].double().t(), a2.double() @ w[n_split:].double().t()], 1) + bias.double()
        assert rel_err(out, ref) < 3e-6


def test_gemm_small_rows_kernel_batched(ops):
    batch, M, N, K = 3, 200, 260, 512
    a, w = rnd(batch, M, K, seed=5), rnd(batch, N, K, seed=6, scale=0.1)
    out = torch.empty(batch, M, N, device="cuda")
    L = ops._L
    ac, wc = a.cuda(), w.cuda()
    L.check(L.load().ff_gemm_f32_batched(ac.data_ptr(), K, None, 0, wc.data_ptr(), K, None, None, 0,
                                         out.data_ptr(), N, M, N, K, 0, 8, batch, M * K, N * K, M * N,
                                         torch.cuda.current_stream().cuda_stream), "ff_gemm_f32_batched")
    assert rel_err(out, a.double() @ w.double().transpose(1, 2)) < 3e-6


# ---- GEMM on the bf16 / fp16 matrix cores (3 x bf16 split; round 6: 2 x fp16 split, half the products) ------------
@pytest.fixture(params=["bf16x3", "fp16x2"])
def split_kind(request):
    """Both split forms run every product test below: three bf16 terms (six products) and two fp16 terms (three products)."""
    return request.param


def test_split_weight_fp16x2_carries_22_bits(ops):
    """w = w1 + w2' 2^-11 with w1 = fp16(w), w2' = fp16((w - w1) 2^11): 22 mantissa bits wherever w1 is a normal fp16 number
    (|w| >= 6.1e-5), an ABSOLUTE error of at most 2^-36 below that; never worse than 2^-21 relative from 1e-4 to 6e4."""
    w = rnd(200, 512, seed=3) * torch.logspace(-4, 4, 512)
    w = w.clamp(-6.0e4, 6.0e4)
    planes = ops.split_weight(w.cuda(), "fp16x2")
    assert planes.dtype == torch.float16 and tuple(planes.shape) == (2, 32, 200, 16)
    back = ops.planes_to_matrix(planes).cpu()
    err = (back - w.double()).abs()
    assert float((err / w.double().abs().clamp_min(6.2e-5)).max()) < 2.0 ** -21
    tiny = rnd(64, 64, seed=4) * 1e-6
    back = ops.planes_to_matrix(ops.split_weight(tiny.cuda(), "fp16x2")).cpu()
    assert float((back - tiny.double()).abs().max()) <= 2.0 ** -35


def test_split_weight_is_exact(ops):
    """The three bf16 planes must reproduce the fp32 weight to 2^-25 relative (exact split)."""
    w = rnd(200, 512, seed=3) * torch.logspace(-6, 3, 512)  # nine decades of magnitudes
    planes = ops.split_weight(w.cuda())
    back = ops.planes_to_matrix(planes).cpu()
    assert float(((back - w.double()).abs() / w.double().abs().clamp_min(1e-30)).max()) < 2.0 ** -24


@pytest.mark.parametrize("M,N,K", [(256, 512, 512), (1, 512, 512), (77, 1536, 512), (333, 512, 1024),
                                   (4100, 1024, 512), (64, 260, 512), (130, 96, 64), (2304, 512, 512),
                                   (9216, 1536, 512)])
def test_gemm_x3_matches_fp64_like_fp32(ops, split_kind, M, N, K):
    """3 x bf16 product vs fp64: bias, ReLU and aliased residual; its error must not exceed the f32-MFMA
    kernel's by more than a small factor (both are 'fp32 dot product' accurate)."""
    a, w, bias, res = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3), rnd(M, N, seed=4)
    planes = ops.split_weight(w.cuda(), split_kind)
    ref0 = a.double() @ w.double().t() + bias.double()
    out = ops.linear_x3(a.cuda(), planes, bias.cuda())
    e_x3 = rel_err(out, ref0)
    e_f32 = rel_err(ops.linear(a.cuda(), w.cuda(), bias.cuda()), ref0)
    assert e_x3 < 3e-6 and e_x3 < 2.0 * e_f32 + 1e-7
    x = res.cuda()
    ops.linear_x3(a.cuda(), planes, bias.cuda(), act=1, residual=x, out=x)
    assert rel_err(x, torch.relu(ref0) + res.double()) < 3e-6
    y = res.cuda()
    ops.linear_x3(a.cuda(), planes, bias.cuda(), act=1, residual=y, out=y)
    assert torch.equal(x, y)  # deterministic


@pytest.fixture(params=[0, 1, 2], ids=lambda p: "shape%d" % p)
def x3_tuning(request, ops):
    """Every launch shape of the 3 x bf16 kernel (0 = the cost model's choice, 1 whole tiles, 2 equal K-unit ranges)."""
    ops.set_x3_tuning(request.param)
    yield request.param
    ops.set_x3_tuning(0)


@pytest.mark.parametrize("M,N,K", [(1, 512, 512), (77, 1536, 512), (333, 512, 1024), (64, 260, 512), (130, 96, 64),
                                   (2304, 512, 512), (5000, 1024, 512), (9216, 1536, 512), (4608, 512, 1024),
                                   # more than half a round of tiles left over: K-pieces beyond one per CU (no-wait exchange)
                                   (6400, 512, 512), (6400, 1536, 512), (6400, 512, 1024)])
def test_gemm_x3_every_tile_and_launch_shape(ops, split_kind, x3_tuning, M, N, K):
    """Whole tiles and equal K-unit ranges (cut tiles exchanged between blocks): same result up to the summation order,
    deterministic, ragged edges in M and N."""
    a, w, bias, res = rnd(M, K, seed=11), rnd(N, K, seed=12, scale=0.1), rnd(N, seed=13), rnd(M, N, seed=14)
    planes = ops.split_weight(w.cuda(), split_kind)
    ref = torch.relu(a.double() @ w.double().t() + bias.double()) + res.double()
    x = res.cuda()
    ops.linear_x3(a.cuda(), planes, bias.cuda(), act=1, residual=x, out=x)
    assert rel_err(x, ref) < 3e-6
    y = res.cuda()
    ops.linear_x3(a.cuda(), planes, bias.cuda(), act=1, residual=y, out=y)
    assert torch.equal(x, y)
    # a padded output (ldc > N) keeps the columns behind N untouched
    wide = torch.full((M, N + 8), 7.0, device="cuda")
    ops.linear_x3(a.cuda(), planes, bias.cuda(), out=wide[:, :N])
    assert torch.equal(wide[:, N:], torch.full((M, 8), 7.0, device="cuda"))
    assert rel_err(wide[:, :N], a.double() @ w.double().t() + bias.double()) < 3e-6


@pytest.mark.parametrize("M,N,K", [(37, 512, 512), (300, 512, 1024), (1300, 512, 512), (5000, 512, 512), (640, 128, 256),
                                   (9216, 512, 1024), (6400, 512, 512), (6656, 512, 1024)])
def test_gemm_x3_emits_layernorm_segment_statistics(hip_lib, ops, split_kind, x3_tuning, M, N, K):
    """ff_gemm_x3_ln, producer side: C = A W^T + b + residual plus (mean, M2) per row and 32-column segment of the stored C."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    res = (3.0 + 2.0 * torch.randn(M, N, generator=g)).cuda()          # non-zero mean: the cancellation trap
    out, stats = ops.linear_x3_ln(A, ops.split_weight(W, split_kind), b, residual=res, want_stats=True)
    ref = A.double() @ W.double().t() + b.double() + res.double()
    assert (out.double() - ref).abs().max() < 2e-5 * ref.abs().max()
    want = _seg_stats(out.double())                                    # statistics of what was actually stored
    assert not torch.isnan(stats).any()
    assert (stats[..., 0].double() - want[..., 0]).abs().max() < 1e-5
    assert ((stats[..., 1].double() - want[..., 1]).abs() / want[..., 1].clamp_min(1e-6)).max() < 1e-5


@pytest.mark.parametrize("in_epilogue", [False, True], ids=["normalise_first", "normalise_in_epilogue"])
@pytest.mark.parametrize("M,N,div", [(37, 1536, 5), (300, 512, 7), (1300, 1536, 64), (5000, 1024, 256), (9216, 1536, 256),
                                     (4352, 512, 256), (6400, 1536, 256), (6400, 512, 256)])
def test_gemm_x3_consumes_layernorm_statistics_with_folded_weights(hip_lib, ops, split_kind, x3_tuning, M, N, div, in_epilogue):
    """ff_gemm_x3_ln, consumer side: act((LN(x) + pos[row // div]) W^T + b) from raw x, its segment statistics, the planes of
    the folded weight, the folded bias and the pos W^T table -- against the unfused arithmetic in float64; rows normalised
    before the product, and the plain product with rstd (x W'^T - mean colsum(W')) in the epilogue (rows with |mean| / sigma
    around 0.75 here: the second form's bound carries that factor)."""
    K = 512
    g = torch.Generator().manual_seed(M * 3 + N + K)
    x = (1.5 + 2.0 * torch.randn(M, K, generator=g)) * (1.0 + torch.rand(M, 1, generator=g))   # row-dependent scale / mean
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    gamma, beta = 1.0 + 0.3 * torch.randn(K, generator=g), 0.3 * torch.randn(K, generator=g)
    npos = (M + div - 1) // div
    pos = torch.randn(npos, K, generator=g)
    pos_cols = 1024 if N >= 1536 else (N // 2 if N >= 1024 else N)
    xd, Wd = x.cuda(), W.cuda()
    Wf, bf, P = ops.fold_layernorm_linear(Wd, b.cuda(), gamma.cuda(), beta.cuda(), pos.cuda(), pos_cols)
    stats = _seg_stats(xd.double()).float().contiguous()
    colsum = Wf.double().sum(dim=1).float().contiguous() if in_epilogue else None
    out = ops.linear_x3_ln(xd, ops.split_weight(Wf, split_kind), bf, act=1, stats_in=stats, row_table=P, row_div=div, row_cols=pos_cols,
                           colsum=colsum)
    x64 = x.double()
    ln = torch.nn.functional.layer_norm(x64, (K,), gamma.double(), beta.double(), 1e-5)
    rows = torch.arange(M) // div
    addp = torch.zeros(M, N, dtype=torch.float64)
    addp[:, :pos_cols] = pos.double()[rows] @ W.double()[:pos_cols].t()
    ref = torch.relu(ln @ W.double().t() + b.double() + addp)
    err = (out.cpu().double() - ref).abs().max()
    assert err < 3e-5 * max(1.0, ref.abs().max()), err
    # the f32 family's folded form on the same operands: the two must agree like two fp32 evaluations
    out32 = ops.linear_ln(xd, Wf, bf, act=1, stats_in=stats, row_table=P, row_div=div, row_cols=pos_cols)
    assert (out - out32).abs().max() < 3e-5 * max(1.0, ref.abs().max())


@pytest.mark.parametrize("scale_a,scale_w", [(1.0, 0.05), (900.0, 0.22), (1.0, 1e-4), (40.0, 0.05), (1.2e4, 0.05), (1e-3, 0.05)])
def test_gemm_x2h_error_is_fp32_class_over_operand_scales(ops, scale_a, scale_w):
    """The 2 x fp16 product against fp64 at the operand scales of profiles/r06/fp16_split_error_table.txt (LayerNorm output x
    xavier weight, gain-4 residual stream, tiny weights, ReLU hidden rows, rows close to fp16's range, small rows): the error
    relative to |A| |W|^T stays within 1.5x the f32-MFMA kernel's -- an fp32 dot product's -- everywhere."""
    M, N, K = 512, 512, 512
    a = rnd(M, K, seed=21) * scale_a
    if scale_a == 40.0:
        a = torch.relu(a)
    w = (torch.rand(N, K, generator=torch.Generator().manual_seed(22)) * 2 - 1) * scale_w
    ref = a.double() @ w.double().t()
    den = a.double().abs() @ w.double().abs().t()
    got = ops.linear_x3(a.cuda(), ops.split_weight(w.cuda(), "fp16x2"), None).cpu().double()
    f32 = ops.linear(a.cuda(), w.cuda(), None).cpu().double()
    e_h, e_f = ((got - ref).abs() / den).max(), ((f32 - ref).abs() / den).max()
    r_h, r_f = (((got - ref) / den) ** 2).mean().sqrt(), (((f32 - ref) / den) ** 2).mean().sqrt()
    assert torch.isfinite(got).all()
    assert e_h < 1.5 * e_f + 2.0 ** -24 and r_h < 1.5 * r_f + 2.0 ** -26, (float(e_h), float(e_f), float(r_h), float(r_f))


def test_gemm_x2h_epilogue_form_takes_raw_rows_beyond_fp16_range(hip_lib, ops):
    """The LayerNorm-in-the-epilogue form multiplies the RAW rows; the fp16 kernel feeds them at 2^-6 (exact), so rows of
    magnitude 2e5 -- far outside fp16's 65504 -- come out finite and as accurate as the normalise-first form."""
    M, N, K = 300, 512, 512
    g = torch.Generator().manual_seed(5)
    x = (2.0e5 * torch.randn(M, K, generator=g))
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    gamma, beta = 1.0 + 0.3 * torch.randn(K, generator=g), 0.3 * torch.randn(K, generator=g)
    xd = x.cuda()
    Wf, bf, _ = ops.fold_layernorm_linear(W.cuda(), b.cuda(), gamma.cuda(), beta.cuda(), None, 0)
    stats = _seg_stats(xd.double()).float().contiguous()
    colsum = Wf.double().sum(dim=1).float().contiguous()
    planes = ops.split_weight(Wf, "fp16x2")
    ref = torch.nn.functional.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-5) @ W.double().t() + b.double()
    for cs in (colsum, None):
        out = ops.linear_x3_ln(xd, planes, bf, stats_in=stats, colsum=cs).cpu().double()
        assert torch.isfinite(out).all()
        assert (out - ref).abs().max() < 3e-5 * max(1.0, ref.abs().max())


def test_gemm_x3_ln_argument_validation(hip_lib, ops):
    from faceformer_amd.hip import lib as L
    x = torch.randn(64, 256).cuda()
    planes = ops.split_weight(torch.randn(512, 256).cuda())
    with pytest.raises(L.HipExtensionError):       # the normalising form is built for K = 512
        ops.linear_x3_ln(x, planes, stats_in=torch.zeros(64, 8, 2).cuda())
    with pytest.raises(L.HipExtensionError):       # a row table needs statistics
        ops.linear_x3_ln(x, planes, row_table=torch.zeros(4, 512).cuda(), row_div=16, row_cols=512)
    with pytest.raises(L.HipExtensionError):
        ops.set_x3_tuning(3)


def test_gemm_x3_split_a(ops, split_kind):
    M, E = 300, 512
    yq, y, w, b = rnd(M, E, seed=1), rnd(M, E, seed=2), rnd(3 * E, E, seed=3, scale=0.05), rnd(3 * E, seed=4)
    out = ops.linear_x3(yq.cuda(), ops.split_weight(w.cuda(), split_kind), b.cuda(), x2=y.cuda(), n_split=2 * E)
    ref = torch.cat([yq.double() @ w[: 2 * E].double().t(), y.double() @ w[2 * E:].double().t()], 1) + b.double()
    assert rel_err(out, ref) < 3e-6


def test_gemm_x3_identity_layout(ops, split_kind):
    K = 64
    eye = torch.eye(K)
    w = rnd(96, K, seed=9)
    out = ops.linear_x3(eye.cuda(), ops.split_weight(w.cuda(), split_kind), None)
    assert float((out.cpu() - w.t()).abs().max()) < (1e-7 if split_kind == "bf16x3" else 2e-6)   # (24 / 22 bits of |w| <= 4.5)


# ---- attention --------------------------------------------------------------------------------------
def ref_attention(q, k, v, mask=None, causal=False):
    """q [G,H,nq,64] k,v [G,H,nk,64] fp64; mask [G,nk] bool"""
    s = (q * 0.125) @ k.transpose(-1, -2)
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    if causal:
        nq, nk = s.shape[-2:]
        s = s.masked_fill(torch.triu(torch.ones(nq, nk, dtype=torch.bool), 1), float("-inf"))
    return torch.softmax(s, -1) @ v


@pytest.mark.parametrize("G,H,nq,nk", [(2, 2, 33, 33), (1, 8, 260, 260), (3, 2, 5, 100), (4, 1, 1, 1),
                                       (2, 8, 140, 70), (1, 2, 64, 64), (1, 1, 65, 129),
                                       (600, 2, 35, 36), (530, 2, 36, 64), (515, 2, 68, 40), (520, 2, 40, 40), (513, 2, 37, 33),   # short-tail paths
                                       # more (group, head) pairs than workgroups AND query tiles cut between the waves of a block:
                                       # the K/V-resident kernel's first partial records (global scratch) with several pairs per block
                                       (130, 2, 300, 100), (40, 8, 200, 260)])
def test_attention_group_major_with_mask(ops, attn_algo, G, H, nq, nk):
    """Encoder-style layout: rows = g*len + i; padding mask + kv_len."""
    E = H * 64
    q, k, v = rnd(G * nq, E, seed=1), rnd(G * nk, E, seed=2), rnd(G * nk, E, seed=3)
    mask = torch.zeros(G, nk, dtype=torch.bool)
    kv_len = torch.full((G,), nk, dtype=torch.int32)
    for g in range(G):
        cut = max(1, nk - 3 * g - (nk // 4))
        mask[g, cut:] = True
        kv_len[g] = cut
        if cut > 2:
            mask[g, 1] = True  # a hole inside the valid range
    out = ops.attention(q.cuda(), k.cuda(), v.cuda(), G, H, nq, nk, q_group_stride=nq, q_inner=nq,
                        q_outer_stride=0, k_group_stride=nk, k_stride=1, kv_len=kv_len.cuda(),
                        key_mask=mask.to(torch.uint8).cuda())
    qd = q.double().view(G, nq, H, 64).transpose(1, 2)
    kd = k.double().view(G, nk, H, 64).transpose(1, 2)
    vd = v.double().view(G, nk, H, 64).transpose(1, 2)
    ref = ref_attention(qd, kd, vd, mask).transpose(1, 2).reshape(G * nq, E)
    assert rel_err(out, ref) < 5e-6


@pytest.mark.parametrize("t,B,H,causal", [(1, 5, 2, False), (7, 40, 8, False), (36, 24, 8, False),
                                          (70, 3, 2, False), (9, 4, 2, True), (258, 2, 8, False),
                                          (33, 140, 8, False), (34, 130, 8, False), (36, 256, 8, False),
                                          (37, 129, 8, False), (35, 140, 8, True), (40, 129, 8, False), (41, 129, 8, False),
                                          (38, 129, 8, True)])   # keys / queries 1..9 past 32
def test_attention_position_major_self(ops, attn_algo, t, B, H, causal):
    """Decoder self-attention layout: rows = j*B + b, packed q|k|v buffer (ld = 3E)."""
    E = H * 64
    qkv = rnd(t * B, 3 * E, seed=5)
    dq = qkv.cuda()
    out = ops.attention(dq[:, :E], dq[:, E:2 * E], dq[:, 2 * E:], B, H, t, t, q_group_stride=1, q_inner=1,
                        q_outer_stride=B, k_group_stride=1, k_stride=B, causal=causal)
    x = qkv.double().view(t, B, 3, H, 64).permute(2, 1, 3, 0, 4)  # [3,B,H,t,64]
    ref = ref_attention(x[0], x[1], x[2], None, causal)           # [B,H,t,64]
    ref = ref.permute(2, 0, 1, 3).reshape(t * B, E)
    assert rel_err(out, ref) < 5e-6


@pytest.mark.parametrize("t,F,W,S", [(1, 3, 2, 30), (5, 7, 3, 50), (12, 40, 1, 44), (36, 33, 2, 260),
                                     # config-E key counts (S > 288: the block-shared kernel) with LONG prefixes: t*F query rows
                                     # per wireframe for t up to 37, F = the 512 anchors of the wide wireframe / 65 = the compact
                                     # width of a 64-edge one, and a 1028-key wireframe
                                     (36, 512, 1, 516), (37, 65, 2, 516), (33, 129, 2, 1028), (36, 40, 1, 1028), (9, 513, 1, 516)])
def test_attention_cross_shared_kv(ops, attn_algo, t, F, W, S):
    """Decoder cross-attention: F sequences of a wireframe share its K/V; queries position-major."""
    H, E = 8, 512
    B = W * F
    q = rnd(t * B, E, seed=1)
    kv = rnd(W * S, 2 * E, seed=2)
    mask = torch.zeros(W, S, dtype=torch.bool)
    kv_len = torch.full((W,), S, dtype=torch.int32)
    for w in range(W):
        cut = S - 5 * w - 3
        mask[w, cut:] = True
        kv_len[w] = cut
    dkv = kv.cuda()
    out = ops.attention(q.cuda(), dkv[:, :E], dkv[:, E:], W, H, F * t, S, q_group_stride=F, q_inner=F,
                        q_outer_stride=B, k_group_stride=S, k_stride=1, kv_len=kv_len.cuda(),
                        key_mask=mask.to(torch.uint8).cuda())
    qd = q.double().view(t, W, F, H, 64).permute(1, 3, 0, 2, 4).reshape(W, H, t * F, 64)
    kd = kv[:, :E].double().view(W, S, H, 64).transpose(1, 2)
    vd = kv[:, E:].double().view(W, S, H, 64).transpose(1, 2)
    ref = ref_attention(qd, kd, vd, mask)                          # [W,H,t*F,64]
    ref = ref.view(W, H, t, F, 64).permute(2, 0, 3, 1, 4).reshape(t * B, E)
    assert rel_err(out, ref) < 5e-6


def test_attention_online_softmax_rescale_branch(ops, attn_algo):
    """Spike one late key so the running max jumps in a later key tile (forces the rescale path)."""
    H, E, nq, nk = 1, 64, 40, 200
    q, k, v = rnd(nq, E, seed=1), rnd(nk, E, seed=2), rnd(nk, E, seed=3)
    k[150] = q[7] * 40.0   # huge score for query 7 at key 150 (third chunk)
    k[3] = q[20] * 25.0    # and an early spike for another query
    out = ops.attention(q.cuda(), k.cuda(), v.cuda(), 1, H, nq, nk, q_group_stride=nq, q_inner=nq,
                        q_outer_stride=0, k_group_stride=nk, k_stride=1)
    ref = ref_attention(q.double()[None, None], k.double()[None, None], v.double()[None, None])[0, 0]
    assert rel_err(out, ref) < 5e-6


# ---- pointer head -----------------------------------------------------------------------------------
@pytest.mark.parametrize("W,F,S,E", [(1, 256, 260, 512), (3, 7, 44, 512), (2, 5, 70, 128), (1, 1, 5, 64)])
def test_pointer_argmax(ops, W, F, S, E):
    B = W * F
    p, mem = rnd(B, E, seed=1), rnd(W, S, E, seed=2)
    mask = torch.zeros(W, S, dtype=torch.bool)
    kv_len = torch.full((W,), S, dtype=torch.int32)
    for w in range(W):
        cut = S - 2 * w - 1
        mask[w, cut:] = True
        kv_len[w] = cut
    counters = torch.zeros(2, dtype=torch.int32).cuda()
    res = ops.pointer_argmax(p.cuda(), mem.cuda(), mask.to(torch.uint8).cuda(), kv_len.cuda(),
                             seqs_per_group=F, want_logits=True, want_rows=True, counters=counters,
                             ge_bound=4, eq_value=3)
    fill = torch.finfo(torch.float32).min
    logit = torch.einsum("be,bse->bs", p.double(), mem.double()[torch.arange(B) // F])
    logit = logit.masked_fill(mask[torch.arange(B) // F], fill)
    got = res["logits"].cpu().double()
    live = logit > fill
    assert torch.equal(got[~live], logit[~live])
    assert float((got[live] - logit[live]).abs().max()) < 1e-4 * float(logit[live].abs().max())
    # argmax must be exactly torch's on the kernel's own logits (first index on ties)
    assert torch.equal(res["next"].cpu().long(), torch.argmax(res["logits"].cpu(), dim=1))
    top2 = torch.sort(res["logits"].cpu(), dim=1, descending=True).values
    assert torch.equal(res["best"].cpu(), top2[:, 0])
    if S > 1:
        assert torch.equal(res["second"].cpu(), top2[:, 1])
    nxt = res["next"].cpu().long()
    assert torch.equal(res["rows"].cpu(), mem[torch.arange(B) // F, nxt])
    assert counters.cpu().tolist() == [int((nxt >= 4).sum()), int((nxt == 3).sum())]
    # the streaming path (no logits buffer) must agree with the GEMM path
    counters.zero_()
    res2 = ops.pointer_argmax(p.cuda(), mem.cuda(), mask.to(torch.uint8).cuda(), kv_len.cuda(),
                              seqs_per_group=F, want_rows=True, counters=counters, ge_bound=4, eq_value=3)
    decisive = (top2[:, 0] - top2[:, 1]) > 1e-3 if S > 1 else torch.ones(B, dtype=torch.bool)
    assert torch.equal(res2["next"].cpu()[decisive], res["next"].cpu()[decisive])
    assert float((res2["best"].cpu() - res["best"].cpu()).abs().max()) < 1e-4 * float(top2[:, 0].abs().max())
    if S > 1:
        assert float((res2["second"].cpu() - res["second"].cpu()).abs().max()) < 1e-4 * float(top2[:, 0].abs().max())


def test_pointer_ties_and_all_masked(ops):
    E, S = 64, 9
    mem = torch.zeros(1, S, E)
    mem[0, 2] = 1.0
    mem[0, 5] = 1.0          # exact tie between keys 2 and 5 -> lowest index wins
    p = torch.ones(2, E)
    res = ops.pointer_argmax(p.cuda(), mem.cuda(), seqs_per_group=2)
    assert res["next"].cpu().tolist() == [2, 2]
    allmask = torch.ones(1, S, dtype=torch.uint8)
    res = ops.pointer_argmax(p.cuda(), mem.cuda(), allmask.cuda(), seqs_per_group=2, want_logits=True)
    assert res["next"].cpu().tolist() == [0, 0]   # torch.argmax of a constant row is 0
    assert float(res["best"][0]) == torch.finfo(torch.float32).min
    extra = torch.zeros(2, S, dtype=torch.uint8)
    extra[1, 2] = 1          # per-sequence extra mask removes key 2 for sequence 1 only
    res = ops.pointer_argmax(p.cuda(), mem.cuda(), extra_mask=extra.cuda(), seqs_per_group=2)
    assert res["next"].cpu().tolist() == [2, 5]


# ---- LayerNorm folded into the neighbouring projections (ff_gemm_f32_ln) -------------------------------------------
def _seg_stats(x64):
    """[M, N] float64 -> [M, N/32, 2] (mean, M2) per 32-column segment."""
    M, N = x64.shape
    seg = x64.view(M, N // 32, 32)
    mean = seg.mean(dim=2)
    m2 = ((seg - mean[..., None]) ** 2).sum(dim=2)
    return torch.stack([mean, m2], dim=2)


@pytest.mark.parametrize("tile", [0, 3, 6, 8, 11, 12])
@pytest.mark.parametrize("M,N,K", [(37, 512, 512), (300, 512, 1024), (1300, 512, 512), (5000, 512, 512), (640, 128, 256),
                                   (2304, 512, 512), (4864, 512, 1024), (6400, 512, 512)])   # (hybrid launches, round 5)
def test_gemm_emits_layernorm_segment_statistics(hip_lib, ops, M, N, K, tile):
    """Producer side: C = A W^T + b + residual plus, per row and 32-column segment, (mean, M2) of the stored C."""
    if tile == 8 and M > 1024:
        pytest.skip("small-M kernel")
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    res = (3.0 + 2.0 * torch.randn(M, N, generator=g)).cuda()          # non-zero mean: the cancellation trap
    out, stats = ops.linear_ln(A, W, b, residual=res, want_stats=True, tile=tile)
    ref = A.double() @ W.double().t() + b.double() + res.double()
    assert (out.double() - ref).abs().max() < 2e-5 * ref.abs().max()
    want = _seg_stats(out.double())                                    # statistics of what was actually stored
    assert not torch.isnan(stats).any()
    assert (stats[..., 0].double() - want[..., 0]).abs().max() < 1e-5
    assert ((stats[..., 1].double() - want[..., 1]).abs() / want[..., 1].clamp_min(1e-6)).max() < 1e-5


@pytest.mark.parametrize("tile", [0, 3, 6, 8, 11, 12])
@pytest.mark.parametrize("M,N,K,div", [(37, 1536, 512, 5), (300, 512, 512, 7), (1300, 1536, 512, 64), (5000, 1024, 512, 256),
                                       (256, 512, 128, 16), (9216, 1536, 512, 256),
                                       # round 5, the hybrid launch of the 64x64 family with this consumer form: unit ranges of
                                       # 1 / 3 / 2 units behind one, two and three whole tiles per CU (a one-unit OWNED piece in
                                       # front of whole tiles once normalised two slices with the previous tile's statistics)
                                       (2304, 512, 512, 256), (4864, 512, 512, 256), (6400, 512, 512, 256), (5632, 512, 512, 256),
                                       (2304, 1024, 512, 256), (2560, 1536, 512, 256)])
def test_gemm_consumes_layernorm_statistics_with_folded_weights(hip_lib, ops, M, N, K, div, tile):
    """Consumer side: act((LN(x) + pos[row // div]) W^T + b) from raw x, its segment statistics, the folded
    weight / bias and the pos W^T table -- against the unfused arithmetic in float64."""
    if tile == 8 and M > 1024:
        pytest.skip("small-M kernel")
    if tile in (11, 12) and K != 512:
        pytest.skip("the LDS-DMA kernel's normalising form is built for K = 512")
    g = torch.Generator().manual_seed(M * 3 + N + K)
    x = (1.5 + 2.0 * torch.randn(M, K, generator=g)) * (1.0 + torch.rand(M, 1, generator=g))   # row-dependent scale / mean
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    gamma, beta = 1.0 + 0.3 * torch.randn(K, generator=g), 0.3 * torch.randn(K, generator=g)
    npos = (M + div - 1) // div
    pos = torch.randn(npos, K, generator=g)
    pos_cols = N // 2 if N >= 1024 else N
    xd, Wd = x.cuda(), W.cuda()
    Wf, bf, P = ops.fold_layernorm_linear(Wd, b.cuda(), gamma.cuda(), beta.cuda(), pos.cuda(), pos_cols)
    stats = _seg_stats(xd.double()).float().contiguous()
    out = ops.linear_ln(xd, Wf, bf, act=1, stats_in=stats, row_table=P, row_div=div, row_cols=pos_cols, tile=tile)
    x64 = x.double()
    ln = torch.nn.functional.layer_norm(x64, (K,), gamma.double(), beta.double(), 1e-5)
    rows = torch.arange(M) // div
    addp = torch.zeros(M, N, dtype=torch.float64)
    addp[:, :pos_cols] = pos.double()[rows] @ W.double()[:pos_cols].t()
    ref = torch.relu(ln @ W.double().t() + b.double() + addp)
    err = (out.cpu().double() - ref).abs().max()
    assert err < 3e-5 * max(1.0, ref.abs().max()), err


def test_gemm_ln_argument_validation(hip_lib, ops):
    from faceformer_amd.hip import lib as L
    x = torch.randn(64, 512).cuda()
    W = torch.randn(512, 512).cuda()
    with pytest.raises(L.HipExtensionError):       # statistics must describe whole rows: nseg * 32 == K
        ops.linear_ln(x, W, stats_in=torch.zeros(64, 8, 2).cuda(), tile=0)
    with pytest.raises(L.HipExtensionError):       # a row table needs statistics
        ops.linear_ln(x, W, row_table=torch.zeros(4, 512).cuda(), row_div=16, row_cols=512)
    with pytest.raises(L.HipExtensionError):       # not on the generic kernels
        ops.linear_ln(x, W, want_stats=True, tile=1)


def test_prepare_mask_equals_process_masks_and_key_lengths(hip_lib, ops):
    """ff_prepare_mask = reference process_masks (four never-masked special-token columns in front, model.py:61-69) + the key
    length per wireframe, in one launch: against the torch formulation it replaces, incl. an all-masked row and a hole."""
    from faceformer_amd.hip.engine import _kv_len_from_mask
    g = torch.Generator().manual_seed(5)
    for N, L_ in ((1, 256), (7, 40), (3, 1028), (130, 16)):
        m = torch.rand(N, L_, generator=g) < 0.4
        keep = torch.randint(0, L_ + 1, (N,), generator=g)
        m |= torch.arange(L_)[None, :] >= keep[:, None]          # padded tails, plus random holes in front of them
        m[0] = True                                               # every edge masked: the special tokens remain
        md = m.cuda()
        S = L_ + 4
        mask_u8 = torch.empty(N, S, dtype=torch.uint8, device="cuda")
        kv = torch.empty(N, dtype=torch.int32, device="cuda")
        ops._L.check(hip_lib.ff_prepare_mask(md.data_ptr(), N, L_, 4, mask_u8.data_ptr(), kv.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream), "ff_prepare_mask")
        want = torch.cat([torch.zeros(N, 4, dtype=torch.bool), m], dim=1).to(torch.uint8)
        assert torch.equal(mask_u8.cpu(), want)
        assert torch.equal(kv.cpu(), _kv_len_from_mask(want))
        assert int(kv[0]) == 4
