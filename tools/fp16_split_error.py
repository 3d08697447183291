"""Op-level error table for a 2-term fp16 split product (round-5 review item 4), emulated in numpy on the CPU.

Question: can  C = A W^T  in fp32 be evaluated on the fp16 matrix cores as THREE products a1 w1 + a1 w2 + a2 w1
(a = a1 + a2, both fp16: 22 mantissa bits) instead of the six bf16 products of gemm_x3_kernel -- and stay "fp32-class",
i.e. as close to the fp64 result as an ordinary fp32 dot product (what tests/test_hip_ops.py::test_gemm_x3_* require)?

Emulation: every product of two fp16 (or bf16) terms is exact in fp32; the matrix core adds products into an fp32
accumulator one after the other, rounding after each (measured for v_mfma_f32_32x32x16_bf16 in round 4), so a K-long
chain is K sequential fp32 additions per partial-product class.  Forms:
  f32      : plain fp32 chain (v_mfma_f32_32x32x2_f32)
  bf16x3   : the shipped split -- x1y1 in its own accumulator, the five small products in a second one
  fp16x2   : a1 w1 in its own accumulator, a1 w2 + a2 w1 in a second one; the second terms are stored scaled by 2^11
             (a2' = (a - a1) 2^11: keeps them out of fp16's subnormal range), the second accumulator is scaled back once
  fp16x2u  : the same without the 2^11 scaling (second terms as they are)
  fp16x2+  : four products (a2 w2 too)
Operand scalings: 'unit' (|a| ~ 1, |w| ~ 0.05: LayerNorm output x xavier weight), 'big' (|a| up to ~3e3: FFN hidden / residual
stream of the gain-4 goldens), 'tiny' (|w| ~ 1e-4), 'mixed' (per-row scales 1e-3 ... 1e3) -- with and without power-of-two ROW
scaling of A and W to [1, 2) maxima (exact, undone in the epilogue).

Error measure (as in the op tests): max over the matrix of |C - C64| / (|A| |W|^T)_ij  (error relative to the sum of magnitudes),
in units of 2^-24, and its RMS.
"""
import sys

import numpy as np


def to_bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split_bf16x3(x):
    x1 = to_bf16(x); r = (x - x1).astype(np.float32)
    x2 = to_bf16(r); r2 = (r - x2).astype(np.float32)
    return x1, x2, to_bf16(r2)


def split_fp16(x, scale2):
    x1 = x.astype(np.float16).astype(np.float32)
    r = (x - x1).astype(np.float32) * np.float32(scale2)
    return x1, r.astype(np.float16).astype(np.float32)


def chain(acc, terms):
    """acc [M,N] fp32; terms: list of (a [M,K], w [N,K]) -- per k add the products of every pair in order (fp32 rounding each)."""
    K = terms[0][0].shape[1]
    for k in range(K):
        for a, w in terms:
            acc = (acc + np.outer(a[:, k], w[:, k]).astype(np.float32)).astype(np.float32)
    return acc


def forms(A, W):
    M, N = A.shape[0], W.shape[0]
    z = lambda: np.zeros((M, N), np.float32)
    out = {}
    out["f32"] = chain(z(), [(A, W)])
    a1, a2, a3 = split_bf16x3(A); w1, w2, w3 = split_bf16x3(W)
    out["bf16x3"] = (chain(z(), [(a1, w1)]) + chain(z(), [(a1, w2), (a2, w1), (a1, w3), (a2, w2), (a3, w1)])).astype(np.float32)
    S = 2048.0
    a1, a2 = split_fp16(A, S); w1, w2 = split_fp16(W, S)
    out["fp16x2"] = (chain(z(), [(a1, w1)]) + chain(z(), [(a1, w2), (a2, w1)]) * np.float32(1 / S)).astype(np.float32)
    out["fp16x2+"] = (chain(z(), [(a1, w1)]) + chain(z(), [(a1, w2), (a2, w1)]) * np.float32(1 / S) +
                      chain(z(), [(a2, w2)]) * np.float32(1 / (S * S))).astype(np.float32)
    a1, a2 = split_fp16(A, 1.0); w1, w2 = split_fp16(W, 1.0)
    out["fp16x2u"] = (chain(z(), [(a1, w1)]) + chain(z(), [(a1, w2), (a2, w1)])).astype(np.float32)
    return out


def row_pow2(X):
    m = np.abs(X).max(axis=1, keepdims=True)
    e = np.floor(np.log2(np.where(m > 0, m, 1.0)))
    s = np.exp2(-e).astype(np.float32)          # row maxima -> [1, 2)
    return (X * s).astype(np.float32), (1.0 / s).astype(np.float32)


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    M = N = 96
    rng = np.random.default_rng(0)
    cases = {
        "unit": (rng.standard_normal((M, K)), rng.uniform(-0.054, 0.054, (N, K))),
        "big": (rng.standard_normal((M, K)) * 900.0, rng.uniform(-0.22, 0.22, (N, K))),
        "tiny": (rng.standard_normal((M, K)), rng.uniform(-1e-4, 1e-4, (N, K))),
        "mixed": (rng.standard_normal((M, K)) * np.exp(rng.uniform(np.log(1e-3), np.log(1e3), (M, 1))),
                  rng.uniform(-0.054, 0.054, (N, K)) * np.exp(rng.uniform(np.log(1e-2), np.log(1e2), (N, 1)))),
        "relu": (np.maximum(rng.standard_normal((M, K)) * 40.0, 0.0), rng.uniform(-0.054, 0.054, (N, K))),
    }
    names = ["f32", "bf16x3", "fp16x2", "fp16x2+", "fp16x2u"]
    print("K = %d, M = N = %d; error = |C - C64| / (|A| |W|^T), in units of 2^-24: max (rms)" % (K, M))
    print("%-14s" % "operands" + "".join("%20s" % n for n in names))
    for cname, (A, W) in cases.items():
        A = A.astype(np.float32); W = W.astype(np.float32)
        C64 = A.astype(np.float64) @ W.astype(np.float64).T
        den = np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T
        for scaled in (False, True):
            if scaled:
                As, sa = row_pow2(A); Ws, sw = row_pow2(W)
                got = {k: (v * sa * sw.T).astype(np.float32) for k, v in forms(As, Ws).items()}
            else:
                got = forms(A, W)
            row = "%-14s" % (cname + (" /rows" if scaled else ""))
            for n in names:
                e = np.abs(got[n].astype(np.float64) - C64) / den * 2.0 ** 24
                row += "%12.2f (%5.2f)" % (e.max(), np.sqrt((e ** 2).mean()))
            print(row)


if __name__ == "__main__":
    main()
