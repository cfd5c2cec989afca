#!/usr/bin/env python3
"""Numerics of a 1-D Winograd / Cook-Toom form F(m, 5) of the WN stack's 5-tap convolutions (reference layers.py:116-124, 146) — a
STUDY for DESIGN.md §8 (what comes next), not a product path: numpy on the CPU, no kernel uses it.

F(4, 5) computes 4 output frames from 8 input frames with 8 products per (row, channel) instead of 20: 2.5 x fewer MFMAs on the
convolutions that are 80 % of the step's matrix work.  The question answered here is whether the transforms (fp32 adds with
coefficients up to 8) cost the fp32-equivalence that the bf16x6 arithmetic has: transforms in fp32, products exact (what six bf16
products give), accumulation in fp32 in the order of the kernel's 32-deep MFMA steps — against the same model of the direct form.

  python tools/winograd_study.py        # profiles/r05_winograd_numerics.txt
"""
import numpy as np
from fractions import Fraction as Fr
def cook_toom(m, r, pts):
    # 1-D Winograd F(m,r) via Cook-Toom with points pts (len m+r-2) + infinity
    n = m + r - 1
    P = [Fr(p) for p in pts]
    # Vandermonde-style: polynomial multiplication of degree m-1 (outputs side uses transpose)
    # A^T (m x n): rows i, cols p: p^i ; last column (inf) = [0,..,0,1]
    AT = [[(P[j] ** i) for j in range(n - 1)] + [Fr(1) if i == m - 1 else Fr(0)] for i in range(m)]
    # G (n x r): row p: p^k / N_p ; last row = [0..0,1]
    def Np(j):
        v = Fr(1)
        for k in range(n - 1):
            if k != j: v *= (P[j] - P[k])
        return v
    G = [[(P[j] ** k) / Np(j) for k in range(r)] for j in range(n - 1)] + [[Fr(0)] * (r - 1) + [Fr(1)]]
    # B^T (n x n): from Lagrange polynomials: row j = coefficients of prod_{k!=j}(x - p_k) ; last row = coefficients of prod_k (x - p_k)
    def polymul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b): out[i + j] += x * y
        return out
    BT = []
    for j in range(n - 1):
        poly = [Fr(1)]
        for k in range(n - 1):
            if k != j: poly = polymul(poly, [-P[k], Fr(1)])
        BT.append(poly + [Fr(0)] * (n - len(poly)))
    poly = [Fr(1)]
    for k in range(n - 1): poly = polymul(poly, [-P[k], Fr(1)])
    BT.append(poly)
    f = lambda M: np.array([[float(x) for x in row] for row in M])
    return f(AT), f(G), f(BT)

def check(m, r, pts):
    AT, G, BT = cook_toom(m, r, pts)
    rng = np.random.default_rng(0)
    d = rng.standard_normal(m + r - 1); g = rng.standard_normal(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    return np.abs(y - ref).max()

for m, pts in ((2, [0, 1, -1, 2, -2]), (2, [0, 1, -1, 0.5, -0.5]), (4, [0, 1, -1, 2, -2, 0.5, -0.5]), (3, [0, 1, -1, 2, -2, 0.5])):
    print(m, pts, "identity err", check(m, 5, pts))

def conv_err(m, pts, C=192, T=400, M=64, trials=1):
    AT, G, BT = cook_toom(m, 5, pts)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((C, T + 4)).astype(np.float32)       # already padded
    x *= np.exp(rng.standard_normal((C, 1))).astype(np.float32)   # per-channel scale spread
    w = (rng.standard_normal((M, C, 5)) / np.sqrt(C * 5)).astype(np.float32)
    ref = np.zeros((M, T))
    for k in range(5): ref += w[:, :, k].astype(np.float64) @ x[:, k:k + T].astype(np.float64)
    # direct fp32 accumulate (numpy float32 matmul ~ pairwise/blocked) as the native stand-in
    d32 = np.zeros((M, T), np.float32)
    for k in range(5): d32 += w[:, :, k] @ x[:, k:k + T]
    # winograd: transforms in fp32, products exact (float64 of fp32 operands), accumulation over channels in fp32
    n = m + 4
    nt = T // m
    U = np.einsum('pk,mck->mcp', G.astype(np.float32), w).astype(np.float32)          # (M,C,n) fp32 transform (could be fp64 on host)
    U64 = np.einsum('pk,mck->mcp', G, w.astype(np.float64)).astype(np.float32)        # transform in fp64 then rounded
    idx = (np.arange(nt)[:, None] * m + np.arange(n)[None, :])                        # (nt, n)
    dt = x[:, idx]                                                                     # (C, nt, n)
    V = np.einsum('pj,ctj->cpt', BT.astype(np.float32), dt).astype(np.float32)        # fp32
    out = {}
    for name, UU in (("U fp32", U), ("U fp64->fp32", U64)):
        Mm = np.einsum('mcp,cpt->mpt', UU.astype(np.float64), V.astype(np.float64))   # exact products, wide accumulate
        Mm32 = Mm.astype(np.float32)                                                   # rounding of the fp32 accumulators (optimistic)
        y = np.einsum('ip,mpt->mti', AT.astype(np.float32), Mm32).reshape(M, nt * m)
        out[name] = y
    sc = np.abs(ref).max()
    print(f"F({m},5) pts={pts}: direct fp32 max/rms err {np.abs(d32-ref).max()/sc:.2e} / {np.sqrt(((d32-ref)**2).mean())/sc:.2e}", end="")
    for k, y in out.items():
        print(f" | wino[{k}] {np.abs(y-ref[:, :nt*m]).max()/sc:.2e} / {np.sqrt(((y-ref[:, :nt*m])**2).mean())/sc:.2e}", end="")
    print()

conv_err(2, [0, 1, -1, 2, -2])
conv_err(2, [0, 1, -1, 0.5, -0.5])
conv_err(2, [0, 1, -1, 2, -0.5])
conv_err(4, [0, 1, -1, 2, -2, 0.5, -0.5])
conv_err(3, [0, 1, -1, 2, -2, 0.5])

print("---- with MFMA-like accumulation: exact 32-deep dots, fp32 accumulator adds (6 partial products per block modelled as 6 adds)")
def blocks_acc(A, Bm, nsplit=6):
    # A (M,K) fp32, Bm (K,N) fp32: sequential fp32 accumulation of exact 32-deep block dots; each block's dot is split in nsplit pieces of random sizes to mimic 6 MFMAs
    M, K = A.shape; N = Bm.shape[1]
    acc = np.zeros((M, N), np.float32)
    for k0 in range(0, K, 32):
        blk = A[:, k0:k0 + 32].astype(np.float64) @ Bm[k0:k0 + 32].astype(np.float64)
        # main product + small corrections: model hh (1-2^-8), others tiny: add as main then 5 small pieces
        main = (blk * (1 - 2.0 ** -9)).astype(np.float32)
        rest = blk - main.astype(np.float64)
        acc = (acc + main).astype(np.float32)
        for j in range(nsplit - 1):
            acc = (acc + (rest / (nsplit - 1)).astype(np.float32)).astype(np.float32)
    return acc
def conv_err2(m, pts, C=192, T=400, M=64):
    AT, G, BT = cook_toom(m, 5, pts)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((C, T + 4)).astype(np.float32)
    x *= np.exp(rng.standard_normal((C, 1))).astype(np.float32)
    w = (rng.standard_normal((M, C, 5)) / np.sqrt(C * 5)).astype(np.float32)
    ref = np.zeros((M, T))
    for k in range(5): ref += w[:, :, k].astype(np.float64) @ x[:, k:k + T].astype(np.float64)
    # direct, K ordered (channel-block, tap) like the kernel
    Wd = np.concatenate([w[:, c0:c0 + 32, k] for c0 in range(0, C, 32) for k in range(5)], axis=1)
    Xd = np.concatenate([x[c0:c0 + 32, k:k + T] for c0 in range(0, C, 32) for k in range(5)], axis=0)
    d32 = blocks_acc(Wd, Xd)
    n = m + 4; nt = T // m
    U = np.einsum('pk,mck->mcp', G, w.astype(np.float64)).astype(np.float32)
    idx = (np.arange(nt)[:, None] * m + np.arange(n)[None, :])
    V = np.einsum('pj,ctj->cpt', BT.astype(np.float32), x[:, idx]).astype(np.float32)
    Mm = np.stack([blocks_acc(U[:, :, p], V[:, p, :]) for p in range(n)], axis=1)      # (M, n, nt) fp32
    y = np.einsum('ip,mpt->mti', AT.astype(np.float32), Mm).astype(np.float32).reshape(M, nt * m)
    sc = np.abs(ref).max()
    e = lambda z, r: (np.abs(z - r).max() / sc, np.sqrt(((z - r) ** 2).mean()) / sc)
    print(f"F({m},5) {pts}: direct {e(d32, ref)[0]:.2e}/{e(d32, ref)[1]:.2e}   winograd {e(y, ref[:, :nt*m])[0]:.2e}/{e(y, ref[:, :nt*m])[1]:.2e}")
conv_err2(2, [0, 1, -1, 2, -0.5])
conv_err2(2, [0, 1, -1, 2, -2])
conv_err2(4, [0, 1, -1, 2, -2, 0.5, -0.5])
conv_err2(3, [0, 1, -1, 2, -2, 0.5])
