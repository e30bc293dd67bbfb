#!/usr/bin/env python
"""CPU study (numpy, no GPU): what would two fp16 limbs per operand cost in accuracy?  DESIGN.md section 10, next step 2.

Three ways to evaluate an fp32 product sum_k x[m,k] * w[k,n] with fp32 accumulation, against float64:
  fp32     the products as fp32 values (what the exact-fp32 matrix pipe does)
  bf16x3   x = hi + mid + lo exactly (three bf16 limbs), the six limb products of weight >= 2^-16 (csrc/limb_gemm.hip)
  fp16x2   x * 2^s = hi + lo + r with hi, lo fp16 and |r| <= 2^-22 |x|, an exact power-of-two scale s per ROW of x (per matrix
           for w), the three products hi*hi, hi*lo, lo*hi
All three accumulate the per-k terms in float32 in the same order, so the differences are the representations'.
Operands: the C2 layer shapes (K = 768, N = 256) with (a) post-ReLU aggregated activations, (b) gradients whose rows differ in
magnitude by 1e4 and whose entries are heavy-tailed."""
import numpy as np

rng = np.random.default_rng(0)


def bf16(x):                                   # round-to-nearest-even to bf16, returned as float32
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def limbs_bf16(x):
    hi = bf16(x); r = x - hi
    mid = bf16(r); lo = bf16(r - mid)
    assert np.array_equal(hi + mid + lo, x)
    return hi, mid, lo


def limbs_fp16(x, axis):
    """(hi, lo, scale): x * scale ~ hi + lo, scale a power of two per row (axis=1) or for the whole matrix (axis=None)."""
    m = np.abs(x).max(axis=axis, keepdims=axis is not None)
    m = np.where(m > 0, m, 1.0)
    s = np.exp2(14 - np.ceil(np.log2(m))).astype(np.float32)
    xs = (x * s).astype(np.float32)
    hi = xs.astype(np.float16).astype(np.float32)
    lo = (xs - hi).astype(np.float16).astype(np.float32)
    return hi, lo, s


def accumulate(terms_of_k, M, N, K):
    acc = np.zeros((M, N), np.float32)
    for k in range(K):
        for t in terms_of_k(k):
            acc += t.astype(np.float32)
    return acc


def study(name, x, w):
    M, K = x.shape
    N = w.shape[1]
    truth = x.astype(np.float64) @ w.astype(np.float64)
    f32 = accumulate(lambda k: [np.outer(x[:, k], w[k]).astype(np.float32)], M, N, K)
    xh, xm, xl = limbs_bf16(x); wh, wm, wl = limbs_bf16(w)
    b3 = accumulate(lambda k: [np.outer(xh[:, k], wl[k]), np.outer(xl[:, k], wh[k]), np.outer(xm[:, k], wm[k]),
                               np.outer(xh[:, k], wm[k]), np.outer(xm[:, k], wh[k]), np.outer(xh[:, k], wh[k])], M, N, K)
    ah, al, sa = limbs_fp16(x, 1); bh, bl, sb = limbs_fp16(w, None)
    h2 = accumulate(lambda k: [np.outer(ah[:, k], bl[k]), np.outer(al[:, k], bh[k]), np.outer(ah[:, k], bh[k])], M, N, K)
    h2 = (h2.astype(np.float64) / (sa.astype(np.float64) * float(sb))).astype(np.float32)
    scale = np.abs(truth).max()
    row = np.maximum(np.abs(truth).max(1, keepdims=True), 1e-300)
    print("%-34s max|out| %.3g" % (name, scale))
    for tag, v in (("fp32", f32), ("bf16x3", b3), ("fp16x2", h2)):
        e = np.abs(v.astype(np.float64) - truth)
        print("   %-7s max abs err %.3e   (%.2e of max|out|)   worst row: %.2e of that row's max" % (
            tag, e.max(), e.max() / scale, (e.max(1, keepdims=True) / row).max()))


M, K, N = 1500, 768, 256
w = (rng.uniform(-1, 1, (K, N)) * np.sqrt(6.0 / (256 + 256))).astype(np.float32)
act = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32) * rng.gamma(2.0, 8.0, (M, 1)).astype(np.float32)
study("aggregated post-ReLU activations", act, w)
g = (rng.standard_t(3, (M, 256)) * 1e-5 * np.exp(rng.uniform(np.log(1e-2), np.log(1e2), (M, 1)))).astype(np.float32)
wt = (rng.uniform(-1, 1, (256, 768)) * np.sqrt(6.0 / 512)).astype(np.float32)
study("gradients (rows over 4 decades)", g, wt)
tiny = act.copy(); tiny[:, ::2] *= 1e-6
study("rows mixing O(1) and O(1e-6) entries", tiny, w)
