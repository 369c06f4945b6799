"""The arithmetic of mlp_mode = "fused_f16x2" restated in numpy (pano-nerf_amd/csrc/pn_chain.hip: scale_exp, split_into,
mfma_split, chain_gemm): every fp32 operand times a power of two as an fp16 pair, three partial products, fp32 accumulate.
Prints the error of a 256-term dot product per operand magnitude against fp64, beside plain fp32 and the bf16 three-term
split.  Used by tests/test_f16x2_model.py (CPU)."""
import numpy as np

EXP_TOP, EXP_CAP, EXP_CAP_W, EXP_CAP_Z = 15, 80, 30, 120  # forward samples, weights, zero-start sums


def scale_exp(amax, cap=EXP_CAP):
    """Exponent e with amax * 2^e in [2^14, 2^15); capped from above (zeros / tiny tensors)."""
    amax = np.asarray(amax, np.float32)
    _, ex = np.frexp(amax)  # amax = m * 2^ex, m in [0.5, 1)  (0 -> ex = 0), like v_frexp_exp_i32_f32
    return np.minimum(EXP_TOP - ex, cap).astype(np.int32)


def split_f16(x, e):
    """x * 2^e = h + l with h, l fp16 (round to nearest even both times)."""
    t = np.ldexp(np.asarray(x, np.float32), e).astype(np.float32)
    h = t.astype(np.float16)
    l = (t - h.astype(np.float32)).astype(np.float16)
    return h, l


def dot_f16x2(w, x, cap=EXP_CAP_Z):
    """sum_k w[k] x[:, k] the way a chain GEMM forms it: one exponent for the weight matrix, one per sample (row of x),
    products h h' + h l' + l h' accumulated in fp32, result scaled back exactly.  cap: EXP_CAP_Z for the sums that start
    at zero (backward-direction chains), EXP_CAP for the forward chain."""
    ew = scale_exp(np.abs(w).max(), EXP_CAP_W)
    ex = scale_exp(np.abs(x).max(axis=1), cap)
    wh, wl = split_f16(w, ew)
    xh, xl = split_f16(x, ex[:, None])
    f = lambda a: a.astype(np.float32)
    acc = np.zeros(x.shape[0], np.float32)
    for k in range(w.shape[0]):  # fp32 accumulation, small terms first within a k like the kernel
        acc = acc + f(wl[k]) * f(xh[:, k])
        acc = acc + f(wh[k]) * f(xl[:, k])
        acc = acc + f(wh[k]) * f(xh[:, k])
    return np.ldexp(acc, -(ew + ex)).astype(np.float32)


def split_bf16x3(x):
    def bf16(v):
        u = v.astype(np.float32).view(np.uint32)
        r = ((u >> 16) & 1) + 0x7FFF
        return ((u + r) & 0xFFFF0000).view(np.float32)
    h = bf16(x); m = bf16(x - h); l = bf16(x - h - m)
    return h, m, l


def dot_bf16x3(w, x):
    wh, wm, wl = split_bf16x3(w.astype(np.float32))
    xh, xm, xl = split_bf16x3(x.astype(np.float32))
    acc = np.zeros(x.shape[0], np.float32)
    for k in range(w.shape[0]):
        for a, b in ((wl, xh), (wh, xl), (wm, xm), (wm, xh), (wh, xm), (wh, xh)):
            acc = acc + a[k] * b[:, k]
    return acc


def dot_f32(w, x):
    acc = np.zeros(x.shape[0], np.float32)
    for k in range(w.shape[0]):
        acc = acc + w[k].astype(np.float32) * x[:, k].astype(np.float32)
    return acc


def errors(seed=0, n=512, k=256, wmag=0.1, xmags=(1e-30, 1e-12, 1e-4, 1.0, 1e4, 1e12)):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal(k) * wmag).astype(np.float32)
    out = {}
    for xm in xmags:
        x = (rng.standard_normal((n, k)) * xm * np.exp(rng.standard_normal((n, 1)) * 3)).astype(np.float32)  # rows over decades
        ref = x.astype(np.float64) @ w.astype(np.float64)
        scale = np.sqrt((x.astype(np.float64) ** 2 * w.astype(np.float64) ** 2).sum(1))  # rms size of a row's sum of terms
        rel = lambda y: float(np.max(np.abs(y.astype(np.float64) - ref) / scale))
        out[xm] = dict(f16x2=rel(dot_f16x2(w, x)), bf16x3=rel(dot_bf16x3(w, x)), f32=rel(dot_f32(w, x)))
    return out


if __name__ == "__main__":
    for xm, e in errors().items():
        print(f"|x| ~ {xm:8.0e}: max error / rms term sum   fp16 pair {e['f16x2']:.2e}   bf16 x3 {e['bf16x3']:.2e}   plain fp32 {e['f32']:.2e}")
    x = np.array([1.0, 3.0e-5, 65504.0, 1e-8, 0.0], np.float32)
    e = scale_exp(np.abs(x).max())
    h, l = split_f16(x, e)
    back = np.ldexp(h.astype(np.float64) + l.astype(np.float64), -int(e))
    print("one column over 13 decades: exponent", int(e), " error per element relative to the column maximum",
          np.abs(back - x) / np.abs(x).max())
