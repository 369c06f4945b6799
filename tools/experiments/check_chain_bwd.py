"""GPU check of the fused reverse sweep / tangent sweep / backward chain against the layer-wise exact-fp32 path."""
import ctypes, sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pano_nerf_amd import _lib

dev = torch.device("cuda:0")
import os as _os
if _os.environ.get("PN_LIB"):  # a variant build of the library (tools/build_variant.sh)
    _lib.LIB_PATH = _os.path.abspath(_os.environ["PN_LIB"])
lib = _lib.load()
st = lambda: torch.cuda.current_stream().cuda_stream
E = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
Z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)


_PLANES = 3



import os as _os
_WGS = int(_os.environ.get("PN_MAX_WGS", "0"))  # workgroup budget of every launch (0: all CUs)


def _tfmt(planes):
    """t_format of the calls: PN_TFMT (0 / 1), default Q24 where the build has it"""
    from pano_nerf_amd import _lib as _l
    want = int(_os.environ.get("PN_TFMT", "1"))
    return want if (planes == 2 and int(_l.load().pn_chain_q24_slots(2, 1, 0))) else 0

def TS(t):
    """Flat view of a T tensor in its element type: bf16 with planes = 1 (the buffers are allocated as floats)."""
    if t.dtype == torch.bfloat16:
        return t.reshape(-1)
    return t.reshape(-1).view(torch.bfloat16) if _PLANES == 1 else t.reshape(-1)


def t32_to_rows(t, Mp, F):
    tile = int(lib.pn_chain_tile())
    return TS(t)[:Mp * F].float().reshape(Mp // tile, F, tile).permute(0, 2, 1).reshape(Mp, F)


def slot_to_rows(buf, l, Mp, kind):
    """256-wide slot l of tensor `kind` (0 acts, 1 tangents, 2 deltas, 3 reverse sweep), fp32 T layout or Q24 as the mode stores it."""
    from pano_nerf_amd import tlayout
    if (int(lib.pn_chain_q24_slots(_PLANES, _tfmt(_PLANES), kind)) >> l) & 1:
        return tlayout.q24_decode(buf.reshape(-1)[l * Mp * 256:(l + 1) * Mp * 256].view(torch.uint8), Mp, 256)
    return t32_to_rows(TS(buf)[l * Mp * 256:(l + 1) * Mp * 256], Mp, 256)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def run(M, rows_per_ray, planes, nc=5, reps=0):
    global _PLANES
    _PLANES = planes
    torch.manual_seed(1)
    off = (ctypes.c_int64 * 24)()
    total = lib.pn_param_layout(nc, off)
    params = (torch.rand(total, device=dev) - 0.5) * 0.25
    R = M // rows_per_ray
    mean = (torch.rand(M, 3, device=dev) - 0.5) * 6
    cov = torch.rand(M, 3, device=dev) * 1e-4
    vd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
    Mp = int(lib.pn_pad_rows(M))
    dbias = -1.0
    wpack = E(int(lib.pn_wpack_floats(nc)))
    _lib.call("pn_pack_weights", params.data_ptr(), nc, wpack.data_ptr(), st())
    enc, venc, vb = E(Mp, 96), E(R, 27), E(R, 128)
    acts = E(10, Mp, 256)
    masks = torch.empty(9, Mp, 8, dtype=torch.int32, device=dev)
    rr, rd = E(M, 3), E(M, nc)
    _lib.call("pn_mlp_forward", M, rows_per_ray, R, nc, params.data_ptr(), wpack.data_ptr(), mean.data_ptr(), cov.data_ptr(),
              vd.data_ptr(), enc.data_ptr(), venc.data_ptr(), vb.data_ptr(), acts.data_ptr(), masks.data_ptr(), rr.data_ptr(),
              rd.data_ptr(), st())
    rsweep, scratch, gmean = E(8, Mp, 256), E(Mp, 96), E(M, 3)
    _lib.call("pn_density_grad", M, nc, dbias, params.data_ptr(), wpack.data_ptr(), mean.data_ptr(), cov.data_ptr(),
              acts.data_ptr(), masks.data_ptr(), rd.data_ptr(), rsweep.data_ptr(), scratch.data_ptr(), gmean.data_ptr(), st())
    d_rgb, d_den, v = torch.randn(M, 3, device=dev), torch.randn(M, nc, device=dev), torch.randn(M, 3, device=dev)
    grads = Z(total)
    nw = int(lib.pn_mlp_backward_work_floats(M, rows_per_ray, R, 0))
    work = E(nw)
    d_mean = E(M, 3)
    _lib.call("pn_mlp_backward", M, rows_per_ray, R, nc, dbias, params.data_ptr(), wpack.data_ptr(), mean.data_ptr(),
              cov.data_ptr(), enc.data_ptr(), venc.data_ptr(), acts.data_ptr(), masks.data_ptr(), rd.data_ptr(),
              d_rgb.data_ptr(), d_den.data_ptr(), rsweep.data_ptr(), v.data_ptr(), d_mean.data_ptr(), grads.data_ptr(),
              work.data_ptr(), 0, 1, 0, None, None, None, None, None, None, st(), None)
    torch.cuda.synchronize()
    dbuf = work[:9 * Mp * 256].view(9, Mp, 256)
    tbuf = work[9 * Mp * 256:17 * Mp * 256].view(8, Mp, 256)
    dbott = work[17 * Mp * 256:18 * Mp * 256].view(Mp, 256)
    edot = work[18 * Mp * 256:18 * Mp * 256 + Mp * 96].view(Mp, 96)

    # ---- fused
    pack = torch.empty(int(lib.pn_chain_pack_bytes(planes)), dtype=torch.uint8, device=dev)
    _lib.call("pn_chain_pack", params.data_ptr(), nc, planes, pack.data_ptr(), st())
    enc_t, acts_t = E(Mp * 96), E(int(lib.pn_chain_acts_floats(M)))
    masks_f = torch.zeros(9, Mp, 8, dtype=torch.int32, device=dev)
    rr2, rd2 = E(M, 3), E(M, nc)
    amax = torch.empty(int(lib.pn_chain_amax_slots()), dtype=torch.int32, device=dev)
    _lib.call("pn_chain_forward", M, rows_per_ray, R, nc, planes, pack.data_ptr(), mean.data_ptr(), cov.data_ptr(),
              vd.data_ptr(), E(R * 32).data_ptr(), enc_t.data_ptr(), acts_t.data_ptr(), masks_f.data_ptr(), rr2.data_ptr(), rd2.data_ptr(),
              amax.data_ptr(), _tfmt(planes), _WGS, st())
    rs_t, gmean2 = E(8, Mp * 256), E(M, 3)
    f_dgrad = lambda: _lib.call("pn_chain_density_grad", M, nc, planes, dbias, params.data_ptr(), pack.data_ptr(), mean.data_ptr(),
                                cov.data_ptr(), masks_f.data_ptr(), rd2.data_ptr(), rs_t.data_ptr(), 1, gmean2.data_ptr(), amax.data_ptr(), _tfmt(planes), _WGS, st())
    f_dgrad()
    edot_t, tang_t, sdot = E(Mp * 96), E(8, Mp * 256), E(M)
    f_tan = lambda: _lib.call("pn_chain_tangent", M, nc, planes, params.data_ptr(), pack.data_ptr(), mean.data_ptr(), cov.data_ptr(),
                              masks_f.data_ptr(), v.data_ptr(), edot_t.data_ptr(), tang_t.data_ptr(), sdot.data_ptr(), amax.data_ptr(), _tfmt(planes), _WGS, st())
    f_tan()
    drgb_t, dhv_t, d8_t, delta_t, coef_t = Z(Mp * 32), E(Mp * 128), Z(Mp * 288), E(8, Mp * 256), Z(Mp * 32)
    d_mean2 = E(M, 3)
    f_bwd = lambda: _lib.call("pn_chain_backward", M, nc, planes, dbias, pack.data_ptr(), masks_f.data_ptr(), rd2.data_ptr(),
                              d_rgb.data_ptr(), d_den.data_ptr(), sdot.data_ptr(), mean.data_ptr(), cov.data_ptr(),
                              drgb_t.data_ptr(), dhv_t.data_ptr(), d8_t.data_ptr(), delta_t.data_ptr(), coef_t.data_ptr(),
                              d_mean2.data_ptr(), amax.data_ptr(), _tfmt(planes), _WGS, st())
    f_bwd()
    torch.cuda.synchronize()
    print(f"M={M} planes={planes}")
    print("  grad_mean ", rel(gmean2, gmean))
    for l in (7, 5, 3, 0):
        print(f"  r{l}        ", rel(slot_to_rows(rs_t, l, Mp, 3)[:M], rsweep[l, :M]))
    print("  edot      ", rel(t32_to_rows(edot_t, Mp, 96)[:M], edot[:M]))
    for l in (0, 4, 5, 7):
        print(f"  hdot{l}     ", rel(slot_to_rows(tang_t, l, Mp, 1)[:M], tbuf[l, :M]))
    d8 = t32_to_rows(d8_t, Mp, 288)[:M]
    print("  d_bott    ", rel(d8[:, :256], dbott[:M]))
    for l in (7, 6, 5, 1, 0):
        print(f"  delta{l}    ", rel(slot_to_rows(delta_t, l, Mp, 2)[:M], dbuf[l, :M]))
    print("  d_mean    ", rel(d_mean2, d_mean))
    # second-order addend and padded tensors
    z = rd2[:, 0] + dbias
    sg = torch.sigmoid(z)
    want_dd = d_den.clone()
    want_dd[:, 0] += sg * (1 - sg) * sdot
    print("  d_den(+2nd)", rel(d8[:, 256:256 + nc], want_dd), float(d8[:, 256 + nc:].abs().max()))
    print("  coef      ", rel(t32_to_rows(coef_t, Mp, 32)[:M, 0], sg), " d_rgb", rel(t32_to_rows(drgb_t, Mp, 32)[:M, :3], d_rgb))
    # ---- weight gradients: fused TN GEMMs vs the layer-wise path (defer_wgrad = 0)
    grads_ref = Z(total)
    work2 = E(nw)
    _lib.call("pn_mlp_backward", M, rows_per_ray, R, nc, dbias, params.data_ptr(), wpack.data_ptr(), mean.data_ptr(),
              cov.data_ptr(), enc.data_ptr(), venc.data_ptr(), acts.data_ptr(), masks.data_ptr(), rd.data_ptr(),
              d_rgb.data_ptr(), d_den.data_ptr(), rsweep.data_ptr(), v.data_ptr(), d_mean.data_ptr(), grads_ref.data_ptr(),
              work2.data_ptr(), 0, 0, 0, None, None, None, None, None, None, st(), None)

    class Ev(ctypes.Structure):
        _fields_ = [("M", ctypes.c_int64)] + [(k, ctypes.c_void_p) for k in
                    ("enc_t", "acts_t", "drgb_t", "dhv_t", "d8_t", "delta_t", "rs_t", "edot_t", "tang_t", "coef_t", "amax")]
    ev = Ev(M, enc_t.data_ptr(), acts_t.data_ptr(), drgb_t.data_ptr(), dhv_t.data_ptr(), d8_t.data_ptr(), delta_t.data_ptr(),
            rs_t.data_ptr(), edot_t.data_ptr(), tang_t.data_ptr(), coef_t.data_ptr(), amax.data_ptr())
    grads_f = Z(total)
    wfl = int(lib.pn_chain_wgrad_work_floats())
    wk = E(wfl)
    f_wg = lambda: _lib.check(lib.pn_chain_wgrad(1, ctypes.byref(ev), nc, planes, grads_f.data_ptr(), wk.data_ptr(), wfl, 3, _tfmt(planes), _WGS, st()),
                              "pn_chain_wgrad")
    f_wg()
    torch.cuda.synchronize()
    names = [f"layers.{i}.{k}" for i in range(8) for k in ("w", "b")] + ["extra.w", "extra.b", "view.w", "view.b", "den.w", "col.w", "den.b", "col.b"]
    offs = list(off) + [total]
    order = sorted(range(24), key=lambda i: offs[i])
    for n_, i in enumerate(order):
        lo_, hi_ = offs[i], (offs[order[n_ + 1]] if n_ + 1 < 24 else total)
        g, r = grads_f[lo_:hi_], grads_ref[lo_:hi_]
        print(f"  grad {names[i]:12s} max-rel {rel(g, r):.2e}   norm-rel {float((g-r).norm()/r.norm().clamp_min(1e-30)):.2e}")
    if reps:
        grads_f.zero_()
        for name, fn, fl in (("wgrad", f_wg, 2 * (508160.0 * 2 + 611328.0 - 508160.0)),):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print(f"  fused {name} {dt*1e3:.3f} ms ({M*fl/dt/1e12:.1f} TF fp32-equivalent)")
        for name, fn, fl in (("dgrad", f_dgrad, 2 * 508160.0), ("tangent", f_tan, 2 * 508160.0), ("backward", f_bwd, 2 * 611328.0)):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print(f"  fused {name} {dt*1e3:.3f} ms ({M*fl/dt/1e12:.1f} TF fp32-equivalent)")


if __name__ == "__main__":
    modes = [int(x) for x in sys.argv[1:]] or [3, 2, 1]
    for pl in modes:
        run(64 * 32, 32, pl)
        run(1000 * 10, 10, pl)
    for pl in modes:
        run(4096 * 128, 128, pl, reps=5)
