"""GPU check of the fused forward chain (pn_chain_forward) against the layer-wise exact-fp32 path (pn_mlp_forward)."""
import sys, time
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


def run(M, rows_per_ray, planes, nc=5, reps=0):
    global _PLANES
    _PLANES = planes
    torch.manual_seed(0)
    import ctypes
    off = (ctypes.c_int64 * 24)()
    total = lib.pn_param_layout(nc, off)
    params = (torch.rand(total, device=dev) - 0.5) * 0.2
    R = M // rows_per_ray
    mean = (torch.rand(M, 3, device=dev) - 0.5) * 6
    cov = torch.rand(M, 3, device=dev) * 1e-4
    vd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
    Mp = int(lib.pn_pad_rows(M))
    # reference: layer-wise fp32 MFMA
    wpack = E(int(lib.pn_wpack_floats(nc)))
    _lib.call("pn_pack_weights", params.data_ptr(), nc, wpack.data_ptr(), st())
    enc, venc, vb = E(Mp, 96), E(R, 27), E(R, 128)
    acts = E(10, Mp, 256)
    masks = torch.empty(9, Mp, 8, dtype=torch.int32, device=dev)
    rr, rd = E(M, 3), E(M, nc)
    _lib.call("pn_mlp_forward", M, rows_per_ray, R, nc, params.data_ptr(), wpack.data_ptr(), mean.data_ptr(), cov.data_ptr(),
              vd.data_ptr(), enc.data_ptr(), venc.data_ptr(), vb.data_ptr(), acts.data_ptr(), masks.data_ptr(), rr.data_ptr(),
              rd.data_ptr(), st())
    # fused chain
    pack = torch.empty(int(lib.pn_chain_pack_bytes(planes)), dtype=torch.uint8, device=dev)
    _lib.call("pn_chain_pack", params.data_ptr(), nc, planes, pack.data_ptr(), st())
    enc_t = E(Mp * 96)
    acts_t = E(int(lib.pn_chain_acts_floats(M)))
    masks_f = torch.zeros(9, Mp, 8, dtype=torch.int32, device=dev)
    rr2, rd2 = E(M, 3), E(M, nc)
    import os
    keep = os.environ.get("PN_CHECK_NO_ACTS") is None  # PN_CHECK_NO_ACTS=1: time the inference variant (activations not kept)
    vtab = E(R * 32)
    call = lambda k=True: _lib.call("pn_chain_forward", M, rows_per_ray, R, nc, planes, pack.data_ptr(), mean.data_ptr(),
                                    cov.data_ptr(), vd.data_ptr(), vtab.data_ptr(), enc_t.data_ptr(), acts_t.data_ptr() if (k or keep) else None,
                                    masks_f.data_ptr(), rr2.data_ptr(), rd2.data_ptr(), None, _tfmt(planes), _WGS, st())
    call()
    torch.cuda.synchronize()

    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))

    print(f"M={M} planes={planes}")
    print("  enc      ", rel(t32_to_rows(enc_t, Mp, 96)[:M], enc[:M]))
    for l in range(8):
        h = slot_to_rows(acts_t, l, Mp, 0)[:M]
        print(f"  h{l}       ", rel(h, acts[l, :M]))
    b = t32_to_rows(TS(acts_t)[8 * Mp * 256:8 * Mp * 256 + Mp * 288], Mp, 288)[:M]
    print("  bott     ", rel(b[:, :256], acts[8, :M]))
    vr = (torch.arange(M, device=dev) // rows_per_ray) % R
    print("  viewenc  ", rel(b[:, 256:283], venc[vr]), float(b[:, 283:].abs().max()))
    hv = t32_to_rows(TS(acts_t)[8 * Mp * 256 + Mp * 288:], Mp, 128)[:M]
    print("  hv       ", rel(hv, acts[9, :M, :128]))
    print("  raw_rgb  ", rel(rr2, rr), " raw_den", rel(rd2, rd))
    # gate bits (pn_chain.hip): lane group g = (f % QB) / 4 holds bit 4 (f / QB) + f % 4 of its 8 / NG words
    tile = int(lib.pn_chain_tile()); ng = 64 // tile; qbs = 4 * ng
    bad = 0
    for l in (0, 5, 7):
        h = slot_to_rows(acts_t, l, Mp, 0)[:M]
        for f in (0, 5, 37, 100, 131, 255):
            qb, g, i = f // qbs, (f % qbs) // 4, f % 4
            d = 4 * (qb >> 1) + ((4 * (qb & 1) + i) >> 1)  # dword of the packed B operand holding the element (pn_chain.hip: gate_word / gate_bit)
            word = masks_f[l, :M, g * (8 // ng) + (d >> 4)].to(torch.int64) & 0xffffffff
            bit = (word >> ((d & 15) + 16 * (i & 1))) & 1
            bad += int((bit.bool() != (h[:, f] > 0)).sum())
    print("  gate-bit mismatches (sampled):", bad)
    if reps:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            call(False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        flops = M * 2 * 611328.0 * {3: 6, 2: 3, 1: 1}[planes]
        print(f"  fused forward {dt*1e3:.3f} ms  ({M*2*611328.0/dt/1e12:.1f} TF fp32-equivalent, {flops/dt/1e12:.0f} TF issued on the matrix cores)")
        ref = lambda: _lib.call("pn_mlp_forward", M, rows_per_ray, R, nc, params.data_ptr(), wpack.data_ptr(), mean.data_ptr(),
                                cov.data_ptr(), vd.data_ptr(), enc.data_ptr(), venc.data_ptr(), vb.data_ptr(), acts.data_ptr(),
                                masks.data_ptr(), rr.data_ptr(), rd.data_ptr(), st())
        ref(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ref()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"  layer-wise fp32 forward {dt*1e3:.3f} ms ({M*2*611328.0/dt/1e12:.1f} TF)")


if __name__ == "__main__":
    modes = [int(x) for x in sys.argv[1:]] or [3, 2, 1]  # planes: 3 bf16 three-term, 2 fp16 pair, 1 plain bf16
    for pl in modes:
        run(64 * 32, 32, pl)
        run(1000 * 10, 10, pl)
    for pl in modes:
        run(4096 * 128, 128, pl, reps=5)
