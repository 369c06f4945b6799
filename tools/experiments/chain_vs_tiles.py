"""Fixed cost of a chain launch: pn_chain_forward / pn_chain_backward back to back at 1, 2, 4, 8, 16 tiles per workgroup
(M = 32 768 x tiles rows on 256 CUs) -> time = a + b x tiles.  usage: python tools/experiments/chain_vs_tiles.py [planes=2]"""
import os, sys, time
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R_)
import torch
from pano_nerf_amd import _lib
from oracle import pano_oracle as orc
from pano_nerf_amd.mlp import ORDER, param_layout

planes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
lib = _lib.load()
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
nc = 5
params = orc.init_params(21, nc)
offs, total = param_layout(nc)
flat = torch.zeros(total)
for k in ORDER:
    flat[offs[k]:offs[k] + params[k].numel()] = params[k].reshape(-1)
flat = flat.to(dev)
tf = 1 if (planes == 2 and int(lib.pn_chain_q24_slots(2, 1, 0))) else 0
pack = torch.empty(int(lib.pn_chain_pack_bytes(planes)), dtype=torch.uint8, device=dev)
_lib.call("pn_chain_pack", flat.data_ptr(), nc, planes, pack.data_ptr(), st())
E = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
rows = []
for tiles in (1, 2, 4, 8, 16):
    M = 32768 * tiles
    Mp = int(lib.pn_pad_rows(M))
    mean, cov = (torch.rand(M, 3, device=dev) - 0.5) * 2, 1e-3 + torch.rand(M, 3, device=dev) * 1e-2
    vd = torch.nn.functional.normalize(torch.randn(M // 128, 3, device=dev), dim=-1)
    enc_t, acts_t = E(Mp * 96), E(int(lib.pn_chain_acts_floats(M)))
    masks = torch.zeros(9, Mp, 8, dtype=torch.int32, device=dev)
    rr, rd, vtab = E(M, 3), E(M, nc), E(M // 128 * 32)
    amax = torch.empty(int(lib.pn_chain_amax_slots()), dtype=torch.int32, device=dev)
    fwd = lambda: _lib.call("pn_chain_forward", M, 128, M // 128, nc, planes, pack.data_ptr(), mean.data_ptr(), cov.data_ptr(), vd.data_ptr(),
                            vtab.data_ptr(), enc_t.data_ptr(), acts_t.data_ptr(), masks.data_ptr(), rr.data_ptr(), rd.data_ptr(),
                            amax.data_ptr(), tf, 0, st())
    drgb, dden = torch.randn(M, 3, device=dev), torch.randn(M, nc, device=dev)
    drgb_t, dhv_t, d8_t, delta_t = E(Mp * 32), E(Mp * 128), E(Mp * 288), E(8, Mp * 256)
    bwd = lambda: _lib.call("pn_chain_backward", M, nc, planes, -1.0, pack.data_ptr(), masks.data_ptr(), rd.data_ptr(), drgb.data_ptr(),
                            dden.data_ptr(), None, mean.data_ptr(), cov.data_ptr(), drgb_t.data_ptr(), dhv_t.data_ptr(), d8_t.data_ptr(),
                            delta_t.data_ptr(), None, None, amax.data_ptr(), tf, 0, st())
    out = [tiles]
    for f in (fwd, bwd):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n * 1e6)
    rows.append(out)
    print(f"tiles/WG {tiles:2d}  M {M:7d}  forward {out[1]:8.1f} us  backward {out[2]:8.1f} us", flush=True)
    del enc_t, acts_t, masks, delta_t
for name, c in (("forward", 1), ("backward", 2)):
    b = (rows[-1][c] - rows[0][c]) / (rows[-1][0] - rows[0][0])
    print(f"{name}: per tile {b:.1f} us, fixed {rows[0][c] - b:.1f} us (from 1 tile), {rows[1][c] - 2 * b:.1f} us (from 2 tiles)")
