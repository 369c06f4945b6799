import ctypes, sys
import torch
sys.path.insert(0, ".")
import tools.check_chain_bwd as cb
from pano_nerf_amd import _lib
lib=cb.lib; dev=cb.dev; E=cb.E; Z=cb.Z; st=cb.st
M, rpr, planes, nc = 2048, 32, 3, 5
torch.manual_seed(1)
off = (ctypes.c_int64 * 24)()
total = lib.pn_param_layout(nc, off)
params = (torch.rand(total, device=dev) - 0.5) * 0.25
R = M // rpr
mean = (torch.rand(M, 3, device=dev) - 0.5) * 6
cov = torch.rand(M, 3, device=dev) * 1e-4
vd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
Mp = int(lib.pn_pad_rows(M))
pack = torch.empty(int(lib.pn_chain_pack_bytes(planes)), dtype=torch.uint8, device=dev)
_lib.call("pn_chain_pack", params.data_ptr(), nc, planes, pack.data_ptr(), st())
enc_t, acts_t = E(Mp * 96), E(int(lib.pn_chain_acts_floats(M)))
masks_f = torch.zeros(9, Mp, 8, dtype=torch.int32, device=dev)
rr2, rd2 = E(M, 3), E(M, nc)
_lib.call("pn_chain_forward", M, rpr, R, nc, planes, pack.data_ptr(), mean.data_ptr(), cov.data_ptr(),
          vd.data_ptr(), enc_t.data_ptr(), acts_t.data_ptr(), masks_f.data_ptr(), rr2.data_ptr(), rd2.data_ptr(), st())
v = torch.randn(M, 3, device=dev)
edot_t, tang_t, sdot = E(Mp * 96), E(8, Mp * 256), E(M)
_lib.call("pn_chain_tangent", M, nc, planes, params.data_ptr(), pack.data_ptr(), mean.data_ptr(), cov.data_ptr(),
          masks_f.data_ptr(), v.data_ptr(), edot_t.data_ptr(), tang_t.data_ptr(), sdot.data_ptr(), st())
torch.cuda.synchronize()
W0 = params[off[0]:off[0]+256*96].view(256,96)
edot = cb.t32_to_rows(edot_t, Mp, 96)[:M]
h0 = cb.t32_to_rows(acts_t[:Mp*256], Mp, 256)[:M]
want = (edot.double() @ W0.double().T).float() * (h0 > 0)
got = cb.t32_to_rows(tang_t[0], Mp, 256)[:M]
print("hdot0 vs torch", cb.rel(got, want))
nog = (edot.double() @ W0.double().T).float()
print("ungated?", cb.rel(got, nog))
print(got[0,:8], want[0,:8], nog[0,:8])
print("frac nonzero got", float((got!=0).float().mean()), "want", float((want!=0).float().mean()))
