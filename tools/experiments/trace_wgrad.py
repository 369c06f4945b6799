"""Phase times of the 256 x 256 weight-gradient tile on Q24 operands (debug build: tools/build_variant.sh trwg -DPN_TRACE_WG;
PN_LIB=pano-nerf_amd/libpanonerf_hip_trwg.so python tools/experiments/trace_wgrad.py): shader-clock cycles workgroup 0 spent staging (incl. the
wait for its loads), at the barrier, issuing the next loads and in the products, per wave, summed over its half blocks."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pano_nerf_amd import _lib
if os.environ.get("PN_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["PN_LIB"])
import tools.check_chain_bwd as cb
cb.run(4096 * 128, 128, 2, reps=3)
lib = _lib.load()
buf = (ctypes.c_uint64 * 64)()
lib.pn_chain_trace_read.argtypes = [ctypes.c_void_p]
assert lib.pn_chain_trace_read(buf) == 0
t = list(buf)
n = max(t[32], 1)
print(f"half blocks of workgroup 0: {n}")
for i, name in enumerate(("stage (+ wait for loads)", "barrier", "issue next loads", "products (+ fragment reads)")):
    print(f"{name:30s}", " ".join(f"{t[8 * i + w] / n:7.0f}" for w in range(8)), " cycles per half block, waves 0-7")
print("sum per wave                  ", " ".join(f"{sum(t[8 * i + w] for i in range(4)) / n:7.0f}" for w in range(8)))
