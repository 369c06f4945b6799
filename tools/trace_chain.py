"""Phase trace of the fused forward chain (debug build: PN_EXTRA=-DPN_TRACE_CHAIN pano-nerf_amd/csrc/build.sh).
Shader-clock stamps of wave 0 of workgroup 0 on its second tile."""
import ctypes, sys
import torch
sys.path.insert(0, ".")
from pano_nerf_amd import _lib
import tools.check_chain as cc

planes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cc.run(4096 * 128, 128, planes, reps=3)
lib = _lib.load()
buf = (ctypes.c_uint64 * 64)()
lib.pn_chain_trace_read.argtypes = [ctypes.c_void_p]
assert lib.pn_chain_trace_read(buf) == 0
t = list(buf)
names = {0: "start", 1: "ipe", 2: "L0 gemm", 3: "L0 epi"}
for l in range(1, 8):
    names[2 + 2 * l] = f"L{l} gemm"
    names[3 + 2 * l] = f"L{l} epi"
names.update({18: "density head", 19: "extra gemm", 20: "extra epi+viewenc", 21: "view gemm", 22: "view epi", 23: "color gemm", 24: "color out"})
prev = t[0]
for i in range(1, 25):
    print(f"{names[i]:20s} {t[i]-prev:8d} cycles")
    prev = t[i]
print("tile total", t[24] - t[0])
