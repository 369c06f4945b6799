"""Phase trace of the fused forward chain (debug build: PN_EXTRA=-DPN_TRACE_CHAIN pano-nerf_amd/csrc/build.sh).
Shader-clock stamps of wave 0 of workgroup 0 on its second tile."""
import ctypes, sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pano_nerf_amd import _lib
import tools.check_chain as cc

planes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cc.run(4096 * 128, 128, planes, reps=int(sys.argv[2]) if len(sys.argv) > 2 else 3)
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
print("inside ipe (any tile of wave 0): loads landed -> math done", t[31] - t[30], " -> stores issued", t[32] - t[31])
for i in range(16):
    v = t[40 + i]
    if v:
        hw = v & 0xfffffff
        print(f"block {(i & 7) + (256 if i >= 8 else 0):4d}: lds_alloc {v >> 32:#010x} xcc {(v >> 28) & 0xf} hw_id {hw:#09x} wave {hw & 0xf} simd {(hw >> 4) & 3} cu {(hw >> 8) & 0xf} sh {(hw >> 12) & 1} se {(hw >> 13) & 7}")
n = max(t[59], 1)
print(f"ring acquires of wave 0 / workgroup 0 over the kernel: {t[59]}; mean cycles waiting for own LDS reads {t[56]/n:.0f}, "
      f"for the chunk's DMA {t[57]/n:.0f}, at the barrier {t[58]/n:.0f} (each stamp costs a scalar-memory round trip)")
if t[63] > t[61]:
    print(f"in-kernel shader clock of workgroup 0 over the last forward launch: {(t[62] - t[60]) / (t[63] - t[61]) * 100:.0f} MHz "
          f"({t[62] - t[60]} shader cycles in {(t[63] - t[61]) / 100:.1f} us)")
