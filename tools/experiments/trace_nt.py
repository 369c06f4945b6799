"""Phase stamps of k_gemm_nt_dma workgroups (needs a build with PN_EXTRA=-DPN_TRACE_NT).  100 MHz real-time counter."""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np, torch
from pano_nerf_amd import _lib as lib
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
M = 524288; K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
FLAGS = int(sys.argv[2], 0) if len(sys.argv) > 2 else 3  # BIAS | RELU; 0x100 adds 'no store'
A = torch.relu(torch.randn(M, 256, device=dev)); W = torch.randn(256, 256, device=dev) * 0.06
C = torch.empty(M, 256, device=dev); bias = torch.randn(256, device=dev)
h = lib.load()
for _ in range(20):
    lib.call("pn_gemm_nt", M, 256, K, A.data_ptr(), 256, W.data_ptr(), 256, C.data_ptr(), 256, bias.data_ptr(), None, 256, FLAGS, st)
torch.cuda.synchronize()
nb = 8192
buf = np.zeros(nb * 8, np.uint64)
h.pn_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert h.pn_trace_read(buf.ctypes.data, nb) == 0
t = buf.reshape(nb, 8).astype(np.int64)
t0 = t[:, 0].min()
ns = lambda x: x * 10  # 100 MHz ticks -> ns
print("K", K, "kernel span us", ns(t[:, 4].max() - t0) / 1e3)
for name, a, b in (("prologue", 0, 1), ("main loop", 1, 2), ("epilogue issue", 2, 3), ("store drain", 3, 4), ("whole WG", 0, 4)):
    d = ns(t[:, b] - t[:, a]) / 1e3
    print(f"{name:15s} mean {d.mean():7.2f} us  p10 {np.percentile(d,10):7.2f}  p50 {np.percentile(d,50):7.2f}  p90 {np.percentile(d,90):7.2f}  max {d.max():7.2f}")
# shader clock sustained inside the K loop: s_memtime ticks (shader cycles) per 100 MHz real-time tick
mhz = (t[:, 6] - t[:, 5]) / np.maximum(t[:, 2] - t[:, 1], 1) * 100.0
print(f"shader clock in the K loop: mean {mhz.mean():.0f} MHz  p10 {np.percentile(mhz,10):.0f}  p50 {np.percentile(mhz,50):.0f}  p90 {np.percentile(mhz,90):.0f}")
start = ns(t[:, 0] - t0) / 1e3
order = np.argsort(start)
print("WG start times (us) every 512th:", np.round(start[order][::512], 1))
print("concurrency check: WGs started in first 2 us:", int((start < 2).sum()), " finished before 50%:", int((ns(t[:,4]-t0)/1e3 < 0.5*ns(t[:,4].max()-t0)/1e3).sum()))
