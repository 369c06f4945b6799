"""Micro-benchmark of the two GEMM kernels at the bench shapes (GPU box)."""
import sys, os, ctypes
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from pano_nerf_amd import _lib as lib

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
st = torch.cuda.current_stream().cuda_stream
A = torch.randn(M, 256, device=dev); W = torch.randn(256, 256, device=dev) * 0.06
C = torch.empty(M, 256, device=dev); gate = torch.randn(M, 256, device=dev); bias = torch.randn(256, device=dev)
for _ in range(30):
    lib.call("pn_gemm_nt", M, 256, 256, A.data_ptr(), 256, W.data_ptr(), 256, C.data_ptr(), 256, bias.data_ptr(), gate.data_ptr(), 256, 0, st)
dbg = int(os.environ.get("PN_DBG", "0"))
lib.load().pn_prof_enable(dbg << 8)
print("dbg", dbg)
for name, N, K, flags in (("nt 256x256 bias+relu", 256, 256, 3), ("nt 256x256 gate", 256, 256, 4), ("nt 256x256 plain", 256, 256, 0),
                          ("nt plain nostore", 256, 256, 0x100), ("nt plain noload", 256, 256, 0x200), ("nt noload nostore", 256, 256, 0x300),
                          ("nt 256x96(K) bias+relu", 256, 96, 3), ("nt N=96 K=256", 96, 256, 0), ("nt N=128 K=256", 128, 256, 2)):
    ms = timeit(lambda: lib.call("pn_gemm_nt", M, N, K, A.data_ptr(), 256, W.data_ptr(), 256, C.data_ptr(), 256, bias.data_ptr(), gate.data_ptr(), 256, flags, st))
    print(f"{name:28s} {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF")
X = torch.randn(M, 256, device=dev); Y = torch.randn(M, 256, device=dev); Cw = torch.zeros(256, 256, device=dev)
work = torch.empty(int(lib.load().pn_gemm_tn_work_floats(M, 256, 256)), device=dev)
for name, N1, N2 in (("tn 256x256", 256, 256), ("tn 256x96", 256, 96), ("tn 128x256", 128, 256)):
    ms = timeit(lambda: lib.call("pn_gemm_tn", M, N1, N2, X.data_ptr(), 256, Y.data_ptr(), 256, Cw.data_ptr(), 256, 1, work.data_ptr(), st))
    print(f"{name:28s} {ms*1e3:8.1f} us  {2*M*N1*N2/ms/1e9:7.1f} TF (incl. slab reduce)")

out = torch.zeros(4, device=dev)
for blocks in (256, 512, 1024):
    iters = 4096
    ms = timeit(lambda: lib.call("pn_mfma_probe", out.data_ptr(), blocks, iters, st), n=5)
    fl = blocks * 4 * iters * 4 * 2 * 32 * 32 * 2
    print(f"mfma probe blocks={blocks:5d}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF")

# leading-dimension experiment: power-of-two row pitch (1 KiB) vs padded pitches
for ld in (256, 264, 272, 288, 320):
    A2 = torch.randn(M, ld, device=dev); C2 = torch.empty(M, ld, device=dev)
    ms = timeit(lambda: lib.call("pn_gemm_nt", M, 256, 256, A2.data_ptr(), ld, W.data_ptr(), 256, C2.data_ptr(), ld, bias.data_ptr(), None, 256, 0, st))
    ms2 = timeit(lambda: lib.call("pn_gemm_nt", M, 256, 256, A2.data_ptr(), ld, W.data_ptr(), 256, C2.data_ptr(), ld, bias.data_ptr(), None, 256, 0x100, st))
    print(f"nt plain lda=ldc={ld}: {ms*1e3:8.1f} us {2*M*256*256/ms/1e9:7.1f} TF   nostore {ms2*1e3:8.1f} us {2*M*256*256/ms2/1e9:7.1f} TF")
    del A2, C2

# weight row pitch experiment (every workgroup streams the same 256 KB weight matrix out of L2)
for ldb in (256, 260, 264, 272, 288, 320):
    W2 = torch.randn(256, ldb, device=dev) * 0.06
    ms = timeit(lambda: lib.call("pn_gemm_nt", M, 256, 256, A.data_ptr(), 256, W2.data_ptr(), ldb, C.data_ptr(), 256, bias.data_ptr(), None, 256, 0, st))
    print(f"nt plain ldb={ldb}: {ms*1e3:8.1f} us {2*M*256*256/ms/1e9:7.1f} TF")
