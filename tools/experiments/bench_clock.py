"""Does operand data change GEMM throughput (power-managed clock)?  Same kernels, zero-filled vs N(0,1) operands."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from pano_nerf_amd import _lib as lib

dev = torch.device("cuda:0")
M = 524288
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for kind in ("randn", "zeros", "randn*1e-3", "ones"):
    mk = {"randn": lambda *s: torch.randn(*s, device=dev), "zeros": lambda *s: torch.zeros(*s, device=dev),
          "randn*1e-3": lambda *s: torch.randn(*s, device=dev) * 1e-3, "ones": lambda *s: torch.ones(*s, device=dev)}[kind]
    A, W, C = mk(M, 256), mk(256, 256), torch.empty(M, 256, device=dev)
    ms = timeit(lambda: lib.call("pn_gemm_nt", M, 256, 256, A.data_ptr(), 256, W.data_ptr(), 256, C.data_ptr(), 256, None, None, 256, 0, st))
    X, Y, Cw = mk(M, 256), mk(M, 256), torch.zeros(256, 256, device=dev)
    work = torch.empty(int(lib.load().pn_gemm_tn_work_floats(M, 256, 256)), device=dev)
    ms2 = timeit(lambda: lib.call("pn_gemm_tn", M, 256, 256, X.data_ptr(), 256, Y.data_ptr(), 256, Cw.data_ptr(), 256, 0, work.data_ptr(), st))
    print(f"{kind:11s} nt {ms*1e3:7.1f} us {2*M*65536/ms/1e9:6.1f} TF   tn {ms2*1e3:7.1f} us {2*M*65536/ms2/1e9:6.1f} TF", flush=True)
out = torch.zeros(4, device=dev)
ms = timeit(lambda: lib.call("pn_mfma_probe", out.data_ptr(), 1024, 4096, st), n=5)
print(f"mfma probe (register-resident constants): {1024*4*4096*4*2*32*32*2/ms/1e9:6.1f} TF")
