"""Runs tools/experiments/proto/gemm_phase.hip: matrix-pipe utilisation of the bare GEMM-phase loop (fragment reads from LDS + three MFMAs per
step) for the two MFMA shapes and 1 / 2 waves per SIMD."""
import ctypes, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libgemm_phase.so"))
lib.gemm_phase_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
seed = torch.randint(0, 2 ** 31 - 1, (256,), dtype=torch.int32, device=dev)
out = torch.zeros(512, device=dev)
st = lambda: torch.cuda.current_stream().cuda_stream
iters = 2000
for shape, name, cyc in ((0, "16x16x32", 16), (1, "32x32x16", 32), (2, "16x16x32 reads a pass ahead", 16), (3, "32x32x16 reads a pass ahead", 32),
                         (4, "16x16x32 asm reads 1 step ahead", 16), (5, "16x16x32 asm reads 3 steps ahead", 16), (6, "16x16x32 asm reads 5 steps ahead", 16)):
    for waves in (4, 8):
        call = lambda: lib.gemm_phase_run(shape, waves, seed.data_ptr(), out.data_ptr(), 256, iters, st())
        assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        mfma_per_simd = iters * 12 * 3 * (waves // 4)
        for clk in (2.1,):
            busy = mfma_per_simd * cyc / (ms * 1e-3 * clk * 1e9)
        flops = 256 * waves * iters * 12 * 3 * (2 * 16 * 16 * 32) * (2 if shape in (1, 3) else 1)
        print(f"{name}  {waves // 4} wave(s) per SIMD: {ms:7.3f} ms   {flops / ms / 1e9:7.1f} TF issued   matrix pipe busy {100 * busy:5.1f} % at 2.1 GHz "
              f"({ms * 1e-3 * 2.1e9 / (iters * 12):6.1f} cycles per step and SIMD)")
