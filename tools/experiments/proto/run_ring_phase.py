"""Runs tools/experiments/proto/ring_phase.hip: the bare GEMM-phase loop plus the pieces of a weight ring, one at a time."""
import ctypes, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libring_phase.so"))
lib.ring_phase_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
nchunk = 96  # 2.3 MB of packed weights, as a chain's pass
w = torch.randint(0, 255, (nchunk * 24 * 1024,), dtype=torch.uint8, device=dev)
w = ((w.view(torch.int16) & 0x03ff) | 0x3c00).view(torch.uint8)  # finite fp16 values in [1, 2)
out = torch.zeros(512, device=dev)
st = lambda: torch.cuda.current_stream().cuda_stream
rounds = 20000
names = {0: "bare loop", 1: "+ s_barrier per chunk", 2: "+ LDS-DMA refill, burst behind the barrier (+ barrier, counted vmcnt)",
         6: "+ LDS-DMA refill, one instruction per four steps behind the MFMAs", 10: "burst, waves 0-3 issue all of it",
         14: "spread, waves 0-3 issue all of it"}
for mode in (0, 1, 2, 6, 10, 14):
    call = lambda: lib.ring_phase_run(mode, w.data_ptr(), nchunk, out.data_ptr(), 256, rounds, st())
    assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        assert call() == 0
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    cyc = ms * 1e-3 * 2.1e9 / rounds
    print(f"{names[mode]:75s} {ms:8.3f} ms  {cyc:7.0f} cycles per chunk (matrix pipe: 1152)  busy {100 * 1152 / cyc:5.1f} %")
