"""Runs tools/experiments/proto/ring_phase2.hip: the hand-scheduled GEMM-phase loop plus the pieces of the chains' weight ring, one at a time."""
import ctypes, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libring_phase2.so"))
lib.ring2_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
nchunk = 96
w = torch.randint(0, 255, (nchunk * 25 * 1024,), dtype=torch.uint8, device=dev)
w = ((w.view(torch.int16) & 0x03ff) | 0x3c00).view(torch.uint8)
out = torch.zeros(512, device=dev)
sink = torch.zeros(4 * 64 * 131072, device=dev)
st = lambda: torch.cuda.current_stream().cuda_stream
rounds = 11 * 1000
names = {0: "hand-scheduled loop, reads 3 steps ahead (chunk base per chunk)", 1: "+ s_barrier at every chunk hand-over",
         2: "+ LDS-DMA refill, 3 x 1 KB per wave and chunk behind MFMAs, counted vmcnt", 6: "+ the 256-B bias piece on waves 0-3",
         8: "no ring, 64 dword stores per wave every 11 chunks", 14: "ring + bias piece + the stores",
         9: "barrier + 64 dword stores as a burst every 11 chunks", 17: "barrier + the 64 stores one per step behind the MFMAs",
         33: "barrier + the same bytes as 16 x 16-B stores, burst",
         65: "barrier + the 64 stores of each wave in its own 16-step window", 70: "ring + bias piece + the stores in per-wave windows"}
names.update({137: "barrier + burst stores, nt", 265: "barrier + burst stores, sc0", 393: "barrier + burst stores, sc1",
              521: "barrier + burst stores, sc0 sc1", 649: "barrier + burst stores, sc0 sc1 nt"})
names[769] = "barrier + 16 x 12-B stores (768 B contiguous per instruction, 96 KB per CU), burst"
for mode in (1, 9, 33, 769):
    call = lambda: lib.ring2_run(mode, w.data_ptr(), nchunk, out.data_ptr(), sink.data_ptr(), 256, rounds, st())
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
