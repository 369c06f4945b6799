"""Runs tools/experiments/proto/ws_layer.hip: correctness of the weight-stationary layer against fp64, then its rate (see the .hip header)."""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libws_layer.so"))
lib.ws_layer_run.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
torch.manual_seed(0)
W = (torch.rand(256, 256, device=dev) - 0.5) * 0.25
b = (torch.rand(256, device=dev) - 0.5) * 0.1
st = lambda: torch.cuda.current_stream().cuda_stream


def run(x_rows, tiles, x_tiles, blocks, mode, reps=0):
    xt = x_rows.reshape(-1, 16, 256).permute(0, 2, 1).contiguous()
    y = torch.empty(tiles, 256, 16, device=dev)
    gates = torch.zeros(tiles * 16, 8, dtype=torch.int32, device=dev)
    call = lambda: lib.ws_layer_run(W.data_ptr(), b.data_ptr(), xt.data_ptr(), y.data_ptr(), gates.data_ptr(), tiles, x_tiles, blocks, mode, st())
    assert call() == 0
    torch.cuda.synchronize()
    if not reps:
        return y.permute(0, 2, 1).reshape(-1, 256), gates
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        call()
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# correctness: 4096 samples of mixed magnitudes
M = 4096
x = torch.relu(torch.randn(M, 256, device=dev)) * torch.exp(3 * torch.randn(M, 1, device=dev))
ORI = 8 * int(sys.argv[1]) if len(sys.argv) > 1 else 0  # 1: rows = samples (16-B stores)
y, gates = run(x, M // 16, M // 16, 64, ORI)
want = torch.relu(x.double() @ W.double().T + b.double())
err = ((y.double() - want).abs().max(1).values / want.abs().max(1).values.clamp_min(1e-30)).max()
print("max per-sample relative error against fp64: %.2e" % float(err))
bits = (want > 0)
got_bits = torch.zeros(M, 256, dtype=torch.bool, device=dev)
if ORI:
    for f in range(256):
        got_bits[:, f] = ((gates[:, f // 32] >> (f % 32)) & 1).bool()
else:
    gw = gates.view(torch.int16).reshape(M, 16).to(torch.int32) & 0xffff  # [sample][wave * 4 + g] half words
    for wv in range(4):
        for g in range(4):
            for ft in range(4):
                for i in range(4):
                    got_bits[:, 64 * wv + 16 * ft + 4 * g + i] = ((gw[:, wv * 4 + g] >> (4 * ft + i)) & 1).bool()
print("gate mismatches:", int((got_bits != (y > 0)).sum()), " vs fp64 sign:", int((got_bits != bits).sum()))

# rate: 524288 samples, 256 workgroups
M = 524288
tiles = M // 16
for x_tiles, label in ((128, "input L2-resident (128 tiles cycled)"), (tiles, "input from HBM")):
    xs = torch.relu(torch.randn(min(x_tiles, tiles) * 16, 256, device=dev))
    for mode, name in ((0, "full"), (1, "no MFMA"), (2, "no stores"), (4, "no loads"), (6, "no loads, no stores")):
        ms = run(xs, tiles, x_tiles, 256, ORI + mode, reps=20)
        cyc = ms * 1e-3 * 2.1e9 / (tiles / 256)
        print(f"{label:40s} {name:22s} {ms:7.3f} ms   {cyc:7.0f} cycles per tile at 2.1 GHz (matrix core alone: 1536)   {2*M*65536/ms/1e9:6.1f} TF fp32-equivalent")

# is the exposed store / load time an aggregate (HBM) limit or a per-CU one?  Same tiles per workgroup on fewer workgroups.
print("-- 128 tiles per workgroup, L2-resident input, fewer workgroups")
xs = torch.relu(torch.randn(128 * 16, 256, device=dev))
for blocks in (256, 128, 64, 32):
    row = []
    for mode in (0, 2, 4, 6):
        row.append(run(xs, 128 * blocks, 128, blocks, ORI + mode, reps=20))
    print(f"{blocks:4d} workgroups: full {row[0]:.3f}  no stores {row[1]:.3f}  no loads {row[2]:.3f}  neither {row[3]:.3f} ms")
