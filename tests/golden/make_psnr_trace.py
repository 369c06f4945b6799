"""Train the IMPORTED reference MipNeRF on CPU for a few hundred steps at BASELINE config[0] scale
(64x128 pano, 32 samples; fixed batches and jitter noise from PCG64) and store its loss-per-step trace and the
PSNR of a held-out view (SURVEY.md 8d 'PSNR parity').  Build container only.  Same import shim as make_golden.py."""
import os, sys, types, time
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference"); sys.path.insert(1, ROOT)
for name in ("cv2", "Imath"):
    sys.modules[name] = types.ModuleType(name)
_exr = types.ModuleType("OpenEXR"); _exr.InputFile = _exr.OutputFile = _exr.Header = object; sys.modules["OpenEXR"] = _exr
import numpy as np, torch
import models.mip_nerf as rmip
import utils.surface_rendering as rsurf
from utils.lr_schedule import MipLRDecay
from datasets.base_datasets import Rays
sys.path.insert(0, HERE)
from make_golden import NoiseTap, ref_loss_mip, load_params
from oracle import pano_oracle as orc

STEPS, B, N, H, W = 200, 256, 32, 64, 128


def schedule(seed=7):
    """Per-step batch indices and noise, regenerated identically by tests/test_gpu_psnr.py."""
    rng = np.random.Generator(np.random.PCG64(seed))
    for _ in range(STEPS):
        idx = rng.integers(0, 2 * H * W, size=B)  # cameras 0 and 1 train, camera 2 is held out
        t_rand = rng.random((B, N + 1), dtype=np.float32)
        u_rand = rng.random((B, N + 1), dtype=np.float32) * np.float32(1.0 / (N + 1) - 1.1920929e-07)
        yield idx, t_rand, u_rand


class FixedNoise:
    """Feed pre-drawn noise to the reference's torch.rand / uniform_ calls (in draw order)."""
    def __init__(self, draws):
        self.draws = list(draws)
    def __enter__(self):
        self._rand, self._uni = torch.rand, torch.Tensor.uniform_
        me = self
        def rand(*shape, **kw):
            return torch.from_numpy(me.draws.pop(0))
        def uniform_(t, *a, **kw):
            t.copy_(torch.from_numpy(me.draws.pop(0))); return t
        torch.rand, torch.Tensor.uniform_ = rand, uniform_
        return self
    def __exit__(self, *a):
        torch.rand, torch.Tensor.uniform_ = self._rand, self._uni


def main():
    torch.set_num_threads(8)
    flat, rgbs, radius, _ = orc.synthetic_scene(H, W, 3, seed=4)
    net = rmip.MipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=1)
    load_params(net.mlp, orc.init_params(4, 1))
    opt = torch.optim.Adam(net.mlp.parameters(), lr=2e-4)
    sch = MipLRDecay(opt, 2e-4, 2e-5, 44000, 120, 0.01)
    losses = []
    t0 = time.time()
    for step, (idx, t_rand, u_rand) in enumerate(schedule()):
        it = torch.from_numpy(idx)
        rays = Rays(*[x[it] for x in flat]); gt = rgbs[it]
        with FixedNoise([t_rand, u_rand]):
            outs = net(rays=rays, randomized=True, white_bkgd=False, use_ort_loss=False)
        loss = ref_loss_mip(outs, rays.lossmult, gt, False)
        opt.zero_grad(); loss.backward(); opt.step(); sch.step()
        losses.append(float(loss))
        if step % 50 == 0: print(step, float(loss), time.time() - t0, flush=True)
    # held-out view: camera 2, every 8th pixel
    hold = torch.arange(2 * H * W, 3 * H * W, 8)
    rays = Rays(*[x[hold] for x in flat])
    with torch.no_grad():
        outs = net(rays=rays, randomized=False, white_bkgd=False, use_ort_loss=False)
    pred = outs[1][0]
    psnr = float(-10.0 * torch.log10(torch.mean((rsurf.hdr_to_ldr(pred) - rsurf.hdr_to_ldr(rgbs[hold])) ** 2)))
    print("psnr", psnr)
    np.savez_compressed(os.path.join(HERE, "psnr_trace_mip.npz"), losses=np.array(losses, np.float64), psnr=np.float64(psnr),
                        steps=np.int64(STEPS), B=np.int64(B), N=np.int64(N), H=np.int64(H), W=np.int64(W),
                        pred_head=pred[:64].numpy())

if __name__ == "__main__":
    main()
