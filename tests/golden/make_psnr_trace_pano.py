"""Train the IMPORTED reference PanoMipNeRF on CPU with the FULL panonerf_system loss (coarse + fine + surface + chromaticity
+ orientation: systems/panonerf_system.py:15-75; second-order gradients through the density-gradient normals) for 100
steps of 64 rays x 32 samples on the synthetic 64x128 scene — fixed batches and all three noise draws from PCG64 — and
store the loss-per-step trace and the held-out-view PSNR.  Build container only; same import shim as make_golden.py."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
from make_psnr_trace import FixedNoise  # noqa: E402
from utils.lr_schedule import MipLRDecay  # noqa: E402

orc, rpano, rsurf, Rays = mg.orc, mg.rpano, mg.rsurf, mg.Rays
STEPS, B, N, H, W = 100, 64, 32, 64, 128


def schedule(seed=11):
    """Per-step batch indices and noise, regenerated identically by tests/test_gpu_psnr.py."""
    rng = np.random.Generator(np.random.PCG64(seed))
    for _ in range(STEPS):
        idx = rng.integers(0, 2 * H * W, size=B)  # cameras 0 and 1 train, camera 2 is held out
        t_rand = rng.random((B, N + 1), dtype=np.float32)
        u_rand = rng.random((B, N + 1), dtype=np.float32) * np.float32(1.0 / (N + 1) - 1.1920929e-07)
        env_rand = rng.random((1, 11), dtype=np.float32)
        yield idx, t_rand, u_rand, env_rand


def perturbed_init(seed):
    """The initial weights with EVERY element moved by one ulp up or down (sign from PCG64(seed)); seed None: unchanged."""
    params = orc.init_params(4, 5)
    if seed is None:
        return params
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for k, v in params.items():
        up = torch.from_numpy(rng.integers(0, 2, size=tuple(v.shape)).astype(np.bool_))
        inf = torch.full_like(v, float("inf"))
        out[k] = torch.where(up, torch.nextafter(v, inf), torch.nextafter(v, -inf))
    return out


def train(dtype, perturb_seed=None):
    torch.set_default_dtype(dtype)
    try:
        flat, rgbs, radius, _ = orc.synthetic_scene(H, W, 3, seed=4)
        flat = Rays(*[x.to(dtype) for x in flat])
        rgbs = rgbs.to(dtype)
        env = orc.generate_lit_rays(10, radius)
        envf = Rays(*[x.float().to(dtype) for x in env])
        net = rpano.PanoMipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5,
                                num_env_samples=10).to(dtype)
        mg.load_params(net.mlp, {k: v.to(dtype) for k, v in perturbed_init(perturb_seed).items()})
        opt = torch.optim.Adam(net.mlp.parameters(), lr=2e-4)
        sch = MipLRDecay(opt, 2e-4, 2e-5, 44000, 120, 0.01)
        losses = []
        t0 = time.time()
        for step, (idx, t_rand, u_rand, env_rand) in enumerate(schedule()):
            it = torch.from_numpy(idx)
            rays = Rays(*[x[it] for x in flat])
            gt = rgbs[it]
            draws = [t_rand, u_rand, env_rand] if dtype == torch.float32 else [d.astype(np.float64) for d in (t_rand, u_rand, env_rand)]
            with FixedNoise(draws):
                outs = net(rays=rays, env_rays=envf, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
            loss = mg.ref_loss_pano(outs, rays.lossmult, gt)
            opt.zero_grad(); loss.backward(); opt.step(); sch.step()
            losses.append(float(loss))
            if step % 10 == 0:
                print(dtype, step, float(loss), time.time() - t0, flush=True)
        hold = torch.arange(2 * H * W, 3 * H * W, 16)
        rays = Rays(*[x[hold] for x in flat])
        outs = net(rays=rays, env_rays=envf, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
        pred, surf = outs[1][0].detach(), outs[1][6].detach()
        psnr = float(-10.0 * torch.log10(torch.mean((rsurf.hdr_to_ldr(pred) - rsurf.hdr_to_ldr(rgbs[hold])) ** 2)))
        psnr_surf = float(-10.0 * torch.log10(torch.mean((rsurf.hdr_to_ldr(surf) - rsurf.hdr_to_ldr(rgbs[hold])) ** 2)))
        return np.array(losses, np.float64), psnr, psnr_surf, pred[:64].float().numpy()
    finally:
        torch.set_default_dtype(torch.float32)


def main():
    torch.set_num_threads(8)
    l32, p32, s32, head = train(torch.float32)
    # the SAME reference, weights, batches and noise in fp64: how far its own trajectory moves under a change of
    # arithmetic precision alone (ReLU-gate flips through the second-order path) — the yardstick of the GPU test
    l64, p64, s64, _ = train(torch.float64)
    print("psnr", p32, s32, "fp64", p64, s64)
    np.savez_compressed(os.path.join(HERE, "psnr_trace_pano.npz"), losses=l32, psnr=np.float64(p32),
                        psnr_surface=np.float64(s32), losses64=l64, psnr64=np.float64(p64), psnr_surface64=np.float64(s64),
                        steps=np.int64(STEPS), B=np.int64(B), N=np.int64(N), H=np.int64(H), W=np.int64(W), pred_head=head)


if __name__ == "__main__":
    main()
