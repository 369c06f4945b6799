"""Trains the same PanoMipNeRF from the same weights on the same synthetic batches in every MLP mode and prints the
held-out PSNR curve: the default fp16-pair arithmetic must train like the exact-fp32 layer-wise path (GPU box).
usage: python3 tools/train_compare.py [steps=300]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import pano_nerf_amd as pn
from oracle import pano_oracle as orc  # (checker: synthetic scene + initial weights only)

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
H, W, B, N = 32, 64, 256, 64
flat, rgbs, radius, _ = orc.synthetic_scene(H, W, 3, seed=4)
flat_d = pn.Rays(*[x.to(dev) for x in flat]); rgbs_d = rgbs.to(dev)
env = pn.generate_lit_rays(10, radius)
hold = torch.arange(2 * H * W, 3 * H * W, 4, device=dev)
for mode in ("layerwise", "fused", "fused_f16x2", "fused_bf16"):
    torch.manual_seed(0)
    model = pn.PanoMipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5, num_env_samples=10)
    model.mlp.load_state_dict(orc.init_params(4, 5)); model = model.to(dev); model.mlp_mode = mode
    opt = pn.FlatAdam(model.mlp, lr=2e-4)
    rng = np.random.Generator(np.random.PCG64(7))
    curve = []
    for step in range(steps):
        it = torch.from_numpy(rng.integers(0, 2 * H * W, size=B)).to(dev)
        rays = pn.Rays(*[x[it] for x in flat_d])
        model.noise_override = dict(t_rand=torch.from_numpy(rng.random((B, N + 1), dtype=np.float32)),
                                    u_rand=torch.from_numpy(rng.random((B, N + 1), dtype=np.float32) * np.float32(1.0 / (N + 1) - 1.2e-7)),
                                    env_rand=torch.from_numpy(rng.random((1, 11), dtype=np.float32)))
        opt.zero_grad()
        outs = model(rays=rays, env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
        loss, _ = pn.pano_loss(outs, rays.lossmult, rgbs_d[it])
        loss.backward()
        opt.step(lr=pn.mip_lr(step * 40))  # (a faster warm-up than the 120-step schedule: the run is short)
        if (step + 1) % (steps // 5) == 0:
            model.noise_override = None
            with torch.no_grad():
                o = model(rays=pn.Rays(*[x[hold] for x in flat_d]), env_rays=env, randomized=False, white_bkgd=False,
                          enable_surf=True, use_ort_loss=True)
            curve.append((step + 1, round(float(loss), 5), round(pn.loss.hdr_to_ldr_psnr(o[1][0], rgbs_d[hold]), 3),
                          round(pn.loss.hdr_to_ldr_psnr(o[1][6], rgbs_d[hold]), 3)))
    print(f"{mode:12s} (step, loss, PSNR, surface PSNR): {curve}")
