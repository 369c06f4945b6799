"""Pins oracle/pano_oracle.py against the golden vectors captured from the
reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import pano_oracle as orc

T = lambda a: torch.from_numpy(np.asarray(a))
CASES = ["B64_N32", "B16_N128"]
NAMES = ("comp_rgb", "distance", "ort_loss", "normal", "albedo", "roughness", "surface_rgb", "diffuse", "shading")


def rays_of(g):
    return orc.Rays(*[T(g["ray_" + k]) for k in orc.Rays._fields])


def env_of(golden):
    g = golden("raygen_8x16")
    return orc.Rays(*[T(g["env_" + k]) for k in orc.Rays._fields])


def test_raygen_matches_reference(golden):
    g = golden("raygen_8x16")
    rays, radius = orc.generate_pano_rays(8, 16, list(g["c2ws"]))
    for k in orc.Rays._fields:
        got = np.stack(getattr(rays, k))
        assert got.shape == g[k].shape
        np.testing.assert_allclose(got, g[k], rtol=1e-6, atol=1e-7, err_msg=k)
    assert abs(float(radius) - float(g["radius"])) < 1e-12
    env = orc.generate_lit_rays(10, radius)
    for k in orc.Rays._fields:
        assert getattr(env, k).dtype == torch.float16
        np.testing.assert_array_equal(getattr(env, k).numpy(), g["env_" + k], err_msg=k)
    g2 = golden("raygen_64x128")
    rays2, radius2 = orc.generate_pano_rays(64, 128, list(g2["c2ws"]))
    for k in orc.Rays._fields:
        np.testing.assert_allclose(getattr(rays2, k)[0][::7, ::9], g2[k], rtol=1e-6, atol=1e-7, err_msg=k)
    assert abs(float(radius2) - float(g2["radius"])) < 1e-12
    assert abs(float(radius2) - 2 * np.pi / (128 * np.sqrt(3))) < 1e-4 * float(radius2) * 10


def test_sampling_helper_roundtrip(golden):
    """utils/sampling.py:157-161 prints 'test' after a direction -> spherical -> direction round trip;
    the property asserted here is unit length + the same formula as the dataset."""
    g = golden("sampling_helpers")
    rays, _ = orc.generate_pano_rays(8, 16, [np.eye(4, dtype=np.float32)])
    np.testing.assert_allclose(rays.directions[0], g["pano_dirs"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(np.linalg.norm(g["pano_dirs"], axis=-1), 1.0, atol=1e-6)
    env = orc.generate_lit_rays(10, 0.01, dtype=torch.float64)
    np.testing.assert_allclose(env.directions.numpy(), g["uniform_dirs"], atol=1e-12)


@pytest.mark.parametrize("case", CASES)
def test_stage_parity(golden, case):
    g = golden("stages_" + case)
    rays = rays_of(g)
    N = g["t_det"].shape[1] - 1
    t, (m, c) = orc.sample_along_rays(rays.origins, rays.directions, rays.radii, N, rays.near, rays.far)
    assert rel_err(t, g["t_det"]) < 1e-6 and rel_err(m, g["mean_det"]) < 1e-6 and rel_err(c, g["cov_det"]) < 1e-5
    t, (m, c) = orc.sample_along_rays(rays.origins, rays.directions, rays.radii, N, rays.near, rays.far,
                                      T(g["t_rand"]))
    assert rel_err(t, g["t_rnd"]) < 1e-6 and rel_err(m, g["mean_rnd"]) < 1e-6 and rel_err(c, g["cov_rnd"]) < 1e-5
    enc = orc.integrated_pos_enc(m, c, 0, 16)
    assert enc.shape[-1] == 96
    assert rel_err(enc[:4], g["enc_head"]) < 1e-6
    venc = orc.pos_enc(rays.viewdirs, 0, 4)
    assert rel_err(venc, g["viewenc"]) < 1e-6
    p = orc.init_params(4, 5)
    raw_rgb, raw_den = orc.mlp_forward(p, enc, venc)
    assert rel_err(raw_rgb, g["raw_rgb"]) < 1e-5 and rel_err(raw_den, g["raw_den"]) < 1e-5
    sp = torch.nn.functional.softplus
    rgb, sig = sp(T(g["raw_rgb"])), sp(T(g["raw_den"])[..., :1] - 1)
    comp, dist, acc, w = orc.volumetric_rendering(rgb, sig, T(g["t_rnd"]), rays.directions, False)
    for a, k in ((comp, "comp_rgb"), (dist, "distance"), (acc, "acc"), (w, "weights")):
        assert rel_err(a, g[k]) < 1e-6, k
    compw = orc.volumetric_rendering(rgb, sig, T(g["t_rnd"]), rays.directions, True)[0]
    assert rel_err(compw, g["comp_rgb_white"]) < 1e-6
    t2, (m2, c2) = orc.resample_along_rays(rays.origins, rays.directions, rays.radii, T(g["t_rnd"]), T(g["weights"]),
                                           0.01)
    assert rel_err(t2, g["t_resample_det"]) < 1e-6 and rel_err(m2, g["mean_resample_det"]) < 1e-6
    assert rel_err(c2, g["cov_resample_det"]) < 1e-5
    t3, _ = orc.resample_along_rays(rays.origins, rays.directions, rays.radii, T(g["t_rnd"]), T(g["weights"]), 0.01,
                                    T(g["u_rand"]))
    assert rel_err(t3, g["t_resample_rnd"]) < 1e-6
    assert bool((t3[:, 1:] >= t3[:, :-1]).all())
    t4, _ = orc.resample_along_rays(rays.origins, rays.directions, rays.radii, T(g["t_rnd"]),
                                    torch.zeros_like(T(g["weights"])), 0.0)
    assert rel_err(t4, g["t_resample_zero"]) < 1e-6
    env = orc.Rays(*[x.float() for x in env_of(golden)])
    xs = rays.origins + rays.directions * T(g["distance"]).view(-1, 1)
    lt, (lm, lc), ld = orc.sample_each_points(xs, env.directions, 10, env.near, env.far, env.radii, T(g["env_rand"]))
    assert rel_err(lt[:40], g["lit_t"]) < 1e-6 and rel_err(lm[:40], g["lit_mean"]) < 1e-6
    assert rel_err(lc[:40], g["lit_cov"]) < 1e-5 and rel_err(ld[:40], g["lit_dirs"]) == 0
    B = rays.origins.shape[0]
    srgb, dif, shd = orc.surface_rendering(T(g["sr_env"]), T(g["sr_albedo"]), T(g["sr_normal"]), ld.view(B, 10, 3),
                                           env.lossmult)
    assert rel_err(srgb, g["sr_rgb"]) < 1e-6 and rel_err(dif, g["sr_diffuse"]) < 1e-6
    assert rel_err(shd, g["sr_shading"]) < 1e-6
    assert rel_err(orc.hdr_to_ldr(T(g["tm_in"])), g["tm_out"]) < 1e-6
    assert rel_err(orc.hdr_to_ldr(T(g["tm_in"]), quantize=True), g["tm_out_u8"]) < 1e-6


def _check_outputs(outs, g, prefix, tol=1e-4, loose=("normal", "surface_rgb", "diffuse", "shading", "ort_loss")):
    for lvl, tup in enumerate(outs):
        for nme, v in zip(NAMES, tup):
            key = f"{prefix}/l{lvl}/{nme}"
            if v is None:
                assert key not in g, key
                continue
            e = rel_err(v.detach(), g[key])
            # outputs derived from the density gradient are ill-conditioned in fp32
            # (the reference disagrees with its own fp64 run by up to 6.5e-2 abs on `normal`).
            assert e < (5e-2 if nme in loose else tol), (key, e)
            if nme in loose and v.dim() > 0:
                med = float(np.median(np.abs(v.detach().numpy() - g[key]) / (np.abs(g[key]) + 1e-6)))
                assert med < 1e-4, (key, med)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mode", ["fast", "faithful"])
def test_pano_forward_and_grad(golden, case, mode):
    if mode == "faithful" and case == "B16_N128":
        pytest.skip("faithful mode covered at B64_N32 (it is slow by construction)")
    g = golden("pano_full_" + case)
    s = golden("stages_" + case)
    rays = rays_of(s)
    N = s["t_det"].shape[1] - 1
    env = env_of(golden)
    p = {k: v.clone().requires_grad_(True) for k, v in orc.init_params(4, 5).items()}
    outs = orc.pano_forward(p, rays, env, num_samples=N, normals_mode=mode)
    _check_outputs(outs, g, "val")
    noise = dict(t_rand=T(g["train_t_rand"]), u_rand=T(g["train_u_rand"]), env_rand=T(g["train_env_rand"]))
    outs = orc.pano_forward(p, rays, env, num_samples=N, noise=noise, normals_mode=mode)
    _check_outputs(outs, g, "train")
    loss = orc.pano_loss(outs, rays.lossmult, T(s["rgbs"]))
    assert abs(float(loss) - float(g["train/loss"])) < 1e-4 * abs(float(g["train/loss"]))
    grads = torch.autograd.grad(loss, list(p.values()))
    for (k, _), gr in zip(p.items(), grads):
        ref_norm = float(g[f"train/grad/{k}/norm"])
        got = gr.reshape(-1)
        assert abs(float(got.double().norm()) - ref_norm) < 2e-2 * ref_norm + 1e-9, k
        idx = g[f"train/grad/{k}/idx"]
        ref = g[f"train/grad/{k}/val"]
        err = np.abs(got[idx].numpy() - ref)
        assert float(np.median(err)) < 1e-3 * max(float(np.max(np.abs(ref))), 1e-12), k
    outs2 = orc.pano_forward(p, rays, env, num_samples=N, white_bkgd=True, enable_surf=False, use_ort_loss=False)
    assert [i for i, v in enumerate(outs2[1]) if v is None] == list(g["nosurf/none_slots"])
    assert rel_err(outs2[1][0].detach(), g["nosurf/l1/comp_rgb"]) < 1e-4


@pytest.mark.parametrize("case", CASES)
def test_mip_forward_and_grad(golden, case):
    g = golden("mip_full_" + case)
    s = golden("stages_" + case)
    rays = rays_of(s)
    N = s["t_det"].shape[1] - 1
    p = {k: v.clone().requires_grad_(True) for k, v in orc.init_params(4, 1).items()}
    names = ("comp_rgb", "distance", "ort_loss", "normal")
    for mode, use_ort, noise_key in (("val", True, None), ("valno", False, None), ("train", False, "train"),
                                     ("trainort", True, "trainort")):
        noise = None
        if noise_key:
            noise = dict(t_rand=T(g[noise_key + "_t_rand"]), u_rand=T(g[noise_key + "_u_rand"]))
        outs = orc.mip_forward(p, rays, num_samples=N, use_ort_loss=use_ort, noise=noise)
        for lvl, tup in enumerate(outs):
            for nme, v in zip(names, tup):
                key = f"{mode}/l{lvl}/{nme}"
                if v is None:
                    assert key not in g
                    continue
                tol = 5e-2 if (nme in ("normal", "ort_loss") and lvl == 1 and use_ort) else 1e-4
                assert rel_err(v.detach(), g[key]) < tol, key
        if noise_key:
            loss = orc.mip_loss(outs, rays.lossmult, T(s["rgbs"]), use_ort=use_ort)
            assert abs(float(loss) - float(g[mode + "/loss"])) < 1e-4 * abs(float(g[mode + "/loss"]))
            grads = torch.autograd.grad(loss, list(p.values()))
            for (k, _), gr in zip(p.items(), grads):
                ref_norm = float(g[f"{mode}/grad/{k}/norm"])
                assert abs(float(gr.double().norm()) - ref_norm) < 2e-2 * ref_norm + 1e-9, (mode, k)


def test_lr_schedule(golden):
    g = golden("lr_schedule")
    for step, lr in zip(g["steps"], g["lrs"]):
        assert abs(orc.mip_lr(int(step)) - float(lr)) < 1e-12 + 1e-9 * float(lr)


def test_package_lr_schedule_matches_the_reference(golden):
    """pano_nerf_amd.optim.mip_lr (the product's copy, used by bench.py and the training tests) against the values the
    reference's MipLRDecay produced (utils/lr_schedule.py:51-59; tests/golden/make_golden.py)."""
    from pano_nerf_amd.optim import mip_lr
    g = golden("lr_schedule")
    assert len(g["steps"]) >= 4
    for step, lr in zip(g["steps"], g["lrs"]):
        assert abs(mip_lr(int(step)) - float(lr)) < 1e-12 + 1e-9 * float(lr), (int(step), mip_lr(int(step)), float(lr))


@pytest.mark.parametrize("case", ["B64_N32", "B16_N128"])
def test_oracle_full_gradients_match_the_reference(golden, case):
    """Every entry of d loss / d params (614 k values) of the oracle against the imported reference's fp32 gradients
    (tests/golden/make_grad_golden.py): first-order tensors to 1e-4 of the tensor max over ALL entries; tensors the
    second-order path touches on the median and the 99 % quantile (the reference differs from its own fp64 run there)."""
    g, s, gg = golden("pano_full_" + case), golden("stages_" + case), golden("grads_pano_" + case)
    N = s["t_det"].shape[1] - 1
    rays = orc.Rays(*[torch.from_numpy(s["ray_" + k]) for k in orc.Rays._fields])
    ge = golden("raygen_8x16")
    env = orc.Rays(*[torch.from_numpy(ge["env_" + k]).float() for k in orc.Rays._fields])
    params = {k: v.clone().requires_grad_(True) for k, v in orc.init_params(4, 5).items()}
    noise = dict(t_rand=torch.from_numpy(g["train_t_rand"]), u_rand=torch.from_numpy(g["train_u_rand"]),
                 env_rand=torch.from_numpy(g["train_env_rand"]))
    outs = orc.pano_forward(params, rays, env, num_samples=N, noise=noise)
    loss = orc.pano_loss(outs, rays.lossmult, torch.from_numpy(s["rgbs"]))
    loss.backward()
    assert abs(float(loss) - float(gg["loss32"])) < 1e-5 * abs(float(gg["loss32"]))
    order = ([f"layers.{i}.0.{k}" for i in range(8) for k in ("weight", "bias")] +
             ["extra_layer.weight", "extra_layer.bias", "view_layers.0.0.weight", "view_layers.0.0.bias",
              "density_layer.weight", "color_layer.weight", "density_layer.bias", "color_layer.bias"])
    o = 0
    ref = gg["g32"].astype(np.float64)
    for k in order:
        got = params[k].grad.detach().reshape(-1).numpy().astype(np.float64)
        r = ref[o:o + got.size]
        o += got.size
        scale = max(float(np.abs(r).max()), 1e-30)
        err = np.abs(got - r) / scale
        if k.startswith(("extra_layer", "view_layers", "color_layer")):
            assert float(err.max()) <= 1e-4, (k, float(err.max()))
        else:
            assert float(np.median(err)) <= 1e-4 and float(np.mean(err <= 1e-3)) >= 0.99, (k, float(np.median(err)))
    assert o == ref.size


def test_oracle_disparity_sampling_and_forward(golden):
    """`disparity=True` (coarse samples linear in inverse depth, models/mip.py:134-136): the oracle against the reference's
    sample_along_rays and against the val-mode forward tuples of both models built with disparity=True
    (tests/golden/make_disparity_golden.py)."""
    g = golden("disparity_B16_N32")
    rays = orc.Rays(*[torch.from_numpy(g["ray_" + k]) for k in orc.Rays._fields])
    N = g["t_det"].shape[1] - 1
    for t_rand, kt, km, kc in ((None, "t_det", "mean_det", "cov_det"), (torch.from_numpy(g["t_rand"]), "t_rnd", "mean_rnd", "cov_rnd")):
        t, (m, c) = orc.sample_along_rays(rays.origins, rays.directions, rays.radii, N, rays.near, rays.far, t_rand, disparity=True)
        assert rel_err(t, g[kt]) < 1e-6 and rel_err(m, g[km]) < 1e-6 and rel_err(c, g[kc]) < 1e-5
    env = orc.generate_lit_rays(10, float(orc.synthetic_scene(8, 16, 3, seed=4)[2]))
    with torch.no_grad():
        outs = orc.pano_forward(orc.init_params(4, 5), rays, env, num_samples=N, disparity=True)
        mouts = orc.mip_forward(orc.init_params(4, 1), rays, num_samples=N, disparity=True)
    names = ("comp_rgb", "distance", "ort_loss", "normal", "albedo", "roughness", "surface_rgb", "diffuse", "shading")
    for lvl, tup in enumerate(outs):
        for n, v in zip(names, tup):
            if v is not None:
                assert rel_err(v, g[f"pano/l{lvl}/{n}"]) < (5e-3 if n in ("normal", "surface_rgb", "diffuse", "shading", "ort_loss") else 1e-5), (lvl, n)
    for lvl in (0, 1):
        assert rel_err(mouts[lvl][0], g[f"mip/l{lvl}/comp_rgb"]) < 1e-5 and rel_err(mouts[lvl][1], g[f"mip/l{lvl}/distance"]) < 1e-5


def test_oracle_disable_integration(golden):
    """`disable_integration=True` (compute_graph zeroes the covariance: models/pano_mip_nerf.py:241-243, models/mip_nerf.py:213-214):
    the oracle against the reference's val-mode tuples of both models, and its train-mode loss and full gradient
    (tests/golden/make_disint_golden.py)."""
    g = golden("disable_integration_B16_N32")
    rays = orc.Rays(*[torch.from_numpy(g["ray_" + k]) for k in orc.Rays._fields])
    N = g["t_rand"].shape[1] - 1
    env = orc.generate_lit_rays(10, float(orc.synthetic_scene(8, 16, 3, seed=4)[2]))
    with torch.no_grad():
        outs = orc.pano_forward(orc.init_params(4, 5), rays, env, num_samples=N, disable_integration=True)
        mouts = orc.mip_forward(orc.init_params(4, 1), rays, num_samples=N, use_ort_loss=True, disable_integration=True)
        plain = orc.pano_forward(orc.init_params(4, 5), rays, env, num_samples=N)
    names = ("comp_rgb", "distance", "ort_loss", "normal", "albedo", "roughness", "surface_rgb", "diffuse", "shading")
    for lvl, tup in enumerate(outs):
        for n, v in zip(names, tup):
            if v is not None:
                assert rel_err(v, g[f"pano/l{lvl}/{n}"]) < (5e-3 if n in ("normal", "surface_rgb", "diffuse", "shading", "ort_loss") else 1e-5), (lvl, n)
    assert rel_err(plain[1][0], g["pano/l1/comp_rgb"]) > 1e-3  # (the flag changes the render: the fixture is not the default's)
    for lvl in (0, 1):
        assert rel_err(mouts[lvl][0], g[f"mip/l{lvl}/comp_rgb"]) < 1e-5 and rel_err(mouts[lvl][1], g[f"mip/l{lvl}/distance"]) < 1e-5
    assert rel_err(mouts[1][3], g["mip/l1/normal"]) < 5e-3
    p = {k: v.clone().requires_grad_(True) for k, v in orc.init_params(4, 5).items()}
    noise = dict(t_rand=torch.from_numpy(g["t_rand"]), u_rand=torch.from_numpy(g["u_rand"]), env_rand=torch.from_numpy(g["env_rand"]))
    touts = orc.pano_forward(p, rays, env, num_samples=N, noise=noise, disable_integration=True)
    loss = orc.pano_loss(touts, rays.lossmult, torch.from_numpy(g["rgbs"]))
    assert abs(float(loss) - float(g["train/loss"])) < 1e-5 * abs(float(g["train/loss"]))
    grads = torch.autograd.grad(loss, list(p.values()))
    for (k, _), gr in zip(p.items(), grads):
        r = g["train/grad/" + k]
        scale = max(float(np.abs(r).max()), 1e-30)
        err = np.abs(gr.numpy() - r) / scale
        if k.startswith(("extra_layer", "view_layers", "color_layer")):
            assert float(err.max()) <= 1e-4, (k, float(err.max()))
        else:
            assert float(np.median(err)) <= 1e-4 and float(np.mean(err <= 1e-3)) >= 0.99, (k, float(np.median(err)))
