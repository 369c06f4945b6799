"""GPU parity of the drop-in modules: PanoMipNeRF / MipNeRF forward tuples, training loss and parameter
gradients against the golden vectors captured from the reference; chunked full-image render;
size-independent properties at the bench size."""
import numpy as np
import pytest
import torch

from conftest import assert_close, forced_gate_sets, rel_err
from oracle import pano_oracle as orc

pytestmark = pytest.mark.gpu
CASES = ["B64_N32", "B16_N128"]
NAMES9 = ("comp_rgb", "distance", "ort_loss", "normal", "albedo", "roughness", "surface_rgb", "diffuse", "shading")
# outputs derived from the density gradient: the reference fp32 disagrees with its own fp64 run by up to
# 6.5e-2 abs, so they are gated on median + vs-fp64 criteria instead of a pointwise 1e-4
LOOSE = ("normal", "surface_rgb", "diffuse", "shading", "ort_loss")


def dev():
    return torch.device("cuda:0")


def to_dev(rays):
    return type(rays)(*[x.to(dev()) for x in rays])


def rays_of(g):
    return orc.Rays(*[torch.from_numpy(g["ray_" + k]) for k in orc.Rays._fields])


def env_of(golden):
    g = golden("raygen_8x16")
    return orc.Rays(*[torch.from_numpy(g["env_" + k]) for k in orc.Rays._fields])


def make_pano(N, nc=5):
    import pano_nerf_amd as pn
    cls = pn.PanoMipNeRF if nc == 5 else pn.MipNeRF
    m = cls(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=nc, num_env_samples=10)
    m.mlp.load_state_dict(orc.init_params(4, nc))
    return m.to(dev())


def check_tuple(outs, g, prefix, names, g64=None):
    for lvl, tup in enumerate(outs):
        for nme, v in zip(names, tup):
            key = f"{prefix}/l{lvl}/{nme}"
            if v is None:
                assert key not in g, key
                continue
            assert key in g, key
            got = v.detach().cpu().numpy()
            e = rel_err(got, g[key])
            if nme in LOOSE and lvl == 1:
                assert e < 5e-2, (key, e)
                if got.ndim > 0:
                    rel = np.abs(got - g[key]) / (np.abs(g[key]) + 1e-6)
                    med = float(np.median(rel))
                    assert med < 1e-4, (key, med)
                    # >= 99 % of the elements within 1e-3.  The goldens hold 16 / 64 rays, where 1 % is less
                    # than one ray: three rays (9 elements) may sit on a flipped ReLU gate (0-3 gates flip per run under ANY
                    # fp32 summation order; one ray does in the reference's own fp32-vs-fp64 comparison on these batches).
                    # The pointwise statement is tests/test_gpu_grads.py::test_gate_consistent_*: with the kernels' gate
                    # decisions forced into the oracle, every output agrees to 1e-4
                    bad = int(np.sum(np.abs(got - g[key]) > 1e-3 * max(float(np.max(np.abs(g[key]))), 1e-12)))
                    assert bad <= max(9, int(0.01 * got.size)), (key, bad, got.size)
                k64 = f"val64/l1/{nme}"
                if g64 is not None and k64 in g64 and prefix == "val" and got.ndim > 0:
                    # vs the reference's own fp64 run: a ReLU gate whose pre-activation is ~1e-7 can flip under
                    # any fp32 summation order (ours or the reference's), which moves ONE ray by ~1e-3; so the
                    # gate is on the per-ray error with the worst 5 % of rays (at least one) set aside
                    ours = np.sort(np.abs(got - g64[k64]).reshape(got.shape[0], -1).max(-1))
                    theirs = np.sort(np.abs(g[key] - g64[k64]).reshape(got.shape[0], -1).max(-1))
                    drop = max(3, int(np.ceil(0.05 * got.shape[0])))  # rays allowed to sit on a flipped gate
                    assert ours[-drop - 1] <= 2 * theirs[-drop - 1] + 1e-5, (key, ours[-3:], theirs[-3:])
            else:
                assert_close(got, g[key], key)  # tensor-scale 1e-4 and element-wise 1e-4 where |ref| > 1e-3 max


def check_grads(model, g, prefix):
    """Per-tensor summaries captured by make_golden.py (norm + 256 sampled entries).  The full-gradient gates — every
    entry of every tensor, fp32 and fp64 references, gate-consistent pointwise parity — are in tests/test_gpu_grads.py."""
    for k, p in model.mlp.named_parameters():
        ref_norm = float(g[f"{prefix}/grad/{k}/norm"])
        got = p.grad.detach().cpu().reshape(-1)
        assert torch.isfinite(got).all(), k
        assert abs(float(got.double().norm()) - ref_norm) < 2e-3 * ref_norm + 1e-9, (k, float(got.double().norm()), ref_norm)
        idx, ref = g[f"{prefix}/grad/{k}/idx"], g[f"{prefix}/grad/{k}/val"]
        err = np.abs(got[idx].numpy() - ref)
        assert float(np.median(err)) < 1e-4 * max(float(np.max(np.abs(ref))), 1e-12), (k, float(np.median(err)))


MODES = ["fused", "fused_f16x2", "fused_f16x2_t32", "layerwise"]  # on-chip chains with the exact 3-term bf16 split (default) / one fp32-MFMA GEMM per layer


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", CASES)
def test_pano_forward_loss_grads(golden, case, mode):
    import pano_nerf_amd as pn
    g, s = golden("pano_full_" + case), golden("stages_" + case)
    N = s["t_det"].shape[1] - 1
    rays, env = to_dev(rays_of(s)), to_dev(env_of(golden))
    model = make_pano(N)
    model.mlp_mode = mode
    with torch.no_grad():
        outs = model(rays=rays, env_rays=env, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    assert isinstance(outs, list) and len(outs) == 2 and all(len(t) == 9 for t in outs)
    assert outs[1][1].shape == (rays.origins.shape[0],)
    check_tuple(outs, g, "val", NAMES9, g64=g)
    model.noise_override = dict(t_rand=torch.from_numpy(g["train_t_rand"]), u_rand=torch.from_numpy(g["train_u_rand"]),
                                env_rand=torch.from_numpy(g["train_env_rand"]))
    outs = model(rays=rays, env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    check_tuple(outs, g, "train", NAMES9)
    loss, _ = pn.pano_loss(outs, rays.lossmult, torch.from_numpy(s["rgbs"]).to(dev()))
    assert abs(float(loss) - float(g["train/loss"])) < 1e-4 * abs(float(g["train/loss"]))
    loss.backward()
    check_grads(model, g, "train")
    assert model.mlp.last_flat_grad is not None and model.mlp.is_flat()
    # None-slot contract with surface / orientation off, white background on
    model.noise_override = None
    with torch.no_grad():
        outs2 = model(rays=rays, env_rays=env, randomized=False, white_bkgd=True, enable_surf=False, use_ort_loss=False)
    assert [i for i, v in enumerate(outs2[1]) if v is None] == list(g["nosurf/none_slots"])
    assert rel_err(outs2[1][0].cpu(), g["nosurf/l1/comp_rgb"]) < 1e-4
    assert all(v is None for v in outs2[0][2:])


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", CASES)
def test_mip_forward_loss_grads(golden, case, mode):
    import pano_nerf_amd as pn
    g, s = golden("mip_full_" + case), golden("stages_" + case)
    N = s["t_det"].shape[1] - 1
    rays = to_dev(rays_of(s))
    model = make_pano(N, nc=1)
    model.mlp_mode = mode
    names = ("comp_rgb", "distance", "ort_loss", "normal")
    for mode, use_ort in (("val", True), ("valno", False)):
        with torch.no_grad():
            outs = model(rays=rays, randomized=False, white_bkgd=False, use_ort_loss=use_ort)
        assert all(len(t) == 4 for t in outs)
        check_tuple(outs, g, mode, names)
    for mode, use_ort in (("train", False), ("trainort", True)):
        model.noise_override = dict(t_rand=torch.from_numpy(g[mode + "_t_rand"]),
                                    u_rand=torch.from_numpy(g[mode + "_u_rand"]))
        for p in model.mlp.parameters():
            p.grad = None
        outs = model(rays=rays, randomized=True, white_bkgd=False, use_ort_loss=use_ort)
        check_tuple(outs, g, mode, names)
        loss, _ = pn.mip_loss(outs, rays.lossmult, torch.from_numpy(s["rgbs"]).to(dev()), use_ort=use_ort)
        assert abs(float(loss) - float(g[mode + "/loss"])) < 1e-4 * abs(float(g[mode + "/loss"]))
        loss.backward()
        check_grads(model, g, mode)


def test_render_image_chunks(golden):
    """render_image contract (systems/panonerf_system.py:133-192): chunk, render, concatenate."""
    import pano_nerf_amd as pn
    g = golden("render_image_8x16")
    rg = golden("raygen_8x16")
    rays = pn.generate_pano_rays(8, 16, rg["c2ws"][0])
    env = pn.generate_lit_rays(10, pn.rays.pano_pixel_radius(rays))
    img_rays = pn.Rays(*[x.view(1, 8, 16, -1) for x in rays])
    chunks, _ = pn.rearrange_render_image(img_rays, 32)
    assert len(chunks) == int(g["n_chunks"])
    model = make_pano(32)
    keep = {k: [] for k in ("coarse_rgb", "fine_rgb", "coarse_dep", "fine_dep", "normal", "albedo", "surface_rgb",
                            "shading")}
    with torch.no_grad():
        for ch in chunks:
            (c_rgb, c_dep, *_), (f_rgb, f_dep, _, f_nor, alb, _, sf, _, sd) = model(
                rays=ch, env_rays=env, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
            for k, v in zip(keep, (c_rgb, f_rgb, c_dep, f_dep, f_nor, alb, sf, sd)):
                keep[k].append(v)
    for k, v in keep.items():
        img = torch.cat(v, 0).view(1, 8, 16, -1).permute(0, 3, 1, 2).cpu().numpy()
        e = rel_err(img, g[k])
        if k in ("normal", "surface_rgb", "shading"):
            assert e < 5e-2, (k, e)
            assert float(np.median(np.abs(img - g[k]) / (np.abs(g[k]) + 1e-6))) < 1e-4, k
        else:
            assert e < 1e-4, (k, e)


def test_render_image_helper_matches_chunk_loop(golden):
    """pano_nerf_amd.render_image with a different chunk size gives the same image as the reference's 32-ray loop."""
    import pano_nerf_amd as pn
    g = golden("render_image_8x16")
    rg = golden("raygen_8x16")
    rays = pn.generate_pano_rays(8, 16, rg["c2ws"][0])
    env = pn.generate_lit_rays(10, pn.rays.pano_pixel_radius(rays))
    model = make_pano(32)
    out = pn.render_image(model, pn.Rays(*[x.view(1, 8, 16, -1) for x in rays]), env, 8, 16, chunk_size=48)
    names = ("coarse_rgb", "fine_rgb", "coarse_dep", "fine_dep", "normal", "albedo", None, "surface_rgb", "shading")
    for k, v in zip(names, out):
        if k is None:
            assert v is None
            continue
        assert v.shape[0] == 1 and v.shape[2:] == (8, 16)
        e = rel_err(v.cpu().numpy(), g[k])
        assert e < (5e-2 if k in ("normal", "surface_rgb", "shading") else 1e-4), (k, e)


def test_bench_size_properties():
    """At the bench configuration (B=512, N=128) the oracle is too slow for a full compare; check
    size-independent properties: weights are a sub-probability, compositing is linear in colour,
    determinism, unit normals, gradient finite and shard-additive."""
    import pano_nerf_amd as pn
    torch.manual_seed(0)
    B, N = 512, 128
    flat, rgbs, radius, _ = orc.synthetic_scene(16, 32, 3, seed=4)
    idx = torch.randint(0, flat.origins.shape[0], (B,), generator=torch.Generator().manual_seed(4))
    rays = to_dev(pn.Rays(*[x[idx] for x in flat]))
    gt = rgbs[idx].to(dev())
    env = pn.generate_lit_rays(10, radius)
    model = make_pano(N)
    S = N + 1
    gen = torch.Generator().manual_seed(1)
    model.noise_override = dict(t_rand=torch.rand(B, S, generator=gen),
                                u_rand=torch.rand(B, S, generator=gen) * (1.0 / S - 1.2e-7),
                                env_rand=torch.rand(1, 11, generator=gen))
    outs = model(rays=rays, env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    loss, terms = pn.pano_loss(outs, rays.lossmult, gt)
    loss.backward()
    full = model.mlp.last_flat_grad.clone()
    assert torch.isfinite(full).all() and float(full.abs().max()) > 0
    (c0, d0, *_), (c1, d1, ort, nrm, alb, _, sf, dif, shd) = outs
    assert torch.allclose(nrm.norm(dim=-1), torch.ones(B, device=dev()), atol=1e-4)
    assert float(alb.min()) >= 0.03 - 1e-6 and float(alb.max()) <= 0.80 + 1e-6
    assert bool((d1 >= 0).all()) and bool((d1 <= 10).all())
    assert torch.equal(sf, dif)
    # determinism: same inputs, same bits
    outs_b = model(rays=rays, env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    assert torch.equal(outs_b[1][0], c1) and torch.equal(outs_b[1][6], sf)
    # shard additivity: the mean-loss gradient of the full batch = average of the two half-batch gradients
    halves = []
    for lo, hi in ((0, B // 2), (B // 2, B)):
        sub = pn.Rays(*[x[lo:hi] for x in rays])
        model.noise_override = dict(t_rand=model.noise_override["t_rand"], u_rand=model.noise_override["u_rand"],
                                    env_rand=model.noise_override["env_rand"])
        ov = model.noise_override
        model.noise_override = dict(t_rand=ov["t_rand"][lo:hi], u_rand=ov["u_rand"][lo:hi], env_rand=ov["env_rand"])
        for p in model.mlp.parameters():
            p.grad = None
        o2 = model(rays=sub, env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
        l2, _ = pn.pano_loss(o2, sub.lossmult, gt[lo:hi])
        l2.backward()
        halves.append(model.mlp.last_flat_grad.clone())
        model.noise_override = ov
    avg = 0.5 * (halves[0] + halves[1])
    assert float((avg - full).norm() / full.norm()) < 1e-4


@pytest.mark.parametrize("mode", MODES)
def test_headline_size_matches_oracle_on_a_ray_subset(mode):
    """The BASELINE / bench.py size - 4096 rays x 128 + 128 samples (M = 524 288 rows per level, 16 tiles per persistent
    workgroup, 409 600 env-light rows), train mode with fixed noise - checked against the oracle: rays are independent, so
    the oracle run on a strided 128-ray subset (with those rays' noise rows) must reproduce the subset of every per-ray
    output of the 4096-ray call.  Plus the size-independent properties of test_bench_size_properties.
    Reference: models/pano_mip_nerf.py:197-363."""
    import pano_nerf_amd as pn
    B, N, K = 4096, 128, 128
    S = N + 1
    flat, rgbs, radius, _ = orc.synthetic_scene(64, 128, 3, seed=4)
    idx = torch.randint(0, flat.origins.shape[0], (B,), generator=torch.Generator().manual_seed(4096))
    rays_c = orc.Rays(*[x[idx] for x in flat])
    rays = to_dev(pn.Rays(*rays_c))
    gt = rgbs[idx].to(dev())
    env = pn.generate_lit_rays(10, radius)
    env_c = orc.Rays(*[x.cpu() for x in env])
    model = make_pano(N)
    model.mlp_mode = mode
    model.mlp.debug_keep = True
    gen = torch.Generator().manual_seed(7)
    noise = dict(t_rand=torch.rand(B, S, generator=gen), u_rand=torch.rand(B, S, generator=gen) * (1.0 / S - 1.2e-7),
                 env_rand=torch.rand(1, 11, generator=gen))
    model.noise_override = noise
    outs = model(rays=rays, env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    loss, _ = pn.pano_loss(outs, rays.lossmult, gt)
    loss.backward()
    grad = model.mlp.last_flat_grad
    assert bool(torch.isfinite(loss)) and bool(torch.isfinite(grad).all()) and float(grad.abs().max()) > 0
    # ---- the oracle on every 32nd ray
    sub = torch.arange(0, B, B // K)
    sub_rays = orc.Rays(*[x[sub] for x in rays_c])
    sub_noise = dict(t_rand=noise["t_rand"][sub], u_rand=noise["u_rand"][sub], env_rand=noise["env_rand"])
    p = orc.init_params(4, 5)
    with torch.no_grad():
        ref = orc.pano_forward(p, sub_rays, env_c, num_samples=N, noise=sub_noise)
        ref64 = orc.pano_forward({k: v.double() for k, v in p.items()}, orc.Rays(*[x.double() for x in sub_rays]),
                                 orc.Rays(*[x.double() for x in env_c]), num_samples=N,
                                 noise={k: v.double() for k, v in sub_noise.items()})
    # (a) the oracle as it stands: well-conditioned outputs at 1e-4 (tensor scale AND element-wise); the outputs derived from
    # the density gradient (a ReLU gate with a pre-activation of ~1e-7 flips under any fp32 summation order and moves one
    # ray: with 128 samples a ray a few of the 32 rays sit on such a gate) on the median only
    for lvl in (0, 1):
        for nme, v, r in zip(NAMES9, outs[lvl], ref[lvl]):
            assert (v is None) == (r is None), (lvl, nme)
            if v is None or nme == "ort_loss":  # (a mean over all 4096 rays: not a per-ray output)
                continue
            got, want = v.detach()[sub.to(dev())].cpu().numpy(), r.numpy()
            key = f"headline/{mode}/l{lvl}/{nme}"
            if nme in LOOSE:
                scale = max(float(np.abs(want).max()), 1e-12)
                per_ray = np.abs(got - want).reshape(K, -1).max(-1) / scale
                assert float(np.median(per_ray)) < 1e-4, (key, float(np.median(per_ray)))
                # SURVEY.md 7's ">= 99 % of the elements within 1e-3": at 128 samples a ray fp32 itself does not deliver it (the
                # fp32 oracle has 93 - 96 % of these elements within 1e-3 of its own fp64 run), so the fraction is stated against
                # the fp64 oracle next to the fp32 oracle's own (tests/test_gpu_scale.py does the same over 1024 rays)
                want64 = ref64[lvl][NAMES9.index(nme)].numpy()
                ours = float(np.mean(np.abs(got - want64) <= 1e-3 * scale))
                theirs = float(np.mean(np.abs(want - want64) <= 1e-3 * scale))
                assert ours >= min(0.99, theirs - 0.03), (key, "fraction of elements within 1e-3 of the fp64 run: ours, the fp32 oracle's", ours, theirs)
            else:
                assert_close(got, want, key)
    # (b) the oracle on the gate decisions the kernels took for these rays: EVERY per-ray output pointwise at 1e-4
    with orc.forced_gates(forced_gate_sets(model, normals=True, surf=True, rays=sub, n=N)), torch.no_grad():
        refg = orc.pano_forward(p, sub_rays, env_c, num_samples=N, noise=sub_noise)
    for nme, v, r in zip(NAMES9, outs[1], refg[1]):
        if v is None or nme == "ort_loss":
            continue
        got, want = v.detach()[sub.to(dev())].cpu().numpy(), r.numpy()
        e = rel_err(got, want)
        assert e < 1e-4, (f"headline/{mode}/l1/{nme} on identical gates", e)
    # ---- size-independent properties at this size
    (c0, d0, *_), (c1, d1, ort, nrm, alb, _, sf, dif, shd) = outs
    assert torch.allclose(nrm.norm(dim=-1), torch.ones(B, device=dev()), atol=1e-4)
    assert float(alb.min()) >= 0.03 - 1e-6 and float(alb.max()) <= 0.80 + 1e-6
    assert bool((d1 >= 0).all()) and bool((d1 <= 10).all()) and bool((d0 >= 0).all()) and bool((d0 <= 10).all())
    assert torch.equal(sf, dif) and bool(torch.isfinite(shd).all()) and float(ort) >= 0
    outs_b = model(rays=rays, env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    assert torch.equal(outs_b[1][0], c1) and torch.equal(outs_b[1][6], sf)  # determinism: same inputs, same bits
    # shard additivity at the 2-GPU share: the mean-loss gradient = the average of the two 2048-ray gradients
    full = grad.clone()
    halves = []
    for lo, hi in ((0, B // 2), (B // 2, B)):
        model.noise_override = dict(t_rand=noise["t_rand"][lo:hi], u_rand=noise["u_rand"][lo:hi], env_rand=noise["env_rand"])
        o2 = model(rays=pn.Rays(*[x[lo:hi] for x in rays]), env_rays=env, randomized=True, white_bkgd=False, enable_surf=True,
                   use_ort_loss=True)
        l2, _ = pn.pano_loss(o2, rays.lossmult[lo:hi], gt[lo:hi])
        l2.backward()
        halves.append(model.mlp.last_flat_grad.clone())
    avg = 0.5 * (halves[0] + halves[1])
    assert float((avg - full).norm() / full.norm()) < 1e-4


def test_state_dict_and_errors():
    import pano_nerf_amd as pn
    m = pn.PanoMipNeRF(num_samples=8, rgb_activation="softplus", mlp_num_density_channels=5)
    keys = list(m.state_dict().keys())
    assert keys[0] == "mlp.layers.0.0.weight" and "mlp.view_layers.0.0.bias" in keys and len(keys) == 24
    assert sum(p.numel() for p in m.mlp.parameters()) == 613768
    with pytest.raises(NotImplementedError):
        pn.PanoMipNeRF(rgb_activation="sigmoid")
    with pytest.raises(NotImplementedError):
        pn.MipNeRF(rgb_activation="softplus", ray_shape="cylinder")
    rays = pn.Rays(*[torch.zeros(4, d) for d in (3, 3, 3, 1, 1, 1, 1, 1)])
    with pytest.raises(RuntimeError, match="no CPU"):
        pn.MipNeRF(num_samples=8, rgb_activation="softplus")(rays=rays, randomized=False, white_bkgd=False,
                                                             use_ort_loss=False)


def test_autocast_call_pattern_is_fp32_inside(golden):
    """Upstream trains under Lightning '16-mixed': forward runs inside torch.autocast and `env_rays` arrives as fp16.
    The drop-in computes in fp32 regardless (custom_fwd casts the inputs), so the outputs and the gradients are those
    of the plain call, bit for bit."""
    import pano_nerf_amd as pn
    s = golden("stages_B64_N32")
    rays, env = to_dev(rays_of(s)), to_dev(env_of(golden))
    env16 = type(env)(*[x.half() for x in env])  # the loader stores the lit rays as fp16 (pano_datasets.py:218-263)
    gt = torch.from_numpy(s["rgbs"]).to(dev())

    def run(autocast):
        model = make_pano(32)
        with torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
            outs = model(rays=rays, env_rays=env16, randomized=False, white_bkgd=False, enable_surf=True,
                         use_ort_loss=True)
            loss, _ = pn.pano_loss(outs, rays.lossmult, gt)
        loss.backward()
        return outs, loss.detach(), model.mlp.last_flat_grad.clone()

    o0, l0, g0 = run(False)
    o1, l1, g1 = run(True)
    for a, b in zip(o0[1], o1[1]):
        if a is not None:
            assert b.dtype == torch.float32 and torch.equal(a, b)
    assert torch.equal(l0, l1) and torch.equal(g0, g1)


def test_training_is_bit_reproducible():
    """No atomics anywhere on the path (split-K slabs, bias column sums and head partials are reduced in a fixed
    order): two runs of the same seeded training steps end with bit-identical parameters."""
    import numpy as np
    import pano_nerf_amd as pn

    def run():
        torch.manual_seed(11)
        cams = [np.eye(4, dtype=np.float32)]
        pool = pn.DeviceRayPool(16, 32, cams, images=[np.linspace(0, 1, 16 * 32 * 3, dtype=np.float32).reshape(16, 32, 3)])
        env = pool.lit_rays(10)
        model = pn.PanoMipNeRF(num_samples=32, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5).to(dev())
        model.mlp.load_state_dict(orc.init_params(4, 5))
        opt = pn.FlatAdam(model.mlp, lr=2e-4)
        g = torch.Generator(device="cuda").manual_seed(3)
        for i in range(4):
            rays, gt = pool.sample(192, generator=g)
            opt.zero_grad()
            outs = model(rays=rays, env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
            loss, _ = pn.pano_loss(outs, rays.lossmult, gt)
            loss.backward()
            opt.step(model.mlp.last_flat_grad, lr=pn.mip_lr(i))
        return model.mlp.flat_params().detach().clone(), float(loss.detach())

    p1, l1 = run()
    p2, l2 = run()
    assert l1 == l2 and torch.equal(p1, p2)


@pytest.mark.parametrize("parts", [2, 3])
def test_concurrent_sub_batches_give_the_whole_batch_gradient(parts):
    """pano_nerf_amd.concurrent_step: sub-batches on separate streams, weighted by their share of the rays, reproduce
    loss and gradient of one call over the whole batch (deterministic sampling, so the only difference is the order
    of the fp32 sums)."""
    import pano_nerf_amd as pn
    B, N = 250, 32  # 250 = 125 + 125 = 84 + 83 + 83: unequal parts for 3
    flat, rgbs, radius, _ = orc.synthetic_scene(16, 32, 3, seed=4)
    idx = torch.randint(0, flat.origins.shape[0], (B,), generator=torch.Generator().manual_seed(8))
    rays = to_dev(pn.Rays(*[x[idx] for x in flat]))
    gt = rgbs[idx].to(dev())
    env = pn.generate_lit_rays(10, radius)
    model = make_pano(N)
    kw = dict(env_rays=env, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    outs = model(rays=rays, **kw)
    loss, _ = pn.pano_loss(outs, rays.lossmult, gt)
    loss.backward()
    full, full_pgrad = model.mlp.last_flat_grad.clone(), [p.grad.clone() for p in model.mlp.parameters()]
    for p in model.mlp.parameters():
        p.grad = None
    l2, g2, first = pn.concurrent_step(model, pn.pano_loss, rays, gt, parts=parts, **kw)
    torch.cuda.synchronize()
    assert abs(float(l2) - float(loss)) < 1e-5 * abs(float(loss))
    assert float((g2 - full).norm() / full.norm()) < 1e-4
    for p, ref in zip(model.mlp.parameters(), full_pgrad):
        assert float((p.grad - ref).norm()) <= 1e-4 * float(ref.norm()) + 1e-12
    assert first[1][0].shape[0] == (B + parts - 1) // parts
