"""Edge cases through the drop-in modules and the C ABI: ragged / tiny / maximum sizes, refused arguments."""
import numpy as np
import pytest
import torch

from conftest import check_flat_grad_per_tensor, check_flat_grad_pointwise, forced_gate_sets, rel_err
from oracle import pano_oracle as orc

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def scene_rays(B, stride=5):
    flat, rgbs, radius, _ = orc.synthetic_scene(8, 16, 3, seed=4)
    idx = (torch.arange(B) * stride) % flat.origins.shape[0]
    return orc.Rays(*[x[idx] for x in flat]), rgbs[idx], radius


@pytest.mark.parametrize("B,N", [(1, 8), (3, 2), (130, 16), (257, 33)])
def test_ragged_and_tiny_batches_match_oracle(B, N):
    """B not a multiple of the 128-row GEMM tile, a single ray, the smallest sample count (N = 2), odd N."""
    import pano_nerf_amd as pn
    rays_c, rgbs, radius = scene_rays(B)
    rays = pn.Rays(*[x.to(dev()) for x in rays_c])
    env = pn.generate_lit_rays(10, radius)
    params = orc.init_params(4, 5)
    model = pn.PanoMipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5)
    model.mlp.load_state_dict(params)
    model = model.to(dev())
    model.mlp.debug_keep = True
    outs = model(rays=rays, env_rays=env, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    loss, _ = pn.pano_loss(outs, rays.lossmult, rgbs.to(dev()))
    loss.backward()
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    env_c = orc.Rays(*[x.cpu() for x in env])
    ref = orc.pano_forward(p, rays_c, env_c, num_samples=N)
    ref_loss = orc.pano_loss(ref, rays_c.lossmult, rgbs)
    for lvl in (0, 1):
        assert rel_err(outs[lvl][0].detach().cpu(), ref[lvl][0].detach()) < 1e-4
        assert rel_err(outs[lvl][1].detach().cpu(), ref[lvl][1].detach()) < 1e-4
    assert rel_err(outs[1][4].detach().cpu(), ref[1][4].detach()) < 1e-4  # albedo
    assert abs(float(loss) - float(ref_loss)) < 2e-3 * abs(float(ref_loss))
    g = model.mlp.last_flat_grad
    assert bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
    ref_g = torch.autograd.grad(ref_loss, list(p.values()))
    check_flat_grad_per_tensor(g.detach().cpu().numpy(), {k: x.detach().numpy() for k, x in zip(p.keys(), ref_g)}, 5,
                               second_order=True, n_rows=B * (2 * N + 100))
    # pointwise, every entry of every tensor: the oracle on the gate decisions the kernels took
    sets = forced_gate_sets(model, normals=True, surf=True)
    with orc.forced_gates(sets):
        fl = orc.pano_loss(orc.pano_forward(p, rays_c, env_c, num_samples=N), rays_c.lossmult, rgbs)
        fg = torch.autograd.grad(fl, list(p.values()))

    def fp64():
        p64 = {k: v.double().requires_grad_(True) for k, v in params.items()}
        r64 = orc.Rays(*[x.double() for x in rays_c])
        with orc.forced_gates(sets):
            l64 = orc.pano_loss(orc.pano_forward(p64, r64, env_c, num_samples=N), r64.lossmult, rgbs.double())
            return {k: x.detach().numpy() for k, x in zip(p64.keys(), torch.autograd.grad(l64, list(p64.values())))}

    check_flat_grad_pointwise(g.detach().cpu().numpy(), {k: x.detach().numpy() for k, x in zip(p.keys(), fg)}, 5, ref64_fn=fp64)


def test_maximum_sample_count():
    """N = 512 is the most one wavefront scans (PN_MAX_SAMPLES); weights stay a sub-probability and sorted t."""
    import pano_nerf_amd as pn
    rays_c, _, _ = scene_rays(4)
    rays = pn.Rays(*[x.to(dev()) for x in rays_c])
    model = pn.MipNeRF(num_samples=512, rgb_activation="softplus", mlp_num_density_channels=1).to(dev())
    with torch.no_grad():
        outs = model(rays=rays, randomized=True, white_bkgd=True, use_ort_loss=False)
    assert torch.isfinite(outs[1][0]).all() and bool((outs[1][1] >= 0).all()) and bool((outs[1][1] <= 10).all())
    with pytest.raises(NotImplementedError):
        pn.MipNeRF(num_samples=513, rgb_activation="softplus")


def test_refused_arguments():
    from pano_nerf_amd import _lib
    h = _lib.load()
    x = torch.zeros(64, device=dev())
    p = x.data_ptr()
    assert h.pn_sample_coarse(0, 8, 0, p, p, p, p, p, None, p, p, p, None) == -1          # empty batch
    assert h.pn_sample_coarse(4, 513, 0, p, p, p, p, p, None, p, p, p, None) == -2        # beyond PN_MAX_SAMPLES
    assert h.pn_sample_coarse(4, 8, 0, None, p, p, p, p, None, p, p, p, None) == -3       # null pointer
    assert h.pn_composite_forward(4, 8, 3, -1.0, 0.0, 0, p, p, p, p, 4, p, p, p, p, None) == -2  # density channels
    assert h.pn_resample(4, 1, p, p, 0.01, None, p, p, p, p, p, p, None) == -1
    with pytest.raises(RuntimeError, match="bad shape"):
        _lib.call("pn_ipe_encode", 0, p, p, p, None)


def test_device_ray_pool():
    import pano_nerf_amd as pn
    cams = [np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32)]
    cams[1][:3, 3] = [0.1, 0.2, -0.3]
    imgs = [np.full((8, 16, 3), 0.25, np.float32), np.full((8, 16, 3), 0.75, np.float32)]
    pool = pn.DeviceRayPool(8, 16, cams, images=imgs)
    assert len(pool) == 2 * 8 * 16
    rays, rgb = pool.sample(64)
    assert rays.origins.shape == (64, 3) and rgb.shape == (64, 3) and rays.radii.shape == (64, 1)
    # a ray's colour comes from the camera its origin belongs to
    cam1 = (rays.origins[:, 0] > 0.05)
    assert torch.allclose(rgb[cam1], torch.full_like(rgb[cam1], 0.75)) and torch.allclose(rgb[~cam1], torch.full_like(rgb[~cam1], 0.25))
    # the single-launch gather is exactly an index of the pool (same generator -> same indices)
    g1 = torch.Generator(device="cuda").manual_seed(5)
    g2 = torch.Generator(device="cuda").manual_seed(5)
    rays2, rgb2 = pool.sample(97, generator=g1)
    idx = torch.randint(0, len(pool), (97,), device=dev(), generator=g2)
    for got, full in zip(rays2, pool.rays):
        assert torch.equal(got, full[idx])
    assert torch.equal(rgb2, pool.rgbs[idx])
    env = pool.lit_rays(10)
    assert env.directions.dtype == torch.float16 and env.directions.shape == (10, 3)
    assert abs(float(env.lossmult[0, 0]) - 4 * np.pi / 10) < 2e-3


def test_optimizer_step_between_forward_and_backward_raises():
    """FlatAdam writes the flat parameter block through raw device pointers (no autograd version bump): a backward whose
    forward saw the older weights must refuse instead of mixing them with the saved activations."""
    import pano_nerf_amd as pn
    rays_c, rgbs, radius = scene_rays(16)
    rays = pn.Rays(*[x.to(dev()) for x in rays_c])
    env = pn.generate_lit_rays(10, radius)
    model = pn.PanoMipNeRF(num_samples=8, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5).to(dev())
    opt = pn.FlatAdam(model.mlp, lr=1e-3)
    for stepper in ("step", "step_dev"):
        outs = model(rays=rays, env_rays=env, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
        loss, _ = pn.pano_loss(outs, rays.lossmult, rgbs.to(dev()))
        g = torch.ones_like(model.mlp.flat_params())
        if stepper == "step":
            opt.step(g)
        else:
            opt.step_dev(g, torch.full((1,), 1e-3, device=dev()))
        with pytest.raises(RuntimeError, match="modified in place"):
            loss.backward()
    # and the ordinary order still works
    outs = model(rays=rays, env_rays=env, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    loss, _ = pn.pano_loss(outs, rays.lossmult, rgbs.to(dev()))
    loss.backward()
    opt.step()


CONFIGS = [
    # model, B, N, white_bkgd, randomized, enable_surf / use_ort
    ("pano", 37, 64, True, True, True, True),
    ("pano", 300, 20, False, True, False, True),
    ("pano", 64, 128, True, False, True, False),
    ("mip", 200, 48, True, True, None, True),
    ("mip", 129, 96, False, False, None, False),
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: "-".join(str(x) for x in c))
def test_assorted_configurations_match_oracle(cfg):
    """Shapes and flag combinations the goldens do not cover (train-mode noise shared with the oracle): outputs at
    1e-4, loss at 1e-4, gradient norm within the tolerance the ill-conditioned normals allow."""
    import pano_nerf_amd as pn
    kind, B, N, white, rnd, surf, ort = cfg
    nc = 5 if kind == "pano" else 1
    rays_c, rgbs, radius = scene_rays(B, stride=3)
    rays = pn.Rays(*[x.to(dev()) for x in rays_c])
    params = orc.init_params(9, nc)
    gen = torch.Generator().manual_seed(B * 131 + N)
    noise = None
    if rnd:
        noise = dict(t_rand=torch.rand(B, N + 1, generator=gen), u_rand=torch.rand(B, N + 1, generator=gen) * (1 / (N + 1) - 1.2e-7),
                     env_rand=torch.rand(1, 11, generator=gen))
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    if kind == "pano":
        model = pn.PanoMipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5)
        env = pn.generate_lit_rays(10, radius)
        kw = dict(rays=rays, env_rays=env, randomized=rnd, white_bkgd=white, enable_surf=surf, use_ort_loss=ort)
        ref = orc.pano_forward(p, rays_c, orc.Rays(*[x.cpu() for x in env]), num_samples=N, white_bkgd=white, enable_surf=surf,
                               use_ort_loss=ort, noise=noise)
        ref_loss = orc.pano_loss(ref, rays_c.lossmult, rgbs, surface=surf)
    else:
        model = pn.MipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=1)
        kw = dict(rays=rays, randomized=rnd, white_bkgd=white, use_ort_loss=ort)
        ref = orc.mip_forward(p, rays_c, num_samples=N, white_bkgd=white, use_ort_loss=ort, noise=noise)
        ref_loss = orc.mip_loss(ref, rays_c.lossmult, rgbs, use_ort=ort)
    model.mlp.load_state_dict(params)
    model = model.to(dev())
    model.mlp.debug_keep = True
    if rnd:
        model.noise_override = noise
    outs = model(**kw)
    if kind == "pano":
        loss, _ = pn.pano_loss(outs, rays.lossmult, rgbs.to(dev()), surface=surf)
    else:
        loss, _ = pn.mip_loss(outs, rays.lossmult, rgbs.to(dev()), use_ort=ort)
    loss.backward()
    for lvl in (0, 1):
        for slot in (0, 1):
            assert rel_err(outs[lvl][slot].detach().cpu(), ref[lvl][slot].detach()) < 1e-4, (lvl, slot)
    for got, want in zip(outs[1], ref[1]):
        assert (got is None) == (want is None)
    assert abs(float(loss) - float(ref_loss)) < 2e-4 * abs(float(ref_loss))
    g = model.mlp.last_flat_grad
    ref_g = torch.autograd.grad(ref_loss, list(p.values()), allow_unused=True)
    assert bool(torch.isfinite(g).all())
    check_flat_grad_per_tensor(g.detach().cpu().numpy(),
                               {k: (None if x is None else x.detach().numpy()) for k, x in zip(p.keys(), ref_g)},
                               5 if kind == "pano" else 1, second_order=bool(ort) or kind == "pano",
                               n_rows=B * (2 * N + (100 if (kind == "pano" and surf) else 0)))
    # pointwise, every entry of every tensor: the oracle on the gate decisions the kernels took (these cases run up to
    # 42 k MLP rows, where a rank-stripped comparison would hide errors of rank up to 64)
    sets = forced_gate_sets(model, normals=(kind == "pano" or bool(ort)), surf=(kind == "pano" and bool(surf)))

    def forced(dt):
        q = {k: v.to(dt).requires_grad_(True) for k, v in params.items()}
        rc = orc.Rays(*[x.to(dt) for x in rays_c])
        nz = None if noise is None else {k: v.to(dt) for k, v in noise.items()}
        with orc.forced_gates(sets):
            if kind == "pano":
                fl = orc.pano_loss(orc.pano_forward(q, rc, orc.Rays(*[x.cpu() for x in env]), num_samples=N, white_bkgd=white,
                                                    enable_surf=surf, use_ort_loss=ort, noise=nz), rc.lossmult, rgbs.to(dt), surface=surf)
            else:
                fl = orc.mip_loss(orc.mip_forward(q, rc, num_samples=N, white_bkgd=white, use_ort_loss=ort, noise=nz),
                                  rc.lossmult, rgbs.to(dt), use_ort=ort)
            fg = torch.autograd.grad(fl, list(q.values()), allow_unused=True)
        return {k: (None if x is None else x.detach().numpy()) for k, x in zip(q.keys(), fg)}

    check_flat_grad_pointwise(g.detach().cpu().numpy(), forced(torch.float32), 5 if kind == "pano" else 1,
                              ref64_fn=lambda: forced(torch.float64))


@pytest.mark.parametrize("mode", ["fused_f16x2", "fused"])
def test_graph_replay_reproduces_eager_step(mode):
    """A captured training step replayed many times gives the eager step's loss and gradient every time.  (The fp16-pair
    mode clears two small scale tables per forward; done with hipMemsetAsync, replays came out NaN or with stale scales in 4
    of 14 runs of the 512-ray bench - although the captured graph had the edges memset -> accumulating kernel
    (profiles/r03_graph_memset_nodes.txt).  The tables are cleared by a kernel now: a workaround, guarded by this test.)"""
    import pano_nerf_amd as pn
    B, N = 96, 32
    rays_c, rgbs, radius = scene_rays(B)
    rays = pn.Rays(*[x.to(dev()) for x in rays_c])
    gt = rgbs.to(dev())
    env = pn.generate_lit_rays(10, radius)
    model = pn.PanoMipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5)
    model.mlp.load_state_dict(orc.init_params(4, 5))
    model = model.to(dev())
    model.mlp_mode = mode

    def fwd_bwd():
        outs = model(rays=rays, env_rays=env, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
        loss, _ = pn.pano_loss(outs, rays.lossmult, gt)
        loss.backward()
        return loss.detach(), model.mlp.last_flat_grad

    loss0, g0 = fwd_bwd()
    loss0, g0 = loss0.clone(), g0.clone()
    side = torch.cuda.Stream(device=dev())
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss_g, g_g = fwd_bwd()
    scale = float(g0.abs().max())
    for i in range(25):
        loss_g.fill_(float("nan"))
        g_g.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(loss_g)) and abs(float(loss_g) - float(loss0)) <= 1e-6 * abs(float(loss0)), (i, float(loss_g))
        assert float((g_g - g0).abs().max()) <= 1e-6 * scale, (i, float((g_g - g0).abs().max()), scale)
