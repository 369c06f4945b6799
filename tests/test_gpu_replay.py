"""The drop-in the way the reference's validation loop calls it (systems/panonerf_system.py:133-192 with val.chunk_size 512,
configs/panonerf.yaml:22): many small no-grad calls.  Inside the module such a call replays a HIP graph captured once per chunk
size (render.py, _RenderBase._replayed); the result must be what the eager launches give, bit for bit, whatever the chunk size,
and must follow the parameters when they change between calls."""
import numpy as np
import pytest
import torch

from oracle import pano_oracle as orc
from test_gpu_full import dev, make_pano

pytestmark = pytest.mark.gpu


def _loop(model, img_rays, env, chunk, H, W):
    import pano_nerf_amd as pn
    chunks, _ = pn.rearrange_render_image(img_rays, chunk)
    keep = [[] for _ in range(8)]
    with torch.no_grad():
        for ch in chunks:
            (c_rgb, c_dep, *_), (f_rgb, f_dep, _, f_nor, alb, rhn, sf, _, sd) = model(
                rays=ch, env_rays=env, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
            assert rhn is None
            for lst, v in zip(keep, (c_rgb, f_rgb, c_dep, f_dep, f_nor, alb, sf, sd)):
                lst.append(v)
    dims = (3, 3, 1, 1, 3, 3, 3, 3)
    return [torch.cat(x, 0).view(1, H, W, dm).permute(0, 3, 1, 2) for x, dm in zip(keep, dims)], len(chunks)


def test_small_chunk_loop_replays_and_matches_the_big_chunk_render():
    import pano_nerf_amd as pn
    H, W, N = 32, 72, 32  # 2304 rays: chunks of 512 (four whole + one ragged chunk of 256), 100 and one chunk of everything
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, 3] = (0.1, -0.2, 0.05)
    rays = pn.generate_pano_rays(H, W, c2w)
    env = pn.generate_lit_rays(10, pn.rays.pano_pixel_radius(rays))
    img_rays = pn.Rays(*[x.view(1, H, W, -1) for x in rays])
    model = make_pano(N)
    model.replay_inference = False
    eager_big, _ = _loop(model, img_rays, env, H * W, H, W)
    eager_512, n512 = _loop(model, img_rays, env, 512, H, W)
    model.replay_inference = True
    assert not model._replays
    rep_512, _ = _loop(model, img_rays, env, 512, H, W)
    assert len(model._replays) == 2  # 512 rays and the ragged 256
    rep_512b, _ = _loop(model, img_rays, env, 512, H, W)  # second pass: replays only
    rep_100, _ = _loop(model, img_rays, env, 100, H, W)
    assert n512 == 5
    for a, b, c, d, e in zip(eager_big, eager_512, rep_512, rep_512b, rep_100):
        assert torch.equal(a, b), "chunking changed a ray's result"
        assert torch.equal(a, c) and torch.equal(a, d) and torch.equal(a, e), "a replayed chunk differs from the eager launches"
    # the helper with 32 768-ray chunks (the bench's inference leg) gives the same image
    big = pn.render_image(model, img_rays, env, H, W, chunk_size=32768)
    for a, i in zip(eager_big, (0, 1, 2, 3, 4, 5, 7, 8)):
        assert torch.equal(a, big[i])
    # parameters changed between two calls (an optimizer step through raw pointers, or a checkpoint load): the captured sequence
    # re-packs the weights from the live block, so the replay follows
    with torch.no_grad():
        model.mlp.flat_params().mul_(1.01)
    model.replay_inference = False
    eager_new, _ = _loop(model, img_rays, env, 512, H, W)
    model.replay_inference = True
    rep_new, _ = _loop(model, img_rays, env, 512, H, W)
    assert not torch.equal(eager_new[1], eager_512[1])
    for a, b in zip(eager_new, rep_new):
        assert torch.equal(a, b)
    # a training-mode call (grad enabled) never replays and still back-propagates
    sub = pn.Rays(*[x[:64] for x in rays])
    outs = model(rays=sub, env_rays=env, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    outs[1][0].sum().backward()
    assert model.mlp.last_flat_grad is not None and bool(torch.isfinite(model.mlp.last_flat_grad).all())


def test_env_cache_can_be_invalidated_or_switched_off():
    """the fp32 copies of the caller's env rays are cached on (address, version, dtype, shape); a write that bypasses
    the version counter needs invalidate_env_cache() (or cache_env_rays = False)."""
    import pano_nerf_amd as pn
    H, W, N = 8, 16, 16
    rays = pn.generate_pano_rays(H, W, np.eye(4, dtype=np.float32))
    env = pn.generate_lit_rays(10, pn.rays.pano_pixel_radius(rays))
    model = make_pano(N)
    call = lambda: model(rays=rays, env_rays=env, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)[1][8]
    with torch.no_grad():
        a = call().clone()
        env.lossmult.data.mul_(2)  # bypasses the version counter: the cached fp32 copy (and the captured graph) are stale
        b = call().clone()
        assert torch.equal(a, b)
        model.invalidate_env_cache()
        c = call().clone()
        assert torch.allclose(c, 2 * a, rtol=1e-6, atol=0)
        model.cache_env_rays = False
        env.lossmult.data.mul_(0.5)
        d = call().clone()
        assert torch.allclose(d, a, rtol=1e-6, atol=0)
