"""Chunked full-image render.

Mirrors ``PanoNeRFSystem.render_image`` (systems/panonerf_system.py:133-192): flatten the H x W rays, slice
them with ``rearrange_render_image`` (models/mip.py:530-547), render every chunk with ``randomized=False,
enable_surf=True, use_ort_loss=True`` under no_grad, concatenate and reshape to ``[1, C, H, W]``.  The
reference's ``val.chunk_size`` is 512 (1 024 Python iterations per 512x1024 panorama, sized for a 2020 GPU's
memory); with 288 GB of HBM a chunk of 32 768 rays fits comfortably, so ``chunk_size`` defaults to that.
With ``world > 1`` chunks are dealt round-robin to the ranks and gathered (SURVEY.md 8e 'Inference'); a rank runs its
chunks on ``streams`` concurrent HIP streams.
"""
import torch

from .rays import Rays, rearrange_render_image

_KEYS = ("coarse_rgb", "fine_rgb", "coarse_dep", "fine_dep", "fine_nor", "albedo", "roughness", "surface_rgb", "shading")


def render_image(model, rays, env_rays, height, width, chunk_size=32768, white_bkgd=False, rank=0, world=1, streams=2):
    """rays: Rays of [1, H, W, C] (or [H*W, C]) device tensors.  Returns the 9-tuple of render_image:
    (coarse_rgb, fine_rgb, coarse_dep, fine_dep, fine_nor, albedo, roughness(None), surface_rgb, shading), each
    [1, C, H, W]."""
    flat = Rays(*[x.reshape(-1, x.shape[-1]) for x in rays])
    chunks, _ = rearrange_render_image(flat, chunk_size)
    outs = {k: [] for k in ("coarse_rgb", "fine_rgb", "coarse_dep", "fine_dep", "fine_nor", "albedo", "surface_rgb", "shading")}
    mine = list(range(rank, len(chunks), world))
    # chunks are independent: deal them to `streams` HIP streams so that one chunk's GEMM chain fills the first-tile /
    # last-tile bubbles of another's (see pano_nerf_amd.parallel)
    dev0 = flat.origins.device
    cur = torch.cuda.current_stream(dev0) if dev0.type == "cuda" else None
    lanes = [cur]
    forked = cur is not None and streams > 1 and len(mine) > 1
    if forked:
        # The packed weight copies are built ONCE, before the fork, and frozen for the chunk loop: every forward otherwise
        # rebuilds them in place on its own stream while the other stream's kernels read them (the fp16-pair pack passes
        # through a cleared table of weight maxima on the way: a chunk rendered during another's re-pack came out as 1e20)
        from .parallel import _streams
        from .render import _planes_of
        planes = _planes_of(model.mlp_mode)
        model.mlp._frozen = False
        if planes:
            model.mlp.chain_packed(cur.cuda_stream, planes)
        else:
            model.mlp.packed(cur.cuda_stream)
        model.mlp._frozen = True
        lanes += _streams(dev0, min(int(streams), len(mine)) - 1)
        for s in lanes[1:]:
            s.wait_stream(cur)
    try:
        with torch.no_grad():
            for n, i in enumerate(mine):
                lane = lanes[n % len(lanes)]
                with (torch.cuda.stream(lane) if lane is not None else torch.no_grad()):
                    (c_rgb, c_dep, *_), (f_rgb, f_dep, _, f_nor, alb, _, sf_rgb, _, sd) = model(
                        rays=chunks[i], env_rays=env_rays, randomized=False, white_bkgd=white_bkgd, enable_surf=True,
                        use_ort_loss=True)
                    for k, v in zip(outs, (c_rgb, f_rgb, c_dep.view(-1, 1), f_dep.view(-1, 1), f_nor, alb, sf_rgb, sd)):
                        outs[k].append(v)
    finally:
        if forked:
            model.mlp._frozen = False
    for s in lanes[1:]:
        cur.wait_stream(s)
    widths_all = dict(coarse_rgb=3, fine_rgb=3, coarse_dep=1, fine_dep=1, fine_nor=3, albedo=3, surface_rgb=3, shading=3)
    dev = flat.origins.device
    # a rank that was dealt no chunk (more ranks than chunks) contributes zero rows
    cat = {k: torch.cat(v, 0) if v else torch.empty(0, widths_all[k], dtype=torch.float32, device=dev)
           for k, v in outs.items()}
    if world > 1:
        from .dist import gather_image
        packed = torch.cat([cat[k] for k in outs], 1)  # [n_local, 20]
        allp = gather_image(packed, world)
        # undo the round-robin dealing: rank r holds chunks r, r+world, ...
        sizes = [chunks[i].origins.shape[0] for i in range(len(chunks))]
        pos = 0
        starts = {}
        for r in range(world):
            for i in range(r, len(chunks), world):
                starts[i] = pos
                pos += sizes[i]
        full = torch.cat([allp[starts[i]:starts[i] + sizes[i]] for i in range(len(chunks))], 0)
        widths = [cat[k].shape[1] for k in outs]
        cols = torch.split(full, widths, 1)
        cat = dict(zip(outs, cols))
    img = lambda x: None if x is None else x.reshape(1, height, width, -1).permute(0, 3, 1, 2)
    return (img(cat["coarse_rgb"]), img(cat["fine_rgb"]), img(cat["coarse_dep"]), img(cat["fine_dep"]), img(cat["fine_nor"]),
            img(cat["albedo"]), None, img(cat["surface_rgb"]), img(cat["shading"]))
