"""Worker of tests/test_gpu_bench_dist.py::test_two_rank_render_matches_single_rank (launched by torch.distributed.run):
every rank renders its round-robin share of the chunks of a small panorama on cuda:0, the parts are gathered (gloo),
and rank 0 compares the assembled image with its own single-rank render."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pano_nerf_amd as pn  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.manual_seed(7)  # identical replicas without a broadcast
model = pn.PanoMipNeRF(num_samples=16, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5).to(dev)
H, W = 8, 16
cam = torch.eye(4).numpy()
rays = pn.generate_pano_rays(H, W, cam, 0.0, 10.0, device=dev)
env = pn.generate_lit_rays(10, pn.rays.pano_pixel_radius(rays), device=dev)
for chunk in (48, 200):  # 3 chunks (ragged last one) / a single chunk: rank 1 gets nothing
    parts = pn.render_image(model, rays, env, H, W, chunk_size=chunk, rank=rank, world=world)
    if rank == 0:
        ref = pn.render_image(model, rays, env, H, W, chunk_size=chunk)
        for a, b in zip(parts, ref):
            assert (a is None) == (b is None)
            if a is not None:
                assert a.shape == b.shape and torch.equal(a, b), (chunk, a.shape)
dist.barrier()
if rank == 0:
    print("RENDER_DIST_OK")
dist.destroy_process_group()
