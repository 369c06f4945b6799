"""Data-parallel host logic on CPU, world_size 2, gloo: ray sharding, ONE sum all-reduce of the flat
gradient with 1/world scaling reproduces the global-batch gradient, and the inference gather.
The per-rank compute here is the oracle (test stand-in: there is no GPU in this container); on the GPU box
the same `pano_nerf_amd.dist` helpers wrap the HIP render in bench.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pano_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat(grads):
    return torch.cat([g.reshape(-1) for g in grads])


def _loss_grad(params, rays, rgbs, noise, n):
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    outs = orc.mip_forward(p, rays, num_samples=n, noise=noise)
    loss = orc.mip_loss(outs, rays.lossmult, rgbs)
    return _flat(torch.autograd.grad(loss, list(p.values())))


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2 if world <= 2 else 1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pano_nerf_amd.dist import allreduce_flat_grad, gather_image, shard_bounds, shard_rays
    flat, rgbs, _, _ = orc.synthetic_scene(8, 16, 3, seed=4)
    B, N = (12 if world == 2 else 2 * world), 8
    idx = torch.arange(0, B * 5, 5) % flat.origins.shape[0]
    rays = orc.Rays(*[x[idx] for x in flat])
    gen = torch.Generator().manual_seed(3)
    noise = dict(t_rand=torch.rand(B, N + 1, generator=gen), u_rand=torch.rand(B, N + 1, generator=gen) * (1 / (N + 1) - 1.2e-7))
    params = orc.init_params(4, 1)
    sub, (lo, hi) = shard_rays(rays, rank, world)
    assert (lo, hi) == shard_bounds(B, rank, world)
    g = _loss_grad(params, sub, rgbs[idx][lo:hi], {k: v[lo:hi] for k, v in noise.items()}, N)
    allreduce_flat_grad(g, world)
    g = g / world
    img = gather_image(torch.full((hi - lo, 3), float(rank)), world)
    if rank == 0:
        full = _loss_grad(params, rays, rgbs[idx], noise, N)
        np.save(os.path.join(out_dir, "res.npy"), np.array([float((g - full).norm() / full.norm()), img.shape[0],
                                                            float(img[:hi - lo].sum()), float(img[hi - lo:].mean())]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_gradient_equals_global_batch(tmp_path, world):
    """world 8 = the rank count of the BASELINE scaling run (train.py:86-92 under DDP): eight gloo ranks on the CPU."""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rel, rows, first_sum, rest_mean = np.load(tmp_path / "res.npy")
    assert rel < 1e-5, rel            # equal shards + lossmult == 1: mean of shard gradients = global gradient
    B = 12 if world == 2 else 2 * world
    assert rows == B and first_sum == 0.0 and rest_mean == np.mean(np.arange(1, world))


def test_shard_bounds_cover_everything():
    from pano_nerf_amd.dist import shard_bounds
    for n in (1, 7, 4096, 4099):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
