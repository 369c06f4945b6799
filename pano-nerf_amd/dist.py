"""Data-parallel step: rays shard across ranks, ONE all-reduce of the flat gradient per step.

Mirrors what Lightning DDP does for the reference (train.py:79-93): every rank holds a
full model replica and renders its own contiguous slice of the ray batch; the only exchange is the
sum all-reduce of the 613 768-float (2.455 MB) gradient block over RCCL/xGMI, folded into Adam as
``grad_scale = 1/world_size``.  Loss terms are per-rank means over equal-sized shards with
lossmult == 1, so the averaged gradient equals the global-batch gradient.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous, balanced [lo, hi) slice of n items for `rank` (first n % world ranks get one more)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_rays(rays, rank, world):
    lo, hi = shard_bounds(rays.origins.shape[0], rank, world)
    return type(rays)(*[x[lo:hi] for x in rays]), (lo, hi)


def allreduce_flat_grad(flat_grad, world):
    """Sum all-reduce in place; the caller divides by `world` (FlatAdam grad_scale)."""
    if world > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


def gather_image(parts, world):
    """Inference: gather per-rank [n_i, C] chunks to every rank (SURVEY.md 8e 'Inference')."""
    if world == 1:
        return parts
    if parts.is_cuda and dist.get_backend() == "gloo":  # rehearsal on fewer GPUs than ranks: gloo gathers on the host
        return gather_image(parts.cpu(), world).to(parts.device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=parts.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([parts.shape[0]], dtype=torch.int64, device=parts.device))
    mx = int(max(s.item() for s in sizes))
    pad = torch.zeros(mx, parts.shape[1], dtype=parts.dtype, device=parts.device)
    pad[:parts.shape[0]] = parts
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[:int(s.item())] for o, s in zip(outs, sizes)], 0)
