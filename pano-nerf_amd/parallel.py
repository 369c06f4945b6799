"""Concurrent sub-batches: one training step as `parts` independent sub-batches on separate HIP streams.

Rays are independent until the gradient sum, so a batch can be cut into sub-batches whose forward + backward chains
run side by side.  Every GEMM of one chain is followed by a GEMM that depends on it, so a chain alone leaves the
matrix cores idle while a kernel's first tiles load and its last tiles store (4.9 us + ~18 us of a 72 us workgroup
life at 4096 rays; a whole wave of tiles at 512 rays); a second, independent chain fills those gaps.  Measured on one
MI355X (`tools/experiments/bench_concurrent.py`, graph replay): 512 rays 9.53 -> 9.06 ms, 1024 rays 16.86 -> 16.23 ms, 4096 rays 60.9 -> 59.6 ms per step;
four parts are no better than two.

The result is the gradient of the mean loss over the whole batch: sub-batch losses are means over their rays, so the
parts are weighted by their share of the rays (`systems/panonerf_system.py:44-67` computes the same means; the
reference's DDP averages equal shards the same way).  Only the order of the random jitter draws differs from one
call over the whole batch.
"""
import torch

from .dist import shard_bounds

_STREAMS = {}


def _streams(dev, n):
    key = (dev.type, dev.index)
    pool = _STREAMS.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


def concurrent_step(model, loss_fn, rays, rgbs, parts=2, **forward_kwargs):
    """forward + loss + backward of `rays` / `rgbs` in `parts` concurrent sub-batches.

    loss_fn(outputs, lossmult, rgbs) -> (loss, terms), e.g. `pano_nerf_amd.pano_loss`.  Returns
    (loss [], flat_grad [n_params], outputs of the first sub-batch); `flat_grad` is already weighted, pass it to
    `FlatAdam.step[_dev]` with the scale the caller would use for one call (1 / world size)."""
    B = rays.origins.shape[0]
    parts = max(1, min(int(parts), B))
    dev = rays.origins.device
    cur = torch.cuda.current_stream(dev)
    # the transposed / split weight copies are rebuilt after every optimizer step: do it once, BEFORE the fork, so no
    # sub-batch reads them while another one's first forward is still writing them
    from .render import _planes_of
    planes = _planes_of(model.mlp_mode)
    model.mlp._frozen = False
    if planes:
        model.mlp.chain_packed(cur.cuda_stream, planes)
    else:
        model.mlp.packed(cur.cuda_stream)
    side = _streams(dev, parts - 1)
    for s in side:
        s.wait_stream(cur)
    grads, losses, first = [], [], None
    model.mlp.defer_param_grads = True  # _RenderFn.backward leaves the flat gradient in mlp.last_flat_grad only
    model.mlp._frozen = True            # the sub-batches reuse the packs built above
    try:
        return _run_parts(model, loss_fn, rays, rgbs, parts, forward_kwargs, B, cur, side, grads, losses)
    finally:
        model.mlp.defer_param_grads = False
        model.mlp._frozen = False


def _run_parts(model, loss_fn, rays, rgbs, parts, forward_kwargs, B, cur, side, grads, losses):
    first = None
    for i in range(parts):
        lo, hi = shard_bounds(B, i, parts)
        sub = type(rays)(*[x[lo:hi] for x in rays])
        with torch.cuda.stream(cur if i == 0 else side[i - 1]):
            outs = model(rays=sub, **forward_kwargs)
            loss, _ = loss_fn(outs, sub.lossmult, rgbs[lo:hi])
            loss.backward()
            w = (hi - lo) / B
            grads.append(model.mlp.last_flat_grad if w == 1.0 else model.mlp.last_flat_grad * w)
            losses.append(loss.detach() * w)
        if i == 0:
            first = outs
    for s in side:
        cur.wait_stream(s)
    g, total = grads[0], losses[0]
    for x, l in zip(grads[1:], losses[1:]):
        g = g + x
        total = total + l
    model.mlp.last_flat_grad = g
    for p, v in zip((q for _, q in model.mlp.named_in_order()), model.mlp.grad_views(g)):
        p.grad = v  # what one backward over the whole batch would have left for a torch optimizer
    return total, g, first
