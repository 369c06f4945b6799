"""Worker of tests/test_gpu_bench_dist.py::test_ddp_wrapped_module_trains (launched by torch.distributed.run, 2 ranks,
gloo, both on cuda:0): the reference's own data-parallel recipe — torch DistributedDataParallel around the model
(Lightning DDP, train.py:92), torch.optim.Adam over model.mlp.parameters() (systems/base_system.py:82), the
training_step call pattern (systems/panonerf_system.py:30-42) — on the flat-view parameters of PanoMipNeRF."""
import os
import sys

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pano_nerf_amd as pn  # noqa: E402
from oracle import pano_oracle as orc  # noqa: E402  (checker: synthetic scene + initial weights only)

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
N, B = 16, 48
flat, rgbs, radius, _ = orc.synthetic_scene(8, 16, 3, seed=4)
model = pn.PanoMipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5,
                       num_env_samples=10)
torch.manual_seed(100 + rank)  # DIFFERENT initial weights per rank: DDP's constructor must broadcast rank 0's
for p in model.mlp.parameters():
    p.data.add_(0.01 * torch.randn_like(p))
model = model.to(dev)
assert model.mlp.is_flat()
ddp = DDP(model, device_ids=None)  # gloo on device tensors
assert model.mlp.is_flat(), "DDP's parameter broadcast must keep the 24 parameters views of the flat block"
opt = torch.optim.Adam(model.mlp.parameters(), lr=2e-4)
env = pn.generate_lit_rays(10, radius)
for step in range(2):
    idx = (torch.arange(B) * 7 + 13 * step + 29 * rank) % flat.origins.shape[0]  # each rank its own rays
    rays = pn.Rays(*[x[idx].to(dev) for x in flat])
    outs = ddp(rays=rays, env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    loss, _ = pn.pano_loss(outs, rays.lossmult, rgbs[idx].to(dev))
    # the packed weights THIS forward read (the pack buffer is rebuilt in place by every forward) must be the pack of the
    # weights as they stand now (the previous optimizer.step() included), and differ from the previous step's
    st = torch.cuda.current_stream().cuda_stream
    planes = pn.render._planes_of(model.mlp_mode)
    used = model.mlp._chain[planes].clone()
    fresh = torch.empty_like(used)
    pn._lib.call("pn_chain_pack", model.mlp.flat.data_ptr(), 5, planes, fresh.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(used, fresh), "the forward ran on stale packed weights"
    if step > 0:
        assert not torch.equal(used, prev_used), "optimizer.step() did not reach the packed weights"
    prev_used = used
    opt.zero_grad()
    loss.backward()
    opt.step()
    assert model.mlp.is_flat() and bool(torch.isfinite(model.mlp.flat).all())
# replicas stay identical: same averaged gradient, same update
mine = model.mlp.flat.detach().cpu()
both = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(both, mine)
assert torch.equal(both[0], both[1]), float((both[0] - both[1]).abs().max())
g = torch.cat([p.grad.reshape(-1) for _, p in model.mlp.named_in_order()]).cpu()
gs = [torch.empty_like(g) for _ in range(world)]
dist.all_gather(gs, g)
assert torch.equal(gs[0], gs[1]) and float(g.abs().max()) > 0
dist.barrier()
if rank == 0:
    print("DDP_OK")
dist.destroy_process_group()
