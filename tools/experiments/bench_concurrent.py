"""A training step as 1 / 2 / 4 concurrent sub-batches on as many HIP streams (graph replay), B rays in total
(argv[1], default 512).  The sub-batches are independent until the gradient sum, so their GEMM chains fill each
other's first-tile / last-tile bubbles; pano_nerf_amd.concurrent_step is the product form of the 2-stream case."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np, torch
import pano_nerf_amd as pn
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(4)
cams = [np.eye(4, dtype=np.float32) for _ in range(3)]
pool = pn.DeviceRayPool(512, 1024, cams, device=dev)
pool.rgbs = torch.rand(len(pool), 3, device=dev)
env = pool.lit_rays(10)
model = pn.PanoMipNeRF(num_samples=128, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5).to(dev)
opt = pn.FlatAdam(model.mlp, lr=2e-4)
lr_dev = torch.full((1,), 2e-4, device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(8)]

def one(nb):
    rays, gt = pool.sample(nb)
    outs = model(rays=rays, env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    loss, _ = pn.pano_loss(outs, rays.lossmult, gt)
    loss.backward()
    return model.mlp.last_flat_grad

def step_single():
    opt.zero_grad()
    g = one(B)
    opt.step_dev(g, lr_dev, grad_scale=1.0)

def make_parts(k):
    def step():
        opt.zero_grad()
        cur = torch.cuda.current_stream(dev)
        for i in range(1, k):
            streams[i].wait_stream(cur)  # fork BEFORE part 0 is enqueued, or the others would wait for it
        gs = [one(B // k)]
        for i in range(1, k):
            with torch.cuda.stream(streams[i]):
                gs.append(one(B // k))
        for i in range(1, k):
            cur.wait_stream(streams[i])
        g = gs[0]
        for x in gs[1:]:
            g = g + x
        opt.step_dev(g, lr_dev, grad_scale=1.0 / k)
    return step

def bench(fn, name, graph):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    if graph:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn(); fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        run = g.replay
    else:
        run = fn
    for _ in range(5): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name:28s} graph={graph}  {dt*1e3:7.3f} ms/step  {B/dt:9.0f} rays/s", flush=True)

for k in (1, 2, 4):
    bench(make_parts(k), f"{k} part(s) on {k} stream(s)", True)
