"""Which ray carries the rank-one gradient difference of the [257-33] case?  Per-ray gradients (one ray per call, GPU layer-wise
exact-fp32 path vs the gate-forced oracle), then the worst ray's per-sample quantities."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import forced_gate_sets  # noqa: E402
from oracle import pano_oracle as orc  # noqa: E402
import pano_nerf_amd as pn  # noqa: E402
from pano_nerf_amd.mlp import param_layout  # noqa: E402

B, N = 257, 33
dev = torch.device("cuda:0")
flat, rgbs_all, radius, _ = orc.synthetic_scene(8, 16, 3, seed=4)
idx = (torch.arange(B) * 5) % flat.origins.shape[0]
env = pn.generate_lit_rays(10, radius)
env_c = orc.Rays(*[x.cpu() for x in env])
params = orc.init_params(4, 5)
model = pn.PanoMipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5)
model.mlp.load_state_dict(params)
model = model.to(dev)
model.mlp_mode = sys.argv[1] if len(sys.argv) > 1 else "layerwise"
model.mlp.debug_keep = True
offs, total = param_layout(5)
lo, n0 = offs["layers.0.0.weight"], 256 * 96
res = []
for b in range(B):
    ii = idx[b:b + 1]
    rays_c, rgbs = orc.Rays(*[x[ii] for x in flat]), rgbs_all[ii]
    rays = pn.Rays(*[x.to(dev) for x in rays_c])
    outs = model(rays=rays, env_rays=env, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    loss, _ = pn.pano_loss(outs, rays.lossmult, rgbs.to(dev))
    loss.backward()
    g = model.mlp.last_flat_grad.detach().cpu().numpy().astype(np.float64)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    with orc.forced_gates(forced_gate_sets(model, normals=True, surf=True)):
        ref = orc.pano_forward(p, rays_c, env_c, num_samples=N)
        fl = orc.pano_loss(ref, rays_c.lossmult, rgbs)
        fg = torch.autograd.grad(fl, list(p.values()))
    r0 = fg[0].detach().numpy().astype(np.float64).reshape(-1)
    res.append((float(np.abs(g[lo:lo + n0] - r0).max()), float(np.abs(r0).max()), int(idx[b])))
res = np.array(res)
order = np.argsort(res[:, 0])[::-1]
print("max |layer-0 grad| over rays:", res[:, 1].max())
for b in order[:6]:
    print(f"ray {b} (pool row {int(res[b, 2])}): abs err {res[b, 0]:.3e}  own max {res[b, 1]:.3e}  rel {res[b, 0] / res[b, 1]:.2e}")
b = int(order[0])
ii = idx[b:b + 1]
rays_c, rgbs = orc.Rays(*[x[ii] for x in flat]), rgbs_all[ii]
p = {k: v.clone() for k, v in params.items()}
with torch.no_grad():
    ref = orc.pano_forward(p, rays_c, env_c, num_samples=N)
print("worst ray outputs: comp", ref[1][0], "dist", ref[1][1], "normal", ref[1][3], "shading", ref[1][8], "ort", ref[1][2])
print("directions", rays_c.directions, "origin", rays_c.origins)
nol = (ref[1][3][:, None, :] * env_c.directions.float()[None]).sum(-1)
print("n.l", nol)
