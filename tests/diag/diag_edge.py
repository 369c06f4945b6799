"""Run one ragged-batch step (B, N from argv) with weight-gradient batching on/off; prints loss and grad norm."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pano_oracle as orc
import pano_nerf_amd as pn

import os
if os.environ.get("PN_TRACE"):
    from pano_nerf_amd import _lib as _L, render as _R
    _call = _L.call
    def traced(name, *a):
        print("call", name, [hex(x) if isinstance(x, int) and x > 1 << 32 else x for x in a if not hasattr(x, "_length_")], flush=True)
        r = _call(name, *a)
        torch.cuda.synchronize()
        return r
    _L.call = traced
    for fn in ("empty", "zeros"):
        def mk(orig):
            def f(*a, **k):
                t = orig(*a, **k)
                if t.is_cuda:
                    print("alloc", hex(t.data_ptr()), hex(t.data_ptr() + t.numel() * t.element_size()), t.numel() * t.element_size(), flush=True)
                return t
            return f
        setattr(torch, fn, mk(getattr(torch, fn)))

B, N, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
flat, rgbs, radius, _ = orc.synthetic_scene(8, 16, 3, seed=4)
idx = (torch.arange(B) * 5) % flat.origins.shape[0]
rays_c = orc.Rays(*[x[idx] for x in flat])
rays = pn.Rays(*[x.to(dev) for x in rays_c])
env = pn.generate_lit_rays(10, radius)
model = pn.PanoMipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5)
model.mlp.load_state_dict(orc.init_params(4, 5))
model = model.to(dev)
model.batch_weight_grads = bool(batch)
if len(sys.argv) > 4:
    model.overlap_weight_grads = bool(int(sys.argv[4]))
outs = model(rays=rays, env_rays=env, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
torch.cuda.synchronize()
print("forward ok", flush=True)
loss, _ = pn.pano_loss(outs, rays.lossmult, rgbs[idx].to(dev))
loss.backward()
torch.cuda.synchronize()
print("B", B, "N", N, "batch", batch, "loss", float(loss), "gnorm", float(model.mlp.last_flat_grad.norm()), flush=True)
