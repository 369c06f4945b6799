import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import pano_nerf_amd as pn
dev = torch.device("cuda:0")
H, W = 512, 1024
rays = pn.generate_pano_rays(H, W, np.eye(4, dtype=np.float32))
env = pn.generate_lit_rays(10, pn.rays.pano_pixel_radius(rays))
model = pn.PanoMipNeRF(num_samples=128, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5).to(dev)
ref = None
for chunk in (8192, 32768):
    for st in (1, 2, 3):
        pn.render_image(model, rays, env, H, W, chunk_size=chunk, streams=st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = pn.render_image(model, rays, env, H, W, chunk_size=chunk, streams=st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if ref is None: ref = out
        same = all((a is None and b is None) or torch.equal(a, b) for a, b in zip(out, ref))
        print(json.dumps({"chunk": chunk, "streams": st, "s_per_pano": round(dt, 4), "rays_per_s": round(H * W / dt), "same_image": same}), flush=True)
