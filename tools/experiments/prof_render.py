"""One warm 512x1024 panorama through pano_nerf_amd.render_image (for rocprofv3 --kernel-trace --stats)."""
import sys, os, time, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np, torch
import pano_nerf_amd as pn
dev = torch.device("cuda:0")
H, W = 512, 1024
rays = pn.generate_pano_rays(H, W, np.eye(4, dtype=np.float32))
env = pn.generate_lit_rays(10, pn.rays.pano_pixel_radius(rays))
model = pn.PanoMipNeRF(num_samples=128, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5).to(dev)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pn.render_image(model, rays, env, H, W, chunk_size=32768)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"seconds_per_pano": dt, "rays_per_s": H * W / dt}))
