"""Inference throughput: one 512x1024 panorama through pano_nerf_amd.render_image (GPU box)."""
import sys, os, time, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np, torch
import pano_nerf_amd as pn
dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 1024)
rays = pn.generate_pano_rays(H, W, np.eye(4, dtype=np.float32))
env = pn.generate_lit_rays(10, pn.rays.pano_pixel_radius(rays))
model = pn.PanoMipNeRF(num_samples=128, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5).to(dev)
for chunk in (512, 8192, 32768, 65536):
    pn.render_image(model, rays, env, H, W, chunk_size=chunk) if chunk >= 8192 else None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = pn.render_image(model, rays, env, H, W, chunk_size=chunk)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"pano": f"{H}x{W}", "chunk_size": chunk, "seconds_per_pano": dt, "rays_per_s": H * W / dt}))
