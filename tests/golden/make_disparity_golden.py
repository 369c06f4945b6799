"""Golden vectors for `disparity=True` (coarse samples linear in inverse depth, models/mip.py:134-136), captured from the IMPORTED
reference: sample_along_rays (deterministic and randomized) on 16 rays with near = 0.5, far = 10, and the val-mode forward tuples
of MipNeRF / PanoMipNeRF constructed with disparity=True.  Build container only (same import shim as make_golden.py)."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

orc, rmip, rpano, rmipnerf, Rays = mg.orc, mg.rmip, mg.rpano, mg.rmipnerf, mg.Rays
B, N = 16, 32


def main():
    flat, rgbs, radius, _ = orc.synthetic_scene(8, 16, 3, seed=4)
    idx = torch.arange(B) * 7
    rays = Rays(*[x[idx] for x in flat])
    rays = rays._replace(near=torch.full_like(rays.near, 0.5))  # disparity needs near > 0
    out = {"ray_" + k: getattr(rays, k) for k in Rays._fields}
    t_det, (m_det, c_det) = rmip.sample_along_rays(rays.origins, rays.directions, rays.radii, N, rays.near, rays.far, False, True, "cone")
    with mg.NoiseTap(5) as tap:
        t_rnd, (m_rnd, c_rnd) = rmip.sample_along_rays(rays.origins, rays.directions, rays.radii, N, rays.near, rays.far, True, True, "cone")
    out.update(t_det=t_det.contiguous(), mean_det=m_det, cov_det=c_det, t_rand=tap.draws[0], t_rnd=t_rnd, mean_rnd=m_rnd, cov_rnd=c_rnd)
    env = orc.generate_lit_rays(10, radius)
    envf = Rays(*[x.float() for x in env])
    net = rpano.PanoMipNeRF(num_samples=N, disparity=True, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5,
                            num_env_samples=10)
    mg.load_params(net.mlp, orc.init_params(4, 5))
    outs = net(rays=rays, env_rays=envf, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    names = ("comp_rgb", "distance", "ort_loss", "normal", "albedo", "roughness", "surface_rgb", "diffuse", "shading")
    for lvl, tup in enumerate(outs):
        for n, v in zip(names, tup):
            if v is not None:
                out[f"pano/l{lvl}/{n}"] = v.detach()
    mnet = rmipnerf.MipNeRF(num_samples=N, disparity=True, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=1)
    mg.load_params(mnet.mlp, orc.init_params(4, 1))
    mouts = mnet(rays=rays, randomized=False, white_bkgd=False, use_ort_loss=False)
    for lvl, tup in enumerate(mouts):
        out[f"mip/l{lvl}/comp_rgb"], out[f"mip/l{lvl}/distance"] = tup[0].detach(), tup[1].detach()
    mg.npz("disparity_B16_N32", **out)


if __name__ == "__main__":
    main()
