"""Golden vectors for `disable_integration=True` (positional encoding instead of the integrated one: compute_graph zeroes the
covariance, models/pano_mip_nerf.py:241-243, models/mip_nerf.py:213-214), captured from the IMPORTED reference: the val-mode forward
tuples of PanoMipNeRF / MipNeRF constructed with disable_integration=True on 16 rays x 32 samples, and the train-mode loss and full
parameter gradient of the Pano model.  Build container only (same import shim as make_golden.py)."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

orc, rpano, rmipnerf, Rays = mg.orc, mg.rpano, mg.rmipnerf, mg.Rays
B, N = 16, 32
NAMES = ("comp_rgb", "distance", "ort_loss", "normal", "albedo", "roughness", "surface_rgb", "diffuse", "shading")


def main():
    flat, rgbs, radius, _ = orc.synthetic_scene(8, 16, 3, seed=4)
    idx = torch.arange(B) * 7
    rays = Rays(*[x[idx] for x in flat])
    out = {"ray_" + k: getattr(rays, k) for k in Rays._fields}
    out["rgbs"] = rgbs[idx]
    env = orc.generate_lit_rays(10, radius)
    envf = Rays(*[x.float() for x in env])
    net = rpano.PanoMipNeRF(num_samples=N, disable_integration=True, rgb_activation="softplus", rgb_padding=0,
                            mlp_num_density_channels=5, num_env_samples=10)
    mg.load_params(net.mlp, orc.init_params(4, 5))
    outs = net(rays=rays, env_rays=envf, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    for lvl, tup in enumerate(outs):
        for n, v in zip(NAMES, tup):
            if v is not None:
                out[f"pano/l{lvl}/{n}"] = v.detach()
    # train mode (recorded noise): loss of systems/panonerf_system.py:15-75 (make_golden.ref_loss_pano: the reference's hdr_to_ldr,
    # its loss arithmetic) and the full parameter gradient
    with mg.NoiseTap(3) as tap:
        touts = net(rays=rays, env_rays=envf, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    out["t_rand"], out["u_rand"], out["env_rand"] = tap.draws[0], tap.draws[1], tap.draws[2]
    loss = mg.ref_loss_pano(touts, rays.lossmult, rgbs[idx])  # the reference's own hdr_to_ldr / loss arithmetic
    net.zero_grad()
    loss.backward()
    out["train/loss"] = loss.detach()
    out["train/comp_rgb"] = touts[1][0].detach()
    for k, p in net.mlp.named_parameters():
        out["train/grad/" + k] = p.grad.detach().clone()
    mnet = rmipnerf.MipNeRF(num_samples=N, disable_integration=True, rgb_activation="softplus", rgb_padding=0,
                            mlp_num_density_channels=1)
    mg.load_params(mnet.mlp, orc.init_params(4, 1))
    mouts = mnet(rays=rays, randomized=False, white_bkgd=False, use_ort_loss=True)
    for lvl, tup in enumerate(mouts):
        out[f"mip/l{lvl}/comp_rgb"], out[f"mip/l{lvl}/distance"] = tup[0].detach(), tup[1].detach()
    out["mip/l1/normal"], out["mip/l1/ort_loss"] = mouts[1][3].detach(), mouts[1][2].detach()
    mg.npz("disable_integration_B16_N32", **out)


if __name__ == "__main__":
    main()
