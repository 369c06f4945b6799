"""Diagnosis of tests/test_gpu_edges.py::test_ragged_and_tiny_batches_match_oracle[257-33]: per-tensor error of the flat
gradient against the gate-forced oracle, rank structure of the layer-0 error, and the margins of the NON-MLP kinks
(relu(n.l) in the Lambertian term, relu(n.d) in the orientation loss) that forcing the MLP gates does not pin.
    python tests/diag/diag_ragged.py [B N]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import forced_gate_sets  # noqa: E402
from oracle import pano_oracle as orc  # noqa: E402
import pano_nerf_amd as pn  # noqa: E402
from pano_nerf_amd.mlp import ORDER, param_layout  # noqa: E402

B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (257, 33)
dev = torch.device("cuda:0")
flat, rgbs_all, radius, _ = orc.synthetic_scene(8, 16, 3, seed=4)
idx = (torch.arange(B) * 5) % flat.origins.shape[0]
rays_c, rgbs = orc.Rays(*[x[idx] for x in flat]), rgbs_all[idx]
rays = pn.Rays(*[x.to(dev) for x in rays_c])
env = pn.generate_lit_rays(10, radius)
env_c = orc.Rays(*[x.cpu() for x in env])
params = orc.init_params(4, 5)
for mode in ("fused_f16x2", "fused", "layerwise"):
    model = pn.PanoMipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5)
    model.mlp.load_state_dict(params)
    model = model.to(dev)
    model.mlp_mode = mode
    model.mlp.debug_keep = True
    outs = model(rays=rays, env_rays=env, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    loss, _ = pn.pano_loss(outs, rays.lossmult, rgbs.to(dev))
    loss.backward()
    g = model.mlp.last_flat_grad.detach().cpu().numpy().astype(np.float64)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    with orc.forced_gates(forced_gate_sets(model, normals=True, surf=True)):
        ref = orc.pano_forward(p, rays_c, env_c, num_samples=N)
        fl = orc.pano_loss(ref, rays_c.lossmult, rgbs)
        fg = torch.autograd.grad(fl, list(p.values()))
    offs, total = param_layout(5)
    print(f"== {mode}: loss {float(loss):.8f} oracle(forced) {float(fl):.8f}")
    for k, r in zip(p.keys(), fg):
        r = r.detach().numpy().astype(np.float64).reshape(-1)
        got = g[offs[k]:offs[k] + r.size]
        e = np.abs(got - r).max() / max(np.abs(r).max(), 1e-30)
        print(f"   {k:28s} err {e:.2e}")
    r0 = fg[0].detach().numpy().astype(np.float64)
    e2 = (g[offs["layers.0.0.weight"]:offs["layers.0.0.weight"] + r0.size].reshape(r0.shape) - r0) / np.abs(r0).max()
    sv = np.linalg.svd(e2, compute_uv=False)
    print("   layers.0.0.weight error singular values:", np.array2string(sv[:6], precision=2))
    # per-ray output errors against the forced oracle
    for nme, i in (("normal", 3), ("albedo", 4), ("surface_rgb", 6), ("shading", 8)):
        a, b = outs[1][i].detach().cpu().numpy(), ref[1][i].detach().numpy()
        pr = np.abs(a - b).max(-1) / max(np.abs(b).max(), 1e-30)
        print(f"   {nme:12s} worst rays {np.argsort(pr)[-3:]} errors {np.sort(pr)[-3:]}")
    print("   ort_loss", float(outs[1][2]), float(ref[1][2]))
    # margins of the non-MLP kinks in the oracle
    nrm = ref[1][3].detach()
    nol = (nrm[:, None, :] * env_c.directions.float()[None]).sum(-1)
    print("   min |n.l| over rays x light dirs:", float(nol.abs().min()), "at ray", int(nol.abs().min(1).values.argmin()))
