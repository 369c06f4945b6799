"""Parameter-gradient parity at the 1e-4 contract, against FULL gradients of the imported reference (fp32 run and
fp64 run; tests/golden/make_grad_golden.py), for every MLP mode of the HIP path.

Three tests:
  * tensors with no trunk ReLU gate below them (extra_layer / view_layers / color_layer; the density head when the
    normals are not in the loss): max |err| over ALL entries <= 1e-4 of the tensor's max, against the reference's fp32
    gradients;
  * tensors upstream of ReLU gates (the trunk; the density head with normals in the loss): a gate whose pre-activation
    is ~1e-7 flips under any fp32 summation order and the GRADIENT jumps (the value does not) — the reference's own
    fp32 run differs from its fp64 run by up to 2.8e-3 of the tensor max there, in first order already.  Gated on
    median <= 1e-4, relative L2 <= 5e-3, every entry <= 2e-4 once the rank-<=4 part of the error that at most four flipped
    gates explain is removed (the reference's own fp32-vs-fp64 error drops from 2.8e-3 to <= 6e-5 under that operation), and
    <= 2x the reference-fp32's own median error against the fp64 gradients (check_second_order);
  * the strong form, test_gate_consistent_gradients_pointwise: with the gate decisions of the GPU kernels forced into
    the oracle, EVERY entry of EVERY tensor (second order included) agrees to 1e-4 of the tensor max.
"""
import numpy as np
import pytest
import torch

from oracle import pano_oracle as orc
from test_gpu_full import dev, env_of, make_pano, rays_of, to_dev

pytestmark = pytest.mark.gpu
CASES = ["B64_N32", "B16_N128"]
MODES = ["fused", "fused_f16x2", "fused_f16x2_t32", "layerwise"]
FIRST_ORDER = ("extra_layer", "view_layers", "color_layer")


def tensors(nc):
    from pano_nerf_amd.mlp import ORDER, param_layout
    offs, total = param_layout(nc)
    order = sorted(ORDER, key=lambda k: offs[k])
    return [(k, offs[k], offs[order[i + 1]] if i + 1 < len(order) else total) for i, k in enumerate(order)]


def check_first_order(name, got, ref):
    scale = max(float(np.abs(ref).max()), 1e-30)
    e = float(np.abs(got - ref).max()) / scale
    assert e <= 1e-4, (name, e)


def strip_gate_flips(err2d, max_rank=4):
    """A flipped ReLU gate of ONE sample changes a weight gradient by a rank-one term (that sample's delta / tangent times
    its input activations), in the flipped layer and in every layer upstream.  Returns the error matrix with its best
    rank-`max_rank` approximation removed: what cannot be explained by at most `max_rank` such flips."""
    u, sv, vt = np.linalg.svd(err2d, full_matrices=False)
    k = min(max_rank, sv.size)
    return err2d - (u[:, :k] * sv[:k]) @ vt[:k]


def check_second_order(name, got, ref32, ref64, shape=None):
    """Tensors upstream of ReLU gates.  Against the reference's fp32 gradients: median <= 1e-4, relative L2 <= 5e-3, and
    — for weight matrices — EVERY entry within 2e-4 after removing the part of the error that at most four gate flips
    explain (a rank-<=4 term; see strip_gate_flips); for bias vectors >= 99 % of the entries <= 1e-3 wherever the
    reference's own fp32 run meets that against its fp64 run.  Against the fp64 gradients: our median error <= 2x the reference-fp32's own (+5e-5: on these
    16 / 64-ray batches both are set by a Poisson-distributed handful of gate flips, not by arithmetic — the reference's
    fp32 run itself differs from its fp64 run by up to 2.8e-3 of the tensor max on FIRST-order gradients,
    tests/golden/grads_mip_B16_N128.npz, train mode, layers.6.0.weight)."""
    scale = max(float(np.abs(ref64).max()), 1e-30)
    err = np.abs(got - ref32) / scale
    assert float(np.median(err)) <= 1e-4, (name, "median", float(np.median(err)))
    assert float(np.linalg.norm(got - ref32) / max(np.linalg.norm(ref32), 1e-30)) <= 5e-3, (name, "relative L2")
    ours = np.abs(got - ref64) / scale
    theirs = np.abs(ref32 - ref64) / scale
    if shape is not None and min(shape) > 8:
        resid = np.abs(strip_gate_flips(((got - ref32) / scale).reshape(shape)))
        # (the reference's own fp32-vs-fp64 error, up to 2.8e-3 of the tensor max, shrinks to <= 6e-5 under the same
        # operation — 4e-7 in first order: the flips ARE the error)
        assert float(resid.max()) <= 2e-4, (name, "max |err| beyond 4 gate flips", float(resid.max()))
    else:
        frac_ours, frac_theirs = float(np.mean(err <= 1e-3)), float(np.mean(theirs <= 1e-3))
        need = 0.99 if frac_theirs >= 0.999 else max(0.5, frac_theirs - 0.3)
        assert frac_ours >= need, (name, "fraction within 1e-3", frac_ours, frac_theirs)
    a, b = float(np.median(ours)), float(np.median(theirs))
    assert a <= 2 * b + 5e-5, (name, "median vs fp64", a, b)


def shape_of(k, nc):
    if not k.endswith("weight"):
        return None
    if k.startswith("layers."):
        l = int(k.split(".")[1])
        return (256, 96 if l == 0 else (352 if l == 5 else 256))
    return {"extra_layer.weight": (256, 256), "view_layers.0.0.weight": (128, 283), "density_layer.weight": (nc, 256),
            "color_layer.weight": (3, 128)}[k]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", CASES)
def test_pano_full_gradients(golden, case, mode):
    import pano_nerf_amd as pn
    g, s, gg = golden("pano_full_" + case), golden("stages_" + case), golden("grads_pano_" + case)
    N = s["t_det"].shape[1] - 1
    rays, env = to_dev(rays_of(s)), to_dev(env_of(golden))
    model = make_pano(N)
    model.mlp_mode = mode
    model.noise_override = dict(t_rand=torch.from_numpy(g["train_t_rand"]), u_rand=torch.from_numpy(g["train_u_rand"]),
                                env_rand=torch.from_numpy(g["train_env_rand"]))
    outs = model(rays=rays, env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    loss, _ = pn.pano_loss(outs, rays.lossmult, torch.from_numpy(s["rgbs"]).to(dev()))
    assert abs(float(loss) - float(gg["loss32"])) < 1e-4 * abs(float(gg["loss32"]))
    loss.backward()
    got = model.mlp.last_flat_grad.detach().cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    g32, g64 = gg["g32"].astype(np.float64), gg["g64"].astype(np.float64)
    for k, lo, hi in tensors(5):
        if k.startswith(FIRST_ORDER):
            check_first_order(k, got[lo:hi], g32[lo:hi])
        else:
            check_second_order(k, got[lo:hi], g32[lo:hi], g64[lo:hi], shape_of(k, 5))


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", CASES)
def test_mip_full_gradients(golden, case, mode):
    import pano_nerf_amd as pn
    g, s, gg = golden("mip_full_" + case), golden("stages_" + case), golden("grads_mip_" + case)
    N = s["t_det"].shape[1] - 1
    rays = to_dev(rays_of(s))
    model = make_pano(N, nc=1)
    model.mlp_mode = mode
    for tag, use_ort in (("train", False), ("trainort", True)):
        model.noise_override = dict(t_rand=torch.from_numpy(g[tag + "_t_rand"]), u_rand=torch.from_numpy(g[tag + "_u_rand"]))
        for p in model.mlp.parameters():
            p.grad = None
        outs = model(rays=rays, randomized=True, white_bkgd=False, use_ort_loss=use_ort)
        loss, _ = pn.mip_loss(outs, rays.lossmult, torch.from_numpy(s["rgbs"]).to(dev()), use_ort=use_ort)
        loss.backward()
        got = model.mlp.last_flat_grad.detach().cpu().numpy().astype(np.float64)
        g32 = gg[tag + "_g32"].astype(np.float64)
        for k, lo, hi in tensors(1):
            # pointwise 1e-4 for the tensors no trunk gate sits below; the trunk (and, with the orientation loss, the
            # density head) is upstream of ReLU gates whose flips make the GRADIENT discontinuous even in first order
            if k.startswith(FIRST_ORDER) or (not use_ort and k.startswith("density_layer")):
                check_first_order(f"{tag}/{k}", got[lo:hi], g32[lo:hi])
            else:
                check_second_order(f"{tag}/{k}", got[lo:hi], g32[lo:hi], gg[tag + "_g64"].astype(np.float64)[lo:hi], shape_of(k, 1))


@pytest.mark.parametrize("case", CASES)
def test_plain_bf16_mode_against_the_autocast_reference(golden, case):
    """mlp_mode = 'fused_bf16' (BASELINE configs[1]): bf16 operands, fp32 accumulate.  Its tolerance is pinned by the
    imported reference run under torch.autocast('cpu', dtype=bfloat16) (tests/golden/bf16_pano_*.npz): our deviation from
    the reference's fp32 outputs may be at most 3x the autocast reference's own deviation (+1e-3 of the scale), and the
    outputs agree with the autocast reference to 1e-2 of the scale; normal-derived outputs (ill-conditioned already in
    fp32) on the median only."""
    import pano_nerf_amd as pn
    g, s, b, gg = golden("pano_full_" + case), golden("stages_" + case), golden("bf16_pano_" + case), golden("grads_pano_" + case)
    N = s["t_det"].shape[1] - 1
    rays, env = to_dev(rays_of(s)), to_dev(env_of(golden))
    model = make_pano(N)
    model.mlp_mode = "fused_bf16"
    with torch.no_grad():
        outs = model(rays=rays, env_rays=env, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    names = ("comp_rgb", "distance", "ort_loss", "normal", "albedo", "roughness", "surface_rgb", "diffuse", "shading")
    for lvl, tup in enumerate(outs):
        for nme, v in zip(names, tup):
            key = f"val/l{lvl}/{nme}"
            if v is None:
                continue
            got, ref32, ref16 = v.detach().cpu().numpy(), g[key], b[key]
            scale = max(float(np.abs(ref32).max()), 1e-30)
            if nme in ("normal", "surface_rgb", "diffuse", "shading"):
                ours_m = float(np.median(np.abs(got - ref32))) / scale
                theirs_m = float(np.median(np.abs(ref16 - ref32))) / scale
                assert ours_m <= 2 * theirs_m + 1e-2, (key, "median", ours_m, theirs_m)
                continue
            ours, theirs = float(np.abs(got - ref32).max()) / scale, float(np.abs(ref16 - ref32).max()) / scale
            assert ours <= 3 * theirs + 1e-3, (key, ours, theirs)
            assert float(np.abs(got - ref16).max()) / scale <= 1e-2, (key, "vs autocast reference")
    model.noise_override = dict(t_rand=torch.from_numpy(g["train_t_rand"]), u_rand=torch.from_numpy(g["train_u_rand"]),
                                env_rand=torch.from_numpy(g["train_env_rand"]))
    outs = model(rays=rays, env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    loss, _ = pn.pano_loss(outs, rays.lossmult, torch.from_numpy(s["rgbs"]).to(dev()))
    ref_loss, ref16_loss = float(gg["loss32"]), float(b["train/loss"])
    assert abs(float(loss) - ref_loss) <= 3 * abs(ref16_loss - ref_loss) + 2e-3 * abs(ref_loss), (float(loss), ref_loss, ref16_loss)
    loss.backward()
    got = model.mlp.last_flat_grad.detach().cpu().numpy().astype(np.float64)
    g32, g16 = gg["g32"].astype(np.float64), b["train/g"].astype(np.float64)
    cos = lambda a, c: float(np.dot(a, c) / (np.linalg.norm(a) * np.linalg.norm(c)))
    # a bf16 gradient is a noisy estimate of the fp32 one (the autocast reference: cosine 0.80-0.85); ours must be at
    # least as well aligned with the fp32 gradient as the autocast reference is, minus a margin of 0.1
    assert np.isfinite(got).all()
    assert cos(got, g32) >= cos(g16, g32) - 0.1, (cos(got, g32), cos(g16, g32))


# ---------------------------------------------------------------------------- gate-consistent pointwise parity
from conftest import gates_of  # noqa: E402  (shared with tests/test_gpu_edges.py, tests/test_gpu_full.py)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", CASES)
def test_gate_consistent_gradients_pointwise(golden, case, mode):
    """The strong form of gradient parity: the oracle is run with the ReLU gate decisions the GPU kernels took (read back
    from their bit masks), which removes the one ill-conditioned ingredient — and then EVERY output (normal, surface_rgb,
    shading, ort_loss included) and EVERY entry of EVERY gradient tensor, first- and second-order alike, must agree to 1e-4
    of the tensor max (measured: ~1e-5 .. 5e-5)."""
    import pano_nerf_amd as pn
    g, s = golden("pano_full_" + case), golden("stages_" + case)
    N = s["t_det"].shape[1] - 1
    rays_c, env_c = rays_of(s), env_of(golden)
    rays, env = to_dev(rays_c), to_dev(env_c)
    model = make_pano(N)
    model.mlp_mode = mode
    model.mlp.debug_keep = True
    noise = dict(t_rand=torch.from_numpy(g["train_t_rand"]), u_rand=torch.from_numpy(g["train_u_rand"]),
                 env_rand=torch.from_numpy(g["train_env_rand"]))
    model.noise_override = noise
    outs = model(rays=rays, env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    pack = model.mlp.debug_pack
    e0, e1, ee = pack[5], pack[6], pack[7]
    gates = [gates_of(e, mode != "layerwise") for e in (e0, e1, ee)]
    rgbs = torch.from_numpy(s["rgbs"])
    loss, _ = pn.pano_loss(outs, rays.lossmult, rgbs.to(dev()))
    loss.backward()
    got = model.mlp.last_flat_grad.detach().cpu().numpy().astype(np.float64)
    p = {k: v.clone().requires_grad_(True) for k, v in orc.init_params(4, 5).items()}
    with orc.forced_gates([gates[0], gates[1], gates[1], gates[2]]):  # level 0, level 1, level-1 normals, env light
        ref = orc.pano_forward(p, rays_c, orc.Rays(*[x.float() for x in env_c]), num_samples=N, noise=noise)
        ref_loss = orc.pano_loss(ref, rays_c.lossmult, rgbs)
        ref_g = torch.autograd.grad(ref_loss, list(p.values()))
    assert abs(float(loss) - float(ref_loss)) < 1e-4 * abs(float(ref_loss))
    # every OUTPUT too, the normal-derived ones included: pointwise 1e-4 of the tensor scale
    names = ("comp_rgb", "distance", "ort_loss", "normal", "albedo", "roughness", "surface_rgb", "diffuse", "shading")
    for lvl in (0, 1):
        for nme, got_o, ref_o in zip(names, outs[lvl], ref[lvl]):
            assert (got_o is None) == (ref_o is None), (lvl, nme)
            if got_o is not None:
                a_, b_ = got_o.detach().cpu().numpy().astype(np.float64), ref_o.detach().numpy().astype(np.float64)
                e_ = float(np.abs(a_ - b_).max()) / max(float(np.abs(b_).max()), 1e-30)
                assert e_ <= 1e-4, (f"l{lvl}/{nme}", e_)
    by_name = {k: x.detach().numpy().astype(np.float64).reshape(-1) for k, x in zip(p.keys(), ref_g)}
    worst = 0.0
    for k, lo, hi in tensors(5):
        r = by_name[k]
        e = float(np.abs(got[lo:hi] - r).max()) / max(float(np.abs(r).max()), 1e-30)
        worst = max(worst, e)
        assert e <= 1e-4, (k, e)
    print(f"gate-consistent gradients {case} {mode}: worst tensor max-error {worst:.2e}")
    from conftest import report_worst
    report_worst(f"gate-consistent gradients vs the oracle, worst tensor max-error / tensor max [{mode}]", worst)
