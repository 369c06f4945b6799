"""Gradient parity AT SCALE: the whole flat gradient of a training step over >= 393 k MLP rows per
level against the oracle, and the two launch schedules of the weight gradients against each other.

The golden-vector tests (tests/test_gpu_grads.py) stop at 64 rays and the edge cases at 42 k MLP rows; the bench job sums
~1.9 M rows per layer.  Here 1024 rays x 128 + 128 samples (131 072 rows per level and 102 400 env-light rows: 393 k rows with
the second-order segment) run through the default kernel mode and through the oracle (CPU, "fast" normals = one reverse sweep,
the same algorithm) ON THE GATE DECISIONS THE KERNELS TOOK, so that every entry of every gradient tensor can be compared
pointwise at 1e-4 of its tensor's max (a ReLU whose pre-activation is ~1e-7 flips under any fp32 summation order and makes
the GRADIENT discontinuous; see tests/test_gpu_grads.py).  Reference: models/pano_mip_nerf.py:295-313 through autograd,
systems/panonerf_system.py:15-75."""
import numpy as np
import pytest
import torch

from conftest import check_flat_grad_pointwise, forced_gate_sets, rel_err, report_worst
from oracle import pano_oracle as orc
from test_gpu_full import LOOSE, NAMES9, dev, make_pano, to_dev

pytestmark = pytest.mark.gpu


def _scene(B, N, seed):
    import pano_nerf_amd as pn
    S = N + 1
    flat, rgbs, radius, _ = orc.synthetic_scene(64, 128, 3, seed=4)
    idx = torch.randint(0, flat.origins.shape[0], (B,), generator=torch.Generator().manual_seed(seed))
    rays_c = orc.Rays(*[x[idx] for x in flat])
    env = pn.generate_lit_rays(10, radius)
    env_c = orc.Rays(*[x.cpu().float() for x in env])
    gen = torch.Generator().manual_seed(seed + 1)
    noise = dict(t_rand=torch.rand(B, S, generator=gen), u_rand=torch.rand(B, S, generator=gen) * (1.0 / S - 1.2e-7),
                 env_rand=torch.rand(1, 11, generator=gen))
    return rays_c, rgbs[idx], env, env_c, noise


# (the default mode at 1024 rays; the fp32-tensor variant and the concurrent weight-gradient schedule at 256 rays: the oracle's
# fp32 + fp64 forward and backward passes take ~2 minutes of host time per 1024 rays)
@pytest.mark.parametrize("mode,B,overlap", [("fused_f16x2", 1024, False), ("fused_f16x2_t32", 256, False), ("fused_f16x2", 256, True)])
def test_flat_gradient_at_scale_against_the_oracle_on_identical_gates(mode, B, overlap):
    import pano_nerf_amd as pn
    N = 128
    rays_c, gt_c, env, env_c, noise = _scene(B, N, 1024)
    rays, gt = to_dev(pn.Rays(*rays_c)), gt_c.to(dev())
    model = make_pano(N)
    model.mlp_mode = mode
    model.overlap_weight_grads = overlap
    model.mlp.debug_keep = True
    model.noise_override = noise
    outs = model(rays=rays, env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    loss, _ = pn.pano_loss(outs, rays.lossmult, gt)
    loss.backward()
    torch.cuda.synchronize()
    got = model.mlp.last_flat_grad.detach().cpu().numpy().astype(np.float64)
    gate_sets = forced_gate_sets(model, normals=True, surf=True)
    model.mlp.debug_pack = None
    p = {k: v.clone().requires_grad_(True) for k, v in orc.init_params(4, 5).items()}
    # (a) the oracle as it stands, forward only, in fp32 AND fp64.  SURVEY.md 7's gate for the outputs derived from the density
    # gradient: median relative error <= 1e-4, >= 99 % of the elements within 1e-3 of the tensor scale, error against an fp64
    # evaluation <= 2 x the reference-fp32's own.  Over 1024 rays x 128 samples the 99 % do NOT hold for fp32 itself: the oracle's
    # fp32 run has only 93 - 96 % of the `normal` elements within 1e-3 of its own fp64 run (a ReLU whose pre-activation is ~1e-7
    # flips under any fp32 summation order and moves a whole ray; with 128 samples a ray, several per cent of the rays hold one) -
    # and every kernel mode, the exact-fp32 layer-wise one included, measures the same 92 - 95 % against the fp32 oracle.  So the
    # fraction is stated against the fp64 run, next to the fp32 oracle's own: ours >= min(0.99, theirs - 0.02).
    with torch.no_grad():
        ref = orc.pano_forward(p, rays_c, env_c, num_samples=N, noise=noise)
        p64 = {k: v.detach().double() for k, v in p.items()}
        r64 = orc.Rays(*[x.double() for x in rays_c])
        e64 = orc.Rays(*[x.double() for x in env_c])
        ref64 = orc.pano_forward(p64, r64, e64, num_samples=N, noise={k: v.double() for k, v in noise.items()})
    for nme, v, r, r6 in zip(NAMES9, outs[1], ref[1], ref64[1]):
        if v is None or nme == "ort_loss":
            continue
        a, b, c = (x.detach().cpu().numpy().astype(np.float64) for x in (v, r, r6))
        scale = max(float(np.abs(c).max()), 1e-12)
        if nme in LOOSE:
            rel = np.abs(a - b) / (np.abs(b) + 1e-6)
            assert float(np.median(rel)) < 1e-4, (nme, "median", float(np.median(rel)))
            ours, theirs = float(np.mean(np.abs(a - c) <= 1e-3 * scale)), float(np.mean(np.abs(b - c) <= 1e-3 * scale))
            assert ours >= min(0.99, theirs - 0.02), (nme, "fraction of elements within 1e-3 of the fp64 run: ours, the fp32 oracle's", ours, theirs)
            assert float(np.median(np.abs(a - c))) <= 2 * float(np.median(np.abs(b - c))) + 1e-6 * scale, (nme, "median error vs fp64")
            report_worst(f"{B}-ray step: fraction of {nme} elements beyond 1e-3 of the fp64 oracle [{mode}]", 1.0 - ours)
            report_worst(f"{B}-ray step: fraction of {nme} elements beyond 1e-3 of the fp64 oracle [the fp32 oracle itself]", 1.0 - theirs)
        else:
            assert rel_err(a, b) < 1e-4, (nme, rel_err(a, b))
    # (b) the oracle on the kernels' gate decisions, with its gradient: every output and every gradient entry pointwise
    with orc.forced_gates(gate_sets):
        refg = orc.pano_forward(p, rays_c, env_c, num_samples=N, noise=noise)
        ref_loss = orc.pano_loss(refg, rays_c.lossmult, gt_c)
        ref_g = torch.autograd.grad(ref_loss, list(p.values()))
    assert abs(float(loss) - float(ref_loss)) < 1e-4 * abs(float(ref_loss)), (float(loss), float(ref_loss))
    refg64 = None
    for i, (nme, v, r) in enumerate(zip(NAMES9, outs[1], refg[1])):
        if v is None:
            continue
        e = rel_err(v.detach().cpu().numpy(), r.detach().numpy())
        if e >= 1e-4 and nme in LOOSE:
            # forcing the gates removes the discontinuity, not every ill-conditioning: a ray whose per-sample normals nearly cancel
            # amplifies fp32 rounding in the fp32 ORACLE too (one element in 3072 at 1.2e-4 here).  Then SURVEY.md 7's third
            # criterion, on the same gates: error against the fp64 evaluation <= max(1e-4, 2 x the fp32 oracle's own)
            if refg64 is None:
                with orc.forced_gates(gate_sets), torch.no_grad():
                    refg64 = orc.pano_forward(p64, r64, e64, num_samples=N, noise={k: x.double() for k, x in noise.items()})
            c = refg64[1][i].numpy()
            ours, theirs = rel_err(v.detach().cpu().numpy(), c), rel_err(r.detach().numpy(), c)
            assert ours <= max(1e-4, 2 * theirs), (f"l1/{nme} on identical gates vs fp64: ours, the fp32 oracle's own", ours, theirs)
            report_worst(f"{B}-ray step: {nme} on identical gates vs the fp64 oracle, tensor-scale error [{mode}]", ours)
            report_worst(f"{B}-ray step: {nme} on identical gates vs the fp64 oracle, tensor-scale error [the fp32 oracle itself]", theirs)
            continue
        assert e < 1e-4, (f"l1/{nme} on identical gates", e)
    by_name = {k: x.detach().numpy() for k, x in zip(p.keys(), ref_g)}

    def forced64():
        # the same gate-forced oracle in fp64: a tensor beyond 1e-4 of the fp32 oracle must be within max(1e-4, 2 x the fp32
        # oracle's own error) of it (conftest.check_flat_grad_pointwise).  Needed here: layers.0.0.weight of the fp32
        # oracle is itself 1.7e-4 of the tensor max off its fp64 evaluation on this batch - in both tensor formats alike
        pp = {k: v.detach().double().requires_grad_(True) for k, v in p.items()}
        with orc.forced_gates(gate_sets):
            o64 = orc.pano_forward(pp, r64, e64, num_samples=N, noise={k: x.double() for k, x in noise.items()})
            l64 = orc.pano_loss(o64, r64.lossmult, gt_c.double())
            g64 = torch.autograd.grad(l64, list(pp.values()))
        return {k: x.detach().numpy() for k, x in zip(pp.keys(), g64)}

    worst = check_flat_grad_pointwise(got, by_name, 5, tol=1e-4, ref64_fn=forced64)
    report_worst(f"{B} rays x 128+128 samples: flat gradient vs the oracle on identical gates, worst tensor [{mode}"
                 f"{', concurrent weight gradients' if overlap else ''}]", worst)
    print(f"gradient at scale {mode} overlap={overlap}: loss {float(loss):.6f} (oracle {float(ref_loss):.6f}), worst tensor {worst:.2e}")


@pytest.mark.parametrize("B,N", [(512, 128), (97, 33)])
def test_concurrent_weight_gradients_match_the_inline_schedule(B, N):
    """overlap_weight_grads: the weight gradients of each evaluation on a side stream beside the next evaluation's chains, each
    kernel family on its share of the CUs (render.py, _RenderFn.backward) - same operands, another order of the partial sums:
    the gradient agrees with the in-line schedule's to fp32 summation noise, is bit-reproducible, and the outputs are
    bit-identical (the forward is the same launch sequence)."""
    import pano_nerf_amd as pn
    rays_c, gt_c, env, env_c, noise = _scene(B, N, 77 + B)
    rays, gt = to_dev(pn.Rays(*rays_c)), gt_c.to(dev())

    def run(overlap, cw=None, ww=None):
        model = make_pano(N)
        model.overlap_weight_grads = overlap
        if cw is not None:
            model.overlap_chain_wgs, model.overlap_wgrad_wgs = cw, ww
        model.noise_override = noise
        outs = model(rays=rays, env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
        loss, _ = pn.pano_loss(outs, rays.lossmult, gt)
        loss.backward()
        torch.cuda.synchronize()
        return outs, loss.detach().clone(), model.mlp.last_flat_grad.clone()

    o0, l0, g0 = run(False)
    o1, l1, g1 = run(True)
    o2, l2, g2 = run(True)
    o3, l3, g3 = run(True, 64, 192)
    assert torch.equal(l0, l1) and torch.equal(o0[1][0], o1[1][0]) and torch.equal(o0[1][6], o1[1][6])
    assert torch.equal(g1, g2)  # deterministic: every reduction into the flat gradient happens on ONE stream, in a fixed order
    scale = float(g0.abs().max())
    assert float((g1 - g0).abs().max()) <= 3e-6 * scale, float((g1 - g0).abs().max()) / scale
    assert float((g3 - g0).abs().max()) <= 3e-6 * scale, float((g3 - g0).abs().max()) / scale
    # per tensor as well: a small tensor must not hide behind the largest one
    from pano_nerf_amd.mlp import ORDER, param_layout
    offs, total = param_layout(5)
    order = sorted(ORDER, key=lambda k: offs[k])
    for i, k in enumerate(order):
        lo, hi = offs[k], offs[order[i + 1]] if i + 1 < len(order) else total
        s = float(g0[lo:hi].abs().max())
        assert float((g1[lo:hi] - g0[lo:hi]).abs().max()) <= 1e-5 * s + 1e-30, (k, float((g1[lo:hi] - g0[lo:hi]).abs().max()) / max(s, 1e-30))


def test_mip_model_with_concurrent_weight_gradients():
    """MipNeRF (no env light; normals only with the orientation loss) through the concurrent schedule."""
    import pano_nerf_amd as pn
    B, N = 256, 64
    rays_c, gt_c, _, _, noise = _scene(B, N, 5)
    rays, gt = to_dev(pn.Rays(*rays_c)), gt_c.to(dev())
    for use_ort in (False, True):
        res = []
        for overlap in (False, True):
            model = make_pano(N, nc=1)
            model.overlap_weight_grads = overlap
            model.noise_override = dict(t_rand=noise["t_rand"], u_rand=noise["u_rand"])
            outs = model(rays=rays, randomized=True, white_bkgd=False, use_ort_loss=use_ort)
            loss, _ = pn.mip_loss(outs, rays.lossmult, gt, use_ort=use_ort)
            loss.backward()
            torch.cuda.synchronize()
            res.append((loss.detach().clone(), model.mlp.last_flat_grad.clone()))
        assert torch.equal(res[0][0], res[1][0])
        scale = float(res[0][1].abs().max())
        assert float((res[0][1] - res[1][1]).abs().max()) <= 3e-6 * scale
