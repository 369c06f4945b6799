"""Full parameter gradients (fp32 and fp64 runs of the IMPORTED reference) and bf16-autocast outputs for the two small
golden batches.  Build container only (needs /root/reference); stores arrays only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_grad_golden.py

Inputs (rays, rgbs, recorded noise draws) are read back from the fixtures make_golden.py wrote, so both generators
describe the same batches.  Per case (B64_N32, B16_N128):
  grads_pano_<case>.npz  g32: d loss / d params of the reference in fp32 (flat, order of pn_param_layout);
                         g64: the same from an fp64 run (stored as float32); loss32 / loss64
  grads_mip_<case>.npz   train_g32 / train_g64 (no orientation loss: first order only), trainort_g32 / trainort_g64
  bf16_pano_<case>.npz   the reference under torch.autocast("cpu", dtype=torch.bfloat16): validation-mode outputs and the
                         training loss (pins the tolerance of the plain-bf16 MLP mode, BASELINE configs[1])
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (sets up sys.path, stubs and the reference imports)
import numpy as np  # noqa: E402
import torch  # noqa: E402

orc, rpano, rmipnerf, Rays = mg.orc, mg.rpano, mg.rmipnerf, mg.Rays
ORDER = ([f"layers.{i}.0.{k}" for i in range(8) for k in ("weight", "bias")] +
         ["extra_layer.weight", "extra_layer.bias", "view_layers.0.0.weight", "view_layers.0.0.bias",
          "density_layer.weight", "color_layer.weight", "density_layer.bias", "color_layer.bias"])


class NoiseReplay:
    """torch.rand / Tensor.uniform_ return the recorded draws, in order, in the current default dtype."""

    def __init__(self, draws):
        self.draws = [torch.as_tensor(d) for d in draws]
        self.i = 0

    def __enter__(self):
        self._rand, self._uni = torch.rand, torch.Tensor.uniform_
        rp = self

        def nxt():
            x = rp.draws[rp.i].to(torch.get_default_dtype())
            rp.i += 1
            return x

        def rand(*shape, **kw):
            return nxt().clone()

        def uniform_(self_t, *a, **kw):
            self_t.copy_(nxt())
            return self_t

        torch.rand, torch.Tensor.uniform_ = rand, uniform_
        return self

    def __exit__(self, *a):
        torch.rand, torch.Tensor.uniform_ = self._rand, self._uni


def flat_grad(model):
    table = dict(model.mlp.named_parameters())
    return torch.cat([table[k].grad.detach().reshape(-1) for k in ORDER])


def load(name):
    with np.load(os.path.join(HERE, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def run_pano(N, rays, env, rgbs, draws, dtype):
    params = orc.init_params(4, 5)
    torch.set_default_dtype(dtype)
    try:
        m = rpano.PanoMipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5,
                              num_env_samples=10).to(dtype)
        mg.load_params(m.mlp, {k: v.to(dtype) for k, v in params.items()})
        r = Rays(*[x.to(dtype) for x in rays])
        e = Rays(*[x.to(dtype) for x in env])
        with NoiseReplay(draws):
            outs = m(rays=r, env_rays=e, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
        loss = mg.ref_loss_pano(outs, r.lossmult, rgbs.to(dtype))
        m.zero_grad()
        loss.backward()
        return float(loss), flat_grad(m)
    finally:
        torch.set_default_dtype(torch.float32)


def run_mip(N, rays, rgbs, draws, use_ort, dtype):
    params = orc.init_params(4, 1)
    torch.set_default_dtype(dtype)
    try:
        m = rmipnerf.MipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=1).to(dtype)
        mg.load_params(m.mlp, {k: v.to(dtype) for k, v in params.items()})
        r = Rays(*[x.to(dtype) for x in rays])
        with NoiseReplay(draws):
            outs = m(rays=r, randomized=True, white_bkgd=False, use_ort_loss=use_ort)
        loss = mg.ref_loss_mip(outs, r.lossmult, rgbs.to(dtype), use_ort)
        m.zero_grad()
        loss.backward()
        return float(loss), flat_grad(m)
    finally:
        torch.set_default_dtype(torch.float32)


def main():
    torch.set_num_threads(8)
    env_g = load("raygen_8x16")
    env = Rays(*[torch.from_numpy(env_g["env_" + k]).float() for k in Rays._fields])
    for tag in ("B64_N32", "B16_N128"):
        st, pf, mf = load("stages_" + tag), load("pano_full_" + tag), load("mip_full_" + tag)
        N = st["t_det"].shape[1] - 1
        rays = Rays(*[torch.from_numpy(st["ray_" + k]) for k in Rays._fields])
        rgbs = torch.from_numpy(st["rgbs"])
        draws = [pf["train_t_rand"], pf["train_u_rand"], pf["train_env_rand"]]
        l32, g32 = run_pano(N, rays, env, rgbs, draws, torch.float32)
        assert abs(l32 - float(pf["train/loss"])) < 1e-6 * abs(l32), (l32, float(pf["train/loss"]))  # same batch, same noise
        l64, g64 = run_pano(N, rays, env, rgbs, draws, torch.float64)
        mg.npz("grads_pano_" + tag, g32=g32, g64=g64.float(), loss32=np.float64(l32), loss64=np.float64(l64))
        out = {}
        for mode, use_ort in (("train", False), ("trainort", True)):
            d = [mf[mode + "_t_rand"], mf[mode + "_u_rand"]]
            l, g = run_mip(N, rays, rgbs, d, use_ort, torch.float32)
            assert abs(l - float(mf[mode + "/loss"])) < 1e-6 * abs(l)
            out[mode + "_g32"] = g
            _, g64m = run_mip(N, rays, rgbs, d, use_ort, torch.float64)
            out[mode + "_g64"] = g64m.float()
        mg.npz("grads_mip_" + tag, **out)
        # ---- reference under bf16 autocast (upstream trains under '16-mixed', train.py:86)
        params = orc.init_params(4, 5)
        m = rpano.PanoMipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5,
                              num_env_samples=10)
        mg.load_params(m.mlp, params)
        cap = {}
        names = ("comp_rgb", "distance", "ort_loss", "normal", "albedo", "roughness", "surface_rgb", "diffuse", "shading")
        with torch.autocast("cpu", dtype=torch.bfloat16):
            outs = m(rays=rays, env_rays=env, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
        for lvl, tup in enumerate(outs):
            for nme, v in zip(names, tup):
                if v is not None:
                    cap[f"val/l{lvl}/{nme}"] = v.float()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            with NoiseReplay(draws):
                outs = m(rays=rays, env_rays=env, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
            loss = mg.ref_loss_pano([tuple(None if v is None else v.float() for v in t) for t in outs], rays.lossmult, rgbs)
        m.zero_grad()
        loss.backward()
        cap["train/loss"] = loss.float()
        cap["train/g"] = flat_grad(m).float()
        mg.npz("bf16_pano_" + tag, **cap)


if __name__ == "__main__":
    main()
