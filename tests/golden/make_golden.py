"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference); never shipped to or
run on the GPU box.  Nothing from the reference is copied: this script feeds the
reference's own functions seeded inputs and stores inputs + outputs as .npz.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Import shim: the reference's `datasets/` must shadow the
HuggingFace package of the same name; cv2 / OpenEXR / Imath are absent and are
stubbed as empty modules (none of their functions is called on this path).
`systems/*` needs pytorch_lightning (absent), so the loss of
systems/panonerf_system.py:15-75 is re-assembled here from the reference's own
`hdr_to_ldr` and the arithmetic written there.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)
for name in ("cv2", "Imath"):
    sys.modules[name] = types.ModuleType(name)
_exr = types.ModuleType("OpenEXR")
_exr.InputFile = _exr.OutputFile = _exr.Header = object
sys.modules["OpenEXR"] = _exr

import numpy as np  # noqa: E402
import torch  # noqa: E402

import models.mip as rmip  # noqa: E402
import models.mip_nerf as rmipnerf  # noqa: E402
import models.pano_mip_nerf as rpano  # noqa: E402
import utils.surface_rendering as rsurf  # noqa: E402
import utils.sampling as rsampling  # noqa: E402
from datasets.pano_datasets import PanoDataset  # noqa: E402
from datasets.base_datasets import Rays  # noqa: E402
from utils.lr_schedule import MipLRDecay  # noqa: E402
from torch.nn.functional import normalize  # noqa: E402

from oracle import pano_oracle as orc  # noqa: E402  (only for init_params / scene poses)

torch.manual_seed(0)
torch.set_num_threads(8)


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if v is None:
            continue
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items()})


class NoiseTap:
    """Replace torch.rand / Tensor.uniform_ by a PCG64 stream and record draws."""

    def __init__(self, seed):
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self.draws = []

    def __enter__(self):
        self._rand, self._uni = torch.rand, torch.Tensor.uniform_
        tap = self

        def rand(*shape, **kw):
            if len(shape) == 1 and isinstance(shape[0], (list, tuple)):
                shape = tuple(shape[0])
            x = torch.from_numpy(tap.rng.random(shape, dtype=np.float32))
            tap.draws.append(x.clone())
            return x

        def uniform_(self_t, from_=0.0, to=1.0, **kw):
            to = kw.get("to", to)
            x = torch.from_numpy(tap.rng.random(tuple(self_t.shape), dtype=np.float32)) * np.float32(to)
            tap.draws.append(x.clone())
            self_t.copy_(x)
            return self_t

        torch.rand, torch.Tensor.uniform_ = rand, uniform_
        return self

    def __exit__(self, *a):
        torch.rand, torch.Tensor.uniform_ = self._rand, self._uni


def make_dataset(h, w, c2ws, near=0.0, far=10.0):
    ds = PanoDataset.__new__(PanoDataset)
    ds.h, ds.w, ds.near, ds.far, ds.reform_cam = h, w, near, far, False
    ds.camtoworlds = [np.asarray(c, dtype=np.float32) for c in c2ws]
    ds._generate_rays()
    return ds


def flat_rays(ds):
    return Rays(*[torch.from_numpy(np.concatenate([a.reshape(-1, a.shape[-1]) for a in getattr(ds.rays, k)], 0)
                                   .astype(np.float32)) for k in Rays._fields])


def pick(rays, idx):
    return Rays(*[x[idx] for x in rays])


def load_params(module, params):
    module.load_state_dict({k: v.clone() for k, v in params.items()})


def ref_loss_pano(outputs, mask, rgbs, hp=orc.DEFAULT_LOSS):
    gt = rsurf.hdr_to_ldr(rgbs[..., :3], dtype="uint8")
    (rgb_c, *_), (rgb_f, _, ort, _, alb, _, sf, _, _) = outputs
    rgb_c, rgb_f = rsurf.hdr_to_ldr(rgb_c), rsurf.hdr_to_ldr(rgb_f)
    vc = (mask * (rgb_c - gt) ** 2).sum() / mask.sum()
    vf = (mask * (rgb_f - gt) ** 2).sum() / mask.sum()
    loss = hp["coarse_loss_mult"] * vc + vf
    if sf is not None:
        sfl = rsurf.hdr_to_ldr(sf)
        loss = loss + hp["surface_loss"] * (mask * (sfl - gt) ** 2).sum() / mask.sum()
        loss = loss + hp["chrom_loss"] * ((normalize(gt, dim=-1) - normalize(alb, dim=-1)) ** 2).mean()
    if ort is not None:
        loss = loss + hp["ort_loss"] * ort
    return loss


def ref_loss_mip(outputs, mask, rgbs, use_ort, hp=orc.DEFAULT_LOSS):
    gt = rsurf.hdr_to_ldr(rgbs[..., :3], dtype="uint8")
    (c, *_), (f, _, ort, _) = outputs
    c, f = rsurf.hdr_to_ldr(c), rsurf.hdr_to_ldr(f)
    loss = hp["coarse_loss_mult"] * (mask * (c - gt) ** 2).sum() / mask.sum() + (mask * (f - gt) ** 2).sum() / mask.sum()
    if use_ort:
        loss = loss + hp["ort_loss"] * ort
    return loss


def grad_summary(model, prefix, rng_seed=11):
    rng = np.random.Generator(np.random.PCG64(rng_seed))
    out = {}
    for k, p in model.mlp.named_parameters():
        g = p.grad.detach().reshape(-1)
        n = g.numel()
        idx = rng.integers(0, n, size=min(256, n))
        out[f"{prefix}/{k}/norm"] = np.float64(g.double().norm())
        out[f"{prefix}/{k}/idx"] = idx.astype(np.int64)
        out[f"{prefix}/{k}/val"] = g[idx].numpy()
    return out


def main():
    # ------------------------------------------------------------------ a2 / a3
    _, _, _, c2ws = orc.synthetic_scene(8, 16, n_cam=3, seed=4)
    rot = np.eye(4, dtype=np.float32)
    th = 0.3
    rot[:3, :3] = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], np.float32)
    rot[:3, 3] = [0.1, -0.2, 0.3]
    ds = make_dataset(8, 16, c2ws + [rot])
    env = ds.generate_lit_rays(num=10)
    npz("raygen_8x16", c2ws=np.stack(c2ws + [rot]), radius=np.float64(ds.radii),
        **{k: np.stack(getattr(ds.rays, k)) for k in Rays._fields},
        **{"env_" + k: getattr(env, k).numpy() for k in Rays._fields})
    ds2 = make_dataset(64, 128, c2ws[:1])
    env2 = ds2.generate_lit_rays(num=10)
    sub = (slice(None, None, 7), slice(None, None, 9))
    npz("raygen_64x128", c2ws=np.stack(c2ws[:1]), radius=np.float64(ds2.radii),
        **{k: getattr(ds2.rays, k)[0][sub] for k in Rays._fields},
        env_radii=env2.radii.numpy())
    d, th_, ph_ = rsampling.sample_dir_by_pano((8, 16))
    npz("sampling_helpers", pano_dirs=d, uniform_dirs=rsampling.sample_dir_by_unifrom(10))

    pool = flat_rays(ds)
    rng = np.random.Generator(np.random.PCG64(4))
    scene_rays, scene_rgbs, _, _ = orc.synthetic_scene(8, 16, n_cam=3, seed=4)

    # ------------------------------------------------------- stage-level captures
    for (B, N) in ((64, 32), (16, 128)):
        tag = f"B{B}_N{N}"
        idx = torch.from_numpy(rng.integers(0, 3 * 8 * 16, size=B))
        rays = pick(pool, idx)
        rgbs = scene_rgbs[idx]
        envf = Rays(*[x.float() for x in env])
        params = orc.init_params(4, 5)
        pano = rpano.PanoMipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0,
                                 mlp_num_density_channels=5, num_env_samples=10)
        load_params(pano.mlp, params)
        cap = dict(idx=idx, rgbs=rgbs, **{"ray_" + k: getattr(rays, k) for k in Rays._fields})

        # a4/a5 deterministic + randomized
        t_det, (m_det, c_det) = rmip.sample_along_rays(rays.origins, rays.directions, rays.radii, N, rays.near,
                                                       rays.far, False, False, "cone")
        with NoiseTap(100 + B) as tap:
            t_rnd, (m_rnd, c_rnd) = rmip.sample_along_rays(rays.origins, rays.directions, rays.radii, N, rays.near,
                                                           rays.far, True, False, "cone")
        cap.update(t_det=t_det.contiguous(), mean_det=m_det, cov_det=c_det, t_rand=tap.draws[0], t_rnd=t_rnd,
                   mean_rnd=m_rnd, cov_rnd=c_rnd)
        # a6/a7
        enc = rmip.integrated_pos_enc((m_rnd, c_rnd), 0, 16)
        venc = rmip.pos_enc(rays.viewdirs, 0, 4, True)
        cap.update(enc_head=enc[:4], viewenc=venc)
        # a8/a9
        with torch.no_grad():
            raw_rgb, raw_den = pano.mlp(enc, venc)
        cap.update(raw_rgb=raw_rgb, raw_den=raw_den)
        rgb = torch.nn.functional.softplus(raw_rgb)
        sig = torch.nn.functional.softplus(raw_den[..., :1] - 1)
        # a10
        comp, dist, acc, w = rmip.volumetric_rendering(rgb, sig, t_rnd, rays.directions, False)
        compw, *_ = rmip.volumetric_rendering(rgb, sig, t_rnd, rays.directions, True)
        cap.update(comp_rgb=comp, distance=dist, acc=acc, weights=w, comp_rgb_white=compw)
        # a11 both modes
        t_re_det, (m_re, c_re) = rmip.resample_along_rays(rays.origins, rays.directions, rays.radii, t_rnd, w.clone(),
                                                          False, "cone", True, 0.01)
        with NoiseTap(200 + B) as tap:
            t_re_rnd, _ = rmip.resample_along_rays(rays.origins, rays.directions, rays.radii, t_rnd, w.clone(),
                                                   True, "cone", True, 0.01)
        cap.update(t_resample_det=t_re_det, mean_resample_det=m_re, cov_resample_det=c_re, u_rand=tap.draws[0],
                   t_resample_rnd=t_re_rnd)
        # degenerate weights (all zero) -> padding path of the PDF
        t_zero, _ = rmip.resample_along_rays(rays.origins, rays.directions, rays.radii, t_rnd, torch.zeros_like(w),
                                             False, "cone", True, 0.0)
        cap.update(t_resample_zero=t_zero)
        # a13
        xs = rays.origins + rays.directions * dist.view(-1, 1)
        with NoiseTap(300 + B) as tap:
            lt, (lm, lc), ld = rmip.sample_each_points(xs.view(-1, 1, 3), envf.directions, 10, envf.near, envf.far,
                                                       envf.radii, True)
        cap.update(env_rand=tap.draws[0], lit_t=lt[:40], lit_mean=lm[:40], lit_cov=lc[:40], lit_dirs=ld[:40])
        # a14
        g = torch.Generator().manual_seed(5)
        env_rgb = torch.rand(B, 10, 3, generator=g)
        alb = torch.rand(B, 3, generator=g)
        nrm = normalize(torch.randn(B, 3, generator=g), dim=-1)
        srgb, dif, _, shd = rsurf.surface_rendering(env_rgb, alb, nrm, None, ld.view(B, 10, 3), rays.viewdirs,
                                                    envf.lossmult, output_sd=True)
        cap.update(sr_env=env_rgb, sr_albedo=alb, sr_normal=nrm, sr_rgb=srgb, sr_diffuse=dif, sr_shading=shd)
        # a15 tone map
        x = torch.rand(B, 3, generator=g) * 4
        cap.update(tm_in=x, tm_out=rsurf.hdr_to_ldr(x), tm_out_u8=rsurf.hdr_to_ldr(x, dtype="uint8"))
        npz("stages_" + tag, **cap)

        # ---------------------------------------------------- full forwards, Pano
        full = dict(idx=idx)
        outs = pano(rays=rays, env_rays=envf, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
        names = ("comp_rgb", "distance", "ort_loss", "normal", "albedo", "roughness", "surface_rgb", "diffuse",
                 "shading")
        for lvl, tup in enumerate(outs):
            for nme, v in zip(names, tup):
                if v is not None:
                    full[f"val/l{lvl}/{nme}"] = v
        with NoiseTap(400 + B) as tap:
            outs = pano(rays=rays, env_rays=envf, randomized=True, white_bkgd=False, enable_surf=True,
                        use_ort_loss=True)
        full.update(train_t_rand=tap.draws[0], train_u_rand=tap.draws[1], train_env_rand=tap.draws[2])
        for lvl, tup in enumerate(outs):
            for nme, v in zip(names, tup):
                if v is not None:
                    full[f"train/l{lvl}/{nme}"] = v
        loss = ref_loss_pano(outs, rays.lossmult, rgbs)
        pano.zero_grad()
        loss.backward()
        full["train/loss"] = loss
        full.update(grad_summary(pano, "train/grad"))
        # surface off / ort off variant (slots must be None)
        outs2 = pano(rays=rays, env_rays=envf, randomized=False, white_bkgd=True, enable_surf=False,
                     use_ort_loss=False)
        full["nosurf/l1/comp_rgb"] = outs2[1][0]
        full["nosurf/l1/normal"] = outs2[1][3]
        full["nosurf/none_slots"] = np.array([i for i, v in enumerate(outs2[1]) if v is None])
        # fp64 reference of the ill-conditioned outputs (gate vs fp64)
        pano64 = rpano.PanoMipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0,
                                   mlp_num_density_channels=5, num_env_samples=10).double()
        load_params(pano64.mlp, {k: v.double() for k, v in params.items()})
        torch.set_default_dtype(torch.float64)
        try:
            r64 = Rays(*[x.double() for x in rays])
            e64 = Rays(*[x.double() for x in env])
            o64 = pano64(rays=r64, env_rays=e64, randomized=False, white_bkgd=False, enable_surf=True,
                         use_ort_loss=True)
        finally:
            torch.set_default_dtype(torch.float32)
        for nme, v in zip(names, o64[1]):
            if v is not None:
                full[f"val64/l1/{nme}"] = v
        npz("pano_full_" + tag, **full)

        # ----------------------------------------------------- full forwards, Mip
        mparams = orc.init_params(4, 1)
        mip = rmipnerf.MipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=1)
        load_params(mip.mlp, mparams)
        mfull = dict(idx=idx)
        mnames = ("comp_rgb", "distance", "ort_loss", "normal")
        for mode, use_ort in (("val", True), ("valno", False)):
            outs = mip(rays=rays, randomized=False, white_bkgd=False, use_ort_loss=use_ort)
            for lvl, tup in enumerate(outs):
                for nme, v in zip(mnames, tup):
                    if v is not None:
                        mfull[f"{mode}/l{lvl}/{nme}"] = v
        for mode, use_ort in (("train", False), ("trainort", True)):
            with NoiseTap(500 + B) as tap:
                outs = mip(rays=rays, randomized=True, white_bkgd=False, use_ort_loss=use_ort)
            mfull[f"{mode}_t_rand"], mfull[f"{mode}_u_rand"] = tap.draws[0], tap.draws[1]
            for lvl, tup in enumerate(outs):
                for nme, v in zip(mnames, tup):
                    if v is not None:
                        mfull[f"{mode}/l{lvl}/{nme}"] = v
            loss = ref_loss_mip(outs, rays.lossmult, rgbs, use_ort)
            mip.zero_grad()
            loss.backward()
            mfull[f"{mode}/loss"] = loss
            mfull.update(grad_summary(mip, f"{mode}/grad"))
        npz("mip_full_" + tag, **mfull)

    # ----------------------------------------------- render_image-style chunking (a16)
    ds1 = make_dataset(8, 16, c2ws[:1])
    img_rays = Rays(*[torch.from_numpy(getattr(ds1.rays, k)[0].astype(np.float32))[None] for k in Rays._fields])
    chunks, _ = rmip.rearrange_render_image(img_rays, 32)
    params = orc.init_params(4, 5)
    pano = rpano.PanoMipNeRF(num_samples=32, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5,
                             num_env_samples=10)
    load_params(pano.mlp, params)
    envf = Rays(*[x.float() for x in ds1.generate_lit_rays(num=10)])
    keep = {k: [] for k in ("coarse_rgb", "fine_rgb", "coarse_dep", "fine_dep", "normal", "albedo", "surface_rgb",
                            "shading")}
    with torch.no_grad():
        for ch in chunks:
            (c_rgb, c_dep, *_), (f_rgb, f_dep, _, f_nor, alb, _, sf, _, sd) = pano(
                rays=ch, env_rays=envf, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
            for k, v in zip(keep, (c_rgb, f_rgb, c_dep, f_dep, f_nor, alb, sf, sd)):
                keep[k].append(v)
    comp = {k: torch.cat(v, 0).view(1, 8, 16, -1).permute(0, 3, 1, 2) for k, v in keep.items()}
    npz("render_image_8x16", n_chunks=np.int64(len(chunks)), **comp)

    # -------------------------------------------------------------- lr schedule (f1)
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=2e-4)
    sch = MipLRDecay(opt, 2e-4, 2e-5, 44000, 120, 0.01)
    lrs = []
    for step in range(0, 44001):
        sch.last_epoch = step
        if step in (0, 1, 60, 120, 121, 1000, 22000, 44000):
            lrs.append((step, sch.get_lr()[0]))
    npz("lr_schedule", steps=np.array([s for s, _ in lrs]), lrs=np.array([l for _, l in lrs], np.float64))


if __name__ == "__main__":
    main()
