"""CPU oracle for the Pano-NeRF volumetric-rendering hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain PyTorch fp32 restatement, written
from the math, of the reference algorithm.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker; the product path (``pano_nerf_amd``) never
routes through it and fails loudly when the HIP library is missing.

Parity pin: every function here is checked in ``tests/test_oracle_golden.py``
against golden vectors captured from the imported reference in the build
container (``tests/golden/make_golden.py`` is the generating script).

Each function cites the reference lines (relative to the upstream tree) it
restates.  All tensors are fp32 unless noted; ``B`` rays, ``N`` samples per
level, ``S = N + 1`` fence posts, ``M = B * N`` samples.
"""
from __future__ import annotations

import collections
import math

import numpy as np
import torch

# datasets/base_datasets.py:13-16
Rays = collections.namedtuple(
    "Rays",
    ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far", "noise_var"),
)

HALF_PI_F32 = float(np.float32(0.5) * np.float32(np.pi))  # fl32(0.5 * fl32(pi)) = 1.5707964
EPS32 = float(torch.finfo(torch.float32).eps)


# --------------------------------------------------------------------------- rays
def generate_pano_rays(h, w, c2ws, near=0.0, far=10.0):
    """Equirectangular ray generation.  datasets/pano_datasets.py:152-216
    (same direction formula as utils/sampling.py:5-20).

    Returns (Rays of lists, one [h, w, C] float32 array per camera; pixel radius).
    ``radii`` is cast to float32 (numpy>=2 promotes it to float64); the scalar
    radius is returned unrounded because the light rays are built from it.
    """
    col = np.arange(w, dtype=np.float32)[None, :].repeat(h, 0)
    row = np.arange(h, dtype=np.float32)[:, None].repeat(w, 1)
    theta = -(col + 0.5) / w * 2 * np.pi
    phi = (row + 0.5) / h * np.pi
    sp = np.sin(phi)
    cam = np.stack([sp * np.sin(theta), np.cos(phi), sp * np.cos(theta)], -1)
    noise = (sp * np.pi / w).reshape(h, w, 1)
    out = {k: [] for k in Rays._fields}
    radius = None
    for c2w in c2ws:
        c2w = np.asarray(c2w)
        d = (cam @ c2w[:3, :3].T).copy()
        o = np.broadcast_to(c2w[:3, -1], d.shape).copy()
        v = d / np.linalg.norm(d, axis=-1, keepdims=True)
        ones = np.ones_like(o[..., :1])
        mid = d[h // 2]
        dx = np.sqrt(np.sum((mid[:-1] - mid[1:]) ** 2, -1))
        dx = np.concatenate([dx, dx[-2:-1]], 0)  # last column repeats column w-3
        rad = np.tile(dx[None, :], (h, 1))[..., None] * 2 / np.sqrt(12)
        if radius is None:
            radius = rad[0, 0, 0]  # numpy>=2: float64 scalar, fed to generate_lit_rays
        out["origins"].append(o)
        out["directions"].append(d)
        out["viewdirs"].append(v)
        out["radii"].append(rad.astype(np.float32))
        out["lossmult"].append(1 * ones)
        out["near"].append(near * ones)
        out["far"].append(far * ones)
        out["noise_var"].append(noise.copy())
    return Rays(**out), radius


def generate_lit_rays(num, radii, near=0.0, far=10.0, dtype=torch.float16):
    """Golden-spiral light directions, stored in fp16.
    datasets/pano_datasets.py:218-263 (== utils/sampling.py:23-38)."""
    ga = np.pi * (3.0 - np.sqrt(5.0))
    i = np.arange(num, dtype=np.float64)
    y = 1 - (i / float(num - 1)) * 2
    r = np.sqrt(1 - y * y)
    th = ga * i
    d = np.stack([np.cos(th) * r, y, np.sin(th) * r], -1)
    v = d / np.linalg.norm(d, axis=-1, keepdims=True)
    one = np.ones((num, 1))
    fields = dict(
        origins=np.zeros_like(d), directions=d, viewdirs=v,
        radii=np.full((num, 1), float(radii)), lossmult=(4 * np.pi / num) * one,
        near=near * one, far=far * one, noise_var=0 * one)
    return Rays(*[torch.tensor(fields[k]).to(dtype) for k in Rays._fields])


# ----------------------------------------------------------------------- sampling
def cast_rays(t, origins, directions, radii):
    """Conical frustum -> diagonal Gaussian.  models/mip.py:67-89, 36-64 (stable
    branch 51-58), 8-22 (diagonal branch)."""
    t0, t1 = t[..., :-1], t[..., 1:]
    mu = (t0 + t1) / 2
    hw = (t1 - t0) / 2
    den = 3 * mu ** 2 + hw ** 2
    t_mean = mu + (2 * mu * hw ** 2) / den
    t_var = (hw ** 2) / 3 - (4 / 15) * ((hw ** 4 * (12 * mu ** 2 - hw ** 2)) / den ** 2)
    r_var = radii ** 2 * ((mu ** 2) / 4 + (5 / 12) * hw ** 2 - 4 / 15 * (hw ** 4) / den)
    d2 = directions ** 2
    null = 1 - d2 / (d2.sum(-1, keepdim=True) + 1e-10)
    mean = directions[..., None, :] * t_mean[..., None] + origins[..., None, :]
    cov = t_var[..., None] * d2[..., None, :] + r_var[..., None] * null[..., None, :]
    return mean, cov


def _jitter(t, t_rand):
    mids = 0.5 * (t[..., 1:] + t[..., :-1])
    upper = torch.cat([mids, t[..., -1:]], -1)
    lower = torch.cat([t[..., :1], mids], -1)
    return lower + (upper - lower) * t_rand


def sample_along_rays(origins, directions, radii, num_samples, near, far, t_rand=None, disparity=False):
    """Stratified coarse sampling, linear in depth or (disparity) in inverse depth.  models/mip.py:113-151.
    ``t_rand`` [B, S] uniform noise (``None`` = deterministic)."""
    lin = torch.linspace(0.0, 1.0, num_samples + 1)
    if disparity:
        t = 1.0 / (1.0 / near * (1.0 - lin) + 1.0 / far * lin)  # models/mip.py:134-136
    else:
        t = near + (far - near) * lin
    if t_rand is not None:
        t = _jitter(t, t_rand)
    else:
        t = t.expand(origins.shape[0], num_samples + 1)
    return t, cast_rays(t, origins, directions, radii)


def sample_each_points(x_surf, env_dirs, num_samples, near, far, radii, t_rand=None):
    """Light rays from every surface point along every env direction.
    models/mip.py:154-194.  x_surf [B,3]; env_* [D,*]; t_rand [1, Ne+1] shared."""
    B, D = x_surf.shape[0], env_dirs.shape[0]
    o = x_surf[:, None, :].expand(B, D, 3).reshape(-1, 3)
    d = env_dirs[None].expand(B, D, 3).reshape(-1, 3)
    rep = lambda a: a[None].expand(B, D, 1).reshape(-1, 1)
    rad, nr, fr = rep(radii), rep(near), rep(far)
    t = nr + (fr - nr) * torch.linspace(0.0, 1.0, num_samples + 1)
    if t_rand is not None:
        t = _jitter(t, t_rand)
    return t, cast_rays(t, o, d, rad), d


def piecewise_constant_pdf(bins, weights, num_samples, u_rand=None):
    """Inverse-CDF sampling from sorted bins.  models/mip.py:240-301.
    ``u_rand`` [B, num_samples] uniform in [0, 1/num_samples - eps) or None."""
    eps = 1e-5
    wsum = weights.sum(-1, keepdim=True)
    pad = torch.clamp(eps - wsum, min=0)
    weights = weights + pad / weights.shape[-1]
    wsum = wsum + pad
    pdf = weights / wsum
    cdf = torch.clamp(torch.cumsum(pdf[..., :-1], -1), max=1.0)
    z = torch.zeros_like(cdf[..., :1])
    cdf = torch.cat([z, cdf, z + 1], -1)
    if u_rand is not None:
        s = 1 / num_samples
        u = (torch.arange(num_samples) * s)[None] + u_rand
        u = torch.clamp(u, max=1.0 - EPS32)
    else:
        u = torch.linspace(0.0, 1.0 - EPS32, num_samples).expand(cdf.shape[0], num_samples)
    u = u.contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    lo = torch.clamp(idx - 1, min=0)
    hi = torch.clamp(idx, max=cdf.shape[-1] - 1)
    c0, c1 = cdf.gather(-1, lo), cdf.gather(-1, hi)
    b0, b1 = bins.gather(-1, lo), bins.gather(-1, hi)
    den = c1 - c0
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    return b0 + (u - c0) / den * (b1 - b0)


def resample_along_rays(origins, directions, radii, t, weights, padding, u_rand=None):
    """Blur-pool + PDF resample, no gradient.  models/mip.py:304-352
    (stop_grad=True branch)."""
    with torch.no_grad():
        wp = torch.cat([weights[..., :1], weights, weights[..., -1:]], -1)
        wmax = torch.maximum(wp[..., :-1], wp[..., 1:])
        blur = 0.5 * (wmax[..., :-1] + wmax[..., 1:])
        new_t = piecewise_constant_pdf(t, blur + padding, t.shape[-1], u_rand)
    return new_t, cast_rays(new_t, origins, directions, radii)


# ----------------------------------------------------------------------- encodings
def integrated_pos_enc(mean, cov, min_deg, max_deg):
    """IPE, diagonal covariance.  models/mip.py:394-428 with expected_sin 355-361.
    Feature index = half*3L + l*3 + c; the 'cos' half is sin(fl32(y + pi/2))."""
    scales = torch.tensor([2.0 ** i for i in range(min_deg, max_deg)])
    y = (mean[..., None, :] * scales[:, None]).flatten(-2)
    v = (cov[..., None, :] * scales[:, None] ** 2).flatten(-2)
    y2 = torch.cat([y, y + HALF_PI_F32], -1)
    v2 = torch.cat([v, v], -1)
    return torch.exp(-0.5 * v2) * torch.sin(y2)


def pos_enc(x, min_deg, max_deg):
    """Plain positional encoding with identity appended in front.  models/mip.py:431-441."""
    scales = torch.tensor([2.0 ** i for i in range(min_deg, max_deg)])
    xb = (x[..., None, :] * scales[:, None]).flatten(-2)
    return torch.cat([x, torch.sin(torch.cat([xb, xb + HALF_PI_F32], -1))], -1)


# ------------------------------------------------------------------------------ MLP
def mlp_param_shapes(num_density_channels=5, width=256, depth=8, skip=4, xyz_dim=96,
                     view_dim=27, width_cond=128, rgb_ch=3):
    """Names/shapes of the state dict.  models/pano_mip_nerf.py:35-76
    (PureMLP models/mip_nerf.py:19-60 is identical apart from the density width)."""
    shapes = collections.OrderedDict()
    for i in range(depth):
        if i == 0:
            k = xyz_dim
        elif (i - 1) % skip == 0 and i > 1:
            k = width + xyz_dim
        else:
            k = width
        shapes[f"layers.{i}.0.weight"] = (width, k)
        shapes[f"layers.{i}.0.bias"] = (width,)
    shapes["density_layer.weight"] = (num_density_channels, width)
    shapes["density_layer.bias"] = (num_density_channels,)
    shapes["extra_layer.weight"] = (width, width)
    shapes["extra_layer.bias"] = (width,)
    shapes["view_layers.0.0.weight"] = (width_cond, width + view_dim)
    shapes["view_layers.0.0.bias"] = (width_cond,)
    shapes["color_layer.weight"] = (rgb_ch, width_cond)
    shapes["color_layer.bias"] = (rgb_ch,)
    return shapes


def init_params(seed, num_density_channels=5):
    """Deterministic weights from numpy PCG64(seed) with the reference's init
    *distributions*: Xavier-uniform weights (pano_mip_nerf.py:10-14), torch-default
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for biases and for color_layer."""
    rng = np.random.Generator(np.random.PCG64(seed))
    params = collections.OrderedDict()
    for name, shape in mlp_param_shapes(num_density_channels).items():
        if name.endswith("weight"):
            fan_out, fan_in = shape
            if name.startswith("color_layer"):
                bound = 1.0 / math.sqrt(fan_in)
            else:
                bound = math.sqrt(6.0 / (fan_in + fan_out))
            last_fan_in = fan_in
        else:
            bound = 1.0 / math.sqrt(last_fan_in)
        a = rng.uniform(-bound, bound, size=shape).astype(np.float32)
        params[name] = torch.from_numpy(a)
    return params


# Test hook (not part of the reference): a queue of ReLU gate decisions, one [9, M, 256] boolean tensor per mlp_forward
# call (h0..h7, view hidden in the first 128 columns).  With gates forced, relu(z) becomes z * gate, so that two
# implementations can be compared on IDENTICAL gate decisions: a ReLU whose pre-activation is ~1e-7 flips under any fp32
# summation order and makes parameter gradients (not values) discontinuous, which no pointwise tolerance survives.
_FORCED_GATES = None


class forced_gates:
    def __init__(self, gates):
        self.gates = list(gates)

    def __enter__(self):
        global _FORCED_GATES
        _FORCED_GATES = list(self.gates)
        return self

    def __exit__(self, *a):
        global _FORCED_GATES
        left, _FORCED_GATES = _FORCED_GATES, None
        if a[0] is None:
            assert not left, f"{len(left)} forced gate sets were not consumed"


def _act(z, gates, slot):
    if gates is None:
        return torch.relu(z)
    g = gates[slot][..., :z.shape[-1]].reshape(z.shape).to(z.dtype)
    return z * g


def mlp_forward(p, enc, viewenc, skip=4, depth=8):
    """models/pano_mip_nerf.py:95-114.  enc [B,N,96], viewenc [B,27]."""
    gates = _FORCED_GATES.pop(0) if _FORCED_GATES else None
    x = enc
    for i in range(depth):
        x = _act(torch.nn.functional.linear(x, p[f"layers.{i}.0.weight"], p[f"layers.{i}.0.bias"]), gates, i)
        if i % skip == 0 and i > 0:
            x = torch.cat([x, enc], -1)
    raw_density = torch.nn.functional.linear(x, p["density_layer.weight"], p["density_layer.bias"])
    bott = torch.nn.functional.linear(x, p["extra_layer.weight"], p["extra_layer.bias"])
    ve = viewenc[:, None, :].expand(-1, enc.shape[1], -1)
    x = _act(torch.nn.functional.linear(torch.cat([bott, ve], -1),
                                        p["view_layers.0.0.weight"], p["view_layers.0.0.bias"]), gates, 8)
    raw_rgb = torch.nn.functional.linear(x, p["color_layer.weight"], p["color_layer.bias"])
    return raw_rgb, raw_density


def radiance_field(p, mean, cov, viewdirs, rgb_padding=0.0, density_bias=-1.0,
                   min_deg=0, max_deg=16, deg_view=4):
    """compute_graph closure, models/pano_mip_nerf.py:235-280 (mip_nerf.py:206-243
    for the 1-channel head).  Returns rgb, sigma[...,1], albedo|None."""
    enc = integrated_pos_enc(mean, cov, min_deg, max_deg)
    venc = pos_enc(viewdirs, 0, deg_view)
    raw_rgb, raw_den = mlp_forward(p, enc, venc)
    sp = torch.nn.functional.softplus
    rgb = sp(raw_rgb) * (1 + 2 * rgb_padding) - rgb_padding
    sigma = sp(raw_den[..., :1] + density_bias)
    albedo = None
    if raw_den.shape[-1] >= 5:
        albedo = torch.sigmoid(raw_den[..., 1:-1]) * 0.77 + 0.03
    return rgb, sigma, albedo


def volumetric_rendering(rgb, sigma, t, dirs, white_bkgd):
    """models/mip.py:444-483."""
    t_mid = 0.5 * (t[..., :-1] + t[..., 1:])
    delta = (t[..., 1:] - t[..., :-1]) * torch.linalg.norm(dirs[..., None, :], dim=-1)
    x = sigma[..., 0] * delta
    alpha = 1 - torch.exp(-x)
    excl = torch.cat([torch.zeros_like(x[..., :1]), torch.cumsum(x[..., :-1], -1)], -1)
    w = alpha * torch.exp(-excl)
    comp = (w[..., None] * rgb).sum(-2)
    acc = w.sum(-1)
    dist = (w * t_mid).sum(-1) / acc
    dist = torch.clamp(torch.nan_to_num(dist), t[:, 0], t[:, -1])
    if white_bkgd:
        comp = comp + (1.0 - acc[..., None])
    return comp, dist, acc, w


def density_normals(p, mean, cov, viewdirs, mode="fast", **kw):
    """-d sigma / d mean per sample.  models/pano_mip_nerf.py:299-304.
    'faithful' keeps the reference's vmap(jacrev) over all outputs (for timing);
    'fast' uses grad of sum(sigma), identical because samples are independent."""
    B, N, _ = mean.shape
    if mode == "faithful":
        from torch.func import jacrev, vmap

        def one(m, c, v):
            r = radiance_field(p, m.view(1, 1, 3), c.view(1, 1, 3), v.view(1, 3), **kw)
            return tuple(x for x in r if x is not None)

        vd = viewdirs.view(-1, 1, 3).repeat(1, N, 1).view(-1, 3)
        jac = vmap(jacrev(one, argnums=0))(mean.reshape(-1, 3), cov.reshape(-1, 3), vd)[1]
        return -jac.reshape(B, N, 3)
    m = mean if mean.requires_grad else mean.detach().requires_grad_(True)
    with torch.enable_grad():
        _, sigma, _ = radiance_field(p, m, cov, viewdirs, **kw)
        (g,) = torch.autograd.grad(sigma.sum(), m, create_graph=True)
    return -g


def surface_rendering(env_rgb, albedo, normal, lit_dir, solid_angle):
    """Lambertian shading.  utils/surface_rendering.py:104-126, 129-165
    (roughness=None branch).  env_rgb, lit_dir [B,D,3]; solid_angle [D,1]."""
    nol = torch.relu((normal[:, None, :] * lit_dir).sum(-1, keepdim=True))
    shading = (env_rgb * nol * solid_angle).sum(1)
    diffuse = albedo / np.pi * shading
    return diffuse, diffuse, shading  # rgb == diffuse (specular is zero)


def hdr_to_ldr(color, gamma=2.2, quantize=False):
    """ACES tone map.  utils/surface_rendering.py:319-344."""
    color = (color * (2.51 * color + 0.03)) / (color * (2.43 * color + 0.59) + 0.14)
    color = torch.clamp(color, 0, 1)
    if quantize:
        color = ((color * 255.0).to(torch.uint8) / 255.0).to(torch.float32)
    return color ** (1 / gamma)


# --------------------------------------------------------------------- full forwards
def pano_forward(p, rays, env_rays, *, num_samples, white_bkgd=False, enable_surf=True,
                 use_ort_loss=True, noise=None, num_env_samples=10, resample_padding=0.01,
                 rgb_padding=0.0, density_bias=-1.0, normals_mode="fast", disparity=False, disable_integration=False):
    """PanoMipNeRF.forward, models/pano_mip_nerf.py:197-363.
    ``noise`` = None (deterministic) or dict(t_rand [B,S], u_rand [B,S], env_rand [1,Ne+1]).
    ``disable_integration``: every encoding sees a zero covariance (models/pano_mip_nerf.py:241-243, inside compute_graph:
    both levels, the density-gradient normals and the env-light evaluation)."""
    kw = dict(rgb_padding=rgb_padding, density_bias=density_bias)
    zc = (lambda c: torch.zeros_like(c)) if disable_integration else (lambda c: c)
    env = Rays(*[x.float() for x in env_rays])
    ret = []
    t, w = None, None
    for level in range(2):
        if level == 0:
            t, (mean, cov) = sample_along_rays(rays.origins, rays.directions, rays.radii, num_samples,
                                               rays.near, rays.far, None if noise is None else noise["t_rand"], disparity)
        else:
            t, (mean, cov) = resample_along_rays(rays.origins, rays.directions, rays.radii, t, w.detach().clone(),
                                                 resample_padding, None if noise is None else noise["u_rand"])
        cov = zc(cov)
        rgb, sigma, albedos = radiance_field(p, mean, cov, rays.viewdirs, **kw)
        comp, dist, acc, w = volumetric_rendering(rgb, sigma, t, rays.directions, white_bkgd)
        normal = surf = albedo = diffuse = ort = shading = None
        if level == 1:
            nw = w[..., None] / w.sum(-1).view(-1, 1, 1)
            normals = torch.nn.functional.normalize(
                density_normals(p, mean, cov, rays.viewdirs, mode=normals_mode, **kw), dim=-1)
            normal = torch.nn.functional.normalize((nw * normals).sum(1), dim=-1)
            if use_ort_loss:
                dot = (normals * rays.directions[:, None, :]).sum(-1, keepdim=True)
                ort = (nw * torch.relu(dot) ** 2).sum(1).mean()
            if enable_surf:
                albedo = (nw * albedos).sum(1)
                x_surf = rays.origins + rays.directions * dist.view(-1, 1)
                lt, (lm, lc), ldirs = sample_each_points(
                    x_surf, env.directions, num_env_samples, env.near, env.far, env.radii,
                    None if noise is None else noise["env_rand"])
                lrgb, lsig, _ = radiance_field(p, lm, zc(lc), ldirs, **kw)
                env_rgb = volumetric_rendering(lrgb, lsig, lt, ldirs, False)[0].view(normal.shape[0], -1, 3)
                surf, diffuse, shading = surface_rendering(env_rgb, albedo, normal, ldirs.view(env_rgb.shape),
                                                           env.lossmult)
        ret.append((comp, dist, ort, normal, albedo, None, surf, diffuse, shading))
    return ret


def mip_forward(p, rays, *, num_samples, white_bkgd=False, use_ort_loss=False, noise=None,
                resample_padding=0.01, rgb_padding=0.0, density_bias=-1.0, normals_mode="fast", disparity=False,
                disable_integration=False):
    """MipNeRF.forward, models/mip_nerf.py:170-283 (disable_integration: :215-216)."""
    kw = dict(rgb_padding=rgb_padding, density_bias=density_bias)
    zc = (lambda c: torch.zeros_like(c)) if disable_integration else (lambda c: c)
    ret = []
    t, w = None, None
    for level in range(2):
        if level == 0:
            t, (mean, cov) = sample_along_rays(rays.origins, rays.directions, rays.radii, num_samples,
                                               rays.near, rays.far, None if noise is None else noise["t_rand"], disparity)
        else:
            t, (mean, cov) = resample_along_rays(rays.origins, rays.directions, rays.radii, t, w.detach().clone(),
                                                 resample_padding, None if noise is None else noise["u_rand"])
        cov = zc(cov)
        rgb, sigma, _ = radiance_field(p, mean, cov, rays.viewdirs, **kw)
        comp, dist, acc, w = volumetric_rendering(rgb, sigma, t, rays.directions, white_bkgd)
        if level == 1 and use_ort_loss:
            nw = w[..., None] / acc.view(-1, 1, 1)
            normals = torch.nn.functional.normalize(
                density_normals(p, mean, cov, rays.viewdirs, mode=normals_mode, **kw), dim=-1)
            normal = torch.nn.functional.normalize((nw * normals).sum(1), dim=-1)
            dot = (normals * rays.directions[:, None, :]).sum(-1, keepdim=True)
            ort = (nw * torch.relu(dot) ** 2).sum(1).mean()
            ret.append((comp, dist, ort, normal))
        else:
            ret.append((comp, dist, None, torch.ones_like(comp)))
    return ret


# ------------------------------------------------------------------------------ loss
DEFAULT_LOSS = dict(coarse_loss_mult=0.1, surface_loss=1.0, ort_loss=0.1, chrom_loss=0.1)


def pano_loss(outputs, lossmult, rgbs, hp=DEFAULT_LOSS, surface=True):
    """PanoNeRFSystem.training_step, systems/panonerf_system.py:15-75."""
    gt = hdr_to_ldr(rgbs[..., :3], quantize=True)
    (rgb_c, *_), (rgb_f, _, ort, _, alb, _, sf, _, _) = outputs
    mse = lambda x: (lossmult * (hdr_to_ldr(x) - gt) ** 2).sum() / lossmult.sum()
    loss = hp["coarse_loss_mult"] * mse(rgb_c) + mse(rgb_f)
    if surface and sf is not None:
        loss = loss + hp["surface_loss"] * mse(sf)
        if hp["chrom_loss"] > 0:
            nz = torch.nn.functional.normalize
            loss = loss + hp["chrom_loss"] * ((nz(gt, dim=-1) - nz(alb, dim=-1)) ** 2).mean()
    if ort is not None:
        loss = loss + hp["ort_loss"] * ort
    return loss


def mip_loss(outputs, lossmult, rgbs, hp=DEFAULT_LOSS, use_ort=False):
    """MipNeRFSystem.training_step, systems/mipnerf_system.py:22-53."""
    gt = hdr_to_ldr(rgbs[..., :3], quantize=True)
    (c, *_), (f, _, ort, _) = outputs
    mse = lambda x: (lossmult * (hdr_to_ldr(x) - gt) ** 2).sum() / lossmult.sum()
    loss = hp["coarse_loss_mult"] * mse(c) + mse(f)
    if use_ort:
        loss = loss + hp["ort_loss"] * ort
    return loss


def mip_lr(step, lr_init=2e-4, lr_final=2e-5, max_steps=44000, delay_steps=120, delay_mult=0.01):
    """utils/lr_schedule.py:51-59."""
    rate = 1.0
    if delay_steps > 0:
        rate = delay_mult + (1 - delay_mult) * np.sin(0.5 * np.pi * np.clip(step / delay_steps, 0, 1))
    t = np.clip(step / max_steps, 0, 1)
    return float(rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t))


def calc_psnr(x, y):
    """utils/metrics.py:231-237."""
    return float(-10.0 * torch.log10(torch.mean((x - y) ** 2)))


# ------------------------------------------------------------------ synthetic inputs
def synthetic_scene(h, w, n_cam=3, seed=4, near=0.0, far=10.0):
    """identity-rotation cameras at U(-.5,.5)^3, analytic HDR radiance."""
    rng = np.random.Generator(np.random.PCG64(seed))
    c2ws = []
    for _ in range(n_cam):
        m = np.eye(4, dtype=np.float32)
        m[:3, 3] = rng.uniform(-0.5, 0.5, 3).astype(np.float32)
        c2ws.append(m)
    rays, radius = generate_pano_rays(h, w, c2ws, near, far)
    flat = Rays(*[torch.from_numpy(np.concatenate([a.reshape(-1, a.shape[-1]) for a in getattr(rays, k)], 0)
                                   .astype(np.float32)) for k in Rays._fields])
    d, o = flat.viewdirs, flat.origins
    f = torch.stack([1.5 + torch.sin(3 * d[:, 0] + o[:, 0]) + torch.cos(2 * d[:, 1]),
                     1.0 + torch.sin(2 * d[:, 1] + o[:, 1]) * torch.cos(d[:, 2]),
                     0.5 + torch.cos(4 * d[:, 2] + o[:, 2]) + d[:, 1]], -1)
    rgbs = torch.clamp(torch.nn.functional.softplus(f), 0, 1000).float()
    return flat, rgbs, radius, c2ws
