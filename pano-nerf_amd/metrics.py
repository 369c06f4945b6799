"""Validation metrics of a rendered panorama (host-side torch ops on whatever device the image lives on).

Mirrors the functions `systems/panonerf_system.py:validation_step` calls from `utils/metrics.py`:
`calc_mse/rmse/l1/psnr` (:210-237), `calc_mae` / `calc_cossimi` (:240-257) and the solid-angle weighted family
`calc_ws_psnr/l1/mse/rmse/mae/cossimi` (:318-397) with `solid_angle_refinement` (`utils/surface_rendering.py:294-316`).
An equirectangular pixel in row i covers sin(phi_i) dtheta dphi steradians, so the weights are sin((i + .5) pi / H),
normalised to sum 1.  Not part of the training hot path; SURVEY.md 8f rank 4.
"""
import math

import torch
import torch.nn.functional as F


def solid_angle_refinement(h=8, w=16, hemisp=False, device=None):
    """[1, h*w, 1] steradians per pixel of an h x w equirectangular image (upper hemisphere only if `hemisp`)."""
    phi_range = math.pi / 2 if hemisp else math.pi
    rows = (torch.arange(h, dtype=torch.float64) + 0.5) / h
    sa = torch.sin(rows * phi_range) * (2 * math.pi / w) * (phi_range / h)
    return sa.reshape(h, 1).expand(h, w).reshape(1, -1, 1).to(torch.float32).to(device)


def _weights(h, w, device):
    wt = solid_angle_refinement(h, w, device=device).reshape(1, h, w)
    return wt / wt.sum()


def calc_mse(x, y):
    return torch.mean((x - y) ** 2)


def calc_rmse(x, y):
    return torch.mean((x - y) ** 2) ** 0.5


def calc_l1(x, y):
    return torch.abs(x - y).mean()


def calc_psnr(x, y):
    return -10.0 * torch.log10(calc_mse(x, y))


def _angles(x, y, dim):
    if dim == 1:
        x, y = x.permute(0, 2, 3, 1), y.permute(0, 2, 3, 1)
    cos = F.cosine_similarity(x.reshape(-1, 3), y.reshape(-1, 3), dim=-1)
    return torch.nan_to_num(torch.acos(cos) / math.pi * 180, nan=0.0), x


def calc_mae(x, y, dim=-1):
    """Mean angular error in degrees between two [B,H,W,3] (dim=-1) or [B,3,H,W] (dim=1) vector images."""
    return _angles(x, y, dim)[0].mean()


def calc_cossimi(x, y, dim=-1):
    return F.cosine_similarity(x, y, dim=dim).mean()


def calc_ws_mse(pred, gt):
    """Solid-angle weighted squared error of [C,H,W] images (summed over channels, as upstream)."""
    _, h, w = pred.shape
    return torch.sum((pred - gt) ** 2 * _weights(h, w, pred.device))


def calc_ws_rmse(pred, gt):
    return torch.sqrt(calc_ws_mse(pred, gt))


def calc_ws_psnr(pred, gt):
    return -10.0 * torch.log10(calc_ws_mse(pred, gt))


def calc_ws_l1(pred, gt):
    _, h, w = pred.shape
    return torch.sum(torch.abs(pred - gt) * _weights(h, w, pred.device))


def calc_ws_mae(x, y, dim=-1, weights=None):
    ang, xp = _angles(x, y, dim)
    if weights is None:
        _, h, w, _ = xp.shape
        weights = solid_angle_refinement(h, w, device=x.device)
    weights = weights.reshape(-1).to(x.device)
    return torch.sum(ang * (weights / weights.sum()))


def calc_ws_cossimi(x, y, dim=0):
    if dim == 0:
        _, h, w = x.shape
    elif dim == -1:
        h, w, _ = x.shape
    elif dim == 1:
        _, _, h, w = x.shape
    else:
        raise ValueError("dim must be 0, 1 or -1")
    cos = F.cosine_similarity(x, y, dim=dim).reshape(1, h, w)
    return torch.sum(cos * _weights(h, w, x.device))
