"""Golden vectors for the validation metrics by IMPORTING the reference's utils/metrics.py.

Build container only (needs /root/reference); stores seeded inputs and the reference's outputs, no reference code.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_metrics_golden.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
for name in ("cv2", "Imath"):
    sys.modules[name] = types.ModuleType(name)
_exr = types.ModuleType("OpenEXR")
_exr.InputFile = _exr.OutputFile = _exr.Header = object
sys.modules["OpenEXR"] = _exr

import numpy as np  # noqa: E402
import torch  # noqa: E402

import utils.metrics as rm  # noqa: E402
from utils.surface_rendering import solid_angle_refinement  # noqa: E402

rng = np.random.Generator(np.random.PCG64(11))
out = {}
for h, w in ((8, 16), (16, 32)):
    pred = torch.tensor(rng.random((3, h, w), dtype=np.float32))
    gt = torch.tensor(rng.random((3, h, w), dtype=np.float32))
    n1 = torch.tensor(rng.standard_normal((1, h, w, 3)).astype(np.float32))
    n2 = n1 + 0.3 * torch.tensor(rng.standard_normal((1, h, w, 3)).astype(np.float32))
    k = f"{h}x{w}/"
    out[k + "pred"], out[k + "gt"], out[k + "n1"], out[k + "n2"] = pred, gt, n1, n2
    out[k + "solid_angle"] = solid_angle_refinement(h=h, w=w)
    out[k + "mse"] = rm.calc_mse(pred, gt)
    out[k + "psnr"] = rm.calc_psnr(pred, gt)
    out[k + "l1"] = rm.calc_l1(pred, gt)
    out[k + "ws_psnr"] = rm.calc_ws_psnr(pred, gt)
    out[k + "ws_l1"] = rm.calc_ws_l1(pred, gt)
    out[k + "ws_mse"] = rm.calc_ws_mse(pred, gt)
    out[k + "ws_rmse"] = rm.calc_ws_rmse(pred, gt)
    out[k + "ws_mae"] = rm.calc_ws_mae(n1, n2, dim=-1)
    out[k + "mae"] = rm.calc_mae(n1, n2, dim=-1)
    out[k + "ws_cossimi"] = rm.calc_ws_cossimi(n1[0].permute(2, 0, 1), n2[0].permute(2, 0, 1), dim=0)
np.savez_compressed(os.path.join(HERE, "metrics.npz"), **{k: np.asarray(v) for k, v in out.items()})
print({k: np.asarray(v).shape for k, v in out.items()})
