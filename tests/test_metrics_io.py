"""Validation metrics against values captured from the reference (tests/golden/metrics.npz, made by
tests/golden/make_metrics_golden.py) and the EXR / PNG writers (round trip; parity with the OpenEXR library itself is
unpinned: it is not installed on either box)."""
import struct
import zlib

import numpy as np
import torch

from pano_nerf_amd import io_exr, metrics


def test_metrics_match_reference(golden):
    g = golden("metrics")
    for h, w in ((8, 16), (16, 32)):
        k = f"{h}x{w}/"
        pred, gt = torch.tensor(g[k + "pred"]), torch.tensor(g[k + "gt"])
        n1, n2 = torch.tensor(g[k + "n1"]), torch.tensor(g[k + "n2"])
        assert np.allclose(metrics.solid_angle_refinement(h, w).numpy(), g[k + "solid_angle"], rtol=1e-6, atol=0)
        got = {
            "mse": metrics.calc_mse(pred, gt), "psnr": metrics.calc_psnr(pred, gt), "l1": metrics.calc_l1(pred, gt),
            "ws_psnr": metrics.calc_ws_psnr(pred, gt), "ws_l1": metrics.calc_ws_l1(pred, gt),
            "ws_mse": metrics.calc_ws_mse(pred, gt), "ws_rmse": metrics.calc_ws_rmse(pred, gt),
            "ws_mae": metrics.calc_ws_mae(n1, n2, dim=-1), "mae": metrics.calc_mae(n1, n2, dim=-1),
            "ws_cossimi": metrics.calc_ws_cossimi(n1[0].permute(2, 0, 1), n2[0].permute(2, 0, 1), dim=0),
        }
        for name, v in got.items():
            assert abs(float(v) - float(g[k + name])) <= 1e-5 * max(1.0, abs(float(g[k + name]))), (k, name)
    # channel-first layout of calc_ws_mae / calc_mae
    a = metrics.calc_ws_mae(n1.permute(0, 3, 1, 2), n2.permute(0, 3, 1, 2), dim=1)
    assert abs(float(a) - float(g["16x32/ws_mae"])) < 1e-4


def test_exr_round_trip_and_header(tmp_path):
    rng = np.random.default_rng(3)
    img = (rng.random((6, 9, 3), dtype=np.float32) * 1000).astype(np.float32)
    img[0, 0] = [0.0, np.float32(1e-30), np.float32(65504.0 * 4)]
    path = str(tmp_path / "a.exr")
    io_exr.write_exr(path, img)
    raw = open(path, "rb").read()
    assert struct.unpack_from("<ii", raw, 0) == (20000630, 2)
    assert b"channels\0chlist\0" in raw and b"compression\0compression\0" in raw and b"dataWindow\0box2i\0" in raw
    assert len(raw) == raw.index(b"screenWindowWidth") + len(b"screenWindowWidth\0float\0") + 4 + 4 + 1 + 8 * 6 + 6 * (8 + 9 * 12)
    back = io_exr.read_exr(path)
    assert back.dtype == np.float32 and np.array_equal(back, img)
    grey = rng.random((4, 5, 1), dtype=np.float32)
    io_exr.write_exr(str(tmp_path / "g.exr"), grey)
    assert np.array_equal(io_exr.read_exr(str(tmp_path / "g.exr")), np.repeat(grey, 3, axis=2))


def test_png_writer(tmp_path):
    img = np.linspace(0, 1, 4 * 5 * 3, dtype=np.float32).reshape(4, 5, 3)
    path = str(tmp_path / "a.png")
    io_exr.write_png(path, img)
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n" and struct.unpack(">II", raw[16:24]) == (5, 4)
    at = raw.index(b"IDAT")
    (n,) = struct.unpack(">I", raw[at - 4:at])
    rows = zlib.decompress(raw[at + 4:at + 4 + n])
    got = np.frombuffer(rows, np.uint8).reshape(4, 1 + 15)[:, 1:].reshape(4, 5, 3)
    assert np.array_equal(got, (img * 255).astype(np.uint8))  # truncation, like hdr_to_ldr(dtype='uint8')
