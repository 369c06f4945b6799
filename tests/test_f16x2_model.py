"""CPU: the arithmetic of mlp_mode = "fused_f16x2" restated in numpy (tools/check_f16x2.py follows pn_chain.hip's scale_exp /
split_into / mfma_split / chain_gemm): a scaled fp16 pair represents an fp32 operand to fp32's own roundoff, and a 256-term
dot product from three partial products is as accurate as the bf16 three-term split (six products) at every magnitude."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import check_f16x2 as m  # noqa: E402


def test_pair_represents_fp32_to_its_roundoff():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(4096) * np.exp(rng.standard_normal(4096) * 2)).astype(np.float32)
    e = int(m.scale_exp(np.abs(x).max()))
    h, l = m.split_f16(x, e)
    back = np.ldexp(h.astype(np.float64) + l.astype(np.float64), -e)
    big = np.abs(x) >= np.abs(x).max() * 2.0 ** -16  # elements whose low half stays a normal fp16
    assert np.all(np.abs(back - x)[big] <= np.abs(x)[big] * 2.0 ** -23)
    assert np.all(np.abs(back - x)[~big] <= np.abs(x).max() * 2.0 ** -38)  # the rest: tiny against the column's maximum
    assert float(np.abs(np.ldexp(h.astype(np.float64), -e)).max()) <= float(np.abs(x).max()) * (1 + 2.0 ** -10)  # no overflow


def test_exponent_rules():
    assert int(m.scale_exp(np.float32(0.0))) == 15  # frexp exponent of 0 is 0: zeros scale harmlessly
    assert int(m.scale_exp(np.float32(1.0))) == 14 and int(m.scale_exp(np.float32(3.0e4))) == 0
    assert int(m.scale_exp(np.float32(1e-38))) == m.EXP_CAP and int(m.scale_exp(np.float32(1e-38), m.EXP_CAP_Z)) == m.EXP_CAP_Z
    assert int(m.scale_exp(np.float32(1e30))) < -80  # large values are never capped
    for v in (1e-30, 1e-3, 1.0, 6.0e4, 1e20):
        e = int(m.scale_exp(np.float32(v), m.EXP_CAP_Z))
        assert 2.0 ** 14 <= v * 2.0 ** e < 2.0 ** 15


def test_dot_products_match_fp32_accuracy_at_every_magnitude():
    for seed in (0, 1):
        errs = m.errors(seed=seed)
        for mag, e in errs.items():
            # three fp16 products: as good as six bf16 products, within a small factor of an fp32 fma chain
            assert e["f16x2"] <= 2.0 * e["bf16x3"] + 1e-6, (mag, e)
            assert e["f16x2"] <= 4.0 * e["f32"] + 1e-6, (mag, e)
