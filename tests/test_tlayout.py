"""Host-side views of the chains' sample tensors (pano_nerf_amd/tlayout.py) against the formats include/panonerf_hip.h documents:
the fp32 T layout elem[Mp / tile][F][tile] and Q24 byte[Mp / 16][F / 4][16][4][3] (fp32 rounded to 16 significant bits)."""
import numpy as np
import torch

from pano_nerf_amd import tlayout as tl


def test_t_layout_round_trip_and_positions():
    Mp, F, tile = 64, 8, 16
    rows = torch.arange(Mp * F, dtype=torch.float32).reshape(Mp, F)
    flat = tl.t_encode(rows, tile)
    assert torch.equal(tl.t_decode(flat, Mp, F, tile), rows)
    s, f = 37, 5  # sample 37 = block 2, lane 5: element [2][f][5]
    assert float(flat[(2 * F + f) * tile + 5]) == float(rows[s, f])


def test_q24_bytes_positions_and_rounding():
    gen = torch.Generator().manual_seed(3)
    Mp, F = 32, 8
    rows = torch.randn(Mp, F, generator=gen) * torch.exp(8 * torch.randn(Mp, 1, generator=gen))
    rows[0, 0], rows[1, 1], rows[2, 2] = 0.0, -0.0, 1.0 + 2.0 ** -16  # a tie on the dropped byte: away from zero
    buf = tl.q24_encode(rows)
    assert buf.dtype == torch.uint8 and buf.numel() == Mp * F * 3
    got = tl.q24_decode(buf, Mp, F)
    # value: the fp32 bit pattern + 0x80, low byte dropped
    bits = rows.view(torch.int32).numpy().astype(np.int64) & 0xFFFFFFFF
    want_bits = ((bits + 0x80) & 0xFFFFFFFF) & 0xFFFFFF00
    assert np.array_equal(got.view(torch.int32).numpy().astype(np.int64) & 0xFFFFFFFF, want_bits)
    assert float(got[2, 2]) == 1.0 + 2.0 ** -15
    rel = ((got - rows).abs() / rows.abs().clamp_min(1e-30)).max()
    assert float(rel) <= 2.0 ** -16
    assert torch.equal(tl.q24_round(got), got)  # a fixed point
    # position: sample s, feature f -> block s // 16, quad f // 4, lane s % 16, element f % 4, bytes 1..3 of the rounded pattern
    s, f = 21, 6
    off = ((((s // 16) * (F // 4) + f // 4) * 16 + s % 16) * 4 + f % 4) * 3
    b = int(want_bits[s, f])
    assert [int(x) for x in buf[off:off + 3]] == [(b >> 8) & 0xFF, (b >> 16) & 0xFF, (b >> 24) & 0xFF]
