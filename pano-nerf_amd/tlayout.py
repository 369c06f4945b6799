"""Host-side views of the sample-minor ("T layout") tensors the fused chains write and pn_chain_wgrad reads - for tests and
tools; the training path never touches them on the host.

fp32 T layout: elem[Mp / tile][F][tile].  Q24 (include/panonerf_hip.h, pn_chain_q24_slots; fp16-pair mode, 16-sample tiles): fp32
rounded to 16 significant bits, three bytes per element, the four features of a quad of a sample together:
byte[Mp / 16][F / 4][16][4][3], byte b of an element = bits 8 (b + 1) .. 8 (b + 1) + 7 of the rounded fp32."""
import torch


def t_encode(rows, tile):
    """[Mp, F] -> flat fp32 T layout."""
    Mp, F = rows.shape
    return rows.reshape(Mp // tile, tile, F).permute(0, 2, 1).contiguous().reshape(-1)


def t_decode(flat, Mp, F, tile):
    return flat[:Mp * F].reshape(Mp // tile, F, tile).permute(0, 2, 1).reshape(Mp, F)


def q24_encode(rows):
    """[Mp, F] fp32 -> uint8 [Mp * F * 3] in the Q24 layout (round to nearest on the dropped byte, ties away from zero)."""
    Mp, F = rows.shape
    bits = rows.contiguous().float().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    bits = (bits + 0x80) & 0xFFFFFFFF
    b = torch.stack([(bits >> 8) & 0xFF, (bits >> 16) & 0xFF, (bits >> 24) & 0xFF], dim=-1).to(torch.uint8)  # [Mp, F, 3]
    return b.reshape(Mp // 16, 16, F // 4, 4, 3).permute(0, 2, 1, 3, 4).contiguous().reshape(-1)


def q24_decode(buf, Mp, F):
    """uint8 Q24 bytes (at least Mp * F * 3) -> [Mp, F] fp32."""
    b = buf[:Mp * F * 3].reshape(Mp // 16, F // 4, 16, 4, 3).permute(0, 2, 1, 3, 4).reshape(Mp, F, 3).to(torch.int64)
    bits = (b[..., 0] << 8) | (b[..., 1] << 16) | (b[..., 2] << 24)
    bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits).to(torch.int32)
    return bits.view(torch.float32)


def q24_round(rows):
    """The values a Q24 tensor holds for `rows`."""
    return q24_decode(q24_encode(rows), rows.shape[0], rows.shape[1])


def slot_rows(lib, slot_floats, Mp, F, q24):
    """Rows [Mp, F] of one tensor slot (a float view of its Mp * F floats), whichever way it is stored."""
    if q24:
        return q24_decode(slot_floats.reshape(-1).view(torch.uint8), Mp, F)
    return t_decode(slot_floats.reshape(-1), Mp, F, int(lib.pn_chain_tile()))


def write_slot(lib, slot_floats, rows, q24):
    """Store `rows` [Mp, F] into a tensor slot (float view of at least Mp * F floats) as the chains would."""
    flat = slot_floats.reshape(-1)
    if q24:
        enc = q24_encode(rows)
        flat.view(torch.uint8)[:enc.numel()] = enc
    else:
        enc = t_encode(rows, int(lib.pn_chain_tile()))
        flat[:enc.numel()] = enc
