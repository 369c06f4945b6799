"""Result writers for validation dumps: float32 OpenEXR and 8-bit PNG, with no third-party codec.

`utils/io_exr.py:30-47` writes an RGB float32 scanline OpenEXR through the OpenEXR python binding and
`utils/io_exr.py:6-27` reads one back; the binding is not available here, so this module emits / parses the
container itself (OpenEXR 2 single-part scanline file, channels B G R as FLOAT, NO_COMPRESSION, increasing Y) —
the layout any OpenEXR reader accepts.  Parity with files written by the real library is unpinned (the library is
absent on both boxes): the tests check the header fields and a write -> read round trip.  SURVEY.md 8f rank 4.
"""
import struct
import zlib

import numpy as np

_MAGIC = 20000630


def _attr(name, typ, payload):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload


def write_exr(filename, data):
    """data: float32 [H, W, 3] or [H, W, 1] (grey is replicated to R, G, B like upstream)."""
    assert filename.endswith(".exr"), "extension must be .exr"
    data = np.asarray(data)
    assert data.dtype == np.float32, f"data type is {data.dtype}, should be float32"
    h, w, c = data.shape
    if c == 1:
        data = np.repeat(data, 3, axis=2)
    chlist = b"".join(n + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1) for n in (b"B", b"G", b"R")) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    header = (_attr("channels", "chlist", chlist) + _attr("compression", "compression", b"\0") +
              _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box) +
              _attr("lineOrder", "lineOrder", b"\0") + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) +
              _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0)) +
              _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    head = struct.pack("<ii", _MAGIC, 2) + header
    line_bytes = 3 * w * 4
    table_at = len(head)
    first = table_at + 8 * h
    offsets = struct.pack("<%dQ" % h, *[first + y * (8 + line_bytes) for y in range(h)])
    bgr = np.ascontiguousarray(data[:, :, ::-1].transpose(0, 2, 1))  # [H][B,G,R][W]
    with open(filename, "wb") as f:
        f.write(head)
        f.write(offsets)
        for y in range(h):
            f.write(struct.pack("<ii", y, line_bytes))
            f.write(bgr[y].tobytes())


def read_exr(filename, channel=3):
    """Reads back an uncompressed FLOAT scanline file (what write_exr produces) -> float32 [H, W, channel]."""
    buf = open(filename, "rb").read()
    magic, version = struct.unpack_from("<ii", buf, 0)
    if magic != _MAGIC or (version & 0xff) != 2 or (version & 0x1e00):
        raise ValueError("not a single-part scanline OpenEXR file")
    pos, attrs = 8, {}
    while buf[pos] != 0:
        end = buf.index(b"\0", pos)
        name = buf[pos:end].decode()
        pos = end + 1
        end = buf.index(b"\0", pos)
        pos = end + 1
        (size,) = struct.unpack_from("<i", buf, pos)
        attrs[name] = buf[pos + 4:pos + 4 + size]
        pos += 4 + size
    pos += 1
    if attrs["compression"] != b"\0":
        raise NotImplementedError("only NO_COMPRESSION files are read here")
    names, p, ch = [], 0, attrs["channels"]
    while ch[p] != 0:
        end = ch.index(b"\0", p)
        names.append(ch[p:end].decode())
        (ptype,) = struct.unpack_from("<i", ch, end + 1)
        if ptype != 2:
            raise NotImplementedError("only FLOAT channels are read here")
        p = end + 1 + 16
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    offsets = struct.unpack_from("<%dQ" % h, buf, pos)
    planes = np.empty((h, len(names), w), np.float32)
    for off in offsets:
        y, nbytes = struct.unpack_from("<ii", buf, off)
        planes[y - y0] = np.frombuffer(buf, np.float32, len(names) * w, off + 8).reshape(len(names), w)
    want = "RGB" if channel == 3 else "A"
    return np.stack([planes[:, names.index(c), :] for c in want], axis=2)


def write_png(filename, img):
    """uint8 [H, W, 3] or [H, W] -> PNG (zlib only); float images in [0, 1] are scaled and TRUNCATED like
    `hdr_to_ldr(dtype='uint8')` (`utils/surface_rendering.py:319-344`)."""
    img = np.asarray(img)
    if img.dtype != np.uint8:
        img = (np.clip(img, 0.0, 1.0) * 255).astype(np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, c = img.shape
    ctype = {1: 0, 3: 2, 4: 6}[c]
    raw = b"".join(b"\0" + img[y].tobytes() for y in range(h))

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)

    with open(filename, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
