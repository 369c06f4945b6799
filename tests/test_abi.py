"""CPU-side checks of the C ABI: the library loads, exports every symbol the header declares, and
the ctypes prototype table agrees with the header (argument count and scalar/pointer kinds).
No compute entry point is called (no GPU here)."""
import ctypes
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "panonerf_hip.h")


def parse_header():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"\b(const char\*|int64_t|int)\s+(pn_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        kinds = ""
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    kinds += "p"
                elif a.startswith("int64_t"):
                    kinds += "l"
                elif a.startswith("int "):
                    kinds += "i"
                elif a.startswith("float "):
                    kinds += "f"
                elif a.startswith("double "):
                    kinds += "d"
                else:
                    raise AssertionError(f"unparsed argument {a!r} of {name}")
        protos[name] = ({"const char*": "s", "int64_t": "l", "int": "i"}[ret], kinds)
    return protos


def test_header_matches_ctypes_table():
    lib = importlib.import_module("pano_nerf_amd._lib")
    protos = parse_header()
    assert len(protos) >= 25
    assert set(protos) == set(lib.SIGNATURES), set(protos) ^ set(lib.SIGNATURES)
    for name, sig in protos.items():
        assert lib.SIGNATURES[name] == sig, (name, lib.SIGNATURES[name], sig)


def test_library_exports_every_symbol():
    lib = importlib.import_module("pano_nerf_amd._lib")
    if not os.path.exists(lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    handle = lib.load()
    for name in parse_header():
        assert hasattr(handle, name), name
    assert handle.pn_abi_version() == 2
    assert handle.pn_pad_rows(1) == 128 and handle.pn_pad_rows(128) == 128 and handle.pn_pad_rows(129) == 256
    off = (ctypes.c_int64 * 24)()
    assert handle.pn_param_layout(5, off) == 613768
    assert handle.pn_param_layout(1, off) == 612740
    assert handle.pn_param_layout(3, off) < 0
    assert handle.pn_strerror(-2).decode().startswith("unsupported")


def test_split_k_workspace_is_monotone_in_rows():
    """A batched weight-gradient GEMM runs over FEWER rows than its workspace was sized for (the segments of one
    layer vary); the split-K slab count must therefore never shrink as rows grow (host arithmetic only, no launch)."""
    handle = importlib.import_module("pano_nerf_amd._lib").load()
    for n1, n2 in ((256, 256), (256, 96), (128, 256), (128, 32)):
        prev = 0
        for rows in list(range(1, 4000, 37)) + list(range(4000, 700000, 4099)):
            cur = handle.pn_gemm_tn_work_floats(rows, n1, n2)
            assert cur >= prev and cur % (n1 * n2) == 0, (rows, n1, n2)
            prev = cur
    # the workspace of one backward call covers its batched row count and every smaller one
    assert handle.pn_mlp_backward_work_floats(8481, 33, 257, 51143) >= handle.pn_mlp_backward_work_floats(8481, 33, 257, 42662) > 0
    assert handle.pn_mlp_backward_work_floats(8481, 32, 257, 0) < 0  # not a whole number of rays


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    lib = importlib.import_module("pano_nerf_amd._lib")
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lib.load()
