"""ctypes binding of libpanonerf_hip.so (the C ABI declared in include/panonerf_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent this
module raises at import of the first entry point, and every non-zero status is raised as
RuntimeError.  Build with ``python __graft_entry__.py`` or ``pano-nerf_amd/csrc/build.sh``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpanonerf_hip.so")

_P, _I, _L, _F, _D = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double
_CODES = {"p": _P, "i": _I, "l": _L, "f": _F, "d": _D}

# name -> (restype code, argument codes); order = include/panonerf_hip.h
SIGNATURES = {
    "pn_strerror": ("s", "i"),
    "pn_abi_version": ("i", ""),
    "pn_pad_rows": ("l", "l"),
    "pn_param_layout": ("l", "ip"),
    "pn_wpack_floats": ("l", "i"),
    "pn_pack_weights": ("i", "pipp"),
    "pn_raygen_pano": ("i", "iipff" + "p" * 8 + "p"),
    "pn_lit_rays": ("i", "idddpp"),
    "pn_gather_rays": ("i", "llpppp"),
    "pn_sample_pano_rays": ("i", "liiippffp" + "p" * 9 + "p"),
    "pn_sample_coarse": ("i", "lii" + "p" * 9 + "p"),
    "pn_resample": ("i", "lippfp" + "p" * 6 + "p"),
    "pn_sample_env": ("i", "lii" + "p" * 11 + "p"),
    "pn_ipe_encode": ("i", "lpppp"),
    "pn_pos_enc_view": ("i", "lppp"),
    "pn_mlp_forward": ("i", "lili" + "p" * 12 + "p"),
    "pn_density_grad": ("i", "lif" + "p" * 10 + "p"),
    "pn_mlp_backward_work_floats": ("l", "lill"),
    "pn_mlp_backward": ("i", "lilif" + "p" * 16 + "lii" + "p" * 6 + "pp"),
    "pn_composite_forward": ("i", "liiffi" + "pppp" + "l" + "pppp" + "p"),
    "pn_composite_backward": ("i", "liiffi" + "pppp" + "l" + "ppppp" + "p"),
    "pn_surf_gather_forward": ("i", "lii" + "p" * 7 + "p"),
    "pn_surf_gather_backward": ("i", "lii" + "p" * 10 + "p"),
    "pn_surface_forward": ("i", "li" + "p" * 7 + "p"),
    "pn_surface_backward": ("i", "li" + "p" * 10 + "p"),
    "pn_env_origin_backward": ("i", "lipppp"),
    "pn_tonemap_loss": ("i", "l" + "p" * 6 + "fff" + "p" * 6 + "p"),
    "pn_adam_step": ("i", "lppppffffifp"),
    "pn_adam_step_dev": ("i", "lpppppfffpfp"),
    "pn_gemm_nt": ("i", "liipipipippiip"),
    "pn_gemm_tn_work_floats": ("l", "lii"),
    "pn_gemm_tn": ("i", "liipipipiipp"),
    "pn_chain_tile": ("i", ""),
    "pn_chain_pack_bytes": ("l", "i"),
    "pn_chain_pack": ("i", "piipp"),
    "pn_chain_acts_floats": ("l", "l"),
    "pn_chain_amax_slots": ("i", ""),
    "pn_chain_forward": ("i", "lilii" + "p" * 11 + "iip"),
    "pn_chain_density_grad": ("i", "liif" + "p" * 7 + "ippiip"),
    "pn_chain_tangent": ("i", "lii" + "p" * 10 + "iip"),
    "pn_chain_backward": ("i", "liif" + "p" * 15 + "iip"),
    "pn_chain_wgrad_work_floats": ("l", ""),
    "pn_chain_q24_slots": ("i", "iii"),
    "pn_chain_wgrad": ("i", "ipiippliiip"),
    "pn_mfma_probe": ("i", "piip"),
    "pn_prof_enable": ("i", "i"),
    "pn_prof_read": ("i", "ippp"),
}

_lib = None


def load():
    """Load the shared library (once) and attach the prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is not built (run `python __graft_entry__.py` "
            "or pano-nerf_amd/csrc/build.sh).  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = ctypes.c_char_p if res == "s" else _CODES[res]
        fn.argtypes = [_CODES[c] for c in args]
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().pn_strerror(int(code)).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {code})")


def call(name, *args):
    """Call an int-returning entry point and raise on a non-zero status."""
    check(getattr(load(), name)(*args), name)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
