"""Zero-edit integration with Lu-Zhan/Pano-NeRF.

`install()` registers the MI355X classes under the two module names the reference's systems import them from
(`from models.pano_mip_nerf import PanoMipNeRF`, `from models.mip_nerf import MipNeRF`: systems/base_system.py:19-24)
WITHOUT shadowing anything else of the reference: `models`, `utils`, `datasets` stay the reference's own packages
(`models.loss`, `models.mip.rearrange_render_image`, `utils.surface_rendering.hdr_to_ldr`, `utils.lr_schedule`,
`utils.vis`, `utils.io_exr`, `datasets.pano_datasets` keep resolving to the reference).

    python -m pano_nerf_amd.run train.py --config configs/panonerf.yaml ...      # from the reference's directory

or, inside any entry script, before the systems are imported:

    import pano_nerf_amd; pano_nerf_amd.install()
"""
import importlib
import sys
import types

_NAMES = {"models.pano_mip_nerf": ("PanoMipNeRF", "MLP"), "models.mip_nerf": ("MipNeRF", "PureMLP")}


def install():
    """Idempotent.  Requires the reference's `models` package to be importable (its directory on sys.path)."""
    from . import render, mlp
    try:
        pkg = importlib.import_module("models")
    except ImportError as e:  # not raised for a missing cv2 etc.: `models/__init__` of the reference is empty
        raise ImportError("pano_nerf_amd.install(): the reference's `models` package is not importable; put the "
                          "Pano-NeRF checkout on sys.path first") from e
    table = {"PanoMipNeRF": render.PanoMipNeRF, "MipNeRF": render.MipNeRF, "MLP": mlp.RadianceMLP,
             "PureMLP": mlp.RadianceMLP}
    for name, exports in _NAMES.items():
        mod = sys.modules.get(name)
        if mod is None or getattr(mod, "__pano_nerf_amd__", False) is False:
            mod = types.ModuleType(name, f"pano_nerf_amd drop-in for the reference's {name}")
            mod.__pano_nerf_amd__ = True
            mod.__package__ = "models"
        for k in exports:
            setattr(mod, k, table[k])
        sys.modules[name] = mod
        setattr(pkg, name.split(".")[1], mod)
    return pkg


def uninstall():
    for name in _NAMES:
        mod = sys.modules.get(name)
        if mod is not None and getattr(mod, "__pano_nerf_amd__", False):
            del sys.modules[name]
            pkg = sys.modules.get("models")
            if pkg is not None and getattr(pkg, name.split(".")[1], None) is mod:
                delattr(pkg, name.split(".")[1])
