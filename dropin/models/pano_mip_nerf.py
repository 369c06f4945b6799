"""Import-path shim: `from models.pano_mip_nerf import PanoMipNeRF` (systems/base_system.py:23) resolves to the
MI355X implementation when `dropin/` precedes the reference on sys.path."""
from pano_nerf_amd.render import PanoMipNeRF  # noqa: F401
from pano_nerf_amd.mlp import RadianceMLP as MLP  # noqa: F401
