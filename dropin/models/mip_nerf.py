"""Import-path shim: `from models.mip_nerf import MipNeRF` (systems/base_system.py:20)."""
from pano_nerf_amd.render import MipNeRF  # noqa: F401
from pano_nerf_amd.mlp import RadianceMLP as PureMLP  # noqa: F401
