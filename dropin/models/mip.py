"""Import-path shim for the one helper the systems import from models.mip (systems/panonerf_system.py:7)."""
from pano_nerf_amd.rays import rearrange_render_image, Rays, Rays_keys  # noqa: F401
