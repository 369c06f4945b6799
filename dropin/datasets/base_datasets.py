"""Import-path shim: the batch container (datasets/base_datasets.py:13-16)."""
from pano_nerf_amd.rays import Rays, Rays_keys, namedtuple_map  # noqa: F401
