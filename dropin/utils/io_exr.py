"""Import-path shim: `from utils.io_exr import write_exr, read_exr` resolves to the codec-free writers."""
from pano_nerf_amd.io_exr import read_exr, write_exr, write_png  # noqa: F401
