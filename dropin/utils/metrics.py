"""Import-path shim: `from utils.metrics import calc_psnr, calc_ws_psnr, ...` resolves to the MI355X build."""
from pano_nerf_amd.metrics import *  # noqa: F401,F403
