"""pano_nerf_amd — MI355X-native Pano-NeRF volumetric-rendering hot path.

Public surface (mirrors the reference's names):
    PanoMipNeRF, MipNeRF          models/pano_mip_nerf.py:117, models/mip_nerf.py:105
    Rays, rearrange_render_image  datasets/base_datasets.py:13-16, models/mip.py:530-547
    generate_pano_rays, generate_lit_rays   datasets/pano_datasets.py:152-263
    pano_loss, mip_loss           systems/panonerf_system.py:15-75, systems/mipnerf_system.py:22-53
    FlatAdam, mip_lr              systems/base_system.py:82-87, utils/lr_schedule.py:51-59
    render_image                  systems/panonerf_system.py:133-192
    metrics, io_exr               utils/metrics.py:210-397 (calc_* / calc_ws_*), utils/io_exr.py:6-47
    concurrent_step               one training step as concurrent sub-batches on separate HIP streams
    install                       register PanoMipNeRF / MipNeRF under the reference's import paths (zero-edit drop-in)
"""
__version__ = "0.1.0"

from .rays import (Rays, Rays_keys, namedtuple_map, rearrange_render_image, generate_pano_rays, generate_lit_rays,  # noqa
                   DeviceRayPool)
from .render import PanoMipNeRF, MipNeRF  # noqa
from .loss import pano_loss, mip_loss  # noqa
from .optim import FlatAdam, mip_lr  # noqa
from .renderer import render_image  # noqa
from . import metrics, io_exr  # noqa
from .parallel import concurrent_step  # noqa
from .install import install, uninstall  # noqa
