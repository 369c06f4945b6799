"""Ray containers and on-device ray generation.

``Rays`` is the reference's batch container (datasets/base_datasets.py:13-16).  ``generate_pano_rays``
and ``generate_lit_rays`` stand in for ``PanoDataset._generate_rays`` / ``.generate_lit_rays``
(datasets/pano_datasets.py:152-216, 218-263; == utils/sampling.py:5-38) and run as HIP kernels, so a
ray pool can be regenerated in HBM from (camera, pixel) instead of being shipped from the host.
"""
import collections

import numpy as np
import torch

from . import _lib

Rays = collections.namedtuple(
    "Rays", ("origins", "directions", "viewdirs", "radii", "lossmult", "near", "far", "noise_var"))
Rays_keys = Rays._fields
_DIMS = (3, 3, 3, 1, 1, 1, 1, 1)


def namedtuple_map(fn, tup):
    return type(tup)(*map(fn, tup))


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def generate_pano_rays(h, w, c2w, near=0.0, far=10.0, device="cuda"):
    """One equirectangular camera -> Rays of [h*w, C] fp32 device tensors (row-major pixels)."""
    device = torch.device(device)
    c = np.ascontiguousarray(np.asarray(c2w, dtype=np.float32).reshape(-1))
    if c.size != 16:
        raise ValueError("c2w must be a 4x4 matrix")
    out = [torch.empty(h * w, d, dtype=torch.float32, device=device) for d in _DIMS]
    with torch.cuda.device(device):
        _lib.call("pn_raygen_pano", int(h), int(w), c.ctypes.data, float(near), float(far),
                  *[t.data_ptr() for t in out], _stream(device))
    return Rays(*out)


def pano_pixel_radius(rays):
    """The constant pixel radius of a pano (datasets/pano_datasets.py:215)."""
    return float(rays.radii[0, 0])


def generate_lit_rays(num, radius, near=0.0, far=10.0, device="cuda"):
    """`num` golden-spiral light rays as fp16 tensors, like the reference's env_rays."""
    device = torch.device(device)
    buf = torch.empty(14 * num, dtype=torch.float16, device=device)
    with torch.cuda.device(device):
        _lib.call("pn_lit_rays", int(num), float(radius), float(near), float(far), buf.data_ptr(), _stream(device))
    v3 = buf[:9 * num].view(3, num, 3)
    s = buf[9 * num:].view(5, num, 1)
    return Rays(v3[0], v3[1], v3[2], s[0], s[1], s[2], s[3], s[4])


def rearrange_render_image(rays, chunk_size=4096):
    """models/mip.py:530-547: flatten [1,H,W,C] rays and slice into chunks."""
    flat = [getattr(rays, k).reshape(-1, getattr(rays, k).shape[-1]) for k in Rays_keys]
    val_mask = flat[-3]
    n = flat[0].shape[0]
    chunks = [Rays(*[a[i:i + chunk_size] for a in flat]) for i in range(0, n, chunk_size)]
    return chunks, val_mask


class DeviceRayPool:
    """HBM-resident ray pool + batch sampler.

    Stands in for the reference's flattened numpy pool and its 28 DataLoader workers
    (datasets/pano_datasets.py:133-150, 271-275; systems/base_system.py:89-96): rays are generated on the device
    from the camera matrices by ``pn_raygen_pano`` (nothing crosses PCIe per step) and a training batch is one
    ``torch.randint`` + gather on the device.  ``images`` (optional) are the [H, W, 3] HDR targets per camera.
    """

    def __init__(self, height, width, c2ws, images=None, near=0.0, far=10.0, device="cuda"):
        self.h, self.w = int(height), int(width)
        pools = [generate_pano_rays(height, width, c, near, far, device=device) for c in c2ws]
        self.rays = Rays(*[torch.cat([getattr(p, k) for p in pools], 0) for k in Rays_keys])
        self.radius = pano_pixel_radius(pools[0])
        self.rgbs = None
        if images is not None:
            self.rgbs = torch.cat([torch.as_tensor(im, dtype=torch.float32).reshape(-1, 3) for im in images], 0).to(
                self.rays.origins.device)
            if self.rgbs.shape[0] != len(self):
                raise ValueError("images must be [H, W, 3] per camera")

    def __len__(self):
        return self.rays.origins.shape[0]

    def sample(self, batch_size, generator=None):
        """-> (Rays of [B, C], rgb [B, 3] or None), all on the device."""
        import ctypes
        dev = self.rays.origins.device
        B = int(batch_size)
        idx = torch.randint(0, len(self), (B,), device=dev, generator=generator)
        outs = [torch.empty(B, x.shape[1], dtype=torch.float32, device=dev) for x in self.rays]
        rgb = torch.empty(B, 3, dtype=torch.float32, device=dev) if self.rgbs is not None else None
        src = (ctypes.c_void_p * 9)(*[x.data_ptr() for x in self.rays], self.rgbs.data_ptr() if rgb is not None else None)
        dst = (ctypes.c_void_p * 9)(*[x.data_ptr() for x in outs], rgb.data_ptr() if rgb is not None else None)
        with torch.cuda.device(dev):
            _lib.call("pn_gather_rays", B, len(self), idx.data_ptr(), src, dst, torch.cuda.current_stream(dev).cuda_stream)
        return Rays(*outs), rgb

    def lit_rays(self, num=10, near=0.0, far=10.0):
        return generate_lit_rays(num, self.radius, near, far, device=self.rays.origins.device)
