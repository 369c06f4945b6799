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
    """Batch sampler over the pixels of a set of equirectangular cameras.

    Stands in for the reference's flattened numpy ray pool and its 28 DataLoader workers
    (datasets/pano_datasets.py:133-150, 271-275; systems/base_system.py:89-96).  NO ray pool is stored: a training
    batch is one ``torch.randint`` over (camera, pixel) plus one kernel (``pn_sample_pano_rays``) that REGENERATES the
    rays of the drawn pixels from the camera matrices with the arithmetic of ``pn_raygen_pano`` - bit-identical to a
    gather out of the materialised pool (56 B/ray of HBM and of reads saved; nothing crosses PCIe per step).  Only the
    target colours ``rgbs`` ([n_cam*H*W, 3], optional: ``images`` = [H, W, 3] HDR arrays per camera) are kept and gathered.
    """

    def __init__(self, height, width, c2ws, images=None, near=0.0, far=10.0, device="cuda"):
        self.h, self.w = int(height), int(width)
        self.near, self.far = float(near), float(far)
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        mats = np.stack([np.asarray(c, dtype=np.float32).reshape(4, 4) for c in c2ws], 0)
        self.c2ws_host = mats
        self.c2ws = torch.from_numpy(np.ascontiguousarray(mats.reshape(len(mats), 16))).to(self.device)
        self.n_cam = len(mats)
        # the constant pixel radius of the pano (datasets/pano_datasets.py:215): the light rays carry it
        self.radius = pano_pixel_radius(self.take(torch.zeros(1, dtype=torch.int64, device=self.device))[0])
        self.rgbs = None
        if images is not None:
            self.rgbs = torch.cat([torch.as_tensor(im, dtype=torch.float32).reshape(-1, 3) for im in images], 0).to(self.device)
            if self.rgbs.shape[0] != len(self):
                raise ValueError("images must be [H, W, 3] per camera")

    def __len__(self):
        return self.n_cam * self.h * self.w

    @property
    def rays(self):
        """The materialised pool (camera-major, row-major pixels) - for tests and one-off uses; NOT cached."""
        pools = [generate_pano_rays(self.h, self.w, c, self.near, self.far, device=self.device) for c in self.c2ws_host]
        return Rays(*[torch.cat([getattr(p, k) for p in pools], 0) for k in Rays_keys])

    def take(self, idx):
        """Rays (and target colours) of the pool rows `idx` (int64 device tensor, row = camera * H * W + pixel)."""
        dev = self.device
        idx = idx.to(device=dev, dtype=torch.int64).contiguous()
        B = int(idx.numel())
        outs = [torch.empty(B, d, dtype=torch.float32, device=dev) for d in _DIMS]
        rgbs = getattr(self, "rgbs", None)
        rgb = torch.empty(B, 3, dtype=torch.float32, device=dev) if rgbs is not None else None
        with torch.cuda.device(dev):
            _lib.call("pn_sample_pano_rays", B, self.n_cam, self.h, self.w, idx.data_ptr(), self.c2ws.data_ptr(), self.near,
                      self.far, _lib.ptr(rgbs), *[x.data_ptr() for x in outs], _lib.ptr(rgb), _stream(dev))
        return Rays(*outs), rgb

    def sample(self, batch_size, generator=None):
        """-> (Rays of [B, C], rgb [B, 3] or None), all on the device."""
        idx = torch.randint(0, len(self), (int(batch_size),), device=self.device, generator=generator)
        return self.take(idx)

    def lit_rays(self, num=10, near=0.0, far=10.0):
        return generate_lit_rays(num, self.radius, near, far, device=self.device)
