"""Parameter container of the radiance MLP.

Mirrors ``MLP`` (models/pano_mip_nerf.py:17-76) and ``PureMLP`` (models/mip_nerf.py:15-60): same
module tree, hence the same ``state_dict`` keys (``layers.{i}.0.weight`` ... ``color_layer.bias``)
and the same init distributions, so reference checkpoints load and ``.mlp.parameters()`` feeds the
caller's Adam (systems/base_system.py:82).  Unlike the reference the 24 tensors are VIEWS of one
flat fp32 block laid out as ``pn_param_layout`` says: the HIP kernels read weights straight from
it, the data-parallel gradient exchange is a single all-reduce and Adam is a single kernel.

This class holds parameters only; evaluation happens in ``pano_nerf_amd.render``.
"""
import ctypes

import torch

from . import _lib

ORDER = ([f"layers.{i}.0.{k}" for i in range(8) for k in ("weight", "bias")] +
         ["extra_layer.weight", "extra_layer.bias", "view_layers.0.0.weight", "view_layers.0.0.bias",
          "density_layer.weight", "color_layer.weight", "density_layer.bias", "color_layer.bias"])


def param_layout(num_density_channels):
    off = (ctypes.c_int64 * 24)()
    total = _lib.load().pn_param_layout(int(num_density_channels), off)
    if total < 0:
        _lib.check(int(total), "pn_param_layout")
    return dict(zip(ORDER, list(off))), int(total)


class RadianceMLP(torch.nn.Module):
    def __init__(self, net_depth=8, net_width=256, net_depth_condition=1, net_width_condition=128, skip_index=4,
                 num_rgb_channels=3, num_density_channels=5, activation="relu", xyz_dim=96, view_dim=27):
        super().__init__()
        if activation != "relu":
            raise NotImplementedError  # models/pano_mip_nerf.py:52-53
        fixed = (net_depth, net_width, net_depth_condition, net_width_condition, skip_index, num_rgb_channels,
                 xyz_dim, view_dim)
        if fixed != (8, 256, 1, 128, 4, 3, 96, 27) or num_density_channels not in (1, 5):
            raise NotImplementedError(
                "the HIP kernels are specialised for the 8x256 trunk, skip 4, 128-wide view layer, 96/27 encodings "
                f"and 1 or 5 density channels of configs/*.yaml; got {fixed}, density channels {num_density_channels}")
        self.skip_index = skip_index
        self.num_density_channels = num_density_channels
        lin = torch.nn.Linear
        layers = []
        for i in range(net_depth):
            k = xyz_dim if i == 0 else (net_width + xyz_dim if (i - 1) % skip_index == 0 and i > 1 else net_width)
            layers.append(torch.nn.Sequential(lin(k, net_width), torch.nn.ReLU(True)))
        self.layers = torch.nn.ModuleList(layers)
        self.density_layer = lin(net_width, num_density_channels)
        self.extra_layer = lin(net_width, net_width)
        self.view_layers = torch.nn.Sequential(
            torch.nn.Sequential(lin(net_width + view_dim, net_width_condition), torch.nn.ReLU(True)))
        self.color_layer = lin(net_width_condition, num_rgb_channels)
        for m in [l[0] for l in self.layers] + [self.density_layer, self.extra_layer, self.view_layers[0][0]]:
            torch.nn.init.xavier_uniform_(m.weight.data)  # models/pano_mip_nerf.py:10-14; color_layer keeps default
        self._offsets, self._total = None, None
        self.flat = None
        self._wpack = None
        self._generation = 0  # bumped by every writer that bypasses autograd's version counters (FlatAdam: raw pointers)
        self._chain = {}  # planes -> packed chains of the fused kernels (buffer reused, contents rebuilt per call)
        self._frozen = False  # concurrent_step: packs built once before the sub-batch streams fork
        self.last_flat_grad = None
        self._flatten()

    # ------------------------------------------------------------------ flat storage
    def named_in_order(self):
        table = dict(self.named_parameters())
        return [(k, table[k]) for k in ORDER]

    def _flatten(self):
        """(Re)create the flat block on the parameters' current device and re-point every parameter at it."""
        params = self.named_in_order()
        dev = params[0][1].device
        if self._offsets is None:
            self._offsets, self._total = param_layout(self.num_density_channels)
        flat = torch.empty(self._total, dtype=torch.float32, device=dev)
        for k, p in params:
            o = self._offsets[k]
            flat[o:o + p.numel()].copy_(p.data.reshape(-1).float())
            p.data = flat[o:o + p.numel()].view(p.shape)
        self.flat = flat
        self._wpack = None
        self._chain = {}

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._flatten()
        return out

    def is_flat(self):
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + 4 * self._offsets[k] and p.dtype == torch.float32
                   for k, p in self.named_in_order())

    def flat_params(self):
        if not self.is_flat():  # e.g. someone replaced .data; restore the invariant
            self._flatten()
        return self.flat

    def _version(self):
        """Autograd version counters of the 24 parameters and of the flat block (in-place writes that go through torch
        bump them; `p.data.copy_`, raw device writes and collectives do not — hence the packs below are rebuilt on every
        call rather than cached on this key)."""
        flat = self.flat_params()
        return tuple(p._version for _, p in self.named_in_order()) + (flat._version, flat.data_ptr(), self._generation)

    def note_raw_write(self):
        """A writer that goes through raw device pointers (FlatAdam's kernels) calls this: a backward whose forward saw
        the older weights then raises instead of mixing them with activations of the newer ones."""
        self._generation += 1

    def packed(self, stream):
        """Transposed / split weight copies used by the layer-wise GEMM path (pn_pack_weights).  Rebuilt on EVERY call:
        a write that bypasses the version counters (p.data.copy_, an EMA update, a custom loader, a collective into
        `flat`) must never leave the backward-direction GEMMs on stale copies, and the pack is one ~2.4 MB pass."""
        flat = self.flat_params()
        if self._wpack is None or self._wpack.device != flat.device:
            n = _lib.load().pn_wpack_floats(self.num_density_channels)
            self._wpack = torch.empty(n, dtype=torch.float32, device=flat.device)
        elif self._frozen:
            return self._wpack
        _lib.call("pn_pack_weights", flat.data_ptr(), self.num_density_channels, self._wpack.data_ptr(), stream)
        return self._wpack

    def chain_packed(self, stream, planes):
        """Fragment-ordered bf16 planes of every weight matrix, in both directions, for the fused chain kernels
        (pn_chain_pack: ~3-7 MB written); rebuilt on every call for the same reason as `packed`.  The rebuild is IN PLACE on
        the caller's stream: code that runs forwards on several streams at once builds the pack once before the fork and
        sets `_frozen` for the duration (parallel.concurrent_step, renderer.render_image)."""
        flat = self.flat_params()
        buf = self._chain.get(planes)
        if buf is None or buf.device != flat.device:
            buf = torch.empty(int(_lib.load().pn_chain_pack_bytes(planes)), dtype=torch.uint8, device=flat.device)
            self._chain[planes] = buf
        elif self._frozen:
            return buf
        _lib.call("pn_chain_pack", flat.data_ptr(), self.num_density_channels, planes, buf.data_ptr(), stream)
        return buf

    def grad_views(self, flat_grad):
        return [flat_grad[self._offsets[k]:self._offsets[k] + p.numel()].view(p.shape) for k, p in self.named_in_order()]

    def forward(self, *a, **kw):
        raise RuntimeError("RadianceMLP holds parameters only; call PanoMipNeRF / MipNeRF (pano_nerf_amd.render)")
