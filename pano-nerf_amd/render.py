"""PanoMipNeRF / MipNeRF — drop-in modules for the reference's ``models/pano_mip_nerf.py`` and
``models/mip_nerf.py`` (same constructor keywords, same ``.mlp`` state-dict keys, same ``forward``
keywords and returned tuples), evaluated by hand-written HIP kernels through the C ABI of
``libpanonerf_hip.so``.  PyTorch is used for device memory, the current stream and autograd glue
only: the whole render (both levels, normals, env light, surface) is ONE ``autograd.Function`` whose
backward calls the hand-written adjoint kernels, including the second-order path through the
density-gradient normals.

There is no CPU / eager fallback: tensors must live on a HIP device and the library must be built.
"""
import ctypes

import torch

from . import _lib
from .mlp import RadianceMLP
from .rays import Rays

_NAMES9 = ("comp_rgb", "distance", "ort_loss", "normal", "albedo", "roughness", "surface_rgb", "diffuse", "shading")


def _f32(x):
    return x.detach().to(torch.float32).contiguous()


class _Eval:
    """Buffers of one MLP evaluation over M sample rows (all caller-owned HBM).  planes = 0: layer-wise GEMM path
    (row-major activations); planes = 3 / 2 / 1: fused chain kernels (T32 sample-minor tensors, see pn_chain.hip)."""

    def __init__(self, M, rows_per_ray, viewdirs, nc, dev, planes=0, keep=True, tfmt=0):
        self.M, self.rows_per_ray, self.nc = M, rows_per_ray, nc
        self.view_rows = viewdirs.shape[0]
        self.viewdirs = viewdirs
        self.planes = planes
        self.tfmt = tfmt  # 1: the 256-wide tensors only the weight gradients read back are Q24 (include/panonerf_hip.h)
        Mp = int(_lib.load().pn_pad_rows(M))
        self.Mp = Mp
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        self.mean, self.cov = e(M, 3), e(M, 3)
        if planes:
            self.enc = e(Mp * 96)                                          # T [96]
            self.viewtab = e(self.view_rows * 32)                          # view encoding per view row (scratch of the forward)
            # T h0..h7, bottleneck | viewenc, view hidden — only the backward re-reads them: not kept in inference
            self.acts = e(int(_lib.load().pn_chain_acts_floats(M))) if keep else None
            # planes = 2, training: largest |x| of every T tensor (one power-of-two scale per tensor in the weight gradients)
            self.amax = (torch.empty(int(_lib.load().pn_chain_amax_slots()), dtype=torch.int32, device=dev)
                         if (planes == 2 and keep) else None)
        else:
            self.enc = e(Mp, 96)
            self.viewenc, self.viewbias = e(self.view_rows, 27), e(self.view_rows, 128)
            self.acts = e(10, Mp, 256)
            self.amax = None
        self.masks = torch.empty(9, Mp, 8, dtype=torch.int32, device=dev)  # ReLU gates as bit masks
        self.raw_rgb, self.raw_den = e(M, 3), e(M, nc)
        self.t = None
        self.rsweep = None
        self.gmean = None


MLP_MODES = ("fused_f16x2", "fused_f16x2_t32", "fused", "fused_bf16", "layerwise")


def _planes_of(mode):
    try:
        return {"fused": 3, "fused_f16x2": 2, "fused_f16x2_t32": 2, "fused_bf16": 1, "layerwise": 0}[mode]
    except KeyError:
        raise ValueError(f"mlp_mode must be one of {MLP_MODES}, got {mode!r}")


def _tfmt_of(mode):
    """T-tensor format of a mode (the t_format argument of the chain entry points): "fused_f16x2" stores the 256-wide
    tensors that only the weight gradients read back in three bytes per element (Q24, 16 significant bits) where the
    build has them; "fused_f16x2_t32" is the same arithmetic with every T tensor in fp32."""
    if mode != "fused_f16x2":
        return 0
    return 1 if int(_lib.load().pn_chain_q24_slots(2, 1, 0)) else 0


class _Cfg:
    """Static configuration of one render call."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def _mlp_forward(ev, params, wpack, st):
    if ev.planes:
        _lib.call("pn_chain_forward", ev.M, ev.rows_per_ray, ev.view_rows, ev.nc, ev.planes, wpack.data_ptr(),
                  ev.mean.data_ptr(), ev.cov.data_ptr(), ev.viewdirs.data_ptr(), ev.viewtab.data_ptr(), ev.enc.data_ptr(),
                  _lib.ptr(ev.acts),
                  ev.masks.data_ptr(), ev.raw_rgb.data_ptr(), ev.raw_den.data_ptr(), _lib.ptr(ev.amax), ev.tfmt, 0, st)
        return
    _lib.call("pn_mlp_forward", ev.M, ev.rows_per_ray, ev.view_rows, ev.nc, params.data_ptr(), wpack.data_ptr(),
              ev.mean.data_ptr(), ev.cov.data_ptr(), ev.viewdirs.data_ptr(), ev.enc.data_ptr(), ev.viewenc.data_ptr(),
              ev.viewbias.data_ptr(), ev.acts.data_ptr(), ev.masks.data_ptr(), ev.raw_rgb.data_ptr(),
              ev.raw_den.data_ptr(), st)


def _composite_forward(ev, R, N, cfg, white, dirs, dir_mod, st):
    dev = ev.raw_rgb.device
    comp = torch.empty(R, 3, dtype=torch.float32, device=dev)
    dist = torch.empty(R, dtype=torch.float32, device=dev)
    acc = torch.empty(R, dtype=torch.float32, device=dev)
    w = torch.empty(R, N, dtype=torch.float32, device=dev)
    _lib.call("pn_composite_forward", R, N, ev.nc, cfg.density_bias, cfg.rgb_padding, int(white), ev.raw_rgb.data_ptr(),
              ev.raw_den.data_ptr(), ev.t.data_ptr(), dirs.data_ptr(), dir_mod, comp.data_ptr(), dist.data_ptr(),
              acc.data_ptr(), w.data_ptr(), st)
    return comp, dist, acc, w


def _composite_backward(ev, R, N, cfg, white, dirs, dir_mod, d_comp, d_dist, d_w, d_raw_rgb, d_raw_den, st):
    _lib.call("pn_composite_backward", R, N, ev.nc, cfg.density_bias, cfg.rgb_padding, int(white),
              ev.raw_rgb.data_ptr(), ev.raw_den.data_ptr(), ev.t.data_ptr(), dirs.data_ptr(), dir_mod,
              d_comp.data_ptr(), _lib.ptr(d_dist), _lib.ptr(d_w), d_raw_rgb.data_ptr(), d_raw_den.data_ptr(), st)


_SIDE = {}


def _side_stream(dev):
    """Second HIP stream per device for the weight-gradient work (forked/joined inside pn_mlp_backward)."""
    key = (dev.type, dev.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


def _mlp_backward(ev, cfg, params, wpack, d_raw_rgb, d_raw_den, v, d_mean, flat_grad, st, defer=False, deferred=()):
    """One evaluation's backward.  `defer=True` leaves its trunk weight-gradient operands in the returned workspace;
    the last call of the step passes those as `deferred` = [(eval, work, had_tangent), ...] and reduces them all in
    one GEMM per layer."""
    import ctypes
    m_batched = ev.M * (2 if v is not None else 1) + sum(e.M * (2 if t else 1) for e, _, t in deferred)
    n = int(_lib.load().pn_mlp_backward_work_floats(ev.M, ev.rows_per_ray, ev.view_rows, m_batched if deferred else 0))
    work = torch.empty(n, dtype=torch.float32, device=flat_grad.device)
    nd = len(deferred)
    if nd:
        dM = (ctypes.c_int64 * nd)(*[e.M for e, _, _ in deferred])
        denc = (ctypes.c_void_p * nd)(*[e.enc.data_ptr() for e, _, _ in deferred])
        dacts = (ctypes.c_void_p * nd)(*[e.acts.data_ptr() for e, _, _ in deferred])
        drs = (ctypes.c_void_p * nd)(*[(e.rsweep.data_ptr() if t else None) for e, _, t in deferred])
        dwork = (ctypes.c_void_p * nd)(*[w.data_ptr() for _, w, _ in deferred])
        dtan = (ctypes.c_int * nd)(*[int(bool(t)) for _, _, t in deferred])
    else:
        dM = denc = dacts = drs = dwork = dtan = None
    _lib.call("pn_mlp_backward", ev.M, ev.rows_per_ray, ev.view_rows, ev.nc, cfg.density_bias, params.data_ptr(),
              wpack.data_ptr(), ev.mean.data_ptr(), ev.cov.data_ptr(), ev.enc.data_ptr(), ev.viewenc.data_ptr(),
              ev.acts.data_ptr(), ev.masks.data_ptr(), ev.raw_den.data_ptr(), d_raw_rgb.data_ptr(), d_raw_den.data_ptr(),
              _lib.ptr(ev.rsweep), _lib.ptr(v), _lib.ptr(d_mean), flat_grad.data_ptr(), work.data_ptr(),
              m_batched if deferred else 0, int(defer), nd, dM, denc, dacts, drs, dwork, dtan, st,
              _side_stream(flat_grad.device).cuda_stream if cfg.overlap else None)
    return work


class _ChainEvalC(ctypes.Structure):
    """PnChainEval of include/panonerf_hip.h"""
    _fields_ = [("M", ctypes.c_int64), ("enc_t", ctypes.c_void_p), ("acts_t", ctypes.c_void_p),
                ("drgb_t", ctypes.c_void_p), ("dhv_t", ctypes.c_void_p), ("d8_t", ctypes.c_void_p),
                ("delta_t", ctypes.c_void_p), ("rs_t", ctypes.c_void_p), ("edot_t", ctypes.c_void_p),
                ("tang_t", ctypes.c_void_p), ("coef_t", ctypes.c_void_p), ("amax", ctypes.c_void_p)]


def _chain_tangent(ev, cfg, params, pack, v, st, wgs=0):
    """Forward-mode tangent sweep along v = dL/d(grad_mean): edot, hdot_0..7 (with r_l: the second-order weight gradients)
    and sdot (the second-order addend of the backward chain's density seed)."""
    dev = v.device
    e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    Mp = ev.Mp
    ev.edot, ev.tang, ev.sdot = e(Mp * 96), e(8 * Mp * 256), e(ev.M)
    _lib.call("pn_chain_tangent", ev.M, ev.nc, ev.planes, params.data_ptr(), pack.data_ptr(), ev.mean.data_ptr(),
              ev.cov.data_ptr(), ev.masks.data_ptr(), v.data_ptr(), ev.edot.data_ptr(), ev.tang.data_ptr(),
              ev.sdot.data_ptr(), _lib.ptr(ev.amax), ev.tfmt, wgs, st)
    z = e if _lib.load().pn_chain_tile() == 16 else (lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev))
    ev.coef = z(Mp * 32)


def _chain_backward(ev, cfg, params, pack, d_raw_rgb, d_raw_den, v, d_mean, st, wgs=0):
    """Fused data-gradient chain of one evaluation (tangent sweep first when v = dL/d(grad_mean) is given, unless the caller
    ran it: ev.sdot set); leaves the T32 tensors the weight-gradient GEMMs read on `ev`.  wgs: workgroup budget (0 = all CUs)."""
    dev = d_raw_rgb.device
    e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    # drgb / d8 / coef hold a padded k-step that the 32-sample build (pn_chain_tile() = 32) writes only half of; the
    # default 16-sample kernels write every feature of every padded row themselves (0.3 ms of fills per step)
    z = e if _lib.load().pn_chain_tile() == 16 else (lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev))
    Mp = ev.Mp
    if v is None:
        ev.edot = ev.tang = ev.coef = ev.sdot = None
    elif getattr(ev, "sdot", None) is None:
        _chain_tangent(ev, cfg, params, pack, v, st, wgs)
    ev.drgb, ev.dhv, ev.d8, ev.delta = z(Mp * 32), e(Mp * 128), z(Mp * 288), e(8 * Mp * 256)
    _lib.call("pn_chain_backward", ev.M, ev.nc, ev.planes, cfg.density_bias, pack.data_ptr(), ev.masks.data_ptr(),
              ev.raw_den.data_ptr(), d_raw_rgb.data_ptr(), d_raw_den.data_ptr(), _lib.ptr(ev.sdot), ev.mean.data_ptr(),
              ev.cov.data_ptr(), ev.drgb.data_ptr(), ev.dhv.data_ptr(), ev.d8.data_ptr(), ev.delta.data_ptr(),
              _lib.ptr(ev.coef), _lib.ptr(d_mean), _lib.ptr(ev.amax), ev.tfmt, wgs, st)


def _chain_wgrad(evals, nc, planes, flat_grad, st, which=3, wgs=0):
    """One TN GEMM per layer over the sample blocks of the given evaluations (all of the step's, or - concurrent schedule -
    one at a time).  which: 1 first-order products, 2 second-order trunk rows, 3 both (include/panonerf_hip.h)."""
    lib = _lib.load()
    arr = (_ChainEvalC * len(evals))()
    for i, ev in enumerate(evals):
        second = ev.tang is not None
        first = bool(which & 1)  # (the second-order trunk rows alone are reduced before the evaluation's backward chain has run)
        arr[i] = _ChainEvalC(ev.M, ev.enc.data_ptr(), ev.acts.data_ptr(), ev.drgb.data_ptr() if first else None,
                             ev.dhv.data_ptr() if first else None, ev.d8.data_ptr() if first else None,
                             ev.delta.data_ptr() if first else None, ev.rsweep.data_ptr() if second else None,
                             ev.edot.data_ptr() if second else None, ev.tang.data_ptr() if second else None,
                             ev.coef.data_ptr() if second else None, _lib.ptr(ev.amax))
    n = int(lib.pn_chain_wgrad_work_floats())
    work = torch.empty(n, dtype=torch.float32, device=flat_grad.device)
    _lib.check(lib.pn_chain_wgrad(len(evals), ctypes.cast(arr, ctypes.c_void_p), nc, planes, flat_grad.data_ptr(),
                                  work.data_ptr(), n, which, evals[0].tfmt, wgs, st), "pn_chain_wgrad")
    return work


class _RenderFn(torch.autograd.Function):
    """(rays, env rays, noise, 24 parameters) -> the ten differentiable outputs of both levels."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, cfg, mlp, o, d, vd, radii, near, far, env_d, env_rad, env_near, env_far, env_omega, t_rand,
                u_rand, env_rand, *plist):
        dev = o.device
        if dev.type != "cuda":
            raise RuntimeError("pano_nerf_amd renders on a HIP device only (tensors are on %s); there is no CPU "
                               "fallback" % dev)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            params = mlp.flat_params()
            planes = cfg.planes
            wpack = mlp.chain_packed(st, planes) if planes else mlp.packed(st)
            B, N, nc = o.shape[0], cfg.num_samples, cfg.nc
            S, M = N + 1, o.shape[0] * cfg.num_samples
            e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            keep = cfg.keep  # inference (no_grad): release the big activation buffers as soon as possible
            # ---- level 0: stratified samples
            e0 = _Eval(M, N, vd, nc, dev, planes, keep, cfg.tfmt)
            e0.t = e(B, S)
            _lib.call("pn_sample_coarse", B, N, int(cfg.disparity), o.data_ptr(), d.data_ptr(), radii.data_ptr(), near.data_ptr(),
                      far.data_ptr(), _lib.ptr(t_rand), e0.t.data_ptr(), e0.mean.data_ptr(), e0.cov.data_ptr(), st)
            if cfg.disable_integration:  # models/pano_mip_nerf.py:241-243: every encoding of compute_graph sees a zero covariance
                e0.cov.zero_()
            _mlp_forward(e0, params, wpack, st)
            comp0, dist0, _, w0 = _composite_forward(e0, B, N, cfg, cfg.white_bkgd, d, B, st)
            if not keep:
                e0.acts = e0.masks = e0.enc = None
            # ---- level 1: PDF resample (no gradient through the weights: stop_resample_grad)
            e1 = _Eval(M, N, vd, nc, dev, planes, keep, cfg.tfmt)
            e1.t = e(B, S)
            _lib.call("pn_resample", B, N, e0.t.data_ptr(), w0.data_ptr(), cfg.resample_padding, _lib.ptr(u_rand),
                      o.data_ptr(), d.data_ptr(), radii.data_ptr(), e1.t.data_ptr(), e1.mean.data_ptr(),
                      e1.cov.data_ptr(), st)
            if cfg.disable_integration:
                e1.cov.zero_()
            _mlp_forward(e1, params, wpack, st)
            comp1, dist1, _, w1 = _composite_forward(e1, B, N, cfg, cfg.white_bkgd, d, B, st)
            normal = ort = albedo = surface = diffuse = shading = None
            ee = env_rgb = None
            if cfg.normals:
                e1.rsweep = e(8 if (keep or not planes) else 1, e1.Mp, 256)
                e1.gmean = e(M, 3)
                scratch = None
                if planes:
                    _lib.call("pn_chain_density_grad", M, nc, planes, cfg.density_bias, params.data_ptr(),
                              wpack.data_ptr(), e1.mean.data_ptr(), e1.cov.data_ptr(), e1.masks.data_ptr(),
                              e1.raw_den.data_ptr(), e1.rsweep.data_ptr(), int(keep), e1.gmean.data_ptr(),
                              _lib.ptr(e1.amax), e1.tfmt, 0, st)
                else:
                    scratch = e(e1.Mp, 96)
                    _lib.call("pn_density_grad", M, nc, cfg.density_bias, params.data_ptr(), wpack.data_ptr(),
                              e1.mean.data_ptr(), e1.cov.data_ptr(), e1.acts.data_ptr(), e1.masks.data_ptr(),
                              e1.raw_den.data_ptr(), e1.rsweep.data_ptr(), scratch.data_ptr(), e1.gmean.data_ptr(), st)
                normal = e(B, 3)
                ort_ray = e(B) if cfg.use_ort else None
                albedo = e(B, 3) if (cfg.surf and nc == 5) else None
                _lib.call("pn_surf_gather_forward", B, N, nc, e1.gmean.data_ptr(), w1.data_ptr(), e1.raw_den.data_ptr(),
                          d.data_ptr(), normal.data_ptr(), _lib.ptr(ort_ray), _lib.ptr(albedo), st)
                if cfg.use_ort:
                    ort = ort_ray.mean()
                if not keep:
                    e1.rsweep = None
                    del scratch
            if not keep:
                e1.acts = e1.masks = e1.enc = None
            if cfg.surf:
                D, Ne = env_d.shape[0], cfg.num_env_samples
                ee = _Eval(B * D * Ne, Ne, env_d, nc, dev, planes, keep, cfg.tfmt)
                ee.t = e(B * D, Ne + 1)
                _lib.call("pn_sample_env", B, D, Ne, o.data_ptr(), d.data_ptr(), dist1.data_ptr(), env_d.data_ptr(),
                          env_rad.data_ptr(), env_near.data_ptr(), env_far.data_ptr(), _lib.ptr(env_rand),
                          ee.t.data_ptr(), ee.mean.data_ptr(), ee.cov.data_ptr(), st)
                if cfg.disable_integration:
                    ee.cov.zero_()
                _mlp_forward(ee, params, wpack, st)
                env_rgb, _, _, _ = _composite_forward(ee, B * D, Ne, cfg, False, env_d, D, st)
                diffuse, shading = e(B, 3), e(B, 3)
                _lib.call("pn_surface_forward", B, D, env_rgb.data_ptr(), albedo.data_ptr(), normal.data_ptr(),
                          env_d.data_ptr(), env_omega.data_ptr(), diffuse.data_ptr(), shading.data_ptr(), st)
                surface = diffuse.clone()
        ctx.cfg, ctx.mlp = cfg, mlp
        # `normal` and `albedo` are OUTPUTS of this Function: they go through save_for_backward (an output kept as a plain
        # ctx attribute makes an output -> grad_fn -> ctx -> output cycle that only backward() would break; a forward
        # that is never back-propagated would leak its activation buffers).  The evaluation buffers are not outputs.
        ctx.has_normal, ctx.has_albedo = normal is not None, albedo is not None
        ctx.save_for_backward(*[x for x in (normal, albedo) if x is not None])
        ctx.pack = (o, d, vd, env_d, env_omega, e0, e1, ee, w1, env_rgb, params, wpack)
        ctx.param_version = mlp._version() if keep else None
        if getattr(mlp, "debug_keep", False):  # diagnostics only (tests/diag/diag_full.py, tests/test_gpu_grads.py)
            mlp.debug_pack = (o, d, vd, env_d, env_omega, e0, e1, ee, w1, env_rgb, albedo, normal, params, wpack)
        outs = (comp0, dist0, comp1, dist1, ort, normal, albedo, surface, diffuse, shading)
        ctx.present = [x is not None for x in outs]
        return tuple(x if x is not None else torch.zeros((), device=dev) for x in outs)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_comp0, g_dist0, g_comp1, g_dist1, g_ort, g_normal, g_albedo, g_surface, g_diffuse, g_shading):
        cfg, mlp = ctx.cfg, ctx.mlp
        if ctx.pack is None:
            raise RuntimeError("pano_nerf_amd: backward through the same render twice (its buffers were released)")
        o, d, vd, env_d, env_omega, e0, e1, ee, w1, env_rgb, params, wpack = ctx.pack
        saved = list(ctx.saved_tensors)
        normal = saved.pop(0) if ctx.has_normal else None
        albedo = saved.pop(0) if ctx.has_albedo else None
        if ctx.param_version is not None and mlp._version() != ctx.param_version:
            raise RuntimeError("pano_nerf_amd: the MLP parameters were modified in place between forward and backward "
                               "(e.g. an optimizer step before backward()); the saved activations belong to the old weights")
        dev = o.device
        B, N, nc = o.shape[0], cfg.num_samples, cfg.nc
        M = B * N
        # every zero-initialised temporary of this backward is a view of ONE zero-filled arena (one fill launch instead of
        # ~14; the step is launch-sensitive at 512 rays per GPU); sizes padded to 64 floats keep the views 256-B aligned
        pad = lambda n: (n + 63) & ~63
        need = 8 * pad(B * 3) + 2 * pad(B)  # stand-ins for absent upstream gradients
        if cfg.surf:
            D_, Ne_ = env_d.shape[0], cfg.num_env_samples
            need += pad(B * D_ * 3) + 2 * pad(B * 3) + 2 * pad(ee.M * 3) + pad(ee.M * nc)
        need += 2 * (pad(M * 3) + pad(M * nc)) + pad(B * N) + pad(M * 3)
        arena = torch.zeros(need, dtype=torch.float32, device=dev)
        used = [0]

        def z(*shape):
            n = 1
            for k in shape:
                n *= k
            if used[0] + pad(n) > need:  # not expected; keeps the function total
                return torch.zeros(*shape, dtype=torch.float32, device=dev)
            out = arena[used[0]:used[0] + n].view(*shape)
            used[0] += pad(n)
            return out

        gz = lambda g, *s: _f32(g) if g is not None else z(*s)
        with torch.cuda.device(dev):
            main = torch.cuda.current_stream(dev)
            st = main.cuda_stream
            flat_grad = torch.zeros(params.numel(), dtype=torch.float32, device=dev)  # outlives this call (.grad views)
            # CONCURRENT WEIGHT GRADIENTS (fused chains, cfg.overlap): the weight-gradient GEMMs of an evaluation start on a side
            # stream as soon as its backward chain is done and run beside the next evaluation's chains, each kernel family on its
            # share of the CUs (cfg.chain_wgs / cfg.wgrad_wgs: a chain workgroup and a 256-wide weight-gradient workgroup fill a
            # CU each, so the two launches occupy disjoint CUs) - the chains are bound by matrix-instruction issue and their T
            # stores, the weight gradients by their operand reads.  Level 0, whose gradient depends on nothing of level 1, goes
            # first so that its weight gradients have the env-light and level-1 chains to run under; the second-order trunk rows
            # of level 1 (r_l^T hdot_{l-1}: complete after the tangent sweep) run under its backward chain; only its first-order
            # products are left for the end, on the whole chip.  All reductions into flat_grad happen on the side stream, in a
            # fixed order.
            conc = bool(cfg.planes) and bool(cfg.overlap)
            side = _side_stream(dev) if conc else None
            cw, ww = (cfg.chain_wgs, cfg.wgrad_wgs) if conc else (0, 0)
            held = []  # workspaces of the side-stream jobs (kept until the join)

            def wg(evals, which=3, wgs=0):
                if not conc:
                    _chain_wgrad(evals, nc, cfg.planes, flat_grad, st, which, wgs)
                    return
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    held.append(_chain_wgrad(evals, nc, cfg.planes, flat_grad, side.cuda_stream, which, wgs))

            def level0():
                d_rr0, d_rd0 = z(M, 3), z(M, nc)
                _composite_backward(e0, B, N, cfg, cfg.white_bkgd, d, B, gz(g_comp0, B, 3), gz(g_dist0, B), None, d_rr0,
                                    d_rd0, st)
                return d_rr0, d_rd0

            pending = []  # (evaluation, its workspace, ran the tangent sweep): weight gradients batched into the last call
            if conc:
                d_rr0, d_rd0 = level0()
                _chain_backward(e0, cfg, params, wpack, d_rr0, d_rd0, None, None, st)  # (alone on the chip: every CU)
                wg([e0], 3, ww)
            d_dist1 = gz(g_dist1, B).clone() if g_dist1 is not None else z(B)  # (accumulated into below: a private buffer)
            d_normal = gz(g_normal, B, 3) if cfg.normals else None
            d_albedo = None
            if cfg.surf:
                D, Ne = env_d.shape[0], cfg.num_env_samples
                if g_surface is not None and g_diffuse is not None:
                    d_dif = _f32(g_surface) + _f32(g_diffuse)
                else:  # (the training loss reads surface_rgb only: no add)
                    d_dif = gz(g_surface if g_surface is not None else g_diffuse, B, 3)
                d_shd = gz(g_shading, B, 3)
                d_env, d_alb_s, d_nrm_s = z(B, D, 3), z(B, 3), z(B, 3)
                _lib.call("pn_surface_backward", B, D, env_rgb.data_ptr(), albedo.data_ptr(), normal.data_ptr(),
                          env_d.data_ptr(), env_omega.data_ptr(), d_dif.data_ptr(), d_shd.data_ptr(), d_env.data_ptr(),
                          d_alb_s.data_ptr(), d_nrm_s.data_ptr(), st)
                d_albedo = gz(g_albedo, B, 3) + d_alb_s
                d_normal = d_normal + d_nrm_s
                d_rr, d_rd = z(ee.M, 3), z(ee.M, nc)
                _composite_backward(ee, B * D, Ne, cfg, False, env_d, D, d_env, None, None, d_rr, d_rd, st)
                d_mean_e = z(ee.M, 3)
                if cfg.planes:
                    _chain_backward(ee, cfg, params, wpack, d_rr, d_rd, None, d_mean_e, st, cw)
                    if conc:
                        wg([ee], 3, ww)
                    else:
                        pending.append(ee)
                else:
                    pending.append((ee, _mlp_backward(ee, cfg, params, wpack, d_rr, d_rd, None, d_mean_e, flat_grad, st,
                                                      defer=cfg.batch_wgrad), False))
                _lib.call("pn_env_origin_backward", B, D * Ne, d_mean_e.data_ptr(), d.data_ptr(), d_dist1.data_ptr(), st)
            d_rr, d_rd = z(M, 3), z(M, nc)
            d_w1 = v = None
            if cfg.normals:
                d_w1, v = z(B, N), z(M, 3)
                d_ort_ray = None
                if cfg.use_ort and g_ort is not None:
                    d_ort_ray = (_f32(g_ort) / B).expand(B).contiguous()
                _lib.call("pn_surf_gather_backward", B, N, nc, e1.gmean.data_ptr(), w1.data_ptr(),
                          e1.raw_den.data_ptr(), d.data_ptr(), d_normal.data_ptr(), _lib.ptr(d_ort_ray),
                          _lib.ptr(d_albedo), d_w1.data_ptr(), v.data_ptr(), d_rd.data_ptr(), st)
            _composite_backward(e1, B, N, cfg, cfg.white_bkgd, d, B, gz(g_comp1, B, 3), d_dist1, d_w1, d_rr, d_rd, st)
            if cfg.planes:
                e1.sdot = None
                if conc and v is not None:
                    _chain_tangent(e1, cfg, params, wpack, v, st, cw)
                    wg([e1], 2, ww)  # second-order trunk rows, under the backward chain
                _chain_backward(e1, cfg, params, wpack, d_rr, d_rd, v, None, st, cw)
                if conc:
                    wg([e1], 1 if v is not None else 3, 0)  # what is left runs alone: every CU
                else:
                    pending.append(e1)
            else:
                pending.append((e1, _mlp_backward(e1, cfg, params, wpack, d_rr, d_rd, v, None, flat_grad, st,
                                                  defer=cfg.batch_wgrad), v is not None))
                if not cfg.batch_wgrad:
                    pending = []
            if conc:
                main.wait_stream(side)
            else:
                d_rr0, d_rd0 = level0()
                if cfg.planes:
                    _chain_backward(e0, cfg, params, wpack, d_rr0, d_rd0, None, None, st)
                    pending.append(e0)
                    _chain_wgrad(pending, nc, cfg.planes, flat_grad, st)
                else:
                    _mlp_backward(e0, cfg, params, wpack, d_rr0, d_rd0, None, None, flat_grad, st, deferred=pending)
            pending = []
            held = []
        mlp.last_flat_grad = flat_grad
        ctx.pack = None
        if getattr(mlp, "defer_param_grads", False):
            # concurrent_step collects `last_flat_grad` per sub-batch and sets the parameters' .grad once at the end;
            # handing views to AccumulateGrad here would make autograd synchronise the sub-batch streams
            return (None,) * (16 + len(list(mlp.named_in_order())))
        return (None,) * 16 + tuple(mlp.grad_views(flat_grad))


class _RenderBase(torch.nn.Module):
    """Constructor keywords of models/pano_mip_nerf.py:120-151 / models/mip_nerf.py:108-134."""

    _NC = 5

    def __init__(self, num_samples=128, num_levels=2, resample_padding=0.01, stop_resample_grad=True,
                 use_viewdirs=True, disparity=False, ray_shape="cone", min_deg_point=0, max_deg_point=16, deg_view=4,
                 density_activation="softplus", density_noise=0.0, density_bias=-1.0, rgb_activation="sigmoid",
                 alb_activation="sigmoid", rgb_padding=0.001, disable_integration=False, append_identity=True,
                 mlp_net_depth=8, mlp_net_width=256, mlp_net_depth_condition=1, mlp_net_width_condition=128,
                 mlp_skip_index=4, mlp_num_rgb_channels=3, mlp_num_density_channels=1, mlp_net_activation="relu",
                 solid_angle_height=8, solid_angle_width=16, num_env_samples=10, **kwargs):
        super().__init__()
        # same error behaviour as the reference for the strings it rejects
        if rgb_activation != "softplus":
            raise NotImplementedError  # models/pano_mip_nerf.py:174-177
        if self._NC == 5 and alb_activation != "sigmoid":
            raise NotImplementedError  # :178-181
        if density_activation != "softplus":
            raise NotImplementedError  # :183-186
        if ray_shape == "cylinder":
            raise NotImplementedError  # models/mip.py:83-84
        assert ray_shape == "cone"  # models/mip.py:86
        unsupported = []
        if num_levels != 2: unsupported.append("num_levels != 2")
        if not stop_resample_grad: unsupported.append("stop_resample_grad=False")
        # (upstream itself cannot run use_viewdirs=False: its MLP then feeds the 256-wide trunk output to the 128-input colour layer,
        # models/pano_mip_nerf.py:99-113 - "mat1 and mat2 shapes cannot be multiplied (32x256 and 128x3)")
        if not use_viewdirs: unsupported.append("use_viewdirs=False")
        if density_noise and density_noise > 0: unsupported.append("density_noise > 0")
        if (min_deg_point, max_deg_point, deg_view) != (0, 16, 4): unsupported.append("encoding degrees != (0,16,4)")
        if not append_identity: unsupported.append("append_identity=False")
        if num_samples > 512 or num_samples < 2: unsupported.append("num_samples outside [2, 512]")
        if unsupported:
            raise NotImplementedError("pano_nerf_amd HIP path supports the configurations of configs/*.yaml only: "
                                      + ", ".join(unsupported))
        self.num_samples, self.num_levels = int(num_samples), int(num_levels)
        self.resample_padding = float(resample_padding)
        self.disparity = bool(disparity)  # coarse samples linear in inverse depth (models/mip.py:134-136)
        self.disable_integration = bool(disable_integration)  # PE instead of IPE: zero covariance in every encoding (:241-243)
        self.density_bias, self.rgb_padding = float(density_bias), float(rgb_padding)
        self.num_env_samples = int(num_env_samples)
        self.mlp = RadianceMLP(mlp_net_depth, mlp_net_width, mlp_net_depth_condition, mlp_net_width_condition,
                               mlp_skip_index, mlp_num_rgb_channels, mlp_num_density_channels, mlp_net_activation,
                               (max_deg_point - min_deg_point) * 6, deg_view * 6 + 3)
        if self.mlp.num_density_channels != self._NC:
            raise NotImplementedError(f"{type(self).__name__} needs mlp_num_density_channels={self._NC}")
        self.noise_override = None  # tests: dict(t_rand=[B,S], u_rand=[B,S], env_rand=[1,Ne+1])
        # weight-gradient GEMMs on a side stream: same results, different launch order.  Off by default: with the
        # weight gradients of the three evaluations batched (below) the chip is power-limited either way and the
        # time-shared run measured 2 % SLOWER (62.4 k vs 61.0 k rays/s at 4096 rays, 50.8 k vs 49.4 k at 512)
        self.overlap_weight_grads = False
        self.cache_env_rays = True          # see invalidate_env_cache
        # see _replayed.  Off by default: measured on the bench's 512 x 1024 panorama in 512-ray chunks (the reference's
        # validation loop) the eager launches are already GPU-bound - 0.884 s per panorama against 0.733 s in 32 768-ray chunks -
        # and the replayed loop, which adds six input copies and ten output clones per chunk, took 0.926 s
        self.replay_inference = False
        self.replay_inference_max_rays = 4096
        self._replays = {}
        self._env_key = self._env_f32 = self._env_src = None
        # fused chains with overlap_weight_grads: workgroups (= CUs) the chains / the weight-gradient jobs may occupy while they
        # run side by side (0 = no limit: the two families then take turns on whole-chip launches)
        self.overlap_chain_wgs = 112
        self.overlap_wgrad_wgs = 144
        self.batch_weight_grads = True    # one weight-gradient GEMM per layer over env + level-1 + level-0 rows
        # MLP arithmetic / kernel family.  "fused_f16x2" (default) = on-chip chains, every fp32 operand as an fp16 pair
        # x 2^e = h + l (|error| < 2^-24 |x|; one power-of-two scale per weight matrix and per sample / tensor), three
        # partial products, fp32 accumulate; "fused" = the same kernels with the exact 3-term bf16 split (six partial
        # products); "fused_bf16" = plain bf16 operands (BASELINE configs[1]); "layerwise" = one exact-fp32 MFMA GEMM
        # per layer (round-1 path).  PN_MLP_MODE overrides the default.
        import os
        self.mlp_mode = os.environ.get("PN_MLP_MODE", "fused_f16x2")

    def _noise(self, randomized, B, dev, want_env):
        if not randomized:
            return None, None, None
        S = self.num_samples + 1
        ov = self.noise_override
        if ov is not None:
            g = lambda k: _f32(ov[k].to(dev)) if ov.get(k) is not None else None
            return g("t_rand"), g("u_rand"), (g("env_rand") if want_env else None)
        # same draws, same order and shapes as the reference: rand(B,S), uniform_(B,S), rand(1,Ne+1)
        t_rand = torch.rand(B, S, device=dev)
        u_rand = torch.empty(B, S, device=dev).uniform_(to=1.0 / S - torch.finfo(torch.float32).eps)
        env_rand = torch.rand(1, self.num_env_samples + 1, device=dev) if want_env else None
        return t_rand, u_rand, env_rand

    def invalidate_env_cache(self):
        """Forget the cached fp32 copies of the caller's env rays (and the captured inference graphs that read them).  The
        cache is keyed on the source tensors' address, version counter, dtype and shape: call this after a write that bypasses
        the version counters (`.data.copy_`, a numpy alias of a CPU tensor, a raw device write) - or set `cache_env_rays =
        False` to convert on every call."""
        self._env_key = self._env_f32 = self._env_src = None
        self._replays = {}

    def _env_inputs(self, env_rays, surf, dev):
        if env_rays is None or not surf:
            return [torch.zeros(1, 3, device=dev)] + [torch.zeros(1, device=dev)] * 4, None
        # the caller's env rays are the same fp16 tensors every step (systems/base_system.py:57-79): their fp32 device
        # copies are cached on the tensors' identity and version (five conversion launches per step otherwise)
        src = (env_rays.directions, env_rays.radii, env_rays.near, env_rays.far, env_rays.lossmult)
        conv = lambda: [_f32(src[0].to(dev))] + [_f32(x.to(dev)).reshape(-1) for x in src[1:]]
        # no cache while a HIP graph is being captured by the CALLER: tensors made during a capture live in the graph's private
        # pool and hold garbage until the first replay - later eager forwards must not find them here
        if not self.cache_env_rays or (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            return conv(), None
        key = tuple((x.data_ptr(), x._version, x.dtype, tuple(x.shape)) for x in src) + (str(dev),)
        if getattr(self, "_env_key", None) != key:
            self._env_f32 = conv()
            self._env_key = key
            self._env_src = src  # keeps the sources alive: their addresses cannot be handed to other tensors meanwhile
        return self._env_f32, key

    def _run(self, rays, env_rays, randomized, white_bkgd, surf, use_ort, normals):
        o, d, vd = _f32(rays.origins), _f32(rays.directions), _f32(rays.viewdirs)
        radii, near, far = _f32(rays.radii).reshape(-1), _f32(rays.near).reshape(-1), _f32(rays.far).reshape(-1)
        dev = o.device
        env, env_key = self._env_inputs(env_rays, surf, dev)
        t_rand, u_rand, env_rand = self._noise(randomized, o.shape[0], dev, surf)
        cfg = _Cfg(num_samples=self.num_samples, nc=self._NC, density_bias=self.density_bias,
                   rgb_padding=self.rgb_padding, resample_padding=self.resample_padding, disparity=self.disparity,
                   disable_integration=self.disable_integration,
                   white_bkgd=bool(white_bkgd), surf=bool(surf), use_ort=bool(use_ort), normals=bool(normals),
                   num_env_samples=self.num_env_samples, overlap=self.overlap_weight_grads, planes=_planes_of(self.mlp_mode),
                   tfmt=_tfmt_of(self.mlp_mode), chain_wgs=int(self.overlap_chain_wgs), wgrad_wgs=int(self.overlap_wgrad_wgs),
                   batch_wgrad=self.batch_weight_grads,
                   keep=torch.is_grad_enabled() and any(p.requires_grad for p in self.mlp.parameters()))
        plist = [p for _, p in self.mlp.named_in_order()]
        if (self.replay_inference and not cfg.keep and not randomized and dev.type == "cuda"
                and 0 < o.shape[0] <= self.replay_inference_max_rays and (env_key is not None or not cfg.surf)
                and not getattr(self.mlp, "debug_keep", False) and not torch.cuda.is_current_stream_capturing()):
            return self._replayed(cfg, (o, d, vd, radii, near, far), env, env_key, plist), cfg
        outs = _RenderFn.apply(cfg, self.mlp, o, d, vd, radii, near, far, *env, t_rand, u_rand,
                               None if env_rand is None else env_rand.reshape(-1), *plist)
        return outs, cfg

    def _replayed(self, cfg, inputs, env, env_key, plist):
        """A no-grad, non-randomized forward of a small ray chunk - the reference's validation loop calls the model 1 024 times
        per 512 x 1024 panorama with 512-ray chunks (systems/panonerf_system.py:133-192, configs/panonerf.yaml:22) - is ~70
        launches of a few microseconds each: launch-bound.  The launch sequence of such a call is captured ONCE per (chunk
        size, flags, kernel mode, stream) into a HIP graph over static input buffers and replayed for every later chunk: copy
        the chunk's rays in, one graph launch, clone the outputs out.  The captured sequence re-packs the weights from the live
        parameter block like every forward, so an optimizer step between two calls is seen; results are bit-identical to the
        eager call's (same kernels, same order)."""
        dev = inputs[0].device
        stream = torch.cuda.current_stream(dev)
        # everything the captured launch sequence depends on: the chunk size, every scalar of the call's configuration, the env
        # rays' identity, the parameter block's address, the stream (two streams must not share static buffers) and whether
        # the caller froze the weight packs (renderer.render_image with several streams: the capture then holds no re-pack)
        key = (inputs[0].shape[0], tuple(sorted(cfg.__dict__.items())), env_key, self.mlp.flat_params().data_ptr(),
               stream.cuda_stream, str(dev), bool(self.mlp._frozen))
        ent = self._replays.get(key)
        if ent is None:
            if len(self._replays) >= 8:  # a few chunk sizes at most (the full chunks and the ragged last one)
                self._replays.pop(next(iter(self._replays)))
            static = [torch.empty_like(x) for x in inputs]
            for sx, x in zip(static, inputs):
                sx.copy_(x)
            run = lambda: _RenderFn.apply(cfg, self.mlp, *static, *env, None, None, None, *plist)
            with torch.no_grad():
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(stream)
                with torch.cuda.stream(side):
                    run()  # warm-up outside the capture (allocator, lazily set kernel attributes)
                stream.wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    outs = run()
            ent = (graph, static, outs)
            self._replays[key] = ent
        graph, static, outs = ent
        for sx, x in zip(static, inputs):
            sx.copy_(x)
        graph.replay()
        # (ort_loss and the absent outputs are zero-dim placeholders: cloned like the rest - the caller may keep them)
        return tuple(x.clone() for x in outs)


class PanoMipNeRF(_RenderBase):
    """Drop-in for models/pano_mip_nerf.py:117 (forward at :197-363)."""

    _NC = 5

    def forward(self, rays: Rays, env_rays: Rays, randomized: bool, white_bkgd: bool, enable_surf: bool,
                use_ort_loss: bool):
        outs, cfg = self._run(rays, env_rays, randomized, white_bkgd, enable_surf, use_ort_loss, True)
        comp0, dist0, comp1, dist1, ort, normal, albedo, surface, diffuse, shading = outs
        lvl0 = (comp0, dist0, None, None, None, None, None, None, None)
        if not cfg.surf:
            albedo = surface = diffuse = shading = None
        lvl1 = (comp1, dist1, ort if cfg.use_ort else None, normal, albedo, None, surface, diffuse, shading)
        return [lvl0, lvl1]


class MipNeRF(_RenderBase):
    """Drop-in for models/mip_nerf.py:105 (forward at :170-283)."""

    _NC = 1

    def forward(self, rays: Rays, randomized: bool, white_bkgd: bool, use_ort_loss: bool):
        outs, cfg = self._run(rays, None, randomized, white_bkgd, False, use_ort_loss, bool(use_ort_loss))
        comp0, dist0, comp1, dist1, ort, normal = outs[:6]
        lvl0 = (comp0, dist0, None, torch.ones_like(comp0))
        if use_ort_loss:
            lvl1 = (comp1, dist1, ort, normal)
        else:
            lvl1 = (comp1, dist1, None, torch.ones_like(comp1))
        return [lvl0, lvl1]
