"""Tone-mapped LDR-supervises-HDR loss.

Restates ``PanoNeRFSystem.training_step`` (systems/panonerf_system.py:15-75) and
``MipNeRFSystem.training_step`` (systems/mipnerf_system.py:22-53): ACES tone map + gamma
(``hdr_to_ldr``, utils/surface_rendering.py:319-344), uint8-truncated ground truth, masked MSE of the
coarse / fine / surface renders, chromaticity loss on the albedo, orientation loss — as ONE fused HIP
pass (``pn_tonemap_loss``) that also produces the gradients of every rendered input.
"""
import torch

from . import _lib

DEFAULT_LOSS = {"loss.coarse_loss_mult": 0.1, "loss.surface_loss": 1.0, "loss.ort_loss": 0.1, "loss.chrom_loss": 0.1}


class _ToneLossFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, weights, gt_hdr, mask, rgb_c, rgb_f, rgb_s, albedo):
        cw, sw, chw = weights
        dev = rgb_c.device
        if dev.type != "cuda":
            raise RuntimeError("pano_nerf_amd loss runs on a HIP device only; there is no CPU fallback")
        B = rgb_c.shape[0]
        f = lambda x: None if x is None else x.detach().float().contiguous()
        gt, mask, rgb_c, rgb_f, rgb_s, albedo = map(f, (gt_hdr[..., :3], mask.reshape(-1), rgb_c, rgb_f, rgb_s, albedo))
        if chw <= 0:
            albedo_in = None
        else:
            albedo_in = albedo
        terms = torch.empty(8, dtype=torch.float32, device=dev)
        work = torch.empty(8 + 8 * ((B + 255) // 256), dtype=torch.float32, device=dev)
        grads = [torch.empty(B, 3, dtype=torch.float32, device=dev) for _ in range(2)]
        g_s = torch.empty(B, 3, dtype=torch.float32, device=dev) if rgb_s is not None else None
        g_a = torch.empty(B, 3, dtype=torch.float32, device=dev) if albedo_in is not None else None
        with torch.cuda.device(dev):
            _lib.call("pn_tonemap_loss", B, gt.data_ptr(), mask.data_ptr(), rgb_c.data_ptr(), rgb_f.data_ptr(),
                      _lib.ptr(rgb_s), _lib.ptr(albedo_in), float(cw), float(sw), float(chw), terms.data_ptr(),
                      grads[0].data_ptr(), grads[1].data_ptr(), _lib.ptr(g_s), _lib.ptr(g_a), work.data_ptr(),
                      torch.cuda.current_stream(dev).cuda_stream)
        total = terms[5].clone()  # cw * terms[0] + terms[1] (+ sw * terms[2]) (+ chw * terms[3]), formed by the kernel
        ctx.grads = (grads[0], grads[1], g_s, g_a)
        ctx.mark_non_differentiable(terms)
        return total, terms

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_total, _g_terms):
        gc, gf, gs, ga = ctx.grads
        live = [g for g in (gc, gf, gs, ga) if g is not None]
        out = iter(torch._foreach_mul(live, g_total))  # ONE launch for the (up to) four [B,3] gradients
        return (None, None, None) + tuple(None if g is None else next(out) for g in (gc, gf, gs, ga))


def pano_loss(outputs, lossmult, rgbs, hparams=DEFAULT_LOSS, surface=True):
    """Total training loss of PanoNeRFSystem.training_step for `outputs = PanoMipNeRF(...)`."""
    (rgb_c, *_), (rgb_f, _, ort, _, alb, _, sf, _, _) = outputs
    use_s = surface and sf is not None
    chw = hparams["loss.chrom_loss"] if use_s else 0.0
    total, terms = _ToneLossFn.apply((hparams["loss.coarse_loss_mult"], hparams["loss.surface_loss"], chw), rgbs,
                                     lossmult, rgb_c, rgb_f, sf if use_s else None, alb if use_s else None)
    if ort is not None:
        total = total + hparams["loss.ort_loss"] * ort
    return total, terms


def mip_loss(outputs, lossmult, rgbs, hparams=DEFAULT_LOSS, use_ort=False):
    """Total training loss of MipNeRFSystem.training_step for `outputs = MipNeRF(...)`."""
    (c, *_), (f, _, ort, _) = outputs
    total, terms = _ToneLossFn.apply((hparams["loss.coarse_loss_mult"], 0.0, 0.0), rgbs, lossmult, c, f, None, None)
    if use_ort:
        total = total + hparams["loss.ort_loss"] * ort
    return total, terms


def hdr_to_ldr_psnr(pred_hdr, gt_hdr):
    """PSNR between tone-mapped images: calc_psnr (utils/metrics.py:231-237) on hdr_to_ldr outputs.
    Evaluation-only helper (tiny tensors): plain torch."""
    def ldr(c):
        c = (c * (2.51 * c + 0.03)) / (c * (2.43 * c + 0.59) + 0.14)
        return torch.clamp(c, 0, 1) ** (1 / 2.2)
    return float(-10.0 * torch.log10(torch.mean((ldr(pred_hdr) - ldr(gt_hdr)) ** 2)))
