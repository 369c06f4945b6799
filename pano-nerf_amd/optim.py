"""Optimizer step either side of the hot path.

``FlatAdam`` = torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8) (systems/base_system.py:82) as ONE HIP
kernel over the flat parameter block; ``mip_lr`` = MipLRDecay.get_lr (utils/lr_schedule.py:51-59).
"""
import math

import torch

from . import _lib


def mip_lr(step, lr_init=2e-4, lr_final=2e-5, max_steps=44000, lr_delay_steps=120, lr_delay_mult=0.01):
    rate = 1.0
    if lr_delay_steps > 0:
        rate = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
    t = min(max(step / max_steps, 0.0), 1.0)
    return rate * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


class FlatAdam:
    def __init__(self, mlp, lr=2e-4, betas=(0.9, 0.999), eps=1e-8):
        self.mlp, self.lr, self.betas, self.eps = mlp, lr, betas, eps
        flat = mlp.flat_params()
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.step_count = 0

    def step(self, flat_grad=None, grad_scale=1.0, lr=None):
        flat = self.mlp.flat_params()
        g = flat_grad if flat_grad is not None else self.mlp.last_flat_grad
        if g is None:
            raise RuntimeError("no gradient: run backward first")
        if self.exp_avg.device != flat.device:
            self.exp_avg, self.exp_avg_sq = self.exp_avg.to(flat.device), self.exp_avg_sq.to(flat.device)
        self.step_count += 1
        with torch.cuda.device(flat.device):
            _lib.call("pn_adam_step", flat.numel(), flat.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(),
                      self.exp_avg_sq.data_ptr(), float(self.lr if lr is None else lr), float(self.betas[0]),
                      float(self.betas[1]), float(self.eps), int(self.step_count), float(grad_scale),
                      torch.cuda.current_stream(flat.device).cuda_stream)
        self.mlp.note_raw_write()

    def step_dev(self, flat_grad, lr_dev, grad_scale=1.0):
        """Graph-replay friendly step: learning rate read from the 1-element device tensor `lr_dev`, step counter
        kept (and incremented) on the device."""
        flat = self.mlp.flat_params()
        if not hasattr(self, "step_dev_t") or self.step_dev_t.device != flat.device:
            self.step_dev_t = torch.full((1,), self.step_count, dtype=torch.int32, device=flat.device)
        if self.exp_avg.device != flat.device:
            self.exp_avg, self.exp_avg_sq = self.exp_avg.to(flat.device), self.exp_avg_sq.to(flat.device)
        self.step_count += 1
        with torch.cuda.device(flat.device):
            _lib.call("pn_adam_step_dev", flat.numel(), flat.data_ptr(), flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                      self.exp_avg_sq.data_ptr(), lr_dev.data_ptr(), float(self.betas[0]), float(self.betas[1]),
                      float(self.eps), self.step_dev_t.data_ptr(), float(grad_scale),
                      torch.cuda.current_stream(flat.device).cuda_stream)
        self.mlp.note_raw_write()

    def zero_grad(self):
        for p in self.mlp.parameters():
            p.grad = None
        self.mlp.last_flat_grad = None
