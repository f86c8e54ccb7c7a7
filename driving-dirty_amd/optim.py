"""``HipAdam``: torch.optim.Adam semantics (reference autoencoder.py:119-120, roadmap_bce_v2.py:154-157)
executed by ``dd_adam_step``, one fused read-modify-write pass per parameter tensor.

``grad_scale`` folds the 1/world_size of data-parallel gradient averaging into the same pass, so the
all-reduced SUM never needs a separate divide kernel.
"""
import torch

from . import ops


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.adam_step_flat(p.data.view(-1), g.view(-1), st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1),
                                   group["lr"], b1, b2, group["eps"], st["step"], grad_scale)
