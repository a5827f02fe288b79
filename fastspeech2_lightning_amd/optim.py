"""Fused AdamW + Noam schedule + global-norm clipping over the flat parameter buffer.

reference ``fs2/model.py:530-549`` (``torch.optim.AdamW`` + ``NoamLR`` stepped every optimizer step),
``fs2/noam.py:20-26`` and ``gradient_clip_val=1.0`` of ``fs2/cli/train.py:38``.  The step counter,
learning rate, bias corrections and clip coefficient live in a 32-byte device record that the
kernels advance themselves, so a whole training step can be replayed from a hipGraph.
"""
from __future__ import annotations

import torch

from . import hip as H


def noam_scale(step: int, warmup: int) -> float:
    s = max(1, step)
    return warmup ** 0.5 * min(s ** -0.5, s * warmup ** -1.5)


class FusedAdamWNoam:
    def __init__(self, store, step_state, lr, betas, eps, weight_decay, warmup_steps, max_grad_norm=1.0):
        self.store, self.state = store, step_state
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(betas), float(eps), float(weight_decay)
        self.warmup_steps, self.max_grad_norm = int(warmup_steps), float(max_grad_norm)
        self.grad_scale = 1.0  # 1/world_size under data parallelism (gradients are summed by the all-reduce)

    def step(self):
        """One optimizer step: schedule advance, clip coefficient, parameter update (3 launches + 1 finish)."""
        S = self.store
        with torch.cuda.device(S.flat.device):
            H.step_advance(self.state, self.lr, self.warmup_steps, self.betas[0], self.betas[1])
            H.grad_clip_coef(S.grad, self.max_grad_norm, self.grad_scale, self.state)
            H.adamw_step(S.flat, S.grad, S.adam_m, S.adam_v, self.state, self.betas[0], self.betas[1], self.eps,
                         self.weight_decay)

    def zero_grad(self, set_to_none=False):
        pass  # every gradient element is overwritten by the next backward pass

    # host-side views of the device record (each is a D2H copy: logging only)
    def record(self):
        r = self.state.cpu()
        f = r.view(torch.float32)
        return {"step": int(r[0]), "lr": float(f[2]), "clip_coef": float(f[5]), "grad_norm": float(f[6])}

    def get_last_lr(self):
        return [self.record()["lr"]]

    def state_dict(self):
        S = self.store
        return {"step_state": self.state.cpu(), "adam_m": S.adam_m.cpu(), "adam_v": S.adam_v.cpu(),
                "hyper": dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay,
                              warmup_steps=self.warmup_steps, max_grad_norm=self.max_grad_norm)}

    def load_state_dict(self, sd):
        S = self.store
        self.state.copy_(sd["step_state"])
        S.adam_m.copy_(sd["adam_m"])
        S.adam_v.copy_(sd["adam_v"])
