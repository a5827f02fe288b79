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
        """Restores the step record and both moment buffers (native layout, written by ``state_dict``).  Sizes must
        match; hyper-parameters stay those of the current config, as in ``torch.optim.Optimizer.load_state_dict``
        followed by Lightning re-applying the configured schedule -- a mismatch is reported, not silently taken."""
        S = self.store
        for k, t in (("adam_m", S.adam_m), ("adam_v", S.adam_v)):
            if sd[k].numel() != t.numel():
                raise RuntimeError(f"optimizer state {k}: {sd[k].numel()} elements, the model has {t.numel()}")
        self.state.copy_(sd["step_state"])
        S.adam_m.copy_(sd["adam_m"])
        S.adam_v.copy_(sd["adam_v"])
        saved = sd.get("hyper") or {}
        diff = {k: (saved[k], getattr(self, k)) for k in saved
                if k in ("lr", "eps", "weight_decay", "warmup_steps") and float(saved[k]) != float(getattr(self, k))}
        if diff:
            import warnings
            warnings.warn(f"optimizer hyper-parameters differ from the checkpoint's (checkpoint, now): {diff}")

    # ---- torch.optim.AdamW / NoamLR state in the reference's own format (Lightning ``optimizer_states`` /
    # ``lr_schedulers``), so that a run can move between the reference and this build in either direction ----------
    def torch_state_dict(self, param_names: list[str]) -> dict:
        """``torch.optim.AdamW(model.parameters()).state_dict()`` as the reference would hold it after the same
        steps.  ``param_names``: the reference's ``named_parameters()`` order (the frozen ``pitch_bins`` /
        ``energy_bins`` parameters are in the list and, like in torch, carry no state)."""
        S = self.store
        step = int(self.state.cpu()[0])
        state = {}
        if step > 0:
            for i, name in enumerate(param_names):
                if name in S.entries:
                    state[i] = {"step": torch.tensor(float(step)), "exp_avg": S.export_flat(S.adam_m, name).cpu(),
                                "exp_avg_sq": S.export_flat(S.adam_v, name).cpu()}
        lr_now = self.lr * noam_scale(step, self.warmup_steps)
        group = {"lr": lr_now, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": True, "initial_lr": self.lr,
                 "params": list(range(len(param_names)))}
        return {"state": state, "param_groups": [group]}

    def torch_scheduler_state_dict(self) -> dict:
        """``NoamLR.state_dict()`` (fs2/noam.py: an ``_LRScheduler``; ``last_epoch`` counts optimizer steps)."""
        step = int(self.state.cpu()[0])
        lr_now = self.lr * noam_scale(step, self.warmup_steps)
        return {"warmup_steps": self.warmup_steps, "base_lrs": [self.lr], "last_epoch": step, "_step_count": step + 1,
                "_get_lr_called_within_step": False, "_last_lr": [lr_now]}

    def load_torch_state_dict(self, opt_sd: dict, param_names: list[str], sched_sd: dict = None) -> None:
        """Takes over a ``torch.optim.AdamW`` state written by the reference (through Lightning): per-parameter
        ``exp_avg`` / ``exp_avg_sq`` go into the flat moment buffers in kernel layout, the common ``step`` (and the
        scheduler's ``last_epoch``) into the device record."""
        S = self.store
        params = opt_sd["param_groups"][0]["params"]
        if len(params) != len(param_names):
            raise RuntimeError(f"optimizer state has {len(params)} parameters, the model has {len(param_names)}")
        S.adam_m.zero_(); S.adam_v.zero_()
        steps = set()
        for idx, name in zip(params, param_names):
            st = opt_sd["state"].get(idx)
            if st is None:
                continue
            if name not in S.entries:
                raise RuntimeError(f"optimizer state for {name!r}, which is not a trainable parameter here")
            S.import_flat(S.adam_m, name, st["exp_avg"])
            S.import_flat(S.adam_v, name, st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise RuntimeError(f"parameters disagree about the step count: {sorted(steps)}")
        step = steps.pop() if steps else 0
        if sched_sd is not None and int(sched_sd.get("last_epoch", step)) != step:
            raise RuntimeError(f"scheduler last_epoch {sched_sd['last_epoch']} != optimizer step {step}")
        self.set_step(step)

    def set_step(self, step: int) -> None:
        """Device record as ``step`` optimizer steps would have left it (lr and bias corrections are recomputed by
        the next ``step_advance``)."""
        rec = H.new_step_state("cpu")
        rec[0] = int(step)
        self.state.copy_(rec)
