"""Fused AdamW + Noam schedule + global-norm clipping over the flat parameter buffer, behind
``torch.optim``'s own interfaces.

reference ``fs2/model.py:530-549`` (``torch.optim.AdamW`` + ``NoamLR`` stepped every optimizer step),
``fs2/noam.py:1-26`` and ``gradient_clip_val=1.0`` of ``fs2/cli/train.py:38``.

* ``FusedAdamWNoam`` IS a ``torch.optim.Optimizer`` (one parameter group holding the model's flat
  ``nn.Parameter``), so Lightning's automatic optimization -- ``optimizer.step(closure)``,
  ``optimizer.zero_grad()``, ``clip_grad_norm_(parameters)``, ``optimizer.state_dict()`` into the
  checkpoint -- runs unchanged; ``step()`` is three launches over the flat buffers.
* ``NoamLR`` IS a ``torch.optim.lr_scheduler.LRScheduler`` with the reference's constructor and state dict.
  The step counter, learning rate, bias corrections and clip coefficient that the kernels use live in a
  32-byte device record that ``step()`` advances itself (a whole training step can be replayed from a
  hipGraph); the scheduler object is the host-side mirror of that schedule -- what ``LearningRateMonitor``
  reads and what goes into ``lr_schedulers`` of a checkpoint -- and costs no launch.
* ``optimizer.state_dict()`` is ``torch.optim.AdamW``'s for the reference's ``named_parameters()``
  (per-parameter ``exp_avg`` / ``exp_avg_sq`` in the reference's layouts), so a checkpoint Lightning writes
  for this model loads in the reference and vice versa.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import hip as H


def noam_scale(step: int, warmup: int) -> float:
    s = max(1, step)
    return warmup ** 0.5 * min(s ** -0.5, s * warmup ** -1.5)


class FusedAdamWNoam(torch.optim.Optimizer):
    def __init__(self, store, step_state, lr, betas, eps, weight_decay, warmup_steps, max_grad_norm=None,
                 param: Optional[torch.nn.Parameter] = None, param_names: Optional[list] = None):
        """``max_grad_norm``: global-norm clip fused into ``step()`` (None / 0: no clipping here -- as in the reference,
        where clipping is the trainer's ``gradient_clip_val``; ``FastSpeech2.configure_gradient_clipping`` is the hook
        that hands that value over).  ``param``: the model's flat ``nn.Parameter`` (aliases ``store.flat``)."""
        self.store, self.step_state = store, step_state
        if param is None:
            param = torch.nn.Parameter(store.flat)
        self.param = param
        self.param_names = list(param_names) if param_names is not None else None
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(float(b) for b in betas), float(eps), float(weight_decay)
        self.warmup_steps = int(warmup_steps)
        self.max_grad_norm = max_grad_norm
        self.grad_scale = 1.0  # 1/world_size under data parallelism (gradients are summed by the all-reduce)
        #: optimizer steps since autograd last delivered a gradient (model._DeliverGrad resets it): a ``param.grad`` that
        #: still aliases the flat gradient buffer has been consumed when this is > 0 (FastSpeech2.training_step)
        self.steps_since_delivery = 0
        defaults = dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay, amsgrad=False,
                        maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                        decoupled_weight_decay=True)
        super().__init__([param], defaults)

    # ---- torch.optim.Optimizer interface -------------------------------------------------------------------------
    def step(self, closure=None):
        """One optimizer step: schedule advance, clip coefficient, parameter update (3 launches + 1 finish).
        The gradient is ``param.grad`` when autograd delivered one (``loss.backward()`` on ``training_step``'s result:
        normally an alias of the flat gradient buffer, possibly clipped in place by the caller), else the flat
        gradient buffer the backward pass filled."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        S = self.store
        g = self.param.grad if self.param.grad is not None else S.grad
        with torch.cuda.device(S.flat.device):
            H.step_advance(self.step_state, self.lr, self.warmup_steps, self.betas[0], self.betas[1])
            H.grad_clip_coef(g, float(self.max_grad_norm or 0.0), self.grad_scale, self.step_state)
            H.adamw_step(S.flat, g, S.adam_m, S.adam_v, self.step_state, self.betas[0], self.betas[1], self.eps,
                         self.weight_decay)
        S.weights_changed()  # (the transposed weight mirrors are stale until the next training forward refreshes them)
        self.steps_since_delivery += 1
        return loss

    def zero_grad(self, set_to_none: bool = True):
        """Every gradient element is overwritten by the next backward pass, so there is nothing to clear; dropping
        ``param.grad`` (an alias of the flat buffer) makes the next ``loss.backward()`` deliver instead of accumulate."""
        if set_to_none:
            self.param.grad = None
        elif self.param.grad is not None:
            H.axpby(self.param.grad, None, 0.0, 0.0, out=self.param.grad)

    # host-side views of the device record (each is a D2H copy: logging only)
    def record(self):
        r = self.step_state.cpu()
        f = r.view(torch.float32)
        return {"step": int(r[0]), "lr": float(f[2]), "clip_coef": float(f[5]), "grad_norm": float(f[6])}

    def get_last_lr(self):
        return [self.record()["lr"]]

    def state_dict(self):
        """``torch.optim.AdamW``'s state dict over the reference's parameters when their names are known (the model
        passes them), else the native form (flat moments + the device record)."""
        if self.param_names is not None:
            return self.torch_state_dict(self.param_names)
        return self.native_state_dict()

    def native_state_dict(self):
        S = self.store
        return {"step_state": self.step_state.cpu(), "adam_m": S.adam_m.cpu(), "adam_v": S.adam_v.cpu(),
                "hyper": dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay,
                              warmup_steps=self.warmup_steps, max_grad_norm=self.max_grad_norm)}

    def load_state_dict(self, sd):
        """Takes either form.  Native: the step record and both moment buffers; sizes must match; hyper-parameters
        stay those of the current config, as in ``torch.optim.Optimizer.load_state_dict`` followed by Lightning
        re-applying the configured schedule -- a mismatch is reported, not silently taken."""
        if "step_state" not in sd:
            if self.param_names is None:
                raise RuntimeError("a torch.optim.AdamW state dict needs the reference's parameter names "
                                   "(FastSpeech2.configure_optimizers passes them)")
            return self.load_torch_state_dict(sd, self.param_names)
        S = self.store
        for k, t in (("adam_m", S.adam_m), ("adam_v", S.adam_v)):
            if sd[k].numel() != t.numel():
                raise RuntimeError(f"optimizer state {k}: {sd[k].numel()} elements, the model has {t.numel()}")
        self.step_state.copy_(sd["step_state"])
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
    def steps_done(self) -> int:
        return int(self.step_state.cpu()[0])

    def torch_state_dict(self, param_names: list) -> dict:
        """``torch.optim.AdamW(model.parameters()).state_dict()`` as the reference would hold it after the same
        steps.  ``param_names``: the reference's ``named_parameters()`` order (the frozen ``pitch_bins`` /
        ``energy_bins`` parameters are in the list and, like in torch, carry no state)."""
        S = self.store
        step = self.steps_done()
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
        step = self.steps_done()
        lr_now = self.lr * noam_scale(step, self.warmup_steps)
        return {"warmup_steps": self.warmup_steps, "base_lrs": [self.lr], "last_epoch": step, "_step_count": step + 1,
                "_get_lr_called_within_step": False, "_last_lr": [lr_now]}

    def load_torch_state_dict(self, opt_sd: dict, param_names: list, sched_sd: dict = None) -> None:
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
        self.step_state.copy_(rec)


class NoamLR(torch.optim.lr_scheduler.LRScheduler):
    """reference ``fs2/noam.py:4-26``: ``lr = base_lr * warmup^0.5 * min(s^-0.5, s * warmup^-1.5)``, ``s = max(1,
    last_epoch)``, stepped once per optimizer step (``{"scheduler": ..., "interval": "step"}``).  Host arithmetic only:
    the kernels take the same schedule from the optimizer's device record, this object keeps ``param_groups[0]["lr"]``,
    ``get_last_lr()`` and the checkpoint's ``lr_schedulers`` entry in step with it.  Loading a state dict moves the
    device record's step counter too, so the two cannot drift apart across a resume."""

    def __init__(self, optimizer, warmup_steps):
        self.warmup_steps = warmup_steps
        super().__init__(optimizer)

    def get_lr(self):
        return [base_lr * noam_scale(self.last_epoch, self.warmup_steps) for base_lr in self.base_lrs]

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        opt = self.optimizer
        if isinstance(opt, FusedAdamWNoam) and opt.steps_done() != int(self.last_epoch):
            opt.set_step(int(self.last_epoch))
