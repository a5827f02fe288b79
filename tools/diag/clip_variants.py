"""Diagnostic (GPU): where do the fused clip and torch's in-place clip_grad_norm_ part ways?"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from tests.test_lightning_loop_gpu import _model, automatic_optimization, native_steps  # noqa: E402

ref_model, ref_losses = native_steps(5)
for clip in ("hook", "torch"):
    model, batch = _model()
    opt, sched, losses, lrs = automatic_optimization(model, batch, 5, clip)
    sd, sd_ref = model.state_dict(), ref_model.state_dict()
    worst = sorted(((float((sd[k].float() - sd_ref[k].float()).abs().max()), k) for k in sd if sd[k].dtype.is_floating_point), reverse=True)[:6]
    print(clip, [f"{a:.3e}" for a in (losses[-1], ref_losses[-1])], worst)
    g, g_ref = model.store.grad_state_dict(), ref_model.store.grad_state_dict()
    k = worst[0][1]
    if k in g:
        print("  grad of", k, g[k].flatten()[:6].tolist(), g_ref[k].flatten()[:6].tolist(), "record", opt.record())
