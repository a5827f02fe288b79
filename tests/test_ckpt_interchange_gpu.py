"""Checkpoint interchange with the reference, both directions (reference ``fs2/model.py:270-378``).

``tests/golden/ref_written.ckpt`` was written by the reference itself (``oracle/make_golden.py ckpt``: its own
``on_save_checkpoint`` over a Lightning-layout dict with torch's AdamW / NoamLR state, after three training steps).
Here: (1) it loads into the HIP model and reproduces the reference's outputs; (2) training resumes from it -- Adam
moments, step count and schedule included -- and the next step's loss and weights are the reference's; (3) a checkpoint
written HERE after the same three steps has the reference's layout key for key and nearly the same numbers.  The
opposite direction's last leg -- the reference loading a file written here -- runs in the build container
(``tests/test_ckpt_reference_cpu.py``) on ``tests/golden/hip_written.ckpt``, which ``write_hip_checkpoint`` below
produced on the GPU box."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import cases as C
from oracle import fs2_oracle as O

pytestmark = pytest.mark.gpu


def golden_batch(g):
    batch = {}
    for k in g.files:
        if k.startswith("batch/"):
            v = g[k]
            batch[k[6:]] = int(v) if v.ndim == 0 else torch.from_numpy(v)
    return batch


def ckpt_case():
    config = C.small_config(learn_alignment=False)
    config.model.use_postnet = False
    config.training.optimizer.learning_rate = 1e-2
    config.training.optimizer.warmup_steps = 2
    return config


def test_reference_written_checkpoint_loads_and_resumes(golden_dir):
    from fastspeech2_lightning_amd.model import FastSpeech2
    g = np.load(golden_dir / "ckpt_interchange.npz")
    batch = golden_batch(g)
    model, ckpt = FastSpeech2.load_from_checkpoint(golden_dir / "ref_written.ckpt", return_checkpoint=True)
    assert ckpt["model_info"] == {"name": "FastSpeech2", "version": "1.2"} and model.postnet is None
    model.eval()
    out = model(batch)
    for k in ("output", "duration_prediction", "pitch_prediction", "energy_prediction"):
        want = g[f"eval/{k}"]
        assert np.abs(out[k].cpu().numpy() - want).max() < 1e-4 * max(1.0, np.abs(want).max()), k
    # resume: optimizer moments / step / schedule from torch's own state dicts
    opt = model.configure_optimizers()[0][0]
    model.configure_gradient_clipping(opt, 1.0, "norm")  # Trainer(gradient_clip_val=1.0), fs2/cli/train.py:38
    step, epoch = model.restore_training_state(ckpt, opt)
    assert (step, epoch) == (3, 0) and opt.record()["step"] == 3
    before = {k: v.clone() for k, v in model.state_dict().items()}
    model.train()
    loss = float(model.training_step(batch))
    opt.step()
    assert abs(loss - float(g["loss_4"])) < 1e-4 * float(g["loss_4"]), (loss, float(g["loss_4"]))
    rec = opt.record()
    lr4 = 1e-2 * O.noam_scale(3, 2)
    assert rec["step"] == 4 and abs(rec["lr"] - lr4) < 1e-9
    after = model.state_dict()
    num = den = 0.0
    for k in g.files:
        if not k.startswith("sd_after_4/"):
            continue
        name = k[11:]
        want = torch.from_numpy(g[k])
        d_ref, d_got = want - before[name].cpu(), after[name].cpu() - before[name].cpu()
        num += float((d_got - d_ref).pow(2).sum())
        den += float(d_ref.pow(2).sum())
        assert float((d_got - d_ref).abs().max()) <= 2.0 * lr4 * 1.01, name   # an Adam sign flip on a noise-level element
    assert (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5


def three_steps_from_seeded_weights(golden_dir):
    from fastspeech2_lightning_amd.config import Stats
    from fastspeech2_lightning_amd.model import FastSpeech2
    g = np.load(golden_dir / "ckpt_interchange.npz")
    batch = golden_batch(g)
    model = FastSpeech2(ckpt_case(), Stats(**C.STATS))
    model.load_state_dict(O.seeded_state_dict(model.state_dict()))
    model.train()
    opt = model.configure_optimizers()[0][0]
    model.configure_gradient_clipping(opt, 1.0, "norm")  # Trainer(gradient_clip_val=1.0), fs2/cli/train.py:38
    losses = []
    for _ in range(3):
        losses.append(float(model.training_step(batch)))
        opt.step()
    return model, opt, batch, losses, g


def test_checkpoint_written_here_has_the_reference_layout(golden_dir):
    model, opt, batch, losses, g = three_steps_from_seeded_weights(golden_dir)
    assert np.abs(np.asarray(losses) - g["losses_1_3"]).max() < 2e-3 * g["losses_1_3"].max()
    mine = model.checkpoint_dict(global_step=3, epoch=0, optimizer=opt)
    ref = torch.load(golden_dir / "ref_written.ckpt", map_location="cpu", weights_only=False)
    assert set(mine) >= set(ref), set(ref) - set(mine)
    assert list(mine["state_dict"]) == list(ref["state_dict"])           # same keys in the same (registration) order
    for k, v in ref["state_dict"].items():
        assert mine["state_dict"][k].shape == v.shape and mine["state_dict"][k].dtype == v.dtype, k
    assert mine["model_info"] == ref["model_info"] and mine["global_step"] == ref["global_step"]
    assert set(mine["hyper_parameters"]) == set(ref["hyper_parameters"])
    assert mine["hyper_parameters"]["config"]["model"] == ref["hyper_parameters"]["config"]["model"]
    assert mine["hyper_parameters"]["stats"] == ref["hyper_parameters"]["stats"]
    mo, ro = mine["optimizer_states"][0], ref["optimizer_states"][0]
    assert sorted(mo["state"]) == sorted(ro["state"])                     # the frozen bins carry no state on either side
    assert mo["param_groups"][0]["params"] == ro["param_groups"][0]["params"]
    assert set(mo["param_groups"][0]) >= set(ro["param_groups"][0]), set(ro["param_groups"][0]) - set(mo["param_groups"][0])
    for key in ("lr", "betas", "eps", "weight_decay", "initial_lr"):
        a, b = mo["param_groups"][0][key], ro["param_groups"][0][key]
        assert np.allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=1e-6), key
    num = den = 0.0
    for i, st in ro["state"].items():
        assert float(mo["state"][i]["step"]) == float(st["step"]) == 3.0
        for key in ("exp_avg", "exp_avg_sq"):
            assert mo["state"][i][key].shape == st[key].shape, (i, key)
        num += float((mo["state"][i]["exp_avg"] - st["exp_avg"]).pow(2).sum())
        den += float(st["exp_avg"].pow(2).sum())
    assert (num / den) ** 0.5 < 2e-2                                      # same trajectory: the moments agree
    ms, rs = mine["lr_schedulers"][0], ref["lr_schedulers"][0]
    assert set(ms) >= set(rs) - {"_is_initial"} and ms["last_epoch"] == rs["last_epoch"] == 3
    assert np.allclose(ms["_last_lr"], rs["_last_lr"], rtol=1e-6) and ms["base_lrs"] == rs["base_lrs"]


def write_hip_checkpoint(golden_dir, out_dir):
    """Run on the GPU box (``python -m tests.test_ckpt_interchange_gpu``): the file the reference must accept."""
    model, opt, batch, losses, g = three_steps_from_seeded_weights(golden_dir)
    model.save_checkpoint(Path(out_dir) / "hip_written.ckpt", global_step=3, epoch=0, optimizer=opt)
    loss4 = float(model.training_step(batch))
    opt.step()
    sd = model.state_dict()
    probe = {k: sd[k].flatten()[:8].cpu().tolist() for k in ("mel_linear.weight", "text_input_layer.weight")}
    (Path(out_dir) / "hip_written.json").write_text(json.dumps({"losses_1_3": losses, "loss_4": loss4, "after_4": probe}))


if __name__ == "__main__":
    import sys
    root = Path(__file__).resolve().parent
    write_hip_checkpoint(root / "golden", sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
