"""Build-container only (needs ``/root/reference``; skipped on the GPU box): the REFERENCE loads a checkpoint written
by the HIP model -- ``tests/golden/hip_written.ckpt``, produced on an MI355X by
``python -m tests.test_ckpt_interchange_gpu`` after three training steps -- through its own ``on_load_checkpoint``
(fs2/model.py:353-367), strict ``load_state_dict``, ``torch.optim.AdamW.load_state_dict`` and ``NoamLR.load_state_dict``,
and its fourth training step reproduces the loss the HIP run recorded for its own fourth step."""
import json
import os
from pathlib import Path

import numpy as np
import pytest
import torch

REFERENCE = Path(os.environ.get("FS2_REFERENCE", "/root/reference"))
pytestmark = pytest.mark.skipif(not (REFERENCE / "fs2" / "model.py").exists(), reason="the reference checkout is not here")


def test_reference_accepts_and_resumes_a_checkpoint_written_by_the_hip_model(golden_dir):
    from oracle import make_golden as MG
    MG.install_stand_ins()
    from fs2.model import FastSpeech2 as RefFastSpeech2

    ckpt = torch.load(golden_dir / "hip_written.ckpt", map_location="cpu", weights_only=False)
    expect = json.loads((golden_dir / "hip_written.json").read_text())
    hp = ckpt["hyper_parameters"]
    ref = RefFastSpeech2(hp["config"], stats=hp["stats"], lang2id=hp["lang2id"], speaker2id=hp["speaker2id"])
    ref.on_load_checkpoint(ckpt)                       # version / model-type checks + config and stats from the file
    assert ref.config.model.use_postnet is False and ref.config.training.optimizer.warmup_steps == 2
    missing, unexpected = ref.load_state_dict(ckpt["state_dict"], strict=True)
    assert not missing and not unexpected
    (opt,), (sched,) = ref.configure_optimizers()
    opt.load_state_dict(ckpt["optimizer_states"][0])
    sched["scheduler"].load_state_dict(ckpt["lr_schedulers"][0])
    assert sched["scheduler"].last_epoch == 3 and ckpt["global_step"] == 3
    g = np.load(golden_dir / "ckpt_interchange.npz")
    batch = {k[6:]: (int(g[k]) if g[k].ndim == 0 else torch.from_numpy(g[k])) for k in g.files if k.startswith("batch/")}
    ref.train()
    _, _, l4 = MG.reference_train_steps(ref, batch, 1, opt, sched["scheduler"])
    assert abs(l4[0] - expect["loss_4"]) < 1e-4 * expect["loss_4"], (l4[0], expect["loss_4"])
    # and the weights after that step are the HIP run's (Adam moments and step count really came across)
    sd = ref.state_dict()
    for k, probe in expect["after_4"].items():
        got = sd[k].flatten()[:8]
        assert float((got - torch.tensor(probe)).abs().max()) < 2e-4, (k, got.tolist(), probe)
    # the two trajectories are the same run: the reference's own three steps gave the same losses
    assert np.abs(np.asarray(expect["losses_1_3"]) - g["losses_1_3"]).max() < 2e-3 * g["losses_1_3"].max()
