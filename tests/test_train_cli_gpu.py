"""``fs2l train`` on the GPU: 12 optimizer steps on synthetic per-utterance feature files (the reference preprocessor's
on-disk layout), validation on ``validation/total_loss``, ``last.ckpt`` / ``best.ckpt`` in Lightning's layout; then the
same run interrupted after 7 steps (mid-epoch) and resumed must reproduce the uninterrupted run's following steps:
weights, Adam moments, the Noam step, the dropout step counter and the position in the epoch all come back."""
import json

import pytest
import torch

from fastspeech2_lightning_amd import cli
from tests.test_cli_cpu import make_project

pytestmark = pytest.mark.gpu


def run(cfg, out, *extra):
    args = ["train", str(cfg), "--output-dir", str(out), "--log-every", "1", "--devices", "1", *extra]
    assert cli.main(args) == 0
    recs = [json.loads(l) for l in (out / "metrics.jsonl").read_text().splitlines()]
    train = {r["step"]: r for r in recs if "training/total_loss" in r}
    val = [r for r in recs if "validation/total_loss" in r]
    return train, val


def test_train_save_resume_reproduces_the_next_steps(tmp_path):
    cfg = make_project(tmp_path, n_train=10, n_val=3, write_features=True)   # batch 4 -> 3 steps per epoch
    straight, val = run(cfg, tmp_path / "a", "--max-steps", "12")
    assert sorted(straight) == list(range(1, 13))
    assert len(val) == 4 and all(v["validation/total_loss"] > 0 for v in val)   # one validation per epoch
    assert straight[12]["training/total_loss"] < straight[1]["training/total_loss"]   # it trains
    assert abs(straight[5]["lr"] - 1e-3 * 10 ** 0.5 * 4 * 10 ** -1.5) < 1e-9    # Noam: step 5 runs at scale(last_epoch=4)
    ck = torch.load(tmp_path / "a" / "checkpoints" / "last.ckpt", map_location="cpu", weights_only=False)
    assert ck["global_step"] == 12 and ck["epoch"] == 4 and ck["model_info"] == {"name": "FastSpeech2", "version": "1.2"}
    assert set(ck["hyper_parameters"]) >= {"config", "stats", "lang2id", "speaker2id"}
    opt = ck["optimizer_states"][0]
    n_params = len(opt["param_groups"][0]["params"])
    assert len(opt["state"]) == n_params - 2 and ck["lr_schedulers"][0]["last_epoch"] == 12   # pitch/energy bins: no state
    assert (tmp_path / "a" / "checkpoints" / "best.ckpt").exists()

    # interrupted after 7 steps (epoch 2, one batch in), resumed to 12
    first, _ = run(cfg, tmp_path / "b", "--max-steps", "7")
    ck7 = torch.load(tmp_path / "b" / "checkpoints" / "last.ckpt", map_location="cpu", weights_only=False)
    assert ck7["global_step"] == 7 and ck7["epoch"] == 2 and ck7["fs2l_batches_in_epoch"] == 1
    resumed, _ = run(cfg, tmp_path / "b", "--max-steps", "12")     # picks up checkpoints/last.ckpt by itself
    for step in range(1, 8):
        assert first[step]["training/total_loss"] == straight[step]["training/total_loss"], step   # deterministic run
    for step in range(8, 13):
        a, b = straight[step], resumed[step]
        assert abs(a["training/total_loss"] - b["training/total_loss"]) < 1e-6 * a["training/total_loss"], (step, a, b)
        assert abs(a["lr"] - b["lr"]) < 1e-12 and abs(a["grad_norm"] - b["grad_norm"]) < 1e-5 * a["grad_norm"]
    end_a = torch.load(tmp_path / "a" / "checkpoints" / "last.ckpt", map_location="cpu", weights_only=False)
    end_b = torch.load(tmp_path / "b" / "checkpoints" / "last.ckpt", map_location="cpu", weights_only=False)
    for k, v in end_a["state_dict"].items():
        assert torch.allclose(v.float(), end_b["state_dict"][k].float(), rtol=1e-5, atol=1e-7), k
    # a weights-only checkpoint cannot be resumed from (it would silently restart Adam and the warm-up)
    del end_b["optimizer_states"]
    torch.save(end_b, tmp_path / "weights_only.ckpt")
    with pytest.raises(RuntimeError, match="no optimizer state"):
        cli.main(["train", str(cfg), "--output-dir", str(tmp_path / "c"), "--resume", str(tmp_path / "weights_only.ckpt"),
                  "--max-steps", "13", "--devices", "1"])


@pytest.mark.parametrize("kind", ["training", "inference"])
def test_fs2l_benchmark_times_the_forward_pass(tmp_path, capsys, kind):
    """``fs2l benchmark`` (reference fs2/cli/benchmark.py): forward-only timing on a batch of the training filelist."""
    cfg = make_project(tmp_path, n_train=6, n_val=2, write_features=True)
    assert cli.main(["benchmark", str(cfg), "--benchmark-type", kind, "--warmup-reps", "2", "--repetitions", "5"]) == 0
    out = capsys.readouterr().out
    assert f"Average forward pass for {kind} duration after 5 repetitions:" in out and "Standard Deviation" in out
    ms = float(out.split("repetitions:")[1].split("ms")[0])
    assert 0.0 < ms < 1000.0


@pytest.mark.tuned_tiles
def test_tune_tiles_runs_trial_steps_and_hands_back_the_starting_state(tmp_path):
    """``fs2l train --tune-tiles``: the tuner's in-step stage takes dozens of REAL steps on the first batch before training;
    weights, Adam moments, the Noam / dropout step record and the BatchNorm buffers must come back, so the run's own steps
    equal those of a run without the flag (to summation-order rounding: the tile table may differ)."""
    cfg = make_project(tmp_path, n_train=10, n_val=3, write_features=True)
    plain, _ = run(cfg, tmp_path / "a", "--max-steps", "4")
    tuned, _ = run(cfg, tmp_path / "b", "--max-steps", "4", "--tune-tiles")
    recs = [json.loads(l) for l in (tmp_path / "b" / "metrics.jsonl").read_text().splitlines()]
    tune = [r["tune_tiles"] for r in recs if "tune_tiles" in r]
    assert len(tune) == 1 and tune[0]["ms_per_step"] > 0 and tune[0]["changed"] >= 0
    assert sorted(tuned) == [1, 2, 3, 4]
    for step in range(1, 5):
        a, b = plain[step], tuned[step]
        assert abs(a["training/total_loss"] - b["training/total_loss"]) < 2e-5 * a["training/total_loss"], (step, a, b)
        assert a["lr"] == b["lr"]
    ck = torch.load(tmp_path / "b" / "checkpoints" / "last.ckpt", map_location="cpu", weights_only=False)
    assert ck["global_step"] == 4 and ck["lr_schedulers"][0]["last_epoch"] == 4
