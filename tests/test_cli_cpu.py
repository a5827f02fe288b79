"""``fs2l train`` plumbing that needs no GPU: config + ``-c`` overrides, filelists, look-up tables, stats.json, the
run directory and resume detection (``--dry-run`` prints the plan and touches no GPU); checkpoint-layout helpers."""
import json
import subprocess
import sys
from pathlib import Path

import pytest
import torch
import yaml

from fastspeech2_lightning_amd import cli

REPO = Path(__file__).resolve().parent.parent


def make_project(tmp: Path, n_train=5, n_val=2, write_features=False, learn_alignment=False, n_mels=16):
    """A preprocessed-data directory as the reference's preprocessor leaves it: filelists, stats.json, config."""
    sym = [chr(ord("a") + i) for i in range(10)]
    pre = tmp / "preprocessed"
    pre.mkdir()
    (pre / "stats.json").write_text(json.dumps(dict(
        pitch=dict(min=0, max=1, std=1, mean=0, norm_min=-3, norm_max=3),
        energy=dict(min=0, max=1, std=1, mean=0, norm_min=-3, norm_max=3))))
    g = torch.Generator().manual_seed(0)
    rows = []
    for i in range(n_train + n_val):
        n_tok = int(torch.randint(4, 12, (1,), generator=g))
        toks = [sym[int(j)] for j in torch.randint(0, len(sym), (n_tok,), generator=g)]
        spk, lang = ("spk1" if i % 2 else "spk0"), "eng"
        rows.append(dict(basename=f"utt{i:03d}", language=lang, speaker=spk, characters="".join(toks),
                         character_tokens="/".join(toks), phones="", phone_tokens=""))
        if write_features:
            dur = torch.randint(1, 5, (n_tok,), generator=g)
            T = int(dur.sum())
            feats = {("spec", "spec-22050-mel-librosa.pt"): torch.randn(n_mels, T, generator=g),
                     ("energy", "energy.pt"): torch.randn(n_tok, generator=g),
                     ("pitch", "pitch.pt"): torch.randn(n_tok, generator=g),
                     ("duration", "duration.pt"): dur}
            for (kind, fn), t in feats.items():
                (pre / kind).mkdir(exist_ok=True)
                torch.save(t, pre / kind / "--".join([rows[-1]["basename"], spk, lang, fn]))
    for name, part in (("training_filelist.psv", rows[:n_train]), ("validation_filelist.psv", rows[n_train:])):
        with open(pre / name, "w", encoding="utf8") as f:
            f.write("|".join(rows[0]) + "\n")
            for r in part:
                f.write("|".join(r.values()) + "\n")
    d = 32
    conf = dict(layers=1, heads=2, input_dim=d, feedforward_dim=64, conv_kernel_size=9, dropout=0.1)
    vp = dict(n_layers=2, kernel_size=3, dropout=0.1, input_dim=d, n_bins=16)
    cfg = dict(model=dict(encoder=conf, decoder=conf, learn_alignment=learn_alignment,
                          variance_predictors=dict(energy=vp, pitch=vp, duration=vp)),
               training=dict(batch_size=4, max_epochs=100, max_steps=1000, training_filelist="preprocessed/training_filelist.psv",
                             validation_filelist="preprocessed/validation_filelist.psv", train_data_workers=0,
                             optimizer=dict(learning_rate=1e-3, warmup_steps=10),
                             logger=dict(save_dir="logs", name="exp", version="v0")),
               preprocessing=dict(save_dir="preprocessed", audio=dict(n_mels=n_mels)),
               text=dict(symbols=dict(letters=sym)))
    (tmp / "config.yaml").write_text(yaml.safe_dump(cfg))
    return tmp / "config.yaml"


def test_dry_run_resolves_the_whole_plan(tmp_path):
    cfg = make_project(tmp_path)
    r = subprocess.run([sys.executable, str(REPO / "fs2l"), "train", str(cfg), "-c", "training.batch_size=3",
                        "-c", "training.optimizer.warmup_steps=77", "--max-steps", "20", "--dry-run"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    assert plan["train_utterances"] == 5 and plan["validation_utterances"] == 2
    assert plan["speaker2id"] == {"spk0": 0, "spk1": 1} and plan["lang2id"] == {"eng": 0}
    assert plan["batch_size"] == 3 and plan["max_steps"] == 20 and plan["max_epochs"] == 100
    assert plan["monitor"] == "validation/total_loss" and plan["gradient_clip_val"] == 1.0
    assert plan["run_dir"] == str(tmp_path / "logs" / "exp" / "v0") and plan["resume"] is None
    # an existing last.ckpt is picked up as the resume point
    ck = tmp_path / "logs" / "exp" / "v0" / "checkpoints"
    ck.mkdir(parents=True)
    (ck / "last.ckpt").write_bytes(b"")
    args = cli.build_parser().parse_args(["train", str(cfg)])
    assert cli.plan(args)["resume"] == ck / "last.ckpt"
    assert cli.plan(args)["config"].training.optimizer.warmup_steps == 10


def test_overrides_and_filelist_errors(tmp_path):
    raw = cli.apply_overrides({"training": {"batch_size": 16}}, ["training.batch_size=2", "model.learn_alignment=false",
                                                                  "training.logger.name=abc"])
    assert raw == {"training": {"batch_size": 2, "logger": {"name": "abc"}}, "model": {"learn_alignment": False}}
    with pytest.raises(SystemExit):
        cli.apply_overrides({}, ["novalue"])
    bad = tmp_path / "x.psv"
    bad.write_text("a|b\n1|2\n")
    with pytest.raises(ValueError, match="basename"):
        cli.read_filelist(bad)
    cfg = make_project(tmp_path)
    (tmp_path / "preprocessed" / "stats.json").unlink()
    with pytest.raises(FileNotFoundError):
        cli.plan(cli.build_parser().parse_args(["train", str(cfg)]))


def test_pre_1_2_embedding_rows_move_to_the_current_symbol_order():
    """fs2/model.py:313-349: row i of a pre-1.2 table belongs to the i-th symbol of [8 hard-coded initial symbols] +
    sorted(rest); it must land on that symbol's row in the model's table, unknown symbols on the padding row."""
    from fastspeech2_lightning_amd.model import OLD_HARDCODED_SYMBOLS, old_symbol_order, remap_pre_1_2_text_embedding
    sym_cfg = {"letters": ["b", "a", "c"], "punctuation": {"exclamations": ["!"], "big_breaks": ["."]}}
    old = old_symbol_order(["b", "a", "c", "!", "."])
    assert old == list(OLD_HARDCODED_SYMBOLS) + ["!", ".", "a", "b", "c"]
    w_old = torch.arange(len(old) * 2, dtype=torch.float32).view(len(old), 2) + 1
    ckpt = {"hyper_parameters": {"config": {"text": {"symbols": sym_cfg}}}, "state_dict": {"text_input_layer.weight": w_old.clone()}}
    model_symbols = ["\x80", " ", "!", ".", "<BB>", "<EPS>", "<EXCL>", "<QINT>", "<QUOTE>", "<SB>", "a", "b", "c", "z"]
    remap_pre_1_2_text_embedding(ckpt, model_symbols, (len(model_symbols), 2))
    new = ckpt["state_dict"]["text_input_layer.weight"]
    for i, s in enumerate(old):
        assert torch.equal(new[model_symbols.index(s)], w_old[i]), s
    assert torch.equal(new[model_symbols.index("z")], torch.zeros(2))
    small = ["\x80", "a"]
    with pytest.raises(AssertionError):
        remap_pre_1_2_text_embedding({"hyper_parameters": ckpt["hyper_parameters"],
                                      "state_dict": {"text_input_layer.weight": w_old}}, small, (2, 2))
