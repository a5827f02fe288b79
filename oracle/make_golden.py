"""TEST INFRASTRUCTURE -- generates ``tests/golden/*.npz`` by executing the
REFERENCE's own Python (``/root/reference/fs2``) in the build container.

Runs only where ``/root/reference`` exists (never on the GPU box).  The
reference's third-party imports that are absent from this image
(pytorch_lightning, everyvoice, torchaudio, numba, loguru) are replaced by tiny
stand-in modules registered in ``sys.modules`` before import; the reference's
own files (model.py, variance_adaptor.py, layers.py, blocks.py, loss.py,
attn/*.py, noam.py) then run unmodified.  The single non-reference piece on the
numeric path is the Conformer body (torchaudio is un-vendored): the stand-in for
``torchaudio.models.Conformer`` is ``oracle.fs2_oracle.Conformer``.

Nothing from the reference is copied: the fixtures hold seeded inputs, the state
dict and the reference's outputs only.

Usage:  python oracle/make_golden.py        (writes tests/golden/)
"""
from __future__ import annotations

import os
import sys
import types
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
REFERENCE = Path(os.environ.get("FS2_REFERENCE", "/root/reference"))
sys.path.insert(0, str(REPO))

from fastspeech2_lightning_amd import config as cfgmod  # noqa: E402
from oracle import cases as C  # noqa: E402
from oracle import fs2_oracle as O  # noqa: E402

STATS = C.STATS


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent:
        if parent not in sys.modules:
            _mod(parent)
        setattr(sys.modules[parent], child, m)
    return m


def install_stand_ins():
    from torch import nn

    class LightningModule(nn.Module):
        current_epoch = 0
        global_step = 0
        logger = None

        def save_hyperparameters(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    _mod("pytorch_lightning", LightningModule=LightningModule)
    _mod("torchaudio")
    _mod("torchaudio.models", Conformer=O.Conformer)

    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    _mod("loguru", logger=_Logger())
    _mod("numba", jit=lambda *a, **k: (lambda f: f), prange=range)

    _mod("everyvoice")
    _mod("everyvoice.config")
    _mod("everyvoice.config.type_definitions",
         TargetTrainingTextRepresentationLevel=cfgmod.TargetTrainingTextRepresentationLevel)
    _mod("everyvoice.model")
    _mod("everyvoice.model.feature_prediction")
    _mod("everyvoice.model.feature_prediction.config", FeaturePredictionConfig=cfgmod.FastSpeech2Config)
    for name in ("everyvoice.model.vocoder", "everyvoice.model.vocoder.HiFiGAN_iSTFT_lightning",
                 "everyvoice.model.vocoder.HiFiGAN_iSTFT_lightning.hfgl"):
        _mod(name)
    _mod("everyvoice.model.vocoder.HiFiGAN_iSTFT_lightning.hfgl.utils",
         load_hifigan_from_checkpoint=None, synthesize_data=None)
    _mod("everyvoice.text")
    _mod("everyvoice.text.features", N_PHONOLOGICAL_FEATURES=cfgmod.N_PHONOLOGICAL_FEATURES)
    _mod("everyvoice.text.lookups", LookupTable=dict)
    _mod("everyvoice.text.text_processor", TextProcessor=cfgmod.TextProcessor)
    _mod("everyvoice.text.utils", get_symbols_from_checkpoint_symbol_dict=None, symbol_sorter=None)
    _mod("everyvoice.utils", pydantic_validation_error_shortener=str, slugify=lambda s, **k: s)
    _mod("everyvoice.utils.heavy", expand=None)
    _mod("everyvoice.exceptions", BadDataError=cfgmod.BadDataError, InvalidConfiguration=ValueError)
    # --- stand-ins needed to import fs2/dataset.py (batch producer) -----------------------------
    import enum

    class DatasetTextRepresentation(str, enum.Enum):  # value strings of the parent toolkit's enum
        characters = "characters"
        ipa_phones = "phones"
        arpabet = "arpabet"

    sys.modules["everyvoice.config.type_definitions"].DatasetTextRepresentation = DatasetTextRepresentation
    _mod("everyvoice.dataloader", BaseDataModule=object)
    _mod("everyvoice.preprocessor", Preprocessor=None)
    sys.modules["everyvoice.text.lookups"].lookuptables_from_config = None
    ev_utils = sys.modules["everyvoice.utils"]
    ev_utils._flatten = dict  # per-utterance items are flat dicts: flattening is a copy
    ev_utils.check_dataset_size = None
    ev_utils.filter_dataset_based_on_target_text_representation_level = None
    # the reference's fs2/config builds on everyvoice.config base classes that do
    # not exist here: the schema stand-in is the build's own re-declaration.
    sys.path.insert(0, str(REFERENCE))
    import fs2  # noqa: F401  (namespace of the reference)

    _mod("fs2.config", FastSpeech2Config=cfgmod.FastSpeech2Config)
    # PostNet hard-codes F.dropout(x, 0.5, self.training) (fs2/layers.py:207-209); torch's
    # dropout RNG stream cannot be reproduced by another implementation, so golden
    # train-mode cases run with dropout as the identity (all config dropouts are 0 too):
    # what train mode pins is BatchNorm batch statistics + running-stat updates.
    import fs2.layers as ref_layers

    class _F:
        def __getattr__(self, name):
            return getattr(torch.nn.functional, name)

        @staticmethod
        def dropout(x, p=0.5, training=True, inplace=False):
            return x

    ref_layers.F = _F()


def _np(v):
    if v is None:
        return None
    if torch.is_tensor(v):
        return v.detach().cpu().numpy()
    return np.asarray(v)


def dump_case(name: str, config, batch, train_mode: bool, out_dir: Path):
    """Reference forward + loss + backward on ``batch``; save everything."""
    from fs2.model import FastSpeech2 as RefFastSpeech2

    torch.manual_seed(0)
    kw = dict(lang2id=C.LANG2ID, speaker2id=C.SPEAKER2ID) if config.model.multispeaker else {}
    ref = RefFastSpeech2(config, stats=STATS, **kw)
    # parameter values come from a name-keyed seeded generator (regenerated by the
    # tests), so the fixture carries no weights
    ref.load_state_dict(O.seeded_state_dict(ref.state_dict()))
    ref.train(train_mode)
    b = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    b["basename"] = [f"utt{i}" for i in range(b["text"].shape[0])]
    out = ref(b)
    losses = ref.loss(out, b, C.EPOCH)
    losses["total"].backward()
    save = {}
    for k, v in ref.state_dict().items():
        if "running_" in k or "num_batches" in k:
            save["sd_after/" + k] = _np(v)
    for k, v in batch.items():
        save["batch/" + k] = _np(v)
    for k, v in out.items():
        if v is not None:
            save["out/" + k] = _np(v)
    for k, v in losses.items():
        save["loss/" + k] = _np(v)
    for k, p in ref.named_parameters():
        if p.grad is not None:
            a = _np(p.grad)
            save["grad/" + k] = O.subsample(a) if a.size > O.GRAD_SUBSAMPLE_THRESHOLD else a
    np.savez_compressed(out_dir / f"{name}.npz", **save)
    print(f"{name}: {len(save)} arrays, total loss {float(losses['total']):.6f}")


def dump_units(out_dir: Path):
    """Known-answer vectors of the integer/small pieces (SURVEY 8c list)."""
    from fs2.attn.alignment import mas_width1
    from fs2.layers import PositionalEmbedding
    from fs2.noam import NoamLR
    from fs2.utils.heavy import mask_from_lens
    from fs2.variance_adaptor import LengthRegulator, VarianceAdaptor

    g = torch.Generator().manual_seed(1234)
    save = {}
    lens = torch.tensor([3, 7, 1, 5], dtype=torch.int32)
    save["mask/lens"], save["mask/out"] = _np(lens), _np(mask_from_lens(lens, 7))
    pe = PositionalEmbedding(8)
    save["pos/out"] = _np(pe(torch.arange(7).float()))
    # length regulator: zero durations, truncation, all-padding row
    x = torch.randn(4, 6, 5, generator=g)
    dur = torch.tensor([[2, 0, 3, 1, 0, 0], [1, 1, 1, 1, 1, 1], [0, 0, 0, 0, 0, 0], [4, 4, 4, 0, 0, 0]],
                       dtype=torch.int32)
    for tag, ml in (("full", 1000), ("trunc", 7)):
        o, m = LengthRegulator()(x, dur, max_length=ml)
        save[f"lr/{tag}/out"], save[f"lr/{tag}/mask"] = _np(o), _np(m)
    save["lr/x"], save["lr/dur"] = _np(x), _np(dur)
    # bucketize incl. values on edges and outside the range
    bins = torch.linspace(-3, 3, 15)
    v = torch.cat([bins[[0, 3, 14]], torch.tensor([-5.0, 5.0, 0.0]), torch.randn(20, generator=g) * 2])
    save["bucket/bins"], save["bucket/v"], save["bucket/out"] = _np(bins), _np(v), _np(torch.bucketize(v, bins))
    # MAS: random, tie-heavy (quantised), T2 == 2, T1 == T2 (T2 == 1 indexes out of bounds in the
    # reference itself -- alignment.py:68 reads column -2 -- so it has no defined answer)
    cases = {"rand": torch.randn(17, 6, generator=g), "ties": torch.round(torch.randn(23, 9, generator=g)),
             "t2_2": torch.randn(5, 2, generator=g), "square": torch.randn(7, 7, generator=g)}
    for tag, c in cases.items():
        lp = torch.log_softmax(c, dim=1).numpy().astype(np.float32)
        save[f"mas/{tag}/in"], save[f"mas/{tag}/out"] = lp, mas_width1(lp.copy())
    # average_variance
    var = torch.randn(3, 12, generator=g)
    var[0, 2:4] = 0.0
    durs = torch.tensor([[2, 0, 4, 6], [3, 3, 3, 3], [12, 0, 0, 0]], dtype=torch.int32)
    save["avg/var"], save["avg/durs"] = _np(var), _np(durs)
    save["avg/out"] = _np(VarianceAdaptor.average_variance(None, var, durs))
    # Noam schedule
    opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    sch = NoamLR(opt, 40)
    lrs = []
    for _ in range(120):
        lrs.append(sch.get_last_lr()[0])
        opt.step()
        sch.step()
    save["noam/lrs"] = np.asarray(lrs, dtype=np.float64)
    # inference duration rounding (half-to-even) -- fs2/variance_adaptor.py:360-366
    logd = torch.log(torch.tensor([0.5, 1.5, 2.5, 3.5, 1.0, 7.49, 0.2]) + 1)
    save["round/logd"] = _np(logd)
    save["round/out"] = _np(torch.clamp(torch.round(torch.exp(logd) - 1) * 1.0, min=0).int())
    # GST free-inference conditioning on one style token -- fs2/gst/model.py:77-85
    from fs2.gst.model import StyleEncoder as RefStyleEncoder
    ref_gst = RefStyleEncoder(idim=80)
    ref_gst.load_state_dict(O.seeded_state_dict(O.StyleEncoder(idim=80).state_dict()))
    ref_gst.eval()
    with torch.no_grad():
        save["gst_cond/out"] = _np(ref_gst.condition_on_gst_tokens(3, index=2))
    np.savez_compressed(out_dir / "units.npz", **save)
    print(f"units: {len(save)} arrays")


def dump_data(out_dir: Path):
    """Batch producer golden: the reference's ``FastSpeechDataset.__getitem__`` + ``collate_method`` run over
    synthetic per-utterance feature files; the fixture holds the raw features and the collated result."""
    import tempfile

    from fs2.dataset import FastSpeech2DataModule, FastSpeechDataset

    save = {}
    g = torch.Generator().manual_seed(1234)
    symbols = [f"s{i}" for i in range(12)] + ["/"]
    utts = [("utt-a", "spk0", "eng", 7, 23), ("utt-b", "spk1", "fra", 4, 11), ("utt-c", "spk0", "eng", 9, 31)]
    for learn_alignment in (True, False):
        tag = "align" if learn_alignment else "noalign"
        config = C.small_config(learn_alignment=learn_alignment, n_mels=8)
        config.text.symbols = {"letters": symbols}
        with tempfile.TemporaryDirectory() as tmp:
            config.preprocessing.save_dir = tmp
            entries = []
            for bn, spk, lang, n_tok, n_frames in utts:
                toks = [symbols[int(i)] for i in torch.randint(0, len(symbols), (n_tok,), generator=g)]
                escaped = "/".join(t.replace("/", "\\/") for t in toks)
                feats = {
                    ("spec", f"spec-{config.preprocessing.audio.input_sampling_rate}-"
                             f"{config.preprocessing.audio.spec_type}.pt"): torch.randn(8, n_frames, generator=g),
                    ("energy", "energy.pt"): torch.randn(n_tok, generator=g),
                    ("pitch", "pitch.pt"): torch.randn(n_tok, generator=g),
                }
                if learn_alignment:
                    feats[("attn", "characters-attn-prior.pt")] = torch.rand(n_frames, n_tok, generator=g)
                else:
                    d = torch.ones(n_tok, dtype=torch.int64)
                    d[0] += n_frames - n_tok
                    feats[("duration", "duration.pt")] = d
                for (kind, fn), t in feats.items():
                    (Path(tmp) / kind).mkdir(exist_ok=True)
                    torch.save(t, Path(tmp) / kind / "--".join([bn, spk, lang, fn]))
                    save[f"{tag}/in/{bn}/{kind}"] = t.numpy()
                save[f"{tag}/in/{bn}/tokens"] = np.array(escaped)
                entries.append({"basename": bn, "speaker": spk, "language": lang, "character_tokens": escaped,
                                "characters": "".join(toks)})
            ds = FastSpeechDataset(entries, config, {"eng": 0, "fra": 1}, {"spk0": 0, "spk1": 1})
            batch = FastSpeech2DataModule.collate_method([ds[i] for i in range(len(ds))],
                                                         learn_alignment=learn_alignment)
        for k, v in batch.items():
            if torch.is_tensor(v):
                save[f"{tag}/out/{k}"] = v.numpy()
                save[f"{tag}/dtype/{k}"] = np.array(str(v.dtype))
            elif isinstance(v, list) and v and isinstance(v[0], str):
                save[f"{tag}/out/{k}"] = np.array(v)
    np.savez_compressed(out_dir / "data_collate.npz", **save)
    print(f"data: {len(save)} arrays")


CKPT_SEED = 31


def ckpt_case():
    """Config + batch of the checkpoint-interchange fixtures: the small model WITHOUT PostNet (the reference
    hard-codes a 512-channel PostNet = 15 MB of weights; without it a full checkpoint with Adam moments is < 1 MB)."""
    config = C.small_config(learn_alignment=False)
    config.model.use_postnet = False
    config.training.optimizer.learning_rate = 1e-2
    config.training.optimizer.warmup_steps = 2
    batch = O.synthetic_batch(B=3, ts_lo=6, ts_hi=12, n_symbols=C.N_SYMBOLS, n_mels=16, dur_hi=4, seed=CKPT_SEED)
    return config, batch


def reference_train_steps(ref, batch, n, opt=None, sched=None):
    """n optimizer steps the way Lightning drives the reference: loss -> backward -> clip_grad_norm_(1.0)
    (fs2/cli/train.py:38) -> AdamW.step -> NoamLR.step (interval "step", fs2/model.py:530-549)."""
    if opt is None:
        opt, scheds = ref.configure_optimizers()
        opt, sched = opt[0], scheds[0]["scheduler"]
    losses = []
    for _ in range(n):
        opt.zero_grad()
        b = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        out = ref(b)
        loss = ref.loss(out, b, 0)["total"]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        opt.step()
        sched.step()
        losses.append(float(loss))
    return opt, sched, losses


def dump_ckpt(out_dir: Path):
    """A checkpoint WRITTEN BY THE REFERENCE: its own ``on_save_checkpoint`` over the dict Lightning's
    ``Trainer.save_checkpoint`` assembles (weights, ``hyper_parameters`` = the constructor arguments, torch's own
    optimizer / scheduler state dicts), after three training steps; plus what the checkpoint must reproduce."""
    from fs2.model import FastSpeech2 as RefFastSpeech2

    config, batch = ckpt_case()
    torch.manual_seed(0)
    ref = RefFastSpeech2(config, stats=STATS)
    ref.load_state_dict(O.seeded_state_dict(ref.state_dict()))
    ref.train()
    opt, sched, losses = reference_train_steps(ref, batch, 3)
    ckpt = {"epoch": 0, "global_step": 3, "pytorch-lightning_version": "2.6.1",
            "state_dict": ref.state_dict(), "callbacks": {},  # (no "loops": Lightning would index an empty dict for "fit_loop")
            "optimizer_states": [opt.state_dict()], "lr_schedulers": [sched.state_dict()],
            "hyper_parameters": {"config": ref.config, "stats": ref.stats, "lang2id": ref.lang2id,
                                 "speaker2id": ref.speaker2id}}
    ref.on_save_checkpoint(ckpt)   # fs2/model.py:369-378
    torch.save(ckpt, out_dir / "ref_written.ckpt")
    save = {f"batch/{k}": _np(v) for k, v in batch.items()}
    save["losses_1_3"] = np.asarray(losses)
    ref.eval()
    with torch.no_grad():
        out = ref({k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()})
    for k in ("output", "duration_prediction", "pitch_prediction", "energy_prediction"):
        save[f"eval/{k}"] = _np(out[k])
    ref.train()
    _, _, l4 = reference_train_steps(ref, batch, 1, opt, sched)
    save["loss_4"] = np.asarray(l4[0])
    for k, v in ref.state_dict().items():
        if v.dtype.is_floating_point:
            save[f"sd_after_4/{k}"] = _np(v)
    np.savez_compressed(out_dir / "ckpt_interchange.npz", **save)
    print(f"ckpt: losses {losses} then {l4[0]:.6f}; {len(ckpt['optimizer_states'][0]['state'])} parameters with state")


def main():
    out_dir = REPO / "tests" / "golden"
    out_dir.mkdir(parents=True, exist_ok=True)
    install_stand_ins()
    if len(sys.argv) <= 1 or "units" in sys.argv[1:]:
        dump_units(out_dir)
    if len(sys.argv) <= 1 or "data" in sys.argv[1:]:
        dump_data(out_dir)
    if len(sys.argv) <= 1 or "ckpt" in sys.argv[1:]:
        dump_ckpt(out_dir)
    only = set(sys.argv[1:])
    for name in C.CASES:
        if only and name not in only:
            continue
        config, batch, train = C.build(name)
        dump_case(name, config, batch, train, out_dir)


if __name__ == "__main__":
    main()
