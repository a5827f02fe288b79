"""Batch producer / spectrogram writer / validation loop, host side, CPU only.

``tests/golden/data_collate.npz`` was produced by the reference's own ``FastSpeechDataset.__getitem__`` and
``FastSpeech2DataModule.collate_method`` (oracle/make_golden.py::dump_data) over synthetic per-utterance feature
files; the fixture carries those raw features, so the files are re-created here and read back by the build's
``FeatureDataset`` + ``collate``.  Bar: bit-exact (padding and dtype work only).
"""
import hashlib
from pathlib import Path

import numpy as np
import pytest
import torch

from fastspeech2_lightning_amd import data as D
from fastspeech2_lightning_amd.config import FastSpeech2Config, TextConfig, TextProcessor

UTTS = [("utt-a", "spk0", "eng"), ("utt-b", "spk1", "fra"), ("utt-c", "spk0", "eng")]
SYMBOLS = [f"s{i}" for i in range(12)] + ["/"]


def _config(tmp, learn_alignment):
    cfg = FastSpeech2Config()
    cfg.model.learn_alignment = learn_alignment
    cfg.preprocessing.save_dir = str(tmp)
    cfg.preprocessing.audio.n_mels = 8
    cfg.text = TextConfig(symbols={"letters": SYMBOLS})
    return cfg


def _materialise(golden, tag, tmp, cfg):
    audio = cfg.preprocessing.audio
    names = {"spec": f"spec-{audio.input_sampling_rate}-{audio.spec_type}.pt", "energy": "energy.pt",
             "pitch": "pitch.pt", "attn": "characters-attn-prior.pt", "duration": "duration.pt"}
    entries = []
    for bn, spk, lang in UTTS:
        for kind, fn in names.items():
            key = f"{tag}/in/{bn}/{kind}"
            if key in golden:
                path = D.feature_path(tmp, kind, bn, spk, lang, fn)
                path.parent.mkdir(exist_ok=True)
                torch.save(torch.from_numpy(golden[key]), path)
        toks = str(golden[f"{tag}/in/{bn}/tokens"])
        entries.append({"basename": bn, "speaker": spk, "language": lang, "character_tokens": toks,
                        "characters": "".join(toks.replace("\\/", "\0").split("/")).replace("\0", "/")})
    return entries


@pytest.mark.parametrize("learn_alignment", [True, False])
def test_dataset_collate_matches_reference(golden_dir, tmp_path, learn_alignment):
    golden = np.load(golden_dir / "data_collate.npz")
    tag = "align" if learn_alignment else "noalign"
    cfg = _config(tmp_path, learn_alignment)
    entries = _materialise(golden, tag, tmp_path, cfg)
    ds = D.FeatureDataset(entries, cfg, {"eng": 0, "fra": 1}, {"spk0": 0, "spk1": 1})
    assert len(ds) == 3
    batch = D.collate([ds[i] for i in range(len(ds))], learn_alignment=learn_alignment)
    checked = 0
    for key in golden.files:
        if not key.startswith(f"{tag}/out/"):
            continue
        name = key.split("/")[-1]
        want = golden[key]
        got = batch[name]
        if torch.is_tensor(got):
            assert str(got.dtype) == str(golden[f"{tag}/dtype/{name}"]), name
            assert tuple(got.shape) == want.shape, name
            assert np.array_equal(got.numpy(), want), name
        else:
            assert list(got) == [str(x) for x in want], name
        checked += 1
    assert checked >= 14
    assert batch["mel"].shape == (3, 31, 8) and batch["text"].shape == (3, 9)
    assert batch["duration"].shape == ((3, 31, 9) if learn_alignment else (3, 9))
    assert batch["mel_style_reference"] == [None] * 3 and batch["is_last_input_chunk"] == [None] * 3


def test_missing_durations_is_a_configuration_error(golden_dir, tmp_path):
    golden = np.load(golden_dir / "data_collate.npz")
    cfg = _config(tmp_path, False)
    entries = _materialise(golden, "align", tmp_path, cfg)  # priors on disk, but no duration/ directory
    ds = D.FeatureDataset(entries, cfg, {"eng": 0, "fra": 1}, {"spk0": 0, "spk1": 1})
    with pytest.raises(ValueError, match="learn_alignment"):
        ds[0]


def test_collate_without_mels_and_ragged_single_item():
    item = {"mel": None, "text": torch.IntTensor([3, 4, 5]), "duration": None, "speaker_id": 2, "raw_text": "abc"}
    out = D.collate([item], learn_alignment=True)
    assert out["mel_lens"] is None and out["max_mel_len"] == 1_000_000
    assert out["src_lens"].tolist() == [3] and int(out["max_src_len"]) == 3
    assert out["speaker_id"].dtype == torch.int32 and out["raw_text"] == ["abc"]
    # numpy features are accepted like tensors
    items = [{"mel": torch.zeros(n, 2), "text": torch.IntTensor(range(n)), "pitch": np.ones(n, np.float32)}
             for n in (1, 4)]
    out = D.collate(items, learn_alignment=False)
    assert out["pitch"].tolist() == [[1, 0, 0, 0], [1, 1, 1, 1]]


def test_escaped_token_sequences():
    tp = TextProcessor(TextConfig(symbols={"a": ["a", "b", "/", "ab"]}))
    assert tp.symbols[0] == "\x80"
    assert tp.encode_escaped_string_sequence("a/b/\\//ab/zz") == [2, 4, 1, 3]
    assert tp.encode_escaped_string_sequence(["a", "ab"]) == [2, 3]
    assert tp.encode_escaped_string_sequence("") == []


def test_truncate_basename():
    short = "Hello, World!"
    assert D.slugify(short) == "hello-world"
    assert D.truncate_basename(short) == "hello-world"
    long = "this is a very long sentence that goes on and on"
    out = D.truncate_basename(long)
    assert out == "this-is-a-very-long-" + "-" + hashlib.sha1(long.encode()).hexdigest()[:8]
    assert D.truncate_basename(long + "x") != out  # same prefix, different hash


def test_spec_writer_trims_and_joins_chunks(tmp_path):
    w = D.SpecWriter(tmp_path, "postnet_output", global_step=77, sampling_rate=22050, spec_type="mel-librosa")
    out = {"postnet_output": torch.arange(2 * 6 * 3, dtype=torch.float32).reshape(2, 6, 3),
           "tgt_lens": torch.tensor([4, 6])}
    batch = {"raw_text": ["hello ", "world"], "speaker": ["s", "s"], "language": ["l", "l"],
             "is_last_input_chunk": [False, True]}
    paths = w.write(out, batch)
    assert [p.name for p in paths] == ["hello-world--s--l--ckpt=77--spec-pred-22050-mel-librosa.pt"]
    assert paths[0].parent == Path(tmp_path) / "synthesized_spec"
    spec = torch.load(paths[0], weights_only=True)
    assert spec.shape == (3, 10)  # [bands, frames]: 4 frames of chunk 0 then 6 of chunk 1
    assert torch.equal(spec[:, :4], out["postnet_output"][0, :4].T)
    assert torch.equal(spec[:, 4:], out["postnet_output"][1].T)
    # the accumulator is reset: an unchunked batch writes one file per utterance
    batch2 = {"raw_text": ["a", "b"], "speaker": ["s", "t"], "language": ["l", "l"], "is_last_input_chunk": None}
    assert len(w.write(out, batch2)) == 2


class _FakeModel:
    def __init__(self, scale):
        self.scale = scale

    def validation_step(self, batch):
        return {"total": torch.tensor(batch * self.scale), "spec": torch.tensor(batch * 2.0)}


def test_validate_is_the_mean_over_batches():
    res = D.validate(_FakeModel(1.0), [1.0, 2.0, 6.0])
    assert res == {"validation/total_loss": pytest.approx(3.0), "validation/spec_loss": pytest.approx(6.0)}
    assert D.validate(_FakeModel(1.0), []) == {}


def _validate_rank(rank, world, port, q):
    import torch.distributed as dist

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    batches = [1.0, 3.0] if rank == 0 else [5.0]  # ragged shards: the mean is over all 3 batches
    q.put((rank, D.validate(_FakeModel(1.0), batches)))
    dist.destroy_process_group()


def test_validate_cross_rank_mean_world2():
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_validate_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(2):
        assert got[r]["validation/total_loss"] == pytest.approx(3.0)
