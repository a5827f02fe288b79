"""GPU: feature files -> FeatureDataset -> collate -> DevicePrefetcher -> FastSpeech2.training_step, and
predict_step -> SpecWriter.  The prefetched path must give the same numbers as handing the collated
CPU batch to the step directly, in the same order."""
import numpy as np
import pytest
import torch

from fastspeech2_lightning_amd import data as D
from fastspeech2_lightning_amd.config import Stats, TextConfig
from oracle import cases as C

pytestmark = pytest.mark.gpu

SYMBOLS = [f"s{i}" for i in range(C.N_SYMBOLS - 1)]  # + pad = N_SYMBOLS


def _write_corpus(tmp, cfg, n_utts=6, learn_alignment=False):
    g = torch.Generator().manual_seed(7)
    audio = cfg.preprocessing.audio
    entries = []
    for u in range(n_utts):
        n_tok, spk, lang = 3 + (u * 5) % 7, "spk0", "l0"
        dur = torch.randint(1, 4, (n_tok,), generator=g)
        n_frames = int(dur.sum())
        bn = f"utt{u}"
        feats = {("spec", f"spec-{audio.input_sampling_rate}-{audio.spec_type}.pt"):
                 torch.randn(audio.n_mels, n_frames, generator=g),
                 # learned alignment stores frame-level pitch / energy (averaged per token by the aligner)
                 ("energy", "energy.pt"): torch.randn(n_frames if learn_alignment else n_tok, generator=g),
                 ("pitch", "pitch.pt"): torch.randn(n_frames if learn_alignment else n_tok, generator=g)}
        if learn_alignment:
            feats[("attn", "characters-attn-prior.pt")] = torch.rand(n_frames, n_tok, generator=g) + 0.1
        else:
            feats[("duration", "duration.pt")] = dur
        for (kind, fn), t in feats.items():
            p = D.feature_path(tmp, kind, bn, spk, lang, fn)
            p.parent.mkdir(exist_ok=True)
            torch.save(t, p)
        toks = [SYMBOLS[int(i)] for i in torch.randint(0, len(SYMBOLS), (n_tok,), generator=g)]
        entries.append({"basename": bn, "speaker": spk, "language": lang, "character_tokens": "/".join(toks),
                        "characters": f"utterance number {u} of the synthetic corpus"})
    return entries


def _model(cfg):
    from fastspeech2_lightning_amd.model import FastSpeech2
    torch.manual_seed(0)
    m = FastSpeech2(cfg, Stats(**C.STATS), lang2id=C.LANG2ID, speaker2id=C.SPEAKER2ID)
    m.postnet.dropout_p = 0.0
    m.configure_optimizers()
    m.configure_gradient_clipping(m.optimizer, 1.0, "norm")
    return m


@pytest.mark.parametrize("learn_alignment", [False, True])
def test_prefetched_training_equals_direct(tmp_path, learn_alignment):
    cfg = C.small_config(learn_alignment=learn_alignment)
    cfg.preprocessing.save_dir = str(tmp_path)
    cfg.text = TextConfig(symbols={"letters": SYMBOLS})
    entries = _write_corpus(tmp_path, cfg, learn_alignment=learn_alignment)
    ds = D.FeatureDataset(entries, cfg, C.LANG2ID, C.SPEAKER2ID)
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=0,
                                         collate_fn=lambda items: D.collate(items, learn_alignment))
    a, b = _model(cfg), _model(cfg)
    b.load_state_dict(a.state_dict())
    a.train(); b.train()
    direct = []
    for batch in loader:
        direct.append(float(a.training_step(batch)))
        a.optimizer.step()
    fetched = []
    for batch in D.DevicePrefetcher(loader, b.prepare_batch, b.device_):
        assert batch["mel"].is_cuda and batch["text"].dtype == torch.int32
        fetched.append(float(b.training_step(batch)))
        b.optimizer.step()
    assert len(direct) == 3 and np.isfinite(direct).all()
    # same kernels, same inputs, same order (tolerance only for atomically accumulated reductions)
    assert np.allclose(direct, fetched, rtol=1e-5, atol=0)
    sa, sb = a.state_dict(), b.state_dict()
    assert all(torch.allclose(sa[k].float(), sb[k].float(), rtol=1e-4, atol=1e-6) for k in sa)
    # validation mean over the same loader
    res = D.validate(b, loader)
    assert set(res) >= {"validation/total_loss", "validation/spec_loss"} and np.isfinite(list(res.values())).all()


def test_predict_and_write_spectrograms(tmp_path):
    cfg = C.small_config(learn_alignment=False)
    cfg.preprocessing.save_dir = str(tmp_path)
    cfg.text = TextConfig(symbols={"letters": SYMBOLS})
    entries = _write_corpus(tmp_path, cfg, n_utts=2)
    ds = D.FeatureDataset(entries, cfg, C.LANG2ID, C.SPEAKER2ID)
    batch = D.collate([ds[0], ds[1]], learn_alignment=False)
    model = _model(cfg)
    out = model.predict_step(batch)  # teacher-forced: mel_lens present
    w = D.SpecWriter(tmp_path / "out", "postnet_output", global_step=3)
    paths = w.write(out, batch)
    assert len(paths) == 2 and len({p.name for p in paths}) == 2
    for i, p in enumerate(paths):
        spec = torch.load(p, weights_only=True)
        n = int(batch["mel_lens"][i])
        assert spec.shape == (cfg.preprocessing.audio.n_mels, n)
        assert torch.equal(spec, out["postnet_output"][i, :n].cpu().T)
        assert p.name.startswith("utterance-number-" + str(i) + "-o-")  # 20-character slug + sha1 suffix
