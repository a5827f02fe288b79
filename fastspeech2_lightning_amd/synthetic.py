"""Synthetic LJSpeech-shaped batches in the layout of the reference's ``collate_method``
(``fs2/dataset.py:257-293``), as specified in SURVEY.md 8d: ``src_lens ~ U{lo..hi}`` with at least
one utterance at ``hi``, per-token durations ``U{1..dur_hi}`` (zero on padding), ``mel_lens`` = the
duration totals, ``mel ~ N(0,1)`` zeroed on padding, pitch/energy ``N(0,1)``.  CPU tensors."""
from __future__ import annotations

import numpy as np
import torch


def synthetic_batch(B=2, ts_lo=48, ts_hi=64, n_symbols=64, n_mels=80, seed=1234,
                    learn_alignment=False, dur_hi=9, frame_level=False, content_seed=None):
    """``seed`` fixes the batch's structure (text lengths, durations, hence every padded shape); ``content_seed``, when
    given, draws the token ids / mel / pitch / energy values from a second generator -- data-parallel ranks then get
    batches of identical shape (the same work per GPU, which is what weak scaling means) with different contents."""
    g = torch.Generator().manual_seed(seed)
    src_lens = torch.randint(ts_lo, ts_hi + 1, (B,), generator=g, dtype=torch.int32)
    src_lens[0] = ts_hi
    Ts = int(src_lens.max())
    text = torch.randint(1, n_symbols, (B, Ts), generator=g, dtype=torch.int32)
    dur = torch.randint(1, dur_hi + 1, (B, Ts), generator=g, dtype=torch.int32)
    if content_seed is not None:
        g = torch.Generator().manual_seed(content_seed)
        text = torch.randint(1, n_symbols, (B, Ts), generator=g, dtype=torch.int32)
    smask = torch.arange(Ts)[None, :] < src_lens[:, None]
    text = text * smask
    dur = dur * smask
    mel_lens = dur.sum(1).to(torch.int32)
    Tm = int(mel_lens.max())
    tmask = torch.arange(Tm)[None, :] < mel_lens[:, None]
    mel = torch.randn(B, Tm, n_mels, generator=g) * tmask[..., None]
    lvl_frames = learn_alignment or frame_level
    if lvl_frames:
        pitch = torch.randn(B, Tm, generator=g) * tmask
        energy = torch.randn(B, Tm, generator=g) * tmask
    else:
        pitch = torch.randn(B, Ts, generator=g) * smask
        energy = torch.randn(B, Ts, generator=g) * smask
    batch = dict(text=text, src_lens=src_lens, max_src_len=Ts, mel=mel, mel_lens=mel_lens,
                 max_mel_len=Tm, pitch=pitch, energy=energy,
                 speaker_id=torch.zeros(B, dtype=torch.int32),
                 language_id=torch.zeros(B, dtype=torch.int32))
    if learn_alignment:
        batch["duration"] = beta_binomial_prior(mel_lens, src_lens, Tm, Ts)
    else:
        batch["duration"] = dur
    return batch


def beta_binomial_prior(mel_lens, src_lens, Tm, Ts, scaling=1.0):
    """Attention prior of the aligner's data pipeline (beta-binomial over text positions for each
    frame), zero padded to (B, Tm, Ts)."""
    from scipy.stats import betabinom

    out = torch.zeros(len(mel_lens), Tm, Ts)
    for b, (t1, t2) in enumerate(zip(mel_lens.tolist(), src_lens.tolist())):
        k = np.arange(t2)
        rows = [betabinom(t2 - 1, scaling * i, scaling * (t1 + 1 - i)).pmf(k) for i in range(1, t1 + 1)]
        out[b, :t1, :t2] = torch.tensor(np.array(rows), dtype=torch.float32)
    return out


DEFAULT_STATS = dict(pitch=dict(min=0, max=1, std=1, mean=0, norm_min=-3, norm_max=3),
                     energy=dict(min=0, max=1, std=1, mean=0, norm_min=-3, norm_max=3))


def default_symbols(n_symbols: int = 64):
    """A text config whose symbol table has ``n_symbols`` entries including the pad symbol."""
    return dict(symbols=dict(letters=[f"s{i:02d}" for i in range(n_symbols - 1)]))
