"""Diagnostic (run by hand on the GPU box): the full-size train step of the HIP path and of the fp32 CPU oracle, both
against the oracle in float64 -- which gradients are ill-conditioned in fp32 on EITHER side, and by how much."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from fastspeech2_lightning_amd.config import FastSpeech2Config, Stats  # noqa: E402
from fastspeech2_lightning_amd.synthetic import DEFAULT_STATS, default_symbols, synthetic_batch  # noqa: E402
from oracle import fs2_oracle as O  # noqa: E402
from tests import dropout_masks as DM  # noqa: E402


def main(dropout_on: bool, tune: bool):
    from fastspeech2_lightning_amd import hip as H
    from fastspeech2_lightning_amd.model import FastSpeech2
    H.GEMM_TUNE = tune
    if dropout_on:
        config = FastSpeech2Config(model=dict(learn_alignment=False), text=default_symbols(64))
    else:
        conf, vp = dict(dropout=0.0), dict(dropout=0.0)
        config = FastSpeech2Config(model=dict(learn_alignment=False, encoder=conf, decoder=conf,
                                              variance_predictors=dict(energy=vp, pitch=vp, duration=vp)),
                                   text=default_symbols(64))
    batch = synthetic_batch(B=32, ts_lo=96, ts_hi=128, n_symbols=64, n_mels=80, seed=1234, dur_hi=9)
    torch.set_num_threads(16)
    model = FastSpeech2(config, Stats(**DEFAULT_STATS), seed=1234)
    o32 = O.FastSpeech2Oracle(config, Stats(**DEFAULT_STATS), n_symbols=64)
    sd = O.seeded_state_dict(o32.state_dict())
    o32.load_state_dict(sd)
    o64 = O.FastSpeech2Oracle(config, Stats(**DEFAULT_STATS), n_symbols=64)
    o64.load_state_dict(sd)
    o64 = o64.double()
    model.load_state_dict(sd)
    model.train(); o32.train(); o64.train()
    p_post = 0.5 if dropout_on else 0.0
    model.postnet.dropout_p = o32.postnet.dropout_p = o64.postnet.dropout_p = p_post
    if dropout_on:
        DM.inject(model, o32, 32, batch["text"].shape[1], 648)
        DM.inject(model, o64, 32, batch["text"].shape[1], 648)
    b64 = {k: (v.double() if torch.is_tensor(v) and v.dtype == torch.float32 else v) for k, v in batch.items()}
    l32 = o32.loss(o32(batch), batch, 0)
    l32["total"].backward()
    l64 = o64.loss(o64(b64), b64, 0)
    l64["total"].backward()
    model.training_step(batch)
    for k in l64:
        print(f"loss {k:10s} hip {float(model.last_losses[k]):.7f} cpu32 {float(l32[k]):.7f} f64 {float(l64[k]):.7f}")
    got = model.store.grad_state_dict()
    g32 = {k: p.grad for k, p in o32.named_parameters() if p.grad is not None}
    rows = []
    for k, p in o64.named_parameters():
        if p.grad is None:
            continue
        s = float(p.grad.abs().max())
        eh = float((got[k].cpu().double() - p.grad).abs().max()) / s
        ec = float((g32[k].double() - p.grad).abs().max()) / s
        rows.append((eh, ec, s, k))
    rows.sort(reverse=True)
    print(f"{'hip err':>10s} {'cpu32 err':>10s} {'|g|max':>10s}  tensor   (errors relative to the tensor's max, vs float64)")
    for eh, ec, s, k in rows[:25]:
        print(f"{eh:10.2e} {ec:10.2e} {s:10.2e}  {k}")
    print("median hip", sorted(r[0] for r in rows)[len(rows) // 2], "median cpu32", sorted(r[1] for r in rows)[len(rows) // 2])


if __name__ == "__main__":
    main("--dropout" in sys.argv, "--notune" not in sys.argv)
