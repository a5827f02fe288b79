import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from fastspeech2_lightning_amd.config import FastSpeech2Config, Stats
from fastspeech2_lightning_amd.model import FastSpeech2
from oracle import cases as C
from oracle import fs2_oracle as O

def run(B, ts_lo, ts_hi, dur_hi, seed, zero_dur=False):
    config = C.small_config(learn_alignment=False)
    batch = O.synthetic_batch(B=B, ts_lo=ts_lo, ts_hi=ts_hi, n_symbols=C.N_SYMBOLS, n_mels=config.preprocessing.audio.n_mels, seed=seed, dur_hi=dur_hi)
    model = FastSpeech2(config, Stats(**C.STATS))
    oracle = O.FastSpeech2Oracle(config, Stats(**C.STATS), n_symbols=C.N_SYMBOLS)
    sd = O.seeded_state_dict(oracle.state_dict())
    oracle.load_state_dict(sd); model.load_state_dict(sd)
    model.train(); oracle.train(); model.postnet.dropout_p = 0.0; oracle.postnet.dropout_p = 0.0
    ref = oracle(batch); rl = oracle.loss(ref, batch, 0); rl["total"].backward()
    tot = model.training_step(batch)
    g = model.store.grad_state_dict()
    worst = 0.0
    gmax = max(float(p.grad.abs().max()) for p in oracle.parameters() if p.grad is not None)
    for k, p in oracle.named_parameters():
        if p.grad is None: continue
        worst = max(worst, float((g[k].cpu() - p.grad).abs().max()) / gmax)
    print(f"B={B} ts=[{ts_lo},{ts_hi}] dur_hi={dur_hi}: Ts={batch['text'].shape[1]} Tm={batch['mel'].shape[1]} loss {float(tot):.6f} vs {float(rl['total']):.6f}  grad err/gmax {worst:.2e}", flush=True)

run(1, 2, 2, 2, 1)  # (one token, train mode: torch BatchNorm itself refuses a single value per channel)
run(1, 2, 3, 1, 2)
run(2, 1, 9, 3, 3)
run(5, 3, 40, 6, 4)
run(1, 130, 140, 9, 5)
