"""bf16-mixed vs the fp32 oracle (and vs the oracle under CPU autocast) on the default-width model: prints error metrics."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import torch
from fastspeech2_lightning_amd.config import FastSpeech2Config, Stats
from fastspeech2_lightning_amd.model import FastSpeech2
from oracle import cases as C
from oracle import fs2_oracle as O

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
conf = dict(layers=layers, dropout=0.0)
vp = dict(dropout=0.0)
config = FastSpeech2Config(
    model=dict(encoder=conf, decoder=conf, learn_alignment=False,
               variance_predictors=dict(energy=vp, pitch=vp, duration=vp)),
    text=dict(symbols=dict(letters=[f"s{i}" for i in range(40)])))
batch = O.synthetic_batch(B=4, ts_lo=20, ts_hi=40, n_symbols=41, n_mels=80, seed=3, dur_hi=6)
oracle = O.FastSpeech2Oracle(config, Stats(**C.STATS), n_symbols=41)
sd = O.seeded_state_dict(oracle.state_dict())
oracle.load_state_dict(sd)
oracle.train(); oracle.postnet.dropout_p = 0.0
ref = oracle(batch)
ref_losses = oracle.loss(ref, batch, 0)
ref_losses["total"].backward()
rg = {k: p.grad.clone() for k, p in oracle.named_parameters() if p.grad is not None}

def report(tag, out, losses, grads):
    key = "postnet_output"
    o = out[key].detach().float().cpu()
    r = ref[key].detach()
    print(f"{tag}: mel mse {float(((o - r) ** 2).mean()):.3e}  max abs {float((o - r).abs().max()):.3e}  (ref rms {float(r.pow(2).mean().sqrt()):.3f})")
    for k in ref_losses:
        print(f"   loss {k}: {float(losses[k]):.6f} ref {float(ref_losses[k]):.6f} rel {abs(float(losses[k]) - float(ref_losses[k])) / max(abs(float(ref_losses[k])), 1e-9):.2e}")
    num = den = dot = 0.0; worst = ("", 0.0)
    gmax = max(float(g.abs().max()) for g in rg.values())
    for k, g in rg.items():
        d = grads[k].float().cpu() - g
        num += float(d.pow(2).sum()); den += float(g.pow(2).sum())
        r_ = float(d.abs().max()) / gmax
        if r_ > worst[1]: worst = (k, r_)
    print(f"   grads: rel L2 {np.sqrt(num / den):.3e}  worst max-abs/gmax {worst}")

for prec in ("32-true", "bf16-mixed"):
    model = FastSpeech2(config, Stats(**C.STATS), precision=prec)
    model.load_state_dict(sd)
    model.train(); model.postnet.dropout_p = 0.0
    total = model.training_step(batch)
    out = model._last_output if hasattr(model, "_last_output") else model(batch)
    report(prec, out, model.last_losses, model.store.grad_state_dict())

# the reference's own bf16-mixed semantics: torch autocast on the CPU oracle
oracle.zero_grad()
with torch.autocast("cpu", dtype=torch.bfloat16):
    out = oracle(batch)
    losses = oracle.loss(out, batch, 0)
losses["total"].backward()
report("cpu autocast(bf16)", out, losses, {k: p.grad for k, p in oracle.named_parameters() if p.grad is not None})
