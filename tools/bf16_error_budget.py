"""Which bf16 tensor group costs what: BASELINE.json configs[2] -- batch 64, Tm = 648, default model,
dropout on with the kernels' masks -- mel MSE / loss error / gradient error against the fp32 CPU oracle, and ms per step,
with each group switched back in turn:

  stored           the product's bf16-mixed mode (bf16 operand STORAGE end to end)
  chain_fp32       FS2_BF16_CHAIN=0: convolution-module value|gate / depthwise result / PostNet inner results fp32 tensors
  register_round   FS2_BF16_STORAGE=0: fp32 tensors everywhere, GEMM operands rounded to bf16 in registers (rounds 1-2)
  <block>_fp32     that block's GEMMs (and attention) in exact fp32 ("32-true"), everything else as `stored`
  all_fp32         precision 32-true (the parity path: what the oracle comparison is worth at this size)

usage (GPU box): python tools/bf16_error_budget.py [batch] > gpurun_out/bf16_error_budget.txt"""
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
from fastspeech2_lightning_amd import hip as H  # noqa: E402
from fastspeech2_lightning_amd import modules as M  # noqa: E402
from fastspeech2_lightning_amd.config import FastSpeech2Config, Stats  # noqa: E402
from fastspeech2_lightning_amd.model import FastSpeech2  # noqa: E402
from fastspeech2_lightning_amd.synthetic import DEFAULT_STATS, default_symbols, synthetic_batch  # noqa: E402
from oracle import fs2_oracle as O  # noqa: E402  (the checker: this is a measurement tool, not the product)
import dropout_masks as DM  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
config = FastSpeech2Config(model=dict(learn_alignment=False), text=default_symbols(64))
batch = synthetic_batch(B=B, ts_lo=96, ts_hi=128, n_symbols=64, n_mels=80, seed=1234, dur_hi=9)
Ts, Tm = batch["text"].shape[1], batch["mel"].shape[1]
torch.set_num_threads(max(torch.get_num_threads(), 16))
oracle = O.FastSpeech2Oracle(config, Stats(**DEFAULT_STATS), n_symbols=64)
sd = O.seeded_state_dict(oracle.state_dict())
oracle.load_state_dict(sd)
oracle.train()
oracle.postnet.dropout_p = 0.5
ref = ref_losses = ref_grads = None


def run(name, precision="bf16-mixed", storage=True, chain=True, overrides=None):
    global ref, ref_losses, ref_grads
    H.BF16_STORAGE, M.BF16_CHAIN = storage, chain
    H._TILE_CACHE.clear()
    model = FastSpeech2(config, Stats(**DEFAULT_STATS), seed=1234, precision=precision)
    model.precision_overrides = dict(overrides or {})
    model.load_state_dict(sd)
    model.train()
    model.postnet.dropout_p = 0.5
    if ref is None:  # the masks depend on (seed, site, step) only: one oracle run serves every configuration
        DM.inject(model, oracle, B, Ts, Tm)
        ref = oracle(batch)
        ref_losses = oracle.loss(ref, batch, 0)
        ref_losses["total"].backward()
        ref_grads = {k: p.grad.clone() for k, p in oracle.named_parameters() if p.grad is not None}
    model.training_step(batch)
    out = model.last_output
    o, r = out["postnet_output"].cpu(), ref["postnet_output"].detach()
    mse = float(((o - r) ** 2).mean())
    ltot = abs(float(model.last_losses["total"]) - float(ref_losses["total"])) / abs(float(ref_losses["total"]))
    lmax = max(abs(float(v) - float(ref_losses[k])) / abs(float(ref_losses[k])) for k, v in model.last_losses.items())
    got = model.store.grad_state_dict()
    num = sum(float((got[k].cpu() - g).pow(2).sum()) for k, g in ref_grads.items())
    den = sum(float(g.pow(2).sum()) for g in ref_grads.values())
    # ms per step: the step counter has moved, timing only
    opt = model.configure_optimizers()[0][0]
    model.configure_gradient_clipping(opt, 1.0, "norm")
    dev = model.prepare_batch(batch)

    def step():
        with torch.no_grad():
            model.training_step(dev)
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(15):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 15 * 1e3
    print(f"{name:16s} mel MSE {mse:9.3e}   total loss {ltot:9.3e}   worst term {lmax:9.3e}   gradients (rel. L2) {(num / den) ** 0.5:7.4f}"
          f"   {ms:6.2f} ms/step", flush=True)
    del model
    torch.cuda.empty_cache()


print(f"bf16 error budget at batch {B}, Ts {Ts}, Tm {Tm}, dropout on (kernel masks), against the fp32 CPU oracle")
run("stored")
run("chain_fp32", chain=False)
run("register_round", storage=False)
for block in ("postnet", "decoder", "encoder", "adaptor", "mel_linear"):
    run(f"{block}_fp32", overrides={block: "32-true"})
run("dec+post_fp32", overrides={"decoder": "32-true", "postnet": "32-true", "mel_linear": "32-true"})
run("all_fp32", precision="32-true")
