"""Forward / loss / backward / optimizer split of the benchmark step (HIP events on the main stream, eager).
usage: python tools/phase_times.py [side=1|0]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from fastspeech2_lightning_amd.config import Stats  # noqa: E402
from fastspeech2_lightning_amd.model import FastSpeech2  # noqa: E402
from fastspeech2_lightning_amd.synthetic import DEFAULT_STATS, synthetic_batch  # noqa: E402

side = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
config = bench.make_config(False)
model = FastSpeech2(config, Stats(**DEFAULT_STATS), device="cuda:0", seed=1234)
model.train()
model.env.side_enabled = side
opt = model.configure_optimizers()[0][0]
model.configure_gradient_clipping(opt, 1.0, "norm")  # Trainer(gradient_clip_val=1.0), fs2/cli/train.py:38
batch = model.prepare_batch(synthetic_batch(B=32, ts_lo=96, ts_hi=128, n_symbols=64, n_mels=80, seed=1234, dur_hi=9))


def step(ev=None):
    def mark():
        if ev is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev.append(e)
    mark()
    out = model(batch)
    mark()
    losses = model.loss(out, model._ctx["batch"], model.current_epoch)
    mark()
    model.backward()
    mark()
    opt.step()
    mark()


for _ in range(4):
    step()
torch.cuda.synchronize()
acc = [0.0] * 4
N = 10
for _ in range(N):
    torch.cuda._sleep(int(0.03 * 2.0e9))
    ev = []
    step(ev)
    torch.cuda.synchronize()
    for i in range(4):
        acc[i] += ev[i].elapsed_time(ev[i + 1])
names = ["forward", "loss", "backward", "optimizer"]
print(f"side stream {'on' if side else 'off'}: " + "  ".join(f"{n} {a / N:.2f} ms" for n, a in zip(names, acc)) + f"  total {sum(acc) / N:.2f} ms")
