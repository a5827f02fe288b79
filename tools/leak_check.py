"""Runs N train steps of the benchmark configuration and prints allocator statistics along the way (GPU only):
memory must be flat after the first steps (usage: python tools/leak_check.py [steps] [precision])."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from fastspeech2_lightning_amd.config import Stats  # noqa: E402
from fastspeech2_lightning_amd.model import FastSpeech2  # noqa: E402
from fastspeech2_lightning_amd.synthetic import DEFAULT_STATS, synthetic_batch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
prec = sys.argv[2] if len(sys.argv) > 2 else "32-true"
torch.cuda.set_device(0)
model = FastSpeech2(bench.make_config(), Stats(**DEFAULT_STATS), device="cuda:0", seed=1234, precision=prec)
model.train()
opt = model.configure_optimizers()[0][0]
model.configure_gradient_clipping(opt, 1.0, "norm")  # Trainer(gradient_clip_val=1.0), fs2/cli/train.py:38
batch = model.prepare_batch(synthetic_batch(B=32, ts_lo=96, ts_hi=128, n_symbols=64, n_mels=80, seed=1234, dur_hi=9))
marks = {}
for i in range(steps):
    model.training_step(batch)
    opt.step()
    if i in (20, steps // 2, steps - 1):
        torch.cuda.synchronize()
        marks[i] = (torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, float(model.last_losses["total"]))
        print(f"step {i}: allocated {marks[i][0]} MiB, reserved {marks[i][1]} MiB, loss {marks[i][2]:.4f}", flush=True)
a = [v[1] for v in marks.values()]
assert a[-1] <= a[0] * 1.05 + 64, f"reserved memory grew: {a}"
print("flat")
