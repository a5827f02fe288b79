"""Host time to enqueue one training step (eager) vs the device time of the step."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from fastspeech2_lightning_amd.config import Stats  # noqa: E402
from fastspeech2_lightning_amd.model import FastSpeech2  # noqa: E402
from fastspeech2_lightning_amd.synthetic import DEFAULT_STATS, synthetic_batch  # noqa: E402

model = FastSpeech2(bench.make_config(False), Stats(**DEFAULT_STATS), device="cuda:0", seed=1234)
model.train()
opt = model.configure_optimizers()[0][0]
model.configure_gradient_clipping(opt, 1.0, "norm")  # Trainer(gradient_clip_val=1.0), fs2/cli/train.py:38
batch = model.prepare_batch(synthetic_batch(B=32, ts_lo=96, ts_hi=128, n_symbols=64, n_mels=80, seed=1234, dur_hi=9))
def step():
    model.training_step(batch); opt.step()
for _ in range(4):
    step()
torch.cuda.synchronize()
# host enqueue time with the GPU blocked by a long spin kernel (so the host never waits for the device)
torch.cuda._sleep(int(0.5 * 2.0e9))
t0 = time.perf_counter()
step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host enqueue time for one step: {(t1 - t0) * 1e3:.2f} ms")
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print(f"steady state: {(time.perf_counter() - t0) * 100:.2f} ms/step")

import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile()
torch.cuda._sleep(int(0.5 * 2.0e9))
pr.enable()
step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
