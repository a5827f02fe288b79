"""Host cost of enqueueing a training step: after a device synchronisation, the wall time of each of the next few step()
calls (nothing waits for the device unless a queue is full) beside the device time of the same steps.
usage (GPU only): python tools/host_probe.py [32-true|bf16-mixed] [batch]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "32-true"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if prec == "bf16-mixed" else 32)
rig = bench.Rig(prec, batch, False, False, 0, 1, 0, False)
for _ in range(4):
    rig.step()
torch.cuda.synchronize()
for rep in range(3):
    ts = [time.perf_counter()]
    for _ in range(6):
        rig.step()
        ts.append(time.perf_counter())
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    print(f"{prec} b{batch}: host ms per step after a sync: " + " ".join(f"{(b - a) * 1e3:6.2f}" for a, b in zip(ts, ts[1:])) +
          f" | all six on the device after {(t_end - ts[0]) * 1e3:6.2f} ms", flush=True)
# the host's own cost: ONE step enqueued into an empty queue (nothing to wait for, nothing pushing back), twelve times
single = []
for rep in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rig.step()
    single.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
single.sort()
print(f"{prec} b{batch}: host ms for one step into an empty queue: median {single[6]:.2f}, min {single[0]:.2f}, max {single[-1]:.2f}", flush=True)
