"""cProfile of the host side of training steps (the device runs beside it; nothing is synchronised inside the profile).
usage (GPU only): python tools/host_profile.py [32-true|bf16-mixed] [batch] [n_steps=20]"""
import cProfile
import pstats
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "32-true"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if prec == "bf16-mixed" else 32)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rig = bench.Rig(prec, batch, False, False, 0, 1, 0, False)
for _ in range(4):
    rig.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    rig.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(45)
