"""LayerNorm forward / backward at the decoder's size (20736 x 256): microseconds and achieved HBM-side TB/s (GPU only)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

M, C = 20736, 256
x = torch.randn(M, C, device="cuda"); dy = torch.randn(M, C, device="cuda"); r = torch.randn(M, C, device="cuda")
g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
dg = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda")
y, mean, rstd = H.layernorm_fwd(x, g, b)


def timeit(fn, n=200):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


t = timeit(lambda: H.layernorm_fwd(x, g, b))
print(f"ln fwd           {t:6.1f} us  {2 * M * C * 4 / t / 1e6:5.2f} TB/s")
t = timeit(lambda: H.layernorm_bwd(dy, x, g, mean, rstd, dg, db))
print(f"ln bwd           {t:6.1f} us  {3 * M * C * 4 / t / 1e6:5.2f} TB/s (incl. the second-stage reduction)")
t = timeit(lambda: H.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dx_add=r))
print(f"ln bwd + dx_add  {t:6.1f} us  {4 * M * C * 4 / t / 1e6:5.2f} TB/s")
