"""Attention forward / backward at the decoder's shape (B=32, T=648, H=2, HD=128), dropout on (GPU only)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
B, T, Hh, HD = 32, 648, 2, 128
D = Hh * HD
g = torch.Generator().manual_seed(0)
lens = torch.randint(430, 649, (B,), generator=g).int().to(dev)
qkv = torch.randn(B * T, 3 * D, device=dev)
dout = torch.randn(B * T, D, device=dev)
step = torch.zeros(4, dtype=torch.int64, device=dev)
drop = H.Drop(0.2, 777, step)
o, lse = H.attention_fwd(qkv, lens, B, T, Hh, drop)
tf = timeit(lambda: H.attention_fwd(qkv, lens, B, T, Hh, drop), 20)
tb = timeit(lambda: H.attention_bwd(qkv, lens, o, dout, lse, B, T, Hh, drop), 20)
frac = float(lens.float().mean()) / T
print(f"fwd {tf * 1e6:7.1f} us  bwd (delta+dQ+dK/dV) {tb * 1e6:7.1f} us  (mean len/T = {frac:.2f})")
