"""Attention forward / backward at the decoder's shape (B=32, T=648, H=2, HD=128) (GPU only).
usage: python tools/bench_attn.py [ragged|full|lenN] [drop_p] [T] [precision]   -- `full`: every utterance T keys (no tail
effects); precision 32-true | bf16-mixed | 32-split: prints the error against the 32-true kernels too"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


mode = sys.argv[1] if len(sys.argv) > 1 else "ragged"
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
dev = "cuda"
B, T, Hh, HD = 32, int(sys.argv[3]) if len(sys.argv) > 3 else 648, 2, 128
D = Hh * HD
g = torch.Generator().manual_seed(0)
if mode == "ragged":
    lens = torch.randint(int(T * 0.66), T + 1, (B,), generator=g).int()
elif mode.startswith("len"):
    lens = torch.full((B,), int(mode[3:]), dtype=torch.int32)
else:
    lens = torch.full((B,), T, dtype=torch.int32)
lens = lens.to(dev)
qkv = torch.randn(B * T, 3 * D, device=dev) if "FS2_UNIFORM" not in __import__("os").environ else torch.rand(B * T, 3 * D, device=dev) * 2 - 1
dout = torch.randn(B * T, D, device=dev)
step = torch.zeros(4, dtype=torch.int64, device=dev)
drop = H.Drop(p, 777, step) if p > 0 else H.NO_DROP
prec = sys.argv[4] if len(sys.argv) > 4 else "32-true"
o0, lse0 = H.attention_fwd(qkv, lens, B, T, Hh, drop)
g0 = H.attention_bwd(qkv, lens, o0, dout, lse0, B, T, Hh, drop)
H.set_precision(prec)
o, lse = H.attention_fwd(qkv, lens, B, T, Hh, drop)
if prec != "32-true":
    g1 = H.attention_bwd(qkv, lens, o, dout, lse, B, T, Hh, drop)
    valid = (torch.arange(T, device=dev)[None, :] < lens[:, None])
    rel = lambda a, b: float((a - b).norm() / b.norm())  # noqa: E731
    mx = lambda a, b: float((a - b).abs().max())  # noqa: E731
    om, o0m = o * valid[..., None], o0 * valid[..., None]
    gm, g0m = g1.view(B, T, -1) * valid[..., None], g0.view(B, T, -1) * valid[..., None]
    print(f"{prec} vs 32-true: o rel {rel(om, o0m):.2e} max {mx(om, o0m):.2e}; lse max {mx(lse * valid[:, None, :], lse0 * valid[:, None, :]):.2e}; "
          f"dqkv rel {rel(gm, g0m):.2e} max {mx(gm, g0m):.2e} (|dqkv| max {float(g0m.abs().max()):.2e})")
tf = timeit(lambda: H.attention_fwd(qkv, lens, B, T, Hh, drop), 20)
tb = timeit(lambda: H.attention_bwd(qkv, lens, o, dout, lse, B, T, Hh, drop), 20)
frac = float(lens.float().mean()) / T
flop_fwd = 4.0 * B * Hh * T * float(lens.float().mean()) * HD  # two products over the valid keys
print(f"{mode:6s} T={T} drop={p:.1f}: fwd {tf * 1e6:7.1f} us ({flop_fwd / tf / 1e12:5.1f} TF)  bwd (delta+dQ+dK/dV) {tb * 1e6:7.1f} us "
      f"({3.5 * flop_fwd / tb / 1e12:5.1f} TF on 7 products)  (mean len/T = {frac:.2f})")
