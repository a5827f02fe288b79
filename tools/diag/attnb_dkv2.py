"""Diagnostic: Q = 0 (uniform probabilities), dO = one-hot query row -> dV[key, :] = 1/T for every key."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

B, T, Hh, hd = 1, int(sys.argv[1]) if len(sys.argv) > 1 else 32, 2, 128
D = Hh * hd
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B, T, 3 * D, generator=g)
qkv[..., :D] = 0
qkv = qkv.bfloat16().cuda()
lens = torch.tensor([T], dtype=torch.int32).cuda()
ob, lse = H.attention_fwd_b(qkv, lens, B, T, Hh)
print("lse", lse[0, 0, :4].tolist(), "expect", torch.log(torch.tensor(float(T))).item())
res = []
for qs in range(T):
    dout = torch.zeros(B, T, D)
    dout[0, qs, :] = 1
    dout = dout.bfloat16().cuda()
    dqkv = H.attention_bwd_b(qkv, lens, ob, dout, lse, B, T, Hh).float()
    dv = dqkv[0, :, 2 * D:2 * D + hd] * T  # head 0: expect all ones
    res.append((qs, dv.min().item(), dv.max().item(), dv[:, 0].tolist()[:8]))
for r in res:
    print(r)
