"""Diagnostic: dO = one-hot query row q* -> dV[key, 0] = P[q*, key]; prints where the kernel's P differs."""
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

B, T, Hh, hd = 1, int(sys.argv[1]) if len(sys.argv) > 1 else 32, 2, 128
D = Hh * hd
g = torch.Generator().manual_seed(1)
qkv = (0.5 * torch.randn(B, T, 3 * D, generator=g)).bfloat16().cuda()
lens = torch.tensor([T], dtype=torch.int32).cuda()
q, k, v = qkv.float().view(B, T, 3, Hh, hd).permute(2, 0, 3, 1, 4)
P = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), -1)[0, 0]  # [q][key], head 0
ob, lse = H.attention_fwd_b(qkv, lens, B, T, Hh)
got = torch.zeros(T, T)
for qs in range(T):
    dout = torch.zeros(B, T, D)
    dout[0, qs, :] = 1
    dqkv = H.attention_bwd_b(qkv, lens, ob, dout.bfloat16().cuda(), lse, B, T, Hh).float()
    got[qs] = dqkv[0, :, 2 * D].cpu()
err = (got - P.cpu()).abs() / P.cpu().abs().max()
torch.set_printoptions(linewidth=250, precision=2, sci_mode=False)
print("max err", err.max().item())
print((err > 0.02).int())
# is the kernel's row q a permutation of the reference rows?
for qs in range(min(T, 8)):
    d = (got[qs][None, :] - P.cpu()).abs().sum(-1)
    print("kernel row", qs, "closest reference row", int(d.argmin()), "dist", round(d.min().item(), 4))
