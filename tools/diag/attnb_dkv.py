"""Diagnostic: where do dK / dV of the bf16-storage attention kernels differ from the reference?"""
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

B, T, Hh, hd = 1, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 2, 128
D = Hh * hd
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B, T, 3 * D, generator=g).bfloat16().cuda()
dout = torch.randn(B, T, D, generator=g).bfloat16().cuda()
lens = torch.tensor([T], dtype=torch.int32).cuda()
qr = qkv.float().requires_grad_(True)
q, k, v = qr.view(B, T, 3, Hh, hd).permute(2, 0, 3, 1, 4)
s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
p = torch.softmax(s, -1)
o = (p @ v).permute(0, 2, 1, 3).reshape(B, T, D)
o.backward(dout.float())
ob, lse = H.attention_fwd_b(qkv, lens, B, T, Hh)
dqkv = H.attention_bwd_b(qkv, lens, ob, dout, lse, B, T, Hh).float()
for name, off in (("dq", 0), ("dk", D), ("dv", 2 * D)):
    a, r = dqkv[0, :, off:off + D], qr.grad[0, :, off:off + D]
    e = (a - r).abs()
    print(name, "rel", (e.norm() / r.norm()).item())
    rows = e.view(T, Hh, hd).amax(-1)  # [T][H]
    print("  per key/query row (head 0), blocks of 8:", [round(x, 3) for x in rows[:, 0].view(-1, 8).amax(-1).tolist()])
    cols = e.view(T, Hh, 4, 32).amax(0)[0].amax(-1)
    print("  per d-block:", [round(x, 3) for x in cols.tolist()])
