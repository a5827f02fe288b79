"""Isolated timing of the fp32 attention backward at the benchmark's decoder shape: recomputing kernels against the spilled-dS
pair (GPU only).  usage: python tools/diag/attn_bwd_time.py [B=32] [T=648]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 648
Hh, hd = 2, 128
D = Hh * hd
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, T, 3 * D, generator=g).cuda()
dout = torch.randn(B, T, D, generator=g).cuda()
lens = torch.tensor([T, 430, 40] + [430 + (7 * i) % (T - 430 + 1) for i in range(B - 3)], dtype=torch.int32)[:B].cuda()
drop = H.Drop(0.2, 77)
o, lse, sc = H.attention_fwd(qkv, lens, B, T, Hh, drop, save_scores=True)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3


for rep in range(2):
    H.ATTN_SPILL = False
    print(f"B={B} T={T} recomputing backward:        {timed(lambda: H.attention_bwd(qkv, lens, o, dout, lse, B, T, Hh, drop)):8.1f} us", flush=True)
    H.ATTN_SPILL = True
    print(f"B={B} T={T} spilled dS:                  {timed(lambda: H.attention_bwd(qkv, lens, o, dout, lse, B, T, Hh, drop)):8.1f} us", flush=True)
    print(f"B={B} T={T} spilled dS + forward scores: {timed(lambda: H.attention_bwd(qkv, lens, o, dout, lse, B, T, Hh, drop, scores=sc)):8.1f} us", flush=True)
    print(f"B={B} T={T} forward:                     {timed(lambda: H.attention_fwd(qkv, lens, B, T, Hh, drop)):8.1f} us", flush=True)
    print(f"B={B} T={T} forward + score store:       {timed(lambda: H.attention_fwd(qkv, lens, B, T, Hh, drop, save_scores=True)):8.1f} us", flush=True)
