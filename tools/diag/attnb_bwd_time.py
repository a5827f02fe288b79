"""Isolated timing of the bf16-storage attention backward at the benchmark's decoder shape (B = 64, T = 648, 2 x 128): recomputing
pair against the spilled-dS pair (GPU only)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

B, T, Hh, hd = 64, 648, 2, 128
D = Hh * hd
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, T, 3 * D, generator=g).bfloat16().cuda()
dout = torch.randn(B, T, D, generator=g).bfloat16().cuda()
lens = torch.tensor([T, 430, 40] + [430 + (7 * i) % (T - 430 + 1) for i in range(B - 3)], dtype=torch.int32).cuda()
drop = H.Drop(0.2, 77)
o, lse = H.attention_fwd_b(qkv, lens, B, T, Hh, drop)
for rep in range(2):
    for spill in (False, True):
        H.ATTN_SPILL_B = spill
        H.attention_bwd_b(qkv, lens, o, dout, lse, B, T, Hh, drop)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            H.attention_bwd_b(qkv, lens, o, dout, lse, B, T, Hh, drop)
        e1.record()
        torch.cuda.synchronize()
        print(f"bf16 storage, B={B} T={T}, spilled dS={spill}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us per backward", flush=True)
