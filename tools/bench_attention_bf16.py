"""Times the attention kernel families at the benchmark's decoder shape (B utterances, T = 648, 2 heads x 128, ragged
lengths 0.66 .. 1, dropout 0.1): fp32 storage with bf16 operands rounded in registers (attention2.hip, what
bf16-mixed ran in rounds 1-2) against bf16 storage (attention_bf16.hip).  Prints microseconds per launch."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from fastspeech2_lightning_amd import hip as H  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--T", type=int, default=648)
    ap.add_argument("--drop", type=float, default=0.1)
    a = ap.parse_args()
    B, T, Hh, hd = a.batch, a.T, 2, 128
    D = Hh * hd
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B, T, 3 * D, generator=g).cuda()
    dout = torch.randn(B, T, D, generator=g).cuda()
    lens = (T * (0.66 + 0.34 * torch.rand(B, generator=g))).int().clamp(1, T)
    lens[0] = T
    lens = lens.cuda()
    drop = H.Drop(a.drop, 7) if a.drop > 0 else H.NO_DROP
    qb, db = qkv.bfloat16(), dout.bfloat16()
    saved = H.get_precision()
    H.set_precision("bf16-mixed")
    try:
        o, lse = H.attention_fwd(qkv, lens, B, T, Hh, drop)
        t_f = timeit(lambda: H.attention_fwd(qkv, lens, B, T, Hh, drop))
        t_b = timeit(lambda: H.attention_bwd(qkv, lens, o, dout, lse, B, T, Hh, drop))
    finally:
        H.set_precision(saved)
    ob, lseb = H.attention_fwd_b(qb, lens, B, T, Hh, drop)
    t_fb = timeit(lambda: H.attention_fwd_b(qb, lens, B, T, Hh, drop))
    t_bb = timeit(lambda: H.attention_bwd_b(qb, lens, ob, db, lseb, B, T, Hh, drop))
    keys = lens.float().sum().item()
    flops_f = 2 * 2 * Hh * hd * T * keys  # two products over (T queries x valid keys)
    print(f"B={B} T={T} drop={a.drop}: valid keys {keys / (B * T):.2f} of padded")
    print(f"  fp32 storage, bf16 operands : fwd {t_f:8.1f} us   bwd {t_b:8.1f} us")
    print(f"  bf16 storage                : fwd {t_fb:8.1f} us ({flops_f / t_fb * 1e-6:.0f} TFLOP/s)   "
          f"bwd {t_bb:8.1f} us ({2.5 * flops_f / t_bb * 1e-6:.0f} TFLOP/s algorithmic)")


if __name__ == "__main__":
    main()
