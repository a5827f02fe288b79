"""Times the Conformer convolution module's memory-bound kernels at the benchmark's decoder shape (B x 648 x 256, K = 9,
GLU): depthwise convolution forward / backward and the BatchNorm passes, on fp32 and on bf16 tensors.  Prints
microseconds and the HBM-side bytes each launch has to move (operands once, results once)."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from fastspeech2_lightning_amd import hip as H  # noqa: E402


def timeit(fn, n=30):
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
    a = ap.parse_args()
    B, T, C, K = a.batch, 648, 256, 9
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B * T, 2 * C, generator=g).cuda()
    w = (0.3 * torch.randn(K, C, generator=g)).cuda()
    bias = torch.zeros(C).cuda()
    dy = torch.randn(B, T, C, generator=g).cuda()
    dw, db = torch.empty(K, C, device="cuda"), torch.empty(C, device="cuda")
    n = B * T * C
    for name, xx, dd, eb in (("fp32", x, dy, 4), ("bf16", x.bfloat16(), dy.bfloat16(), 2)):
        t_f = timeit(lambda: H.dwconv_fwd(xx, w, bias, B, T, glu=True, stats=True))
        t_b = timeit(lambda: H.dwconv_bwd(dd, xx, w, dw, db, B, T, glu=True, out_dtype=torch.bfloat16))
        by_f, by_b = n * (2 * eb + eb), n * (eb + 2 * eb + 2 * 2)
        print(f"{name}: dwconv fwd {t_f:6.1f} us ({by_f / 1e6:5.0f} MB, {by_f / t_f / 1e6:4.2f} TB/s)   "
              f"bwd {t_b:6.1f} us ({by_b / 1e6:5.0f} MB, {by_b / t_b / 1e6:4.2f} TB/s)")


if __name__ == "__main__":
    main()
