"""Per-shape timing of the fs2hip GEMM family on the shapes of the benchmark step (GPU only)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    M = 20736
    T = 648
    dev = "cuda"
    shapes = [  # (name, kind, M, N, K, taps)
        ("ffn1 fwd", "nt", M, 1024, 256, 1), ("ffn2 fwd", "nt", M, 256, 1024, 1), ("qkv fwd", "nt", M, 768, 256, 1),
        ("proj fwd", "nt", M, 256, 256, 1), ("pw1 fwd", "nt", M, 512, 256, 1), ("mel fwd", "nt", M, 80, 256, 1),
        ("post0 fwd", "nt", M, 512, 80, 5), ("post1 fwd", "nt", M, 512, 512, 5), ("post4 fwd", "nt", M, 80, 512, 5),
        ("ffn2 dx", "nn", M, 1024, 256, 1), ("ffn1 dx", "nn", M, 256, 1024, 1), ("qkv dx", "nn", M, 256, 768, 1),
        ("proj dx", "nn", M, 256, 256, 1), ("post1 dx", "nn", M, 512, 512, 5),
        ("ffn1 dw", "tn", M, 1024, 256, 1), ("ffn2 dw", "tn", M, 256, 1024, 1), ("proj dw", "tn", M, 256, 256, 1),
        ("qkv dw", "tn", M, 768, 256, 1), ("post1 dw", "tn", M, 512, 512, 5), ("enc ffn1 fwd", "nt", 4096, 1024, 256, 1),
        ("enc proj fwd", "nt", 4096, 256, 256, 1),
    ]
    tot_t = tot_f = 0
    for name, kind, m, n, k, taps in shapes:
        g = torch.Generator(device=dev).manual_seed(0)
        if kind == "nt":
            x = torch.randn(m, k, device=dev, generator=g)
            w = torch.randn(*( (taps, n, k) if taps > 1 else (n, k)), device=dev, generator=g)
            out = torch.empty(m, n, device=dev)
            fn = lambda: H.linear_fwd(x, w, taps=taps, T=T, out=out)
        elif kind == "nn":  # dx[m, n] = dy[m, k] @ w[k, n]
            dy = torch.randn(m, k, device=dev, generator=g)
            w = torch.randn(*((taps, k, n) if taps > 1 else (k, n)), device=dev, generator=g)
            out = torch.empty(m, n, device=dev)
            fn = lambda: H.linear_bwd_data(dy, w, taps=taps, T=T, out=out)
        else:  # dw[n, k] = dy[m, n]^T x[m, k]
            dy = torch.randn(m, n, device=dev, generator=g)
            x = torch.randn(m, k, device=dev, generator=g)
            out = torch.empty(*((taps, n, k) if taps > 1 else (n, k)), device=dev)
            fn = lambda: H.linear_bwd_weight(dy, x, out, taps=taps, T=T)
        t = timeit(fn)
        fl = 2.0 * m * n * k * taps
        tot_t += t
        tot_f += fl
        print(f"{name:14s} {kind} M={m:6d} N={n:5d} K={k * taps:5d}  {t * 1e6:8.1f} us  {fl / t / 1e12:6.1f} TF/s", flush=True)
    print(f"total {tot_f / tot_t / 1e12:.1f} TF/s")


if __name__ == "__main__":
    main()
