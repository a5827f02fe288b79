"""Kernel-efficiency probe: shapes whose tile count divides the CU count evenly."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = "cuda"
for m, n, k in [(16384, 512, 256), (16384, 512, 1024), (16384, 512, 4096), (16384, 1024, 256), (32768, 1024, 1024),
                (16384, 256, 256), (16384, 256, 1024), (16384, 256, 4096), (32768, 256, 256), (32768, 256, 2048)]:
    x = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev)
    out = torch.empty(m, n, device=dev)
    t = timeit(lambda: H.linear_fwd(x, w, out=out))
    tiles128 = (m // 128) * ((n + 127) // 128)
    print(f"nt M={m} N={n} K={k}: {t*1e6:8.1f} us {2.0*m*n*k/t/1e12:6.1f} TF/s  (128x128 tiles {tiles128}, 128x64 tiles {(m//128)*(n//64)})", flush=True)
