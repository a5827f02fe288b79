"""Does the fp32 MFMA rate of one GEMM hold when it runs for seconds instead of a burst?  (clock / power management)
PostNet-shaped 5-tap convolution, tile 13, TFLOP/s over bursts of growing length (GPU only)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

M, N, K, taps, T = 20736, 512, 512, 5, 648
x = torch.randn(M, K, device="cuda"); w = torch.randn(taps, N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
fl = 2.0 * M * N * K * taps
H.linear_fwd(x, w, taps=taps, T=T, out=out)
torch.cuda.synchronize()
for n in (10, 100, 1000, 4000, 10, 1000):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        H.linear_fwd(x, w, taps=taps, T=T, out=out)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"{n:5d} launches: {ms / n * 1e3:7.1f} us each, {fl * n / ms / 1e9:6.1f} TFLOP/s", flush=True)

# cold caches: a 1 GiB fill between launches (evicts L2 and the 256 MB Infinity Cache); event pair around the GEMM only
big = torch.empty(256 << 20, device="cuda")
for label, flush in (("warm", False), ("cold", True), ("warm", False), ("cold", True)):
    tot = 0.0
    for _ in range(20):
        if flush:
            big.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        H.linear_fwd(x, w, taps=taps, T=T, out=out)
        e1.record(); e1.synchronize()
        tot += e0.elapsed_time(e1)
    print(f"{label}: {tot / 20 * 1e3:7.1f} us per launch (event pair included)", flush=True)
