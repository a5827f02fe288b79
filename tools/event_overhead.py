"""What a HIP event pair around ONE launch measures beyond the kernel: events around a near-empty launch, around two
of them, and the rocprof-visible kernel time of that launch for comparison (run under rocprofv3 for the latter)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

x = torch.zeros(64, device="cuda")
y = torch.empty_like(x)
big = torch.randn(20736, 256, device="cuda"); w = torch.randn(1024, 256, device="cuda"); out = torch.empty(20736, 1024, device="cuda")
H.linear_fwd(big, w, out=out)
for n in (1, 2, 4):
    tot = 0.0
    for _ in range(200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            H.axpby(x, None, 1.0, 0.0, out=y)
        e1.record()
        e1.synchronize()
        tot += e0.elapsed_time(e1)
    print(f"events around {n} tiny launch(es): {tot / 200 * 1e3:.2f} us")
# a 100 us GEMM alone vs inside a queue of other work
torch.cuda.synchronize()
for label, pre in (("idle queue", 0), ("busy queue", 3)):
    tot = 0.0
    for _ in range(50):
        torch.cuda._sleep(200000)
        for _ in range(pre):
            H.linear_fwd(big, w, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        H.linear_fwd(big, w, out=out)
        e1.record()
        for _ in range(pre):
            H.linear_fwd(big, w, out=out)
        e1.synchronize()
        tot += e0.elapsed_time(e1)
    print(f"events around one GEMM, {label}: {tot / 50 * 1e3:.2f} us")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(50):
    H.linear_fwd(big, w, out=out)
e1.record(); e1.synchronize()
print(f"same GEMM, 50 back to back under one event pair: {e0.elapsed_time(e1) / 50 * 1e3:.2f} us each")
