"""Host time per wrapper call (GPU held busy so that the host never waits)."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

dev = "cuda"
x = torch.randn(512, 256, device=dev); w = torch.randn(256, 256, device=dev); b = torch.randn(256, device=dev)
g = torch.ones(256, device=dev); out = torch.empty(512, 256, device=dev)
H.linear_fwd(x, w, b); H.layernorm_fwd(x, g, b); H.axpby(x, None, 1.0, 0.0)
torch.cuda.synchronize()
def t(fn, n=150):
    torch.cuda._sleep(int(2e9))
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt
print(f"linear_fwd        {t(lambda: H.linear_fwd(x, w, b)):6.2f} us")
print(f"linear_fwd(out=)  {t(lambda: H.linear_fwd(x, w, b, out=out)):6.2f} us")
print(f"layernorm_fwd     {t(lambda: H.layernorm_fwd(x, g, b)):6.2f} us")
print(f"axpby             {t(lambda: H.axpby(x, None, 1.0, 0.0)):6.2f} us")
print(f"torch.empty       {t(lambda: torch.empty(512, 256, device=dev)):6.2f} us")
L = H.lib()
s = H._stream()
px, po = x.data_ptr(), out.data_ptr()
print(f"raw ctypes axpby  {t(lambda: L.fs2hip_axpby(px, None, po, x.numel(), 1.0, 0.0, 0.0, 0, None, s)):6.2f} us")
