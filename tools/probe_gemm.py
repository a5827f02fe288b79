"""Runs one GEMM shape on one workgroup tile N times (for rocprofv3 --pmc passes).
usage: python3 tools/probe_gemm.py KIND M N K TILE [launches]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

kind, m, n, k, tile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
launches = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = "cuda"
H.GEMM_TUNE = False
H._TILE_CACHE.clear()
import ctypes as C  # noqa: E402
orig = H._tune_tile
H._tune_tile = lambda a: tile
if kind == "nt":
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); out = torch.empty(m, n, device=dev)
    fn = lambda: H.linear_fwd(x, w, out=out)
elif kind == "nn":
    dy = torch.randn(m, k, device=dev); w = torch.randn(k, n, device=dev); out = torch.empty(m, n, device=dev)
    fn = lambda: H.linear_bwd_data(dy, w, out=out)
elif kind == "cdw":  # 5-tap conv weight gradient, rows = 32 utterances of m / 32 frames
    dy = torch.randn(m, n, device=dev); x = torch.randn(m, k, device=dev); out = torch.empty(5, n, k, device=dev)
    fn = lambda: H.linear_bwd_weight(dy, x, out, taps=5, T=m // 32)
else:
    dy = torch.randn(m, n, device=dev); x = torch.randn(m, k, device=dev); out = torch.empty(n, k, device=dev)
    fn = lambda: H.linear_bwd_weight(dy, x, out)
for _ in range(launches):
    fn()
torch.cuda.synchronize()
print("done", kind, m, n, k, tile)
