import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from fastspeech2_lightning_amd import hip as H
M = 20736
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N in (1024, 768, 256):
    x = torch.randn(M, 256, device="cuda"); w = torch.randn(N, 256, device="cuda") / 16; b = torch.randn(N, device="cuda")
    u = torch.empty(M, N, device="cuda"); drop = H.Drop(0.2, 5)
    for tile in (7, 32):
        H.GEMM_TILES = (tile,); H._TILE_CACHE.clear()
        a = t(lambda: H.linear_fwd(x, w, b))
        c = t(lambda: H.linear_fwd(x, w, b, epi=H.EPI_ACT, act="silu", out_pre=u, drop=drop))
        print(f"N={N} tile {tile}: store {a:6.1f} us ({2*M*N*256/a/1e6:5.1f} TF)   silu+drop+pre {c:6.1f} us", flush=True)
