"""Where does an epilogue-heavy bf16-storage GEMM spend its time?  The FFN-1 forward shape (41 472 x 1024 x 256) with the
pieces of its epilogue switched on one at a time.  GPU only."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

H.set_precision("bf16-mixed")
dev, bf = "cuda", torch.bfloat16
M, N, K = 41472, 1024, 256
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 22
H.GEMM_TILES_B = (tile,)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


x = torch.randn(M, K, device=dev).to(bf)
w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
b = torch.randn(N, device=dev)
u = torch.empty(M, N, device=dev, dtype=bf)
u32 = torch.empty(M, N, device=dev)
res = torch.randn(M, N, device=dev)
drop = H.Drop(0.2, 5, torch.zeros(1, dtype=torch.int64, device=dev))
cases = [
    ("store fp32", lambda: H.linear_fwd(x, w, b)),
    ("store bf16", lambda: H.linear_fwd(x, w, b, out_dtype=bf)),
    ("relu bf16", lambda: H.linear_fwd(x, w, b, epi=H.EPI_ACT, act="relu", out_dtype=bf)),
    ("silu bf16", lambda: H.linear_fwd(x, w, b, epi=H.EPI_ACT, act="silu", out_dtype=bf)),
    ("silu + dropout bf16", lambda: H.linear_fwd(x, w, b, epi=H.EPI_ACT, act="silu", drop=drop, out_dtype=bf)),
    ("silu + pre-activation bf16", lambda: H.linear_fwd(x, w, b, epi=H.EPI_ACT, act="silu", out_pre=u, out_dtype=bf)),
    ("silu + dropout + pre-activation bf16", lambda: H.linear_fwd(x, w, b, epi=H.EPI_ACT, act="silu", out_pre=u, drop=drop, out_dtype=bf)),
    ("silu + dropout + pre-activation fp32", lambda: H.linear_fwd(x, w, b, epi=H.EPI_ACT, act="silu", out_pre=u32, drop=drop)),
    ("residual + dropout fp32", lambda: H.linear_fwd(x, w, b, epi=H.EPI_RESID, resid=res, drop=drop)),
]
print(f"tile {tile}, {M} x {N} x {K}")
for name, fn in cases:
    H._TILE_CACHE.clear()
    t = timeit(fn)
    print(f"  {name:40s} {t * 1e6:7.1f} us")
