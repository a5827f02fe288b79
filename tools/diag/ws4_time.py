"""Isolated timing of the K = 1024 GEMM forms by tile (GPU only).  usage: python tools/diag/ws4_time.py [rows=43008]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 43008
H.set_precision("bf16-mixed")
g = torch.Generator().manual_seed(0)
x = torch.randn(M, 1024, generator=g).bfloat16().cuda()
w = (torch.randn(256, 1024, generator=g) / 32).bfloat16().cuda()
b = torch.randn(256, generator=g).cuda()
r = torch.randn(M, 256, generator=g).cuda()
step = torch.zeros(1, dtype=torch.int64, device="cuda")
drop = H.Drop(0.2, 5, step)
big = torch.empty(512 * 1024 * 1024 // 4, device="cuda")  # cache flush between timings


def timed(fn, cold):
    """warm: 20 launches back to back (the host enqueues faster than they run); cold: one launch after a 512 MB write, minus
    the same bracket around nothing but the wrapper's host time is still inside -- compare tiles, not absolute numbers"""
    fn(); torch.cuda.synchronize()
    if not cold:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / 20
    ts = []
    for _ in range(6):
        big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


forms = {
    "ffn2 fwd resid+drop fp32": lambda: H.linear_fwd(x, w, b, epi=H.EPI_RESID, resid=r, res_scale=0.5, drop=drop),
    "ffn1 dgrad bf16": lambda: H.linear_bwd_data(x, None, out_dtype=torch.bfloat16, wt=w),
}
fl = 2.0 * M * 256 * 1024
for name, fn in forms.items():
    for tile in (20, 22, 33):
        saved = H.GEMM_TILES_B
        H.GEMM_TILES_B = (tile,)
        H._TILE_CACHE.clear()
        H.GEMM_TUNE = True
        try:
            warm, cold = timed(fn, False), timed(fn, True)
        finally:
            H.GEMM_TILES_B = saved
        print(f"{name:28s} M={M} tile {tile}: warm {warm:7.1f} us {fl / warm / 1e6:7.1f} TF | after a 512 MB write {cold:7.1f} us "
              f"{fl / cold / 1e6:7.1f} TF", flush=True)
