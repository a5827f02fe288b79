import sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
orig = H.pick_splitk
for M, N, K in ((20736, 256, 256), (20736, 1024, 256), (20736, 256, 1024), (20736, 768, 256), (20736, 512, 256), (4096, 256, 256), (4096, 1024, 256), (20736, 80, 256)):
    dy, x = torch.randn(M, N, device="cuda"), torch.randn(M, K, device="cuda")
    dw = torch.empty(N * K, device="cuda")
    line = f"M={M} N={N} K={K} default S={orig(N, K, M)}:"
    for S in (4, 8, 16, 32, 64):
        if M // S < 64: continue
        H.pick_splitk = lambda *a, S=S, **k: S
        H._TILE_CACHE.clear()
        t = timeit(lambda: H.linear_bwd_weight(dy, x, dw))
        line += f"  S={S}: {t*1e6:6.1f} us ({2.0*M*N*K/t/1e12:5.1f} TF)"
    print(line)
