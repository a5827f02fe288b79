"""One bf16-storage GEMM case, a few launches (for rocprofv3 --pmc passes).  usage: probe_gemm_bf16.py CASE TILE [M]
CASE: store_bf16 | store_fp32 | ffn1 | resid | dgrad | wgrad"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

case, tile = sys.argv[1], int(sys.argv[2])
M = int(sys.argv[3]) if len(sys.argv) > 3 else 41472
N, K = 1024, 256
H.set_precision("bf16-mixed")
H.GEMM_TILES_B = (tile,)
dev, bf = "cuda", torch.bfloat16
x = torch.randn(M, K, device=dev).to(bf)
w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
b = torch.randn(N, device=dev)
u = torch.empty(M, N, device=dev, dtype=bf)
res = torch.randn(M, N, device=dev)
dy = torch.randn(M, N, device=dev).to(bf)
dw, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
drop = H.Drop(0.2, 5, torch.zeros(1, dtype=torch.int64, device=dev))
fn = {
    "store_bf16": lambda: H.linear_fwd(x, w, b, out_dtype=bf),
    "store_fp32": lambda: H.linear_fwd(x, w, b),
    "ffn1": lambda: H.linear_fwd(x, w, b, epi=H.EPI_ACT, act="silu", out_pre=u, drop=drop, out_dtype=bf),
    "resid": lambda: H.linear_fwd(x, w, b, epi=H.EPI_RESID, resid=res, drop=drop),
    "dgrad": lambda: H.linear_bwd_data(dy, w, out_dtype=bf),
    "wgrad": lambda: H.linear_bwd_weight(dy, x, dw, bias_grad=db),
}[case]
for _ in range(6):
    fn()
    H._PENDING_REDUCTIONS.clear()
torch.cuda.synchronize()
