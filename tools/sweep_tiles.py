"""Every workgroup tile on the step's main GEMM shapes, one line per (shape, tile): where each tile stands, in isolation
(GPU only).  usage: python tools/sweep_tiles.py [32-true|bf16-stored] [batch]
The shapes are the decoder's (M = batch x 648 rows), the encoder's (batch x 128) and the PostNet's, forward /
data gradient / weight gradient with the epilogues the step uses."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fastspeech2_lightning_amd import hip as H  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "32-true"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if prec != "32-true" else 32)
stored = prec == "bf16-stored"
H.set_precision("bf16-mixed" if stored else prec)
dev = "cuda"
H.GEMM_TUNE = True
drop = H.Drop(0.2, 5)


def sweep(label, flops, fn):
    H._TILE_CACHE.clear(); H._TILE_TIMINGS.clear()
    fn()
    torch.cuda.synchronize()
    (key, tim), = H._TILE_TIMINGS.items()
    best = tim[0][0]
    cells = "  ".join(f"{t}:{ms / 4 * 1e3:6.1f}us" for ms, t in sorted(tim, key=lambda x: x[1]))
    print(f"{label:44s} best tile {tim[0][1]:2d} {best / 4 * 1e3:7.1f} us {flops / (best / 4 * 1e-3) / 1e12:7.1f} TF | {cells}", flush=True)


def cast(t):
    return H.cast_bf16(t) if stored else t


odt = torch.bfloat16 if stored else torch.float32
for tag, M in (("dec", B * 648), ("enc", B * 128)):
    for name, N, K in (("ffn1", 1024, 256), ("ffn2", 256, 1024), ("qkv", 768, 256), ("proj", 256, 256), ("pw1", 512, 256)):
        x = cast(torch.randn(M, K, device=dev)); w = cast(torch.randn(N, K, device=dev) * K ** -0.5); b = torch.randn(N, device=dev)
        dy = cast(torch.randn(M, N, device=dev)); res = torch.randn(M, N, device=dev)
        dw = torch.empty(N * K, device=dev); bg = torch.empty(N, device=dev)
        fl = 2.0 * M * N * K
        if name == "ffn1":
            u = torch.empty(M, N, device=dev, dtype=odt)
            sweep(f"{tag} {name} fwd silu+drop+pre  {M}x{N}x{K}", fl,
                  lambda: H.linear_fwd(x, w, b, epi=H.EPI_ACT, act="silu", out_pre=u, drop=drop, out_dtype=odt))
            sweep(f"{tag} {name} dgrad               {M}x{K}x{N}", fl, lambda: H.linear_bwd_data(dy, w, out_dtype=odt))
        elif name in ("ffn2", "proj"):
            sweep(f"{tag} {name} fwd resid+drop     {M}x{N}x{K}", fl,
                  lambda: H.linear_fwd(x, w, b, epi=H.EPI_RESID, resid=res, res_scale=0.5, drop=drop))
            if name == "ffn2":
                aux = cast(torch.randn(M, K, device=dev))
                sweep(f"{tag} {name} dgrad dact+drop    {M}x{K}x{N}", fl,
                      lambda: H.linear_bwd_data(dy, w, epi=H.EPI_DACT, act="silu", aux=aux, drop=drop, out_dtype=odt))
            else:
                sweep(f"{tag} {name} dgrad               {M}x{K}x{N}", fl, lambda: H.linear_bwd_data(dy, w, out_dtype=odt))
        else:
            sweep(f"{tag} {name} fwd bias            {M}x{N}x{K}", fl, lambda: H.linear_fwd(x, w, b, out_dtype=odt))
            sweep(f"{tag} {name} dgrad               {M}x{K}x{N}", fl, lambda: H.linear_bwd_data(dy, w, out_dtype=odt))
        sweep(f"{tag} {name} wgrad+bias          {N}x{K}x{M}", fl, lambda: (H.linear_bwd_weight(dy, x, dw, bias_grad=bg), H.drop_pending_reductions()))
T, C = 648, 512
M = B * T
x = cast(torch.randn(M, C, device=dev)); w = cast(torch.randn(5, C, C, device=dev) * (5 * C) ** -0.5); b = torch.randn(C, device=dev)
dy = cast(torch.randn(M, C, device=dev)); dw = torch.empty(5 * C * C, device=dev); bg = torch.empty(C, device=dev)
fl = 2.0 * M * C * C * 5
sweep(f"postnet conv5 fwd   {M}x{C}x{5 * C}", fl, lambda: H.linear_fwd(x, w, b, taps=5, T=T, out_dtype=odt))
sweep(f"postnet conv5 dgrad {M}x{C}x{5 * C}", fl, lambda: H.linear_bwd_data(dy, w, taps=5, T=T, out_dtype=odt))
sweep(f"postnet conv5 wgrad {C}x{C}x5x{M}", fl, lambda: (H.linear_bwd_weight(dy, x, dw, taps=5, T=T, bias_grad=bg), H.drop_pending_reductions()))
