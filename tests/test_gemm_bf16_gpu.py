"""GPU parity of the "bf16-mixed" GEMM mode (Fs2GemmArgs.operand_bf16): operands are rounded to bf16 (round to
nearest even) in registers, products and sums are fp32.  The reference rounds the same operands with torch
(``.bfloat16()`` is RNE as well) and multiplies in fp64, so what remains is fp32 accumulation error: the same
4e-6 * sqrt(K) bound as the fp32 tests -- a wrong k-slot pairing between the A and B fragments, a dropped
reduction group or a fp32 fall-back would all show at 1e-3 or more."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.tuned_tiles]


@pytest.fixture()
def H():
    from fastspeech2_lightning_amd import hip
    assert torch.cuda.is_available()
    hip.lib()
    saved = hip.GEMM_TILES, dict(hip._TILE_CACHE), hip.get_precision()
    hip.set_precision("bf16-mixed")
    yield hip
    hip.set_precision(saved[2])
    hip.GEMM_TILES = saved[0]
    hip._TILE_CACHE.clear()
    hip._TILE_CACHE.update(saved[1])


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def q(t):  # the rounding the kernel applies to its operands
    return t.bfloat16().double()


def close(a, b, K, msg):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = max(float(b.abs().max()), 1e-6)
    tol = 4e-6 * math.sqrt(K) + 1e-6
    err = float((a - b).abs().max()) / scale
    assert err < tol, f"{msg}: rel err {err:.3e} > {tol:.3e}"


def test_precision_switch_is_validated(H):
    with pytest.raises(ValueError):
        H.set_precision("fp8")
    H.set_precision("32-true")
    assert H.get_precision() == "32-true" and not H.GEMM_BF16
    H.set_precision("bf16-mixed")
    assert H.get_precision() == "bf16-mixed" and H.GEMM_BF16


@pytest.mark.parametrize("tile", [4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15])
def test_bf16_operands_every_direct_to_lds_tile(H, tile):
    """Forward (NT), backward-data (NN), weight-gradient (TN, split-K) and a 5-tap convolution (forward, data and
    weight gradient) with one tile forced; ragged M/N edges and a reduction that is not a multiple of the K-tile."""
    H.GEMM_TILES = (tile,)
    H._TILE_CACHE.clear()
    M, N, K = 20736 + 40, 1024, 256
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    y = H.linear_fwd(x, w, b)
    close(y, q(x) @ q(w).t() + b.double(), K, f"tile {tile} fwd")
    dy, w2 = rnd(M, 512, seed=4), rnd(512, 272, seed=7, scale=512 ** -0.5)
    dx = H.linear_bwd_data(dy, w2)
    close(dx, q(dy) @ q(w2), 512, f"tile {tile} bwd data")
    xs, dys = rnd(4100, 272, seed=11), rnd(4100, 80, seed=12)
    dw = torch.empty(80, 272, device="cuda")
    H.linear_bwd_weight(dys, xs, dw)
    close(dw, q(dys).t() @ q(xs), 4100, f"tile {tile} bwd weight")
    # pre-activation output + SiLU epilogue read the fp32 accumulators
    pre = torch.empty(4100, 1024, device="cuda")
    xa = rnd(4100, K, seed=13)
    ya = H.linear_fwd(xa, w, b, epi=H.EPI_ACT, act="silu", out_pre=pre)
    ref = q(xa) @ q(w).t() + b.double()
    close(pre, ref, K, f"tile {tile} pre")
    close(ya, F.silu(ref), K, f"tile {tile} silu")
    B, T, Cin, Cout, taps = 8, 648, 64, 512, 5
    xc = rnd(B, T, Cin, seed=5)
    wc = rnd(Cout, Cin, taps, seed=6, scale=(Cin * taps) ** -0.5)
    bc = rnd(Cout, seed=8)
    xr = q(xc).requires_grad_(True)
    wr = q(wc).requires_grad_(True)
    ref = F.conv1d(xr.transpose(1, 2), wr, bc.double(), padding=2).transpose(1, 2)
    wp = wc.permute(2, 0, 1).contiguous()
    yc = H.linear_fwd(xc, wp, bc, taps=taps, T=T)
    close(yc, ref, Cin * taps, f"tile {tile} conv fwd")
    dyc = rnd(B, T, Cout, seed=9)
    # the gradients of the rounded-operand product w.r.t. its operands, with dy rounded as the kernel rounds it
    ref.backward(q(dyc))
    dxc = H.linear_bwd_data(dyc, wp, taps=taps, T=T)
    close(dxc, xr.grad, Cout * taps, f"tile {tile} conv bwd data")
    dwc = torch.empty(taps, Cout, Cin, device="cuda")
    H.linear_bwd_weight(dyc, xc, dwc, taps=taps, T=T)
    close(dwc, wr.grad.permute(2, 0, 1), B * T, f"tile {tile} conv bwd weight")
    assert all(k[9] == 1 for k in H._TILE_CACHE), "every launch above must carry operand_bf16"
    assert tile in set(H._TILE_CACHE.values()), H._TILE_CACHE


def test_bf16_mode_differs_from_fp32_by_operand_rounding_only(H):
    """Sanity on magnitudes: against the exact fp32 product the bf16-mixed result is off by ~2^-9 per operand."""
    M, N, K = 2048, 512, 256
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    y_bf = H.linear_fwd(x, w)
    H.set_precision("32-true")
    y_32 = H.linear_fwd(x, w)
    exact = x.double() @ w.double().t()
    e32 = float((y_32.double() - exact).abs().max())
    ebf = float((y_bf.double() - exact).abs().max())
    assert e32 < 1e-4 and 1e-4 < ebf < 5e-2, (e32, ebf)
