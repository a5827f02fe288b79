"""bf16 operand STORAGE (Fs2GemmArgs.operand_bf16 == 3): A and B are bf16 in memory, k-contiguous, and go HBM -> LDS ->
v_mfma_f32_32x32x16_bf16 without a conversion.  The result is the fp32-accumulated sum of exact bf16 x bf16 products,
so against float64 on the bf16 values the bound is the fp32 kernels' own (4e-6 sqrt(K)); every tile of the
one-tile-per-workgroup direct-to-LDS core, ragged edges, the 5-tap convolution, fused epilogues, the transposed-weight
form of the data gradient, and the casts that produce the operands."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.tuned_tiles]


@pytest.fixture(scope="module")
def H():
    from fastspeech2_lightning_amd import hip
    hip.lib()
    return hip


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def close(got, want, K, what):
    tol = 4e-6 * (K ** 0.5) * max(1.0, float(want.abs().max()))
    err = float((got.detach().cpu().double() - want.double()).abs().max())
    assert err < tol, f"{what}: max err {err:.3e} > {tol:.3e}"


def test_casts(H):
    x = rnd(1000, 264, seed=1)
    xb = H.cast_bf16(x.cuda())
    assert xb.dtype == torch.bfloat16 and torch.equal(xb.cpu(), x.bfloat16())
    w = rnd(264, 100, seed=2)
    wt = H.transpose_cast_bf16(w.cuda())
    assert wt.shape == (100, 264) and torch.equal(wt.cpu(), w.t().contiguous().bfloat16())


@pytest.mark.parametrize("tile", [4, 5, 6, 7, 8, 9, None])
@pytest.mark.parametrize("M,N,K", [(1000, 256, 256), (4100, 1024, 264), (129, 80, 64), (20736, 264, 1024)])
def test_forward_and_data_gradient(H, tile, M, N, K):
    saved = H.GEMM_TILES
    try:
        if tile is not None:
            H.GEMM_TILES = (tile,)
        H._TILE_CACHE.clear()
        x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
        xb, wb = x.bfloat16(), w.bfloat16()
        y = H.linear_fwd(xb.cuda(), wb.cuda(), b.cuda())
        close(y, xb.double() @ wb.double().t() + b.double(), K, "forward")
        # activation epilogue with the pre-activation output, residual epilogue
        u = torch.empty(M, N, device="cuda")
        a = H.linear_fwd(xb.cuda(), wb.cuda(), b.cuda(), epi=H.EPI_ACT, act="silu", out_pre=u)
        ref_u = (xb.double() @ wb.double().t() + b.double())
        close(u, ref_u, K, "pre-activation")
        close(a, F.silu(ref_u), K, "silu")
        r = rnd(M, N, seed=4)
        y2 = H.linear_fwd(xb.cuda(), wb.cuda(), b.cuda(), epi=H.EPI_RESID, resid=r.cuda(), res_scale=0.5)
        close(y2, r.double() + 0.5 * ref_u, K, "residual")
        # data gradient: dy [M, N] bf16 against the transposed weight [K, N] bf16
        if N % 8 == 0:
            dy = rnd(M, N, seed=5).bfloat16()
            wt = H.transpose_cast_bf16(w.cuda())  # [K, N]
            dx = H.linear_bwd_data(dy.cuda(), wt)
            close(dx, dy.double() @ wb.double(), N, "data gradient")
            aux = rnd(M, K, seed=6)
            du = H.linear_bwd_data(dy.cuda(), wt, epi=H.EPI_DACT, act="silu", aux=aux.cuda())
            xa = aux.double().requires_grad_(True)
            F.silu(xa).backward(dy.double() @ wb.double())
            close(du, xa.grad, N, "data gradient with act'")
    finally:
        H.GEMM_TILES = saved
        H._TILE_CACHE.clear()


@pytest.mark.parametrize("tile", [5, 8, 9, None])
def test_five_tap_convolution(H, tile):
    saved = H.GEMM_TILES
    try:
        if tile is not None:
            H.GEMM_TILES = (tile,)
        H._TILE_CACHE.clear()
        B, T, Cin, Cout, taps = 3, 77, 128, 192, 5
        x, w, b = rnd(B, T, Cin, seed=1), rnd(Cout, Cin, taps, seed=2, scale=(Cin * taps) ** -0.5), rnd(Cout, seed=3)
        xb, wb = x.bfloat16(), w.bfloat16()
        wk = wb.permute(2, 0, 1).contiguous()  # [taps, Cout, Cin]
        y = H.linear_fwd(xb.cuda().view(B * T, Cin), wk.cuda(), b.cuda(), taps=taps, T=T)
        ref = F.conv1d(xb.double().transpose(1, 2), wb.double(), b.double(), padding=2).transpose(1, 2)
        close(y.view(B, T, Cout), ref, Cin * taps, "conv forward")
    finally:
        H.GEMM_TILES = saved
        H._TILE_CACHE.clear()


def test_refused_shapes(H):
    x, w = rnd(64, 100, seed=1).bfloat16().cuda(), rnd(32, 100, seed=2).bfloat16().cuda()
    with pytest.raises((RuntimeError, ValueError)):
        H.linear_fwd(x, w)  # 100 bf16 per row: not whole 16-byte pieces
