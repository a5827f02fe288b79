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
    w3 = rnd(5, 72, 130, seed=3)
    w3t = H.transpose_cast_bf16(w3.cuda())
    assert w3t.shape == (5, 130, 72) and torch.equal(w3t.cpu(), w3.transpose(1, 2).contiguous().bfloat16())


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


def test_postnet_reads_bf16_operands_from_memory_in_bf16_mixed(H):
    """bf16-mixed, PostNet: the four 512-channel convolutions take their input (forward) and three of them their output
    gradient (data gradient) as the bf16 copy the BatchNorm kernels write, instead of rounding the fp32 tensor in
    registers.  Same rounded operands, same products; the summation order differs (64-deep K-tiles), and an fp32
    difference of 1e-6 upstream flips a bf16 rounding (4e-3) downstream here and there -> a whole train step (dropout on:
    the copies carry the masks) agrees with the register-rounding mode at the bf16 level: losses 1e-4, all gradients
    together 1e-2 relative L2 (measured 1.8e-3; the mode's bound against the fp32 oracle is 0.1), and the stored mode
    really ran."""
    from fastspeech2_lightning_amd.config import FastSpeech2Config, Stats
    from fastspeech2_lightning_amd.model import FastSpeech2
    from fastspeech2_lightning_amd.synthetic import DEFAULT_STATS, synthetic_batch
    conf = dict(layers=1)
    config = FastSpeech2Config(model=dict(encoder=conf, decoder=conf, learn_alignment=False),
                               text=dict(symbols=dict(letters=[f"s{i}" for i in range(40)])))
    batch = synthetic_batch(B=3, ts_lo=20, ts_hi=40, n_symbols=41, n_mels=80, seed=3, dur_hi=6)
    res = {}
    saved = H.BF16_STORAGE
    try:
        for stored in (False, True):
            H.BF16_STORAGE = stored
            H._TILE_CACHE.clear()
            model = FastSpeech2(config, Stats(**DEFAULT_STATS), seed=11, precision="bf16-mixed")
            model.train()
            model.training_step(batch)
            res[stored] = (dict(model.last_losses), {k: v.clone() for k, v in model.store.grad_state_dict().items()},
                           {key[9] for key in H._TILE_CACHE})
    finally:
        H.BF16_STORAGE = saved
        H._TILE_CACHE.clear()
    assert 3 in res[True][2] and 3 not in res[False][2], "operand_bf16 == 3 launches: only in the stored mode"
    for k, v in res[False][0].items():
        assert abs(float(res[True][0][k]) - float(v)) < 1e-4 * max(1.0, abs(float(v))), k
    num = sum(float((res[True][1][k] - g).pow(2).sum()) for k, g in res[False][1].items())
    den = sum(float(g.pow(2).sum()) for g in res[False][1].values())
    assert (num / den) ** 0.5 < 1e-2, (num / den) ** 0.5
