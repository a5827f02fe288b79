"""The bf16-storage GEMM core (Fs2GemmArgs.operand_bf16 == 4, csrc/gemm_bf16.hip): A and B are bf16 in memory in ANY
orientation and go HBM -> LDS -> v_mfma_f32_32x32x16_bf16 without a conversion and without a transposed copy -- forward
(both k-contiguous), data gradient (the weight as stored: reduction-major, read with ds_read_b64_tr_b16), weight
gradient (both reduction-major, split-K, + the bias gradient from the same launch).  The result is the
fp32-accumulated sum of exact bf16 x bf16 products, so against float64 on the bf16 values the bound is the fp32
kernels' own (4e-6 sqrt(K)); bf16 results are that, rounded once.  Every tile, ragged edges, the 5-tap convolution in all
three forms, the fused epilogues with fp32 and bf16 outputs, dropout masks identical to the fp32 kernels'."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.tuned_tiles]

TILES = [20, 22, 23, 24, 25, None]


@pytest.fixture(scope="module")
def H():
    from fastspeech2_lightning_amd import hip
    hip.lib()
    return hip


class only_tile:
    def __init__(self, H, tile):
        self.H, self.tile = H, tile

    def __enter__(self):
        self.saved = self.H.GEMM_TILES_B
        if self.tile is not None:
            self.H.GEMM_TILES_B = (self.tile,)
        self.H._TILE_CACHE.clear()

    def __exit__(self, *exc):
        self.H.GEMM_TILES_B = self.saved
        self.H._TILE_CACHE.clear()
        return False


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def close(got, want, K, what, bf16_out=False):
    got = got.detach().cpu().double()
    scale = max(1.0, float(want.abs().max()))
    tol = 4e-6 * (K ** 0.5) * scale
    if bf16_out:  # one rounding to 8 significant bits on top
        err = float(((got - want.double()).abs() - want.double().abs() * 2.0 ** -8).clamp_min(0).max())
    else:
        err = float((got - want.double()).abs().max())
    assert err < tol, f"{what}: max err {err:.3e} > {tol:.3e}"


def test_casts(H):
    x = rnd(1000, 264, seed=1)
    xb = H.cast_bf16(x.cuda())
    assert xb.dtype == torch.bfloat16 and torch.equal(xb.cpu(), x.bfloat16())


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("M,N,K", [(1000, 256, 256), (4100, 1024, 264), (129, 80, 64), (20736, 264, 1024), (111, 96, 32)])
def test_forward_data_gradient_weight_gradient(H, tile, M, N, K):
    with only_tile(H, tile):
        x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
        xb, wb = x.bfloat16(), w.bfloat16()
        xd, wd = xb.cuda(), wb.cuda()
        ref_u = xb.double() @ wb.double().t() + b.double()
        y = H.linear_fwd(xd, wd, b.cuda())
        assert y.dtype == torch.float32
        close(y, ref_u, K, "forward")
        yb = H.linear_fwd(xd, wd, b.cuda(), out_dtype=torch.bfloat16)
        assert yb.dtype == torch.bfloat16
        close(yb, ref_u, K, "forward, bf16 result", bf16_out=True)
        # activation epilogue with the pre-activation output (fp32 and bf16), residual epilogue
        u = torch.empty(M, N, device="cuda")
        a = H.linear_fwd(xd, wd, b.cuda(), epi=H.EPI_ACT, act="silu", out_pre=u)
        close(u, ref_u, K, "pre-activation")
        close(a, F.silu(ref_u), K, "silu")
        ub = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ab = H.linear_fwd(xd, wd, b.cuda(), epi=H.EPI_ACT, act="silu", out_pre=ub, out_dtype=torch.bfloat16)
        close(ub, ref_u, K, "pre-activation, bf16", bf16_out=True)
        # the activation is applied to the ROUNDED pre-activation (what the backward pass reads back)
        close(ab, F.silu(ub.cpu().double()), K, "silu of the stored pre-activation", bf16_out=True)
        r = rnd(M, N, seed=4)
        y2 = H.linear_fwd(xd, wd, b.cuda(), epi=H.EPI_RESID, resid=r.cuda(), res_scale=0.5)
        close(y2, r.double() + 0.5 * ref_u, K, "residual")
        # data gradient: dy [M, N] bf16 against the weight AS STORED [N, K]
        dy = rnd(M, N, seed=5).bfloat16()
        dyd = dy.cuda()
        ref_dx = dy.double() @ wb.double()
        dx = H.linear_bwd_data(dyd, wd)
        close(dx, ref_dx, N, "data gradient")
        dxb = H.linear_bwd_data(dyd, wd, out_dtype=torch.bfloat16)
        close(dxb, ref_dx, N, "data gradient, bf16 result", bf16_out=True)
        for aux in (rnd(M, K, seed=6), rnd(M, K, seed=6).bfloat16()):
            du = H.linear_bwd_data(dyd, wd, epi=H.EPI_DACT, act="silu", aux=aux.cuda(), out_dtype=torch.bfloat16)
            xa = aux.double().requires_grad_(True)
            F.silu(xa).backward(ref_dx)
            close(du, xa.grad, N, f"data gradient with act' ({aux.dtype} aux)", bf16_out=True)
        # weight gradient (split-K inside) + bias gradient from the same launch
        dw = torch.empty(N, K, device="cuda")
        db = torch.full((N,), 7.0, device="cuda")
        H.linear_bwd_weight(dyd, xd, dw, bias_grad=db)
        H.flush_grad_reductions()
        close(dw, dy.double().t() @ xb.double(), M, "weight gradient")
        close(db, dy.double().sum(0), M, "bias gradient")


@pytest.mark.parametrize("tile", TILES)
def test_five_tap_convolution_all_three_forms(H, tile):
    with only_tile(H, tile):
        B, T, Cin, Cout, taps = 3, 77, 128, 192, 5
        x, w, b = rnd(B, T, Cin, seed=1), rnd(Cout, Cin, taps, seed=2, scale=(Cin * taps) ** -0.5), rnd(Cout, seed=3)
        xb, wb = x.bfloat16(), w.bfloat16()
        wk = wb.permute(2, 0, 1).contiguous().cuda()  # [taps, Cout, Cin]: the kernels' layout of a conv weight
        xd = xb.cuda().view(B * T, Cin)
        y = H.linear_fwd(xd, wk, b.cuda(), taps=taps, T=T)
        xr = xb.double().transpose(1, 2).requires_grad_(True)
        wr = wb.double().requires_grad_(True)
        ref = F.conv1d(xr, wr, b.double(), padding=2)
        close(y.view(B, T, Cout), ref.transpose(1, 2), Cin * taps, "conv forward")
        dy = rnd(B, T, Cout, seed=4).bfloat16()
        ref.backward(dy.double().transpose(1, 2))
        dyd = dy.cuda().view(B * T, Cout)
        if Cout % 64 == 0:
            dx = H.linear_bwd_data(dyd, wk, taps=taps, T=T)
            close(dx.view(B, T, Cin), xr.grad.transpose(1, 2), Cout * taps, "conv data gradient")
        dw = torch.empty(taps, Cout, Cin, device="cuda")
        db = torch.empty(Cout, device="cuda")
        H.linear_bwd_weight(dyd, xd, dw, taps=taps, T=T, bias_grad=db)
        H.flush_grad_reductions()
        close(dw, wr.grad.permute(2, 0, 1), B * T, "conv weight gradient")
        close(db, dy.double().sum((0, 1)), B * T, "conv bias gradient")


def test_dropout_masks_are_the_fp32_kernels(H):
    """Same site, same step, same element index -> the same mask whichever core (and whichever result type) wrote the
    element: forward and backward kernels of different precisions can share a site."""
    M, N, K = 777, 256, 128
    x, w = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, scale=K ** -0.5).bfloat16()
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    drop = H.Drop(0.3, 0x1234567, step)
    y32 = H.linear_fwd(x.float().cuda(), w.float().cuda(), epi=H.EPI_ACT, act="relu", drop=drop)
    yb = H.linear_fwd(x.cuda(), w.cuda(), epi=H.EPI_ACT, act="relu", drop=drop, out_dtype=torch.bfloat16)
    yf = H.linear_fwd(x.cuda(), w.cuda(), epi=H.EPI_ACT, act="relu", drop=drop)
    ref = torch.relu(x.double() @ w.double().t())
    live = ref.abs() > 1e-3
    for name, got in (("bf16 result", yb), ("fp32 result", yf)):
        same = ((got.float().cpu() != 0) == (y32.cpu() != 0)) | ~live
        assert bool(same.all()), name
    keep = float((y32.cpu()[live] != 0).float().mean())
    assert abs(keep - 0.7) < 0.02, keep
    close(yf, y32.cpu().double(), K, "masked values")


def test_refused_shapes(H):
    x, w = rnd(64, 100, seed=1).bfloat16().cuda(), rnd(32, 100, seed=2).bfloat16().cuda()
    with pytest.raises((RuntimeError, ValueError)):
        H.linear_fwd(x, w)  # 100 bf16 per row: not whole 16-byte pieces
    x, w = rnd(64, 64, seed=1).bfloat16().cuda(), rnd(30, 64, seed=2).bfloat16().cuda()
    with pytest.raises((RuntimeError, ValueError)):
        H.linear_fwd(x, w)  # 30 output columns: not whole quads


@pytest.mark.parametrize("dropout", [0.0, 0.2])
def test_train_step_with_bf16_activation_storage(H, dropout):
    """bf16-mixed with operand STORAGE end to end (Conformer layers + PostNet; activations between GEMMs exist only in
    bf16, weights read from the bf16 mirror, bias gradients from the weight-gradient GEMMs) against the same mode with
    fp32 storage and register rounding (``FS2_BF16_STORAGE=0``): both are bf16 products with fp32 accumulation of the
    same network, they differ by where values are rounded (stored activations are rounded once more than operands
    rounded on the fly), so a train step agrees at the bf16 level -- losses 2e-3, all gradients together 3e-2 relative
    L2 (the mode's own bound against the fp32 oracle is 0.1) -- and the storage core really ran."""
    from fastspeech2_lightning_amd.config import FastSpeech2Config, Stats
    from fastspeech2_lightning_amd.model import FastSpeech2
    from fastspeech2_lightning_amd.synthetic import DEFAULT_STATS, synthetic_batch
    conf = dict(layers=1, dropout=dropout)
    vp = dict(dropout=0.0)
    config = FastSpeech2Config(model=dict(encoder=conf, decoder=conf, learn_alignment=False,
                                          variance_predictors=dict(energy=vp, pitch=vp, duration=vp)),
                               text=dict(symbols=dict(letters=[f"s{i}" for i in range(40)])))
    batch = synthetic_batch(B=3, ts_lo=20, ts_hi=40, n_symbols=41, n_mels=80, seed=3, dur_hi=6)
    assert batch["mel"].shape[1] >= 64
    res = {}
    saved = H.BF16_STORAGE
    try:
        for stored in (False, True):
            H.BF16_STORAGE = stored
            H._TILE_CACHE.clear()
            model = FastSpeech2(config, Stats(**DEFAULT_STATS), seed=11, precision="bf16-mixed")
            model.train()
            with torch.no_grad():
                model.training_step(batch)
            res[stored] = (dict(model.last_losses), {k: v.clone() for k, v in model.store.grad_state_dict().items()},
                           {key[9] for key in H._TILE_CACHE})
    finally:
        H.BF16_STORAGE = saved
        H._TILE_CACHE.clear()
    assert 4 in res[True][2] and 4 not in res[False][2], "operand_bf16 == 4 launches: only in the stored mode"
    for k, v in res[False][0].items():
        assert abs(float(res[True][0][k]) - float(v)) < 2e-3 * max(1.0, abs(float(v))), (k, float(res[True][0][k]), float(v))
    num = sum(float((res[True][1][k] - g).pow(2).sum()) for k, g in res[False][1].items())
    den = sum(float(g.pow(2).sum()) for g in res[False][1].values())
    print(f"\nstored vs register-rounding bf16-mixed: gradient relative L2 {(num / den) ** 0.5:.4f}")
    assert (num / den) ** 0.5 < 3e-2, (num / den) ** 0.5
    # every bias gradient that now comes out of a weight-gradient GEMM
    for k in ("decoder.conformer_layers.0.ffn1.sequential.1.bias", "decoder.conformer_layers.0.self_attn.in_proj_bias",
              "decoder.conformer_layers.0.conv_module.sequential.0.bias", "decoder.conformer_layers.0.ffn2.sequential.1.bias"):
        a, b = res[True][1][k], res[False][1][k]
        assert float((a - b).norm() / b.norm().clamp_min(1e-12)) < 5e-2, k
