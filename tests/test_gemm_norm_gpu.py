"""GPU parity of the GEMM family and LayerNorm against plain PyTorch fp32 (CPU) references.
Tolerances: fp32 MFMA is a k-ordered fmaf chain; against a differently-ordered fp32 sum the
error bound is ~K * eps * |a||b|, so results are compared at 2e-5 * K^0.5 of the output scale."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.tuned_tiles]


@pytest.fixture(scope="module")
def H():
    import os
    from fastspeech2_lightning_amd import hip
    assert torch.cuda.is_available()
    hip.lib()
    if os.environ.get("FS2_TEST_GEMM_TILES"):  # e.g. "4" or "5,6": restrict the autotuner to these tiles
        hip.GEMM_TILES = tuple(int(t) for t in os.environ["FS2_TEST_GEMM_TILES"].split(","))
        hip._TILE_CACHE.clear()
    return hip


def close(a, b, K=256, msg=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = max(float(b.abs().max()), 1e-6)
    tol = 4e-6 * math.sqrt(K) + 1e-6
    err = float((a - b).abs().max()) / scale
    assert err < tol, f"{msg}: rel err {err:.3e} > {tol:.3e}"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (2048, 1024, 256), (513, 80, 256), (130, 768, 64), (64, 64, 16)])
def test_linear_fwd_plain(H, M, N, K):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    y = H.linear_fwd(x.cuda(), w.cuda(), b.cuda())
    close(y, F.linear(x, w, b), K, "linear")


def test_linear_fwd_act_and_resid(H):
    M, N, K = 777, 256, 1024
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    pre = torch.empty(M, N, device="cuda")
    y = H.linear_fwd(x.cuda(), w.cuda(), b.cuda(), epi=H.EPI_ACT, act="silu", out_pre=pre)
    ref = F.linear(x, w, b)
    close(pre, ref, K, "pre")
    close(y, F.silu(ref), K, "silu")
    y = H.linear_fwd(x.cuda(), w.cuda(), b.cuda(), epi=H.EPI_ACT, act="relu")
    close(y, F.relu(ref), K, "relu")
    y = H.linear_fwd(x.cuda(), w.cuda(), b.cuda(), epi=H.EPI_RESID, resid=r.cuda(), res_scale=0.5)
    close(y, r + 0.5 * ref, K, "resid")


@pytest.mark.parametrize("taps,Cin,Cout", [(5, 80, 512), (5, 512, 80), (3, 256, 256), (9, 64, 128)])
def test_conv_taps_fwd_bwd(H, taps, Cin, Cout):
    B, T = 3, 37
    x = rnd(B, T, Cin, seed=1)
    w = rnd(Cout, Cin, taps, seed=2, scale=(Cin * taps) ** -0.5)
    b = rnd(Cout, seed=3)
    x.requires_grad_(True); w.requires_grad_(True)
    ref = F.conv1d(x.transpose(1, 2), w, b, padding=(taps - 1) // 2).transpose(1, 2)
    wp = w.detach().permute(2, 0, 1).contiguous().cuda()  # [taps, Cout, Cin]
    y = H.linear_fwd(x.detach().cuda(), wp, b.cuda(), taps=taps, T=T)
    close(y, ref, Cin * taps, "conv fwd")
    dy = rnd(B, T, Cout, seed=5)
    ref.backward(dy)
    dx = H.linear_bwd_data(dy.cuda(), wp, taps=taps, T=T)
    close(dx, x.grad, Cout * taps, "conv bwd data")
    dw = torch.empty(taps, Cout, Cin, device="cuda")
    H.linear_bwd_weight(dy.cuda(), x.detach().cuda(), dw, taps=taps, T=T)
    close(dw, w.grad.permute(2, 0, 1), B * T, "conv bwd weight")


@pytest.mark.parametrize("tile", [13, 14, 15, 12, 11, 10])
def test_persistent_and_split_tail_tiles_at_full_size(H, tile):
    """Benchmark-size shapes (more tiles than workgroup slots, with a partial last round): the persistent cores and
    the variants that cut the tail tiles along the reduction (finished by the fix-up pass), forced one at a time.
    Forward GEMM with bias, backward-data GEMM, and a 5-tap convolution whose tail slices start inside a tap."""
    saved = H.GEMM_TILES, dict(H._TILE_CACHE)
    try:
        H.GEMM_TILES = (tile,)
        H._TILE_CACHE.clear()
        M, N, K = 20736 + 40, 512, 256  # ragged last M-tile as well
        x, w, b = rnd(M, K, seed=1).cuda(), rnd(N, K, seed=2, scale=K ** -0.5).cuda(), rnd(N, seed=3).cuda()
        y = H.linear_fwd(x, w, b)
        close(y, (x.double() @ w.double().t() + b.double()).float(), K, f"tile {tile} fwd")
        w4, b4 = rnd(1024, K, seed=8, scale=K ** -0.5).cuda(), rnd(1024, seed=9).cuda()  # 128x64 tiles: 304-tile tail
        y4 = H.linear_fwd(x, w4, b4)
        close(y4, (x.double() @ w4.double().t() + b4.double()).float(), K, f"tile {tile} fwd N=1024")
        dy, w2 = rnd(M, N, seed=4).cuda(), rnd(N, 512, seed=7, scale=N ** -0.5).cuda()
        dx = H.linear_bwd_data(dy, w2)
        close(dx, (dy.double() @ w2.double()).float(), N, f"tile {tile} bwd data")
        B, T, Cin, Cout, taps = 32, 648, 64, 512, 5
        xc = rnd(B, T, Cin, seed=5).cuda()
        wc = rnd(Cout, Cin, taps, seed=6, scale=(Cin * taps) ** -0.5).cuda()
        ref = F.conv1d(xc.transpose(1, 2).double(), wc.double(), b.double(), padding=2).transpose(1, 2).float()
        yc = H.linear_fwd(xc, wc.permute(2, 0, 1).contiguous(), b, taps=taps, T=T)
        close(yc, ref, Cin * taps, f"tile {tile} conv fwd")
        # the forced tile really ran (0 = the tile does not take that shape and the built-in heuristic chose)
        assert set(H._TILE_CACHE.values()) <= {tile, 0} and tile in H._TILE_CACHE.values(), H._TILE_CACHE
    finally:
        H.GEMM_TILES = saved[0]
        H._TILE_CACHE.clear()
        H._TILE_CACHE.update(saved[1])


@pytest.mark.parametrize("tile", [13, 14, 15])
def test_split_tail_tiles_with_fused_epilogues(H, tile):
    """The tail tiles of tiles 13-15 are finished by a second pass that has to reproduce the fused epilogue: SiLU with the
    pre-activation output and dropout (FFN first linear), residual + dropout (FFN second linear), activation derivative
    + dropout (backward through the first linear).  Reference = the same call on a one-tile-per-workgroup kernel
    (tile 8) with the same dropout seed: the masks are functions of the element index, so the two results must agree
    element by element to fp32 summation-order tolerance, zeros included."""
    saved = H.GEMM_TILES, dict(H._TILE_CACHE)
    M = 20736 + 40
    step = torch.zeros(4, dtype=torch.int64, device="cuda")
    drop = H.Drop(0.2, 4242, step)
    # reductions of 512 and more: slices of an epilogue-carrying GEMM are at least 8 K-tiles (256) long
    x, w1, b1 = rnd(M, 512, seed=1).cuda(), rnd(1024, 512, seed=2, scale=1 / 22).cuda(), rnd(1024, seed=3).cuda()
    a, w2, b2 = rnd(M, 1024, seed=4).cuda(), rnd(512, 1024, seed=5, scale=1 / 32).cuda(), rnd(512, seed=6).cuda()
    r2 = rnd(M, 512, seed=7).cuda()
    dz, w3, u = rnd(M, 512, seed=8).cuda(), rnd(512, 1024, seed=9, scale=1 / 22).cuda(), rnd(M, 1024, seed=10).cuda()

    def run():
        pre = torch.empty(M, 1024, device="cuda")
        y1 = H.linear_fwd(x, w1, b1, epi=H.EPI_ACT, act="silu", out_pre=pre, drop=drop)
        y2 = H.linear_fwd(a, w2, b2, epi=H.EPI_RESID, resid=r2, res_scale=0.5, drop=drop)
        y3 = H.linear_bwd_data(dz, w3, epi=H.EPI_DACT, act="silu", aux=u, alpha=0.5, drop=drop)
        return dict(pre=pre, act=y1, resid=y2, dact=y3)

    try:
        H.GEMM_TILES = (8,)
        H._TILE_CACHE.clear()
        ref = run()
        H.GEMM_TILES = (tile,)
        H._TILE_CACHE.clear()
        got = run()
        ran = sorted(H._TILE_CACHE.values())
        assert tile in ran, (tile, H._TILE_CACHE)  # at least one of the three shapes has a tail this tile cuts
        for k, K in (("pre", 512), ("act", 512), ("resid", 1024), ("dact", 512)):
            close(got[k], ref[k], K, f"tile {tile} {k}")
            assert torch.equal(got[k] == 0, ref[k] == 0), f"tile {tile} {k}: dropout pattern differs"
    finally:
        H.GEMM_TILES = saved[0]
        H._TILE_CACHE.clear()
        H._TILE_CACHE.update(saved[1])


@pytest.mark.parametrize("M,N,K", [(4100, 256, 1024), (333, 1024, 256), (20000, 80, 256)])
def test_linear_bwd(H, M, N, K):
    x, w, dy = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(M, N, seed=3)
    dx = H.linear_bwd_data(dy.cuda(), w.cuda())
    close(dx, dy @ w, N, "dx")
    dw = torch.empty(N, K, device="cuda")
    H.linear_bwd_weight(dy.cuda(), x.cuda(), dw)
    close(dw, dy.t() @ x, M, "dw")
    db = torch.empty(N, device="cuda")
    H.colsum(dy.cuda(), db)
    close(db, dy.sum(0), M, "db")
    # backward through an activation (EPI_DACT)
    pre = rnd(M, K, seed=9)
    du = H.linear_bwd_data(dy.cuda(), w.cuda(), epi=H.EPI_DACT, act="silu", aux=pre.cuda(), alpha=0.5)
    p = pre.clone().requires_grad_(True)
    F.silu(p).backward(0.5 * (dy @ w))
    close(du, p.grad, N, "dact")


def test_dropout_epilogue_statistics_and_replay(H):
    M, N, K = 512, 256, 64
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2)
    y0 = H.linear_fwd(x.cuda(), w.cuda())
    y1 = H.linear_fwd(x.cuda(), w.cuda(), epi=H.EPI_ACT, act="none", drop=H.Drop(0.2, 77))
    y2 = H.linear_fwd(x.cuda(), w.cuda(), epi=H.EPI_ACT, act="none", drop=H.Drop(0.2, 77))
    assert torch.equal(y1, y2)  # same seed -> same mask (the backward regenerates it)
    kept = (y1 != 0).float().mean().item()
    assert abs(kept - 0.8) < 0.01
    m = y1 != 0
    close(y1[m], y0[m] / 0.8, K, "scaled survivors")
    # the same mask multiplies the gradient in EPI_DACT
    g = H.linear_bwd_data(torch.ones(M, K, device="cuda"), torch.ones(K, N, device="cuda"),
                          epi=H.EPI_DACT, act="none", aux=torch.zeros(M, N, device="cuda"), drop=H.Drop(0.2, 77))
    assert torch.equal(g != 0, y1 != 0)
    assert g.shape == (M, N)


@pytest.mark.parametrize("M,C", [(1000, 256), (37, 512), (5, 80), (129, 1024)])
def test_layernorm(H, M, C):
    x, g, b, dy, add = rnd(M, C, seed=1), 1 + 0.1 * rnd(C, seed=2), rnd(C, seed=3), rnd(M, C, seed=4), rnd(M, C, seed=5)
    y, mean, rstd = H.layernorm_fwd(x.cuda(), g.cuda(), b.cuda())
    xr, gr, br = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-5)
    close(y, ref, 16, "ln fwd")
    ref.backward(dy)
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dx = H.layernorm_bwd(dy.cuda(), x.cuda(), g.cuda(), mean, rstd, dg, db, dx_add=add.cuda())
    close(dx, xr.grad + add, 16, "ln dx")
    close(dg, gr.grad, M, "ln dgamma")
    close(db, br.grad, M, "ln dbeta")


@pytest.mark.parametrize("M,C,p", [(1000, 256, 0.1), (37, 512, 0.0), (5, 80, 0.5), (129, 1024, 0.2)])
def test_layernorm_bwd_second_output(H, M, C, p):
    """fs2hip_layernorm_bwd_dz: dx, dgamma, dbeta as the plain backward; dz = scale * dropmask * dx with exactly the
    mask fs2hip_axpby draws for the same Drop record (bit-identical: same products in the same order), and dz's column
    sums, finished with the deferred LayerNorm parameter sums."""
    x, g, b, dy, add = rnd(M, C, seed=1), 1 + 0.1 * rnd(C, seed=2), rnd(C, seed=3), rnd(M, C, seed=4), rnd(M, C, seed=5)
    _, mean, rstd = H.layernorm_fwd(x.cuda(), g.cuda(), b.cuda())
    dg0, db0 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dx0 = H.layernorm_bwd(dy.cuda(), x.cuda(), g.cuda(), mean, rstd, dg0, db0, dx_add=add.cuda())
    drop = H.Drop(p, 31337) if p > 0 else H.NO_DROP
    dg, db, dzsum = (torch.full((C,), float("nan"), device="cuda") for _ in range(3))
    dx, dz = H.layernorm_bwd(dy.cuda(), x.cuda(), g.cuda(), mean, rstd, dg, db, dx_add=add.cuda(), defer=True,
                             dz_scale=0.5, dz_drop=drop, dz_colsum=dzsum)
    H.flush_grad_reductions()
    assert torch.equal(dx, dx0)
    want = H.axpby(dx0, None, 0.5, 0.0, drop)
    assert torch.equal(dz, want)
    if p > 0:
        assert abs((dz != 0).float().mean().item() - (1 - p)) < (0.15 if M * C < 1000 else 0.02)
    close(dg, dg0, M, "dgamma")
    close(db, db0, M, "dbeta")
    close(dzsum, want.double().sum(0).float(), M, "colsum(dz)")


@pytest.mark.parametrize("M,N,K,taps,T", [(20736, 256, 1024, 1, 0), (20736, 1024, 256, 1, 0), (4096, 80, 256, 1, 0),
                                           (2592, 512, 512, 5, 648), (4000, 256, 84, 1, 0), (1024, 64, 64, 1, 0)])
@pytest.mark.parametrize("tile", [None, 3, 6, 7])
def test_split_reduction_finished_in_kernel_is_bit_identical(H, monkeypatch, M, N, K, taps, T, tile):
    """Weight gradients cut along the reduction: the last workgroup of every output tile sums the tile's slabs itself
    (Fs2GemmArgs.counters).  Same slabs, same order as fs2hip_reduce_slabs -> the same bits on the same tile (tile None:
    the autotuner may pick different tiles for the two variants, whose summation orders differ -> fp32 tolerance); and
    the counters come back to zero, so the next launch on the stream (here: the same one, three times) starts clean."""
    dy, x = rnd(M, N, seed=7).cuda(), rnd(M, K, seed=8).cuda()
    if H.pick_splitk(N, K, M, taps) == 1:
        pytest.skip("shape is not split")
    saved = H.GEMM_TILES
    try:
        if tile is not None:
            H.GEMM_TILES = (tile,)
        H._TILE_CACHE.clear()
        want = torch.empty(taps * N * K, device="cuda")
        monkeypatch.setattr(H, "SPLITK_IN_KERNEL", False)
        H.linear_bwd_weight(dy, x, want, taps=taps, T=T)
        monkeypatch.setattr(H, "SPLITK_IN_KERNEL", True)
        for _ in range(3):
            got = torch.full_like(want, float("nan"))
            H.linear_bwd_weight(dy, x, got, taps=taps, T=T)
            if tile is not None:
                assert torch.equal(got, want)
            else:
                close(got, want, M, "in-kernel finish, tuned tile")
        assert int(H.splitk_counters(dy.device).abs().sum()) == 0
    finally:
        H.GEMM_TILES = saved
        H._TILE_CACHE.clear()


def test_predictor_layer_layernorm_fusions_equal_the_separate_launches():
    """``fs2hip_layernorm_fwd_drop`` = LayerNorm then the dropout pass; ``fs2hip_layernorm_bwd_pred`` = the dropout pass over
    dy, LayerNorm backward, then relu' of the normalised ReLU output -- the variance predictors' layers
    (fs2/layers.py:30-48) in two launches instead of five, bit for bit (same mask element index, same arithmetic)."""
    import torch
    from fastspeech2_lightning_amd import hip as H
    g = torch.Generator().manual_seed(5)
    for M, C in ((4096, 256), (777, 64), (33, 1024)):
        x = torch.relu(torch.randn(M, C, generator=g)).cuda()
        gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
        dy = torch.randn(M, C, generator=g).cuda()
        step = torch.full((1,), 7, dtype=torch.int64, device="cuda")
        for p in (0.5, 0.0):
            drop = H.Drop(p, 0x1234, step) if p else H.NO_DROP
            n, mean, rstd = H.layernorm_fwd(x, gamma, beta)
            want = H.axpby(n, None, 1.0, 0.0, drop) if p else n
            got, mean2, rstd2 = H.layernorm_fwd_drop(x, gamma, beta, drop)
            assert torch.equal(got, want) and torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
            dg1, db1, dg2, db2 = (torch.zeros(C, device="cuda") for _ in range(4))
            d = H.axpby(dy, None, 1.0, 0.0, drop) if p else dy
            d = H.layernorm_bwd(d, x, gamma, mean, rstd, dg1, db1)
            want_dx = H.dact_mul(d, x, "relu")
            got_dx = H.layernorm_bwd_pred(dy, x, gamma, mean, rstd, dg2, db2, drop)
            H.flush_grad_reductions()
            assert torch.equal(got_dx, want_dx)
            assert torch.allclose(dg1, dg2, rtol=1e-5, atol=1e-5) and torch.allclose(db1, db2, rtol=1e-5, atol=1e-5)
