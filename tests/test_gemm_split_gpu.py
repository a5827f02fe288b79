"""precision "32-split": the GEMM family on the bf16 matrix pipe with every operand cut exactly into three bf16 planes
(csrc/gemm2_core.h).  The claim is fp32 accuracy, so the reference is float64 on the UNROUNDED fp32 operands and the
bar is the fp32 kernels' own: (a) the tolerance of tests/test_gemm_norm_gpu.py, 4e-6 * sqrt(K) of the output scale,
and (b) at most 3x the error the exact fp32 MFMA kernel makes on the same call (measured: about equal).  Every
direct-to-LDS tile (4-15), all three layouts, 5-tap convolutions forward / data / weight gradient, split-K, and the
fused epilogues.  A result that matches a bf16-ROUNDED reference better than the fp32 one fails (that would be mode 1)."""
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
    yield hip
    hip.set_precision(saved[2])
    hip.GEMM_TILES = saved[0]
    hip._TILE_CACHE.clear()
    hip._TILE_CACHE.update(saved[1])


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def err(a, ref64):
    return float((a.double().cpu() - ref64).abs().max()) / max(float(ref64.abs().max()), 1e-6)


SEEN = {}


def both(H, fn, tile):
    """fn() under the exact fp32 MFMA and under the split mode, the same tile forced (a shape the tile's core does not
    take runs on the library's own choice in both modes; SEEN records what really ran in split mode)."""
    out = {}
    for mode in ("32-true", "32-split"):
        H.set_precision(mode)
        H.GEMM_TILES = (tile,)
        H._TILE_CACHE.clear()
        out[mode] = fn()
    SEEN.setdefault(tile, []).extend((k[9], v) for k, v in H._TILE_CACHE.items())
    return out["32-true"], out["32-split"]


def check(exact, split, ref64, K, what):
    e_exact, e_split = err(exact, ref64), err(split, ref64)
    tol = 4e-6 * math.sqrt(K) + 1e-6
    assert e_split < tol, f"{what}: split error {e_split:.2e} > fp32 tolerance {tol:.2e}"
    assert e_split < 3 * e_exact + 2e-7, f"{what}: split error {e_split:.2e} vs exact fp32 MFMA {e_exact:.2e}"
    # and it is NOT the one-plane bf16 mode: that one is ~1e-3 off
    assert e_split < 2e-5 * math.sqrt(K), what


@pytest.mark.parametrize("tile", [4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15])
def test_split_mode_meets_the_fp32_bound_on_every_tile(H, tile):
    M, N, K = 20736 + 40, 1024, 256
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ex, sp = both(H, lambda: H.linear_fwd(x, w, b), tile)
    check(ex, sp, x.double().cpu() @ w.double().cpu().t() + b.double().cpu(), K, f"tile {tile} fwd")
    dy, w2 = rnd(M, 512, seed=4), rnd(512, 272, seed=7, scale=512 ** -0.5)
    ex, sp = both(H, lambda: H.linear_bwd_data(dy, w2), tile)
    check(ex, sp, dy.double().cpu() @ w2.double().cpu(), 512, f"tile {tile} bwd data")
    xs, dys = rnd(4100, 272, seed=11), rnd(4100, 80, seed=12)

    def wgrad():
        dw = torch.empty(80, 272, device="cuda")
        H.linear_bwd_weight(dys, xs, dw)
        return dw
    ex, sp = both(H, wgrad, tile)
    check(ex, sp, dys.double().cpu().t() @ xs.double().cpu(), 4100, f"tile {tile} bwd weight")
    # fused epilogues read the fp32 accumulators: SiLU + pre-activation output, residual
    xa, r = rnd(4100, K, seed=13), rnd(4100, N, seed=14)

    def act():
        pre = torch.empty(4100, N, device="cuda")
        y = H.linear_fwd(xa, w, b, epi=H.EPI_ACT, act="silu", out_pre=pre)
        y2 = H.linear_fwd(xa, w, b, epi=H.EPI_RESID, resid=r, res_scale=0.5)
        return torch.cat([pre, y, y2], 1)
    ref = xa.double().cpu() @ w.double().cpu().t() + b.double().cpu()
    ex, sp = both(H, act, tile)
    check(ex, sp, torch.cat([ref, F.silu(ref), r.double().cpu() + 0.5 * ref], 1), K, f"tile {tile} epilogues")
    # 5-tap convolution: forward, data gradient, weight gradient
    B, T, Cin, Cout, taps = 8, 648, 64, 512, 5
    xc, wc, bc = rnd(B, T, Cin, seed=5), rnd(Cout, Cin, taps, seed=6, scale=(Cin * taps) ** -0.5), rnd(Cout, seed=8)
    xr = xc.double().cpu().requires_grad_(True)
    wr = wc.double().cpu().requires_grad_(True)
    refc = F.conv1d(xr.transpose(1, 2), wr, bc.double().cpu(), padding=2).transpose(1, 2)
    dyc = rnd(B, T, Cout, seed=9)
    refc.backward(dyc.double().cpu())
    wp = wc.permute(2, 0, 1).contiguous()
    ex, sp = both(H, lambda: H.linear_fwd(xc, wp, bc, taps=taps, T=T), tile)
    check(ex, sp, refc.detach(), Cin * taps, f"tile {tile} conv fwd")
    ex, sp = both(H, lambda: H.linear_bwd_data(dyc, wp, taps=taps, T=T), tile)
    check(ex, sp, xr.grad, Cout * taps, f"tile {tile} conv bwd data")

    def cw():
        dwc = torch.empty(taps, Cout, Cin, device="cuda")
        H.linear_bwd_weight(dyc, xc, dwc, taps=taps, T=T)
        return dwc
    ex, sp = both(H, cw, tile)
    check(ex, sp, wr.grad.permute(2, 0, 1), B * T, f"tile {tile} conv bwd weight")
    ran = [v for mode, v in SEEN[tile] if mode == 2]
    assert ran.count(tile) >= 1, (tile, SEEN[tile])  # the forced tile carried split-mode launches (0 = library's choice)


def test_split_planes_are_exact_on_hard_operands(H):
    """Operands built to punish a sloppy split: wide dynamic range inside a row, exact powers of two, values whose low
    mantissa bits are all ones, subnormal-adjacent magnitudes and a large common offset (cancellation in the sum)."""
    M, N, K = 512, 256, 512
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, K, generator=g) * 4)
    x[:, ::7] = 2.0 ** torch.randint(-20, 20, (M, (K + 6) // 7), generator=g).float()
    x[:, 1::7] = torch.nextafter(x[:, 1::7] * 0 + 3.0, torch.tensor(0.0))   # 2.9999998: mantissa all ones
    w = torch.randn(N, K, generator=g) + 100.0                                 # large offset: heavy cancellation
    w[:, ::5] *= 1e-30
    x, w = x.cuda(), w.cuda()
    ref = x.double().cpu() @ w.double().cpu().t()
    H.set_precision("32-true")
    exact = H.linear_fwd(x, w)
    H.set_precision("32-split")
    split = H.linear_fwd(x, w)
    # per element against the magnitude that bounds fp32 rounding: sum_k |x||w|
    bound = (x.double().cpu().abs() @ w.double().cpu().abs().t())
    e_split = float(((split.double().cpu() - ref).abs() / bound).max())
    e_exact = float(((exact.double().cpu() - ref).abs() / bound).max())
    assert e_split < 3 * e_exact + 1e-7 and e_split < 2e-6, (e_split, e_exact)


def test_whole_train_step_in_split_mode_meets_the_fp32_parity_bounds():
    """A default-width train step with precision="32-split" against the CPU oracle at the fp32 tolerances of
    tests/test_model_gpu.py: loss terms 1e-4, smooth-path gradients 2e-3 of each tensor's max."""
    from fastspeech2_lightning_amd.config import FastSpeech2Config, Stats
    from fastspeech2_lightning_amd.model import FastSpeech2
    from oracle import cases as C
    from oracle import fs2_oracle as O
    conf, vp = dict(layers=2, dropout=0.0), dict(dropout=0.0)
    config = FastSpeech2Config(model=dict(encoder=conf, decoder=conf, learn_alignment=False,
                                          variance_predictors=dict(energy=vp, pitch=vp, duration=vp)),
                               text=dict(symbols=dict(letters=[f"s{i}" for i in range(40)])))
    batch = O.synthetic_batch(B=6, ts_lo=20, ts_hi=40, n_symbols=41, n_mels=80, seed=3, dur_hi=6)
    oracle = O.FastSpeech2Oracle(config, Stats(**C.STATS), n_symbols=41)
    sd = O.seeded_state_dict(oracle.state_dict())
    oracle.load_state_dict(sd)
    oracle.train()
    oracle.postnet.dropout_p = 0.0
    ref = oracle(batch)
    ref_losses = oracle.loss(ref, batch, 0)
    ref_losses["total"].backward()
    model = FastSpeech2(config, Stats(**C.STATS), precision="32-split")
    assert model.precision == "32-split"
    model.load_state_dict(sd)
    model.train()
    model.postnet.dropout_p = 0.0
    model.training_step(batch)
    for k, v in model.last_losses.items():
        want = float(ref_losses[k])
        assert abs(float(v) - want) < 1e-4 * max(1.0, abs(want)), (k, float(v), want)
    o, r = model.last_output["postnet_output"].cpu(), ref["postnet_output"].detach()
    assert float(((o - r) ** 2).mean()) < 1e-8
    got = model.store.grad_state_dict()
    gmax = max(float(p.grad.abs().max()) for p in oracle.parameters() if p.grad is not None)
    for k, p in oracle.named_parameters():
        if p.grad is None or float(p.grad.abs().max()) < 1e-4 * gmax:
            continue
        if k.startswith(("encoder.", "variance_adaptor.", "text_input_layer.")):
            continue  # (downstream of the predictors' ReLUs: tests/test_fullsize_gpu.py)
        assert float((got[k].cpu() - p.grad).abs().max()) < 2e-3 * float(p.grad.abs().max()), k
