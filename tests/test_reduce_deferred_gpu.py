"""Deferred split-K finishes (fs2hip_reduce_slabs_multi): one launch for the slab sums of many weight-gradient GEMMs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from fastspeech2_lightning_amd import hip
    hip.lib()
    return hip


@pytest.mark.parametrize("precision", ["32-true", "bf16-mixed"])
def test_deferred_split_reductions_match_the_immediate_ones(H, precision):
    """FastSpeech2.backward leaves the split-K slabs of its weight-gradient GEMMs unsummed until the flush (one
    multi-job launch, fs2hip_reduce_slabs_multi): same slabs, same summation order -> the same bits, except for small
    outputs whose immediate path sums 16 rows at a time (1e-6 there)."""
    saved = H.get_precision()
    H.set_precision(precision)
    try:
        g = torch.Generator().manual_seed(5)
        cases = [(41472, 256, 1024), (41472, 1024, 256), (8192, 256, 256), (4100, 264, 1024), (20736, 80, 256)]
        outs_now, outs_later, biases_now, biases_later = [], [], [], []
        args = []
        for M, N, K in cases:
            dy = torch.randn(M, N, generator=g).cuda()
            x = torch.randn(M, K, generator=g).cuda()
            if precision == "bf16-mixed" and H.BF16_STORAGE and N % 8 == 0 and K % 8 == 0:
                dy, x = dy.bfloat16(), x.bfloat16()
            args.append((dy, x, N, K))
        for dy, x, N, K in args:
            w, b = torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")
            H.linear_bwd_weight(dy, x, w, bias_grad=b)
            H.flush_grad_reductions()
            outs_now.append(w); biases_now.append(b)
        prev = H.defer_slab_reductions(True)
        try:
            for dy, x, N, K in args:
                w = torch.full((N, K), float("nan"), device="cuda")
                b = torch.empty(N, device="cuda")
                H.linear_bwd_weight(dy, x, w, bias_grad=b)
                outs_later.append(w); biases_later.append(b)
            H.flush_grad_reductions()
        finally:
            H.defer_slab_reductions(prev)
        torch.cuda.synchronize()
        for (M, N, K), a, b in zip(cases, outs_now, outs_later):
            assert torch.isfinite(b).all()
            scale = a.abs().max().item()
            assert (a - b).abs().max().item() <= 1e-6 * scale, (M, N, K)
        for a, b in zip(biases_now, biases_later):
            assert torch.equal(a, b)
    finally:
        H.set_precision(saved)
