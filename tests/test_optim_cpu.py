"""Host side of the optimizer / scheduler pair (no GPU): they are torch's own interfaces, the Noam values are the
reference's (golden ``noam/lrs`` from the reference's ``NoamLR``), the state dicts have torch's shapes."""
import numpy as np
import torch

from fastspeech2_lightning_amd import params as P
from fastspeech2_lightning_amd.optim import FusedAdamWNoam, NoamLR, noam_scale


def _store():
    S = P.ParamStore()
    S.add("a.weight", (4, 3), "id", P.init_normal)
    S.add("a.bias", (4,), "id", P.init_zeros)
    S.add("conv.weight", (5, 4, 3), "convk", P.init_normal)
    S.finalize("cpu", seed=0)
    return S


def test_interfaces_and_schedule(golden_dir):
    S = _store()
    state = torch.zeros(4, dtype=torch.int64)
    opt = FusedAdamWNoam(S, state, 1e-3, (0.9, 0.98), 1e-9, 0.01, 4000, param_names=list(S.entries))
    assert isinstance(opt, torch.optim.Optimizer) and opt.max_grad_norm is None
    assert opt.param_groups[0]["params"][0].data_ptr() == S.flat.data_ptr()
    sched = NoamLR(opt, 4000)
    assert isinstance(sched, torch.optim.lr_scheduler.LRScheduler)
    # the reference's own NoamLR(AdamW(lr=1e-3), warmup_steps=40).get_last_lr() before each of 120 steps (make_golden.py)
    want = np.load(golden_dir / "units.npz")["noam/lrs"]
    s2 = NoamLR(FusedAdamWNoam(S, state.clone(), 1e-3, (0.9, 0.98), 1e-9, 0.0, 40), 40)
    lrs = []
    for _ in range(len(want)):
        lrs.append(s2.get_last_lr()[0])
        s2.step()
    np.testing.assert_allclose(lrs, want, rtol=1e-12)
    assert sched.get_last_lr()[0] == 1e-3 * noam_scale(0, 4000)
    for k in range(1, 6):
        sched.step()
        assert abs(sched.get_last_lr()[0] - 1e-3 * noam_scale(k, 4000)) < 1e-15
        assert opt.param_groups[0]["lr"] == sched.get_last_lr()[0]


def test_state_dicts_have_torch_shapes_and_round_trip():
    S = _store()
    state = torch.zeros(4, dtype=torch.int64)
    names = list(S.entries)
    opt = FusedAdamWNoam(S, state, 1e-3, (0.9, 0.98), 1e-9, 0.01, 10, param_names=names)
    S.adam_m.copy_(torch.arange(S.total, dtype=torch.float32))
    S.adam_v.copy_(torch.arange(S.total, dtype=torch.float32) * 2)
    opt.set_step(7)
    sd = opt.state_dict()
    assert set(sd) == {"state", "param_groups"} and len(sd["state"]) == 3
    assert sd["state"][2]["exp_avg"].shape == (5, 4, 3) and float(sd["state"][0]["step"]) == 7.0
    assert sd["param_groups"][0]["initial_lr"] == 1e-3 and sd["param_groups"][0]["decoupled_weight_decay"] is True
    # torch's own AdamW accepts it for parameters of those shapes
    ref_params = [torch.nn.Parameter(torch.zeros(S.entries[n].ref_shape)) for n in names]
    ref = torch.optim.AdamW(ref_params, 1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.01)
    ref.load_state_dict(sd)
    assert torch.equal(ref.state[ref_params[2]]["exp_avg"], sd["state"][2]["exp_avg"])
    # and back
    S2 = _store()
    opt2 = FusedAdamWNoam(S2, torch.zeros(4, dtype=torch.int64), 1e-3, (0.9, 0.98), 1e-9, 0.01, 10, param_names=names)
    opt2.load_state_dict(ref.state_dict())
    for n in names:  # (the alignment padding between entries carries no state)
        assert torch.equal(S2.export_flat(S2.adam_m, n), S.export_flat(S.adam_m, n))
        assert torch.equal(S2.export_flat(S2.adam_v, n), S.export_flat(S.adam_v, n))
    assert opt2.steps_done() == 7
    sched = NoamLR(opt2, 10)
    sched.load_state_dict(opt.torch_scheduler_state_dict())
    assert sched.last_epoch == 7 and opt2.steps_done() == 7
    # the native form round-trips too
    S3 = _store()
    opt3 = FusedAdamWNoam(S3, torch.zeros(4, dtype=torch.int64), 1e-3, (0.9, 0.98), 1e-9, 0.01, 10)
    opt3.load_state_dict(opt.native_state_dict())
    assert torch.equal(S3.adam_m, S.adam_m) and opt3.steps_done() == 7
    # zero_grad drops the delivered gradient
    opt.param.grad = torch.zeros_like(opt.param)
    opt.zero_grad()
    assert opt.param.grad is None
