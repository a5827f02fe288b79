"""The module under the reference's own call sites (boundary row (b)): Lightning's AUTOMATIC optimization
-- ``configure_optimizers`` -> ``optimizer.step(closure)`` where the closure is ``zero_grad`` / ``training_step`` /
``loss.backward()`` / ``configure_gradient_clipping`` -- restated here in a few lines (Lightning is not in the image),
must reproduce the native loop's five steps (``fs2l train`` / ``bench.py``: flat gradient buffer read directly) to
1e-6 and cost at most 2 % per step.

reference: ``fs2/model.py:384-390`` (``training_step`` returns the loss, Lightning differentiates it),
``fs2/model.py:530-549`` (``torch.optim.AdamW`` + ``NoamLR``, interval "step"), ``fs2/cli/train.py:33-41``
(``gradient_clip_val=1.0``)."""
import time

import pytest
import torch

from fastspeech2_lightning_amd.config import Stats
from oracle import cases as C
from oracle import fs2_oracle as O

pytestmark = pytest.mark.gpu


def _model(seed=3):
    from fastspeech2_lightning_amd.model import FastSpeech2
    config = C.small_config(learn_alignment=False)
    config.training.optimizer.learning_rate = 1e-2
    config.training.optimizer.warmup_steps = 3
    model = FastSpeech2(config, Stats(**C.STATS), seed=seed)
    model.train()
    batch = O.synthetic_batch(B=4, ts_lo=6, ts_hi=12, n_symbols=C.N_SYMBOLS, n_mels=config.preprocessing.audio.n_mels,
                              seed=11, dur_hi=4)
    return model, batch


def native_steps(n):
    model, batch = _model()
    opt = model.configure_optimizers()[0][0]
    model.configure_gradient_clipping(opt, 1.0, "norm")
    losses = []
    for _ in range(n):
        with torch.no_grad():
            losses.append(float(model.training_step(batch)))
        opt.step()
    return model, losses


def automatic_optimization(model, batch, n, clip="hook", accumulate=1):
    """Lightning's fit loop for one optimizer, automatic optimization, ``gradient_clip_val=1.0``."""
    opts, scheds = model.configure_optimizers()
    opt, sched = opts[0], scheds[0]
    assert isinstance(opt, torch.optim.Optimizer)
    assert isinstance(sched["scheduler"], torch.optim.lr_scheduler.LRScheduler) and sched["interval"] == "step"
    assert [p for g in opt.param_groups for p in g["params"]] == list(model.parameters())
    losses, lrs = [], []
    for i in range(n):
        def closure():
            # Lightning's closure order: training_step -> zero_grad (first micro-batch of an accumulation window
            # only) -> backward.  ``param.grad`` therefore still aliases the previous step's gradient buffer when
            # training_step is entered.
            total = 0.0
            for j in range(accumulate):
                loss = model.training_step(batch, i)
                assert loss.requires_grad and loss.grad_fn is not None and loss.dim() == 0
                if j == 0:
                    opt.zero_grad()
                (loss / accumulate if accumulate > 1 else loss).backward()
                model.on_train_batch_end(loss, batch, i)
                total += float(loss.detach())
                logged = model.callback_metrics  # fs2/model.py:387-389: training/{k}_loss every step
                assert float(logged["training/total_loss"]) == pytest.approx(float(loss.detach()), rel=1e-6)
                assert {"training/spec_loss", "training/postnet_loss", "training/duration_loss", "training/pitch_loss",
                        "training/energy_loss"} <= set(logged)
            if clip == "hook":      # LightningModule.configure_gradient_clipping(optimizer, gradient_clip_val, algorithm)
                model.configure_gradient_clipping(opt, 1.0, "norm")
            else:                   # what the default hook does: clip_grad_norm_ over the optimizer's parameters
                torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            return total / accumulate
        losses.append(opt.step(closure))
        lrs.append(opt.param_groups[0]["lr"])
        sched["scheduler"].step()
    return opt, sched["scheduler"], losses, lrs


@pytest.mark.parametrize("clip", ["hook", "torch"])
def test_automatic_optimization_reproduces_the_native_steps(clip):
    ref_model, ref_losses = native_steps(5)
    model, batch = _model()
    opt, sched, losses, lrs = automatic_optimization(model, batch, 5, clip)
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), (losses, ref_losses)
    if clip == "hook":
        w, w_ref = model.store.flat, ref_model.store.flat
        assert float((w - w_ref).abs().max()) <= 1e-6 * float(w_ref.abs().max()), float((w - w_ref).abs().max())
    else:
        # torch's clip_grad_norm_ rounds its coefficient differently (last bit), so weights differ by ~1e-7 after a
        # step -- and a parameter whose TRUE gradient is zero (a bias in front of a BatchNorm: depthwise-conv and
        # PostNet conv biases; the key third of an attention in-projection bias) has only rounding noise for a gradient,
        # which Adam normalises to steps of +-lr whose sign is chaotic.  Those are held to "moved by at most the five
        # learning rates"; everything else to 2e-6.
        sd, sd_ref, g = model.state_dict(), ref_model.state_dict(), ref_model.store.grad_state_dict()
        gmax = max(float(v.abs().max()) for v in g.values())
        lr_sum = sum(lrs)
        checked = 0
        for k, v in g.items():
            diff = (sd[k] - sd_ref[k]).abs()
            if k.endswith("self_attn.in_proj_bias"):
                # the KEY third of the attention in-projection bias is such a parameter too: a constant added to every key
                # shifts all of a query's scores alike and cancels in the softmax, so its true gradient is zero and what the
                # kernels produce is the rounding of dK's column sums
                n = v.numel() // 3
                assert float(diff[n:2 * n].max()) <= 2.0 * lr_sum, (k, float(diff[n:2 * n].max()))
                diff = torch.cat([diff[:n], diff[2 * n:]])
            d = float(diff.max())
            if float(v.abs().max()) < 1e-5 * gmax:
                assert d <= 2.0 * lr_sum, (k, d)
            else:
                assert d <= 2e-6 * max(1.0, float(sd_ref[k].abs().max())), (k, d)
                checked += 1
        assert checked > 100
    # the scheduler object mirrors the device-resident schedule (what LearningRateMonitor and the checkpoint see)
    o = model.config.training.optimizer
    assert lrs == pytest.approx([o.learning_rate * O.noam_scale(k, o.warmup_steps) for k in range(5)])
    assert sched.last_epoch == 5 and opt.record()["step"] == 5
    assert abs(opt.record()["lr"] - o.learning_rate * O.noam_scale(4, o.warmup_steps)) < 1e-9
    # param.grad is the flat gradient buffer itself: nothing was copied
    p = model.flat_param
    assert p.grad is not None and p.grad.data_ptr() == model.store.grad.data_ptr()


def test_gradient_accumulation_and_loss_scaling():
    """``accumulate_grad_batches=2``: Lightning divides the loss (upstream gradient 1/2) and calls ``zero_grad`` every
    second batch only.  The same batch twice must give the single-batch gradient, i.e. the same five steps."""
    ref_model, ref_losses = native_steps(3)
    model, batch = _model()
    _, _, losses, _ = automatic_optimization(model, batch, 3, "hook", accumulate=2)
    assert losses == pytest.approx(ref_losses, rel=1e-5)
    w, w_ref = model.store.flat, ref_model.store.flat
    assert float((w - w_ref).abs().max()) <= 1e-5 * float(w_ref.abs().max())


def test_optimizer_and_scheduler_state_dicts_are_torch_adamw_and_noamlr():
    model, batch = _model()
    opt, sched, _, _ = automatic_optimization(model, batch, 2)
    sd = opt.state_dict()
    names = model.reference_parameter_names()
    assert set(sd) == {"state", "param_groups"} and sd["param_groups"][0]["params"] == list(range(len(names)))
    name = "mel_linear.weight"
    st = sd["state"][names.index(name)]
    assert float(st["step"]) == 2.0 and st["exp_avg"].shape == model.state_dict()[name].shape
    ssd = sched.state_dict()
    assert ssd["last_epoch"] == 2 and ssd["warmup_steps"] == 3 and ssd["base_lrs"] == [1e-2]
    # a fresh model resumes from them through the same two objects (what Lightning's checkpoint connector does)
    other, _ = _model(seed=4)
    other.load_state_dict(model.state_dict())
    o_opts, o_scheds = other.configure_optimizers()
    o_opts[0].load_state_dict(sd)
    o_scheds[0]["scheduler"].load_state_dict(ssd)
    assert o_opts[0].record()["step"] == 2 and o_scheds[0]["scheduler"].last_epoch == 2
    assert torch.equal(other.store.adam_m, model.store.adam_m) and torch.equal(other.store.adam_v, model.store.adam_v)


def test_cost_of_the_autograd_hand_over():
    """<= 2 % per step over the native loop (same model, same batch, interleaved rounds, medians)."""
    from fastspeech2_lightning_amd.synthetic import DEFAULT_STATS, default_symbols, synthetic_batch
    from fastspeech2_lightning_amd.config import FastSpeech2Config
    from fastspeech2_lightning_amd.model import FastSpeech2
    config = FastSpeech2Config(model=dict(learn_alignment=False), text=default_symbols(64))
    model = FastSpeech2(config, Stats(**DEFAULT_STATS), seed=1)
    model.train()
    batch = model.prepare_batch(synthetic_batch(B=16, ts_lo=96, ts_hi=128, n_symbols=64, n_mels=80, seed=1234, dur_hi=9))
    opts, scheds = model.configure_optimizers()
    opt, sched = opts[0], scheds[0]["scheduler"]
    model.configure_gradient_clipping(opt, 1.0, "norm")

    def native():
        with torch.no_grad():
            model.training_step(batch)
        opt.step()

    def automatic():
        def closure():  # Lightning's order: the stale alias of the last step's gradient is dropped, not cloned
            loss = model.training_step(batch)
            opt.zero_grad()
            loss.backward()
            model.configure_gradient_clipping(opt, 1.0, "norm")
            return loss
        opt.step(closure)
        sched.step()

    def timed(fn, n=10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    for _ in range(3):
        native(); automatic()
    a, b = [], []
    for _ in range(5):
        a.append(timed(native)); b.append(timed(automatic))
    a, b = sorted(a)[2], sorted(b)[2]
    print(f"\nnative {a * 1e3:.3f} ms/step, automatic optimization {b * 1e3:.3f} ms/step (+{(b / a - 1) * 100:.2f} %)")
    assert b <= 1.02 * a + 1e-4, (a, b)


def test_monitored_validation_metric_and_trainer_precision():
    """The reference's call site (fs2/cli/train.py:33-41): ``ModelCheckpoint(monitor="validation/total_loss")`` reads
    ``trainer.callback_metrics`` after a validation pass; ``validation_step`` must have logged it (fs2/model.py:524-528,
    ``sync_dist=True``) as the epoch mean -- the value ``data.validate`` computes.  ``Trainer(precision=...)`` reaches
    the module through ``self.trainer.precision``."""
    from types import SimpleNamespace
    from fastspeech2_lightning_amd import data as D
    model, batch = _model()
    batch2 = O.synthetic_batch(B=3, ts_lo=5, ts_hi=9, n_symbols=C.N_SYMBOLS,
                               n_mels=model.config.preprocessing.audio.n_mels, seed=5, dur_hi=3)
    want = D.validate(model, [batch, batch2])
    model._val_acc = {}
    model.callback_metrics.clear()
    for i, b in enumerate((batch, batch2)):   # Lightning's validation loop
        model.validation_step(b, i)
    model.on_validation_epoch_end()
    monitor = "validation/total_loss"         # ModelCheckpoint(monitor=...) looks it up in callback_metrics
    assert monitor in model.callback_metrics
    for k, v in want.items():
        assert float(model.callback_metrics[k]) == pytest.approx(v, rel=1e-6), k
    # Trainer(precision="bf16-mixed") -> module.trainer.precision -> the kernels' precision, without a constructor kwarg
    assert model.precision == "32-true"
    model.trainer = SimpleNamespace(precision="bf16-mixed")
    model.setup("fit")
    assert model.precision == "bf16-mixed"
    loss_b = float(model.training_step(batch).detach())
    model.trainer = SimpleNamespace(precision="32-true")
    model.on_fit_start()
    assert model.precision == "32-true"
    loss_f = float(model.training_step(batch).detach())
    assert loss_b != loss_f and abs(loss_b - loss_f) < 2e-2 * abs(loss_f)
    model.trainer = SimpleNamespace(precision="16-mixed")
    with pytest.raises(ValueError):
        model.setup("fit")


def test_constructor_precision_survives_a_default_trainer():
    """``Trainer()`` defaults to "32-true".  A module built with an explicit precision keeps it under such a
    trainer (with a warning) -- ``FastSpeech2(config, precision="bf16-mixed")`` under a plain Trainer was the documented
    usage and "32-split" is not a string Trainer accepts; a non-default trainer value still wins."""
    import warnings
    from types import SimpleNamespace
    from fastspeech2_lightning_amd import hip as H
    from fastspeech2_lightning_amd.config import Stats
    from fastspeech2_lightning_amd.model import FastSpeech2
    _, batch = _model()
    for chosen in ("bf16-mixed", "32-split"):
        model = FastSpeech2(C.small_config(learn_alignment=False), Stats(**C.STATS), precision=chosen)
        model.train()
        model.trainer = SimpleNamespace(precision="32-true")
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            model.setup("fit")
            model.on_fit_start()
        assert model.precision == chosen and H.get_precision() == chosen
        assert any("keeping" in str(x.message) for x in w)
        model.training_step(batch)
        assert model.precision == chosen and H.get_precision() == chosen
    model.trainer = SimpleNamespace(precision="bf16-mixed")      # asked of the Trainer explicitly: adopted
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        model.setup("fit")
    assert model.precision == "bf16-mixed" and any("overrides" in str(x.message) for x in w)


def test_module_casts_keep_or_refuse_the_flat_parameter_alias():
    """``nn.Module._apply`` (``.half()``, ``.to(dtype)``, Lightning's precision plugins) would replace ``flat_param``
    with a converted copy and cut its alias with ``store.flat``: dtype casts raise, same-device moves are no-ops, and
    a re-homing (forced here onto the same GPU: the box has one) keeps alias, weights and training intact."""
    ref_model, ref_losses = native_steps(3)
    model, batch = _model()
    for cast in (lambda m: m.half(), lambda m: m.bfloat16(), lambda m: m.to(torch.float64), lambda m: m.cpu()):
        with pytest.raises(RuntimeError):
            cast(model)
    assert model.to("cuda") is model and model.cuda() is model and model.float() is model
    assert model.flat_param.data_ptr() == model.store.flat.data_ptr()
    old_ptr = model.store.flat.data_ptr()
    model.move_to(model.device_, force=True)
    assert model.store.flat.data_ptr() != old_ptr
    assert model.flat_param.data_ptr() == model.store.flat.data_ptr()
    opt = model.configure_optimizers()[0][0]
    model.configure_gradient_clipping(opt, 1.0, "norm")
    losses = []
    for _ in range(3):
        with torch.no_grad():
            losses.append(float(model.training_step(batch)))
        opt.step()
    assert losses == pytest.approx(ref_losses, rel=1e-6)
    assert float((model.store.flat - ref_model.store.flat).abs().max()) <= 1e-6 * float(ref_model.store.flat.abs().max())
