"""Exports the HIP kernels' dropout masks and injects them into the CPU oracle (test infrastructure).

The kernels never store a mask: every dropout site draws ``keep = hash(seed(site, step), element index) >= p * 2^32``
(``csrc/common.h``) on the fly, in the forward kernel and again in the backward kernel, the element index being the
linear index of the site's tensor in its natural layout ([B*T, C] rows for GEMM epilogues / BatchNorm+activation /
elementwise kernels, [B, H, T, Tp] for the attention probabilities, Tp = T rounded up to even).  ``fs2hip_axpby`` applied to a vector of ones with a
site's ``Drop`` record writes exactly those factors (0 or 1/(1-p)) -- the same device function, the same seed and the
same device-resident step counter -- so ``site_factors`` is the debug export of a site's mask, and
``inject`` puts each one into the oracle's matching ``MaskedDropout`` / attention / PostNet site in that site's layout.
A dropout-ON train step can then be compared element by element with the usual tolerances.
"""
from __future__ import annotations

import torch

from fastspeech2_lightning_amd import hip as H


def site_factors(model, p: float, site: int, *shape) -> torch.Tensor:
    """The factors the kernels apply at dropout site ``site`` at the model's CURRENT device step, as a CPU tensor."""
    drop = model.env.drop(p, site)
    n = 1
    for s in shape:
        n *= s
    if drop.p <= 0:
        return torch.ones(*shape)
    ones = torch.ones(n, device=model.device_, dtype=torch.float32)
    return H.axpby(ones, None, 1.0, 0.0, drop).view(*shape).cpu()


def _conformer(model, stack, ostack, cfg, B, T):
    p, heads, D, Fd = cfg.dropout, cfg.heads, cfg.input_dim, cfg.feedforward_dim
    for layer, olayer in zip(stack.layers, ostack.conformer_layers):
        tb = lambda t: t.permute(1, 0, 2).contiguous()  # (B, T, C) -> the oracle's (T, B, C)  # noqa: E731
        for ffn, offn in ((layer.ffn1, olayer.ffn1), (layer.ffn2, olayer.ffn2)):
            offn.sequential[3].factor = tb(site_factors(model, p, ffn.s1, B, T, Fd))
            offn.sequential[5].factor = tb(site_factors(model, p, ffn.s2, B, T, D))
        # (rows of the attention mask index are padded to an even length: one hash serves two neighbouring keys)
        olayer.attn_prob_factor = (site_factors(model, p, layer.attn.sa, B, heads, T, T + (T & 1))[..., :T].contiguous()
                                   if p > 0 else None)
        olayer.self_attn_dropout.factor = tb(site_factors(model, p, layer.attn.so, B, T, D))
        # conv module: the oracle's Sequential runs on (B, D, T)
        olayer.conv_module.sequential[6].factor = site_factors(model, p, layer.conv.site, B, T, D).permute(0, 2, 1).contiguous()


def inject(model, oracle, B: int, Ts: int, Tm: int) -> dict:
    """Gives every dropout site of ``oracle`` the mask the HIP model draws at its current step.  Returns
    {site name: keep fraction} for the sites that are on."""
    m = model.config.model
    _conformer(model, model.encoder, oracle.encoder, m.encoder, B, Ts)
    _conformer(model, model.decoder, oracle.decoder, m.decoder, B, Tm)
    va, ova = model.variance_adaptor, oracle.variance_adaptor
    seen = {}
    for name in ("energy", "pitch", "duration"):
        c = getattr(m.variance_predictors, name)
        T = Tm if (name != "duration" and c.level.value == "frame") else Ts
        pred, opred = getattr(va, f"{name}_predictor"), getattr(ova, f"{name}_predictor")
        for L, ol in zip(pred.layers, opred.conv):
            f = site_factors(model, c.dropout, L["site"], B, T, c.input_dim)
            ol.layers[3].factor = f
            seen[f"{name}.{L['site']}"] = float((f > 0).float().mean())
    if model.postnet is not None:
        facs = []
        for (w, b, bn, site) in model.postnet.convs:
            cout = model.store.p(b).numel()
            f = site_factors(model, model.postnet.dropout_p, site, B, Tm, cout)
            facs.append(f.permute(0, 2, 1).contiguous())
            seen[f"postnet.{site}"] = float((f > 0).float().mean())
        oracle.postnet.drop_factors = facs if model.postnet.dropout_p > 0 else None
    for i, ol in enumerate(oracle.decoder.conformer_layers):
        if ol.attn_prob_factor is not None:
            seen[f"decoder.{i}.attn_prob"] = float((ol.attn_prob_factor > 0).float().mean())
    return seen


class ReluCapture:
    """Sign patterns of the variance predictors' Conv -> ReLU layers on both sides of a parity step: the CPU oracle's
    pre-activations (forward hooks on its ``nn.ReLU`` modules) and the HIP step's ReLU outputs (the input of each
    ``hip.layernorm_fwd_drop`` call, told apart by the LayerNorm weight it is handed: the predictors of a training step
    walk their layers in lockstep, ``modules.predictors_fwd``).  ``flips()`` names every element where
    the two disagree -- a pre-activation within rounding of zero that fell on different sides of it under the two
    summation orders, which switches that (token, channel)'s gradient path on in one model and off in the other
    (tools/relu_flip_diag.py is the stand-alone form, with a float64 run for scale)."""

    def __init__(self, model, oracle):
        self.model, self.oracle = model, oracle
        self.pre, self.hip, self._hooks, self._real = {}, {}, [], None
        self.order = [(n, li) for n in ("energy", "pitch", "duration")
                      for li in range(len(getattr(model.variance_adaptor, f"{n}_predictor").layers))]

    def __enter__(self):
        for name in ("energy", "pitch", "duration"):
            pred = getattr(self.oracle.variance_adaptor, f"{name}_predictor")
            for li, layer in enumerate(pred.conv):
                def hook(mod, inp, out, key=(name, li)):
                    self.pre[key] = inp[0].detach().clone()
                self._hooks.append(layer.layers[1].register_forward_hook(hook))
        self._real = H.layernorm_fwd_drop
        S = self.model.store
        which = {}  # LayerNorm weight (address in the flat parameter) -> (predictor, layer)
        for name, li in self.order:
            L = getattr(self.model.variance_adaptor, f"{name}_predictor").layers[li]
            which[S.p(L["ln"].w).data_ptr()] = (name, li)

        def wrapped(r, gamma, *a, **k):
            key = which[gamma.data_ptr()]
            assert key not in self.hip, key
            self.hip[key] = (r > 0).cpu()
            return self._real(r, gamma, *a, **k)
        H.layernorm_fwd_drop = wrapped
        return self

    def __exit__(self, *exc):
        H.layernorm_fwd_drop = self._real
        for h in self._hooks:
            h.remove()
        return False

    def flips(self):
        assert len(self.hip) == len(self.order) == len(self.pre), (len(self.hip), len(self.order), len(self.pre))
        out = []
        for key in self.order:
            pat, pre = self.hip[key], self.pre[key]
            for b, t, ch in (pat.view_as(pre) != (pre > 0)).nonzero().tolist():
                out.append((f"{key[0]}_predictor.conv.{key[1]}", b, t, ch, float(pre[b, t, ch])))
        return out
