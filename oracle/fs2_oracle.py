"""TEST INFRASTRUCTURE -- CPU restatement (pure PyTorch, fp32) of the reference's
feature-prediction path.  NOT part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import it.

Every class cites the reference ``file:line`` it restates (paths relative to
the reference checkout).  Module/attribute names are chosen so that
``state_dict()`` keys equal the reference's, which lets one
state dict drive the reference glue, this oracle and the HIP path.

Parity pinning
  * everything except the Conformer body is pinned against the reference's own
    Python, executed in the build container by ``oracle/make_golden.py`` (the
    vectors live in ``tests/golden/``; ``tests/test_oracle_golden.py`` checks them);
  * the Conformer body (``torchaudio.models.Conformer``, torchaudio==2.7.1 per
    the reference's ``uv.lock:3632-3634``) is an un-vendored third-party
    dependency absent from the reference checkout and from this image and no
    reference test pins its numerics -> **parity unpinned** at that boundary;
    ``Conformer`` below restates torchaudio's published layer structure
     from ``torch.nn`` primitives.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------- #
# Dropout with an injectable mask (test infrastructure only)
# --------------------------------------------------------------------------- #
class MaskedDropout(nn.Dropout):
    """``nn.Dropout`` whose random mask can be replaced by a given one: ``factor`` (0 or 1/(1-p) per element, in the
    layout of this site's input) is what the HIP kernels' stateless hash draws for the site, exported by
    ``tests/dropout_masks.py`` -- the only way to compare a dropout-ON train step element by element.  With
    ``factor = None`` it is ``nn.Dropout``.  No parameters: state-dict keys are unchanged."""
    factor: Optional[torch.Tensor] = None

    def forward(self, x):
        if self.factor is None or not self.training:
            return super().forward(x)
        assert self.factor.shape == x.shape, (tuple(self.factor.shape), tuple(x.shape))
        return x * self.factor


def _mha_with_prob_factor(mha: nn.MultiheadAttention, x, key_padding_mask, factor):
    """``nn.MultiheadAttention`` (seq-first self-attention, need_weights=False) written out so that the dropout on the
    attention probabilities can take an injected mask: ``factor`` [B, H, T, T].  Same arithmetic as
    ``F.multi_head_attention_forward``: in_proj -> split heads -> q / sqrt(hd) -> softmax(q k^T + key mask) -> dropout
    -> @ v -> out_proj (``tests/test_oracle_golden.py`` checks it against the module with dropout off)."""
    T, B, D = x.shape
    H = mha.num_heads
    hd = D // H
    qkv = F.linear(x, mha.in_proj_weight, mha.in_proj_bias)
    q, k, v = qkv.chunk(3, dim=-1)
    q = q.reshape(T, B * H, hd).transpose(0, 1) * (hd ** -0.5)
    k = k.reshape(T, B * H, hd).transpose(0, 1)
    v = v.reshape(T, B * H, hd).transpose(0, 1)
    s = torch.bmm(q, k.transpose(1, 2)).view(B, H, T, T)
    s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    if factor is not None:
        p = p * factor
    o = torch.bmm(p.view(B * H, T, T), v).transpose(0, 1).reshape(T, B, D)
    return F.linear(o, mha.out_proj.weight, mha.out_proj.bias)


# --------------------------------------------------------------------------- #
# Conformer (torchaudio.models.Conformer restated; call sites fs2/model.py:95-119)
# --------------------------------------------------------------------------- #
class _FeedForward(nn.Module):
    def __init__(self, d: int, f: int, p: float):
        super().__init__()
        self.sequential = nn.Sequential(
            nn.LayerNorm(d), nn.Linear(d, f), nn.SiLU(), MaskedDropout(p),
            nn.Linear(f, d), MaskedDropout(p),
        )

    def forward(self, x):
        return self.sequential(x)


class _ConvModule(nn.Module):
    def __init__(self, d: int, k: int, p: float):
        super().__init__()
        self.layer_norm = nn.LayerNorm(d)
        self.sequential = nn.Sequential(
            nn.Conv1d(d, 2 * d, 1), nn.GLU(dim=1),
            nn.Conv1d(d, d, k, padding=(k - 1) // 2, groups=d),
            nn.BatchNorm1d(d), nn.SiLU(), nn.Conv1d(d, d, 1), MaskedDropout(p),
        )

    def forward(self, x):  # x: (B, T, D)
        x = self.layer_norm(x).transpose(1, 2)
        return self.sequential(x).transpose(1, 2)


class _ConformerLayer(nn.Module):
    def __init__(self, d: int, f: int, heads: int, k: int, p: float):
        super().__init__()
        self.ffn1 = _FeedForward(d, f, p)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.self_attn = nn.MultiheadAttention(d, heads, dropout=p)
        self.self_attn_dropout = MaskedDropout(p)
        self.attn_prob_factor = None  # injected attention-probability dropout mask [B, H, T, T] (tests only)
        self.conv_module = _ConvModule(d, k, p)
        self.ffn2 = _FeedForward(d, f, p)
        self.final_layer_norm = nn.LayerNorm(d)

    def forward(self, x, key_padding_mask):  # x: (T, B, D)
        x = 0.5 * self.ffn1(x) + x
        r = x
        h = self.self_attn_layer_norm(x)
        if self.attn_prob_factor is not None and self.training:
            h = _mha_with_prob_factor(self.self_attn, h, key_padding_mask, self.attn_prob_factor)
        else:
            h, _ = self.self_attn(h, h, h, key_padding_mask=key_padding_mask, need_weights=False)
        x = self.self_attn_dropout(h) + r
        x = x + self.conv_module(x.transpose(0, 1)).transpose(0, 1)
        x = 0.5 * self.ffn2(x) + x
        return self.final_layer_norm(x)


class Conformer(nn.Module):
    def __init__(self, input_dim, num_heads, ffn_dim, num_layers,
                 depthwise_conv_kernel_size, dropout=0.0):
        super().__init__()
        self.conformer_layers = nn.ModuleList(
            _ConformerLayer(input_dim, ffn_dim, num_heads, depthwise_conv_kernel_size, dropout)
            for _ in range(num_layers)
        )

    def forward(self, x, lengths):  # x: (B, T, D), T == max(lengths)
        T = x.shape[1]
        pad = torch.arange(T, device=x.device)[None, :] >= lengths[:, None].to(x.device)
        x = x.transpose(0, 1)
        for layer in self.conformer_layers:
            x = layer(x, pad)
        return x.transpose(0, 1), lengths


# --------------------------------------------------------------------------- #
# fs2/utils/heavy.py:11-15, fs2/layers.py:123-140
# --------------------------------------------------------------------------- #
def mask_from_lens(lens: torch.Tensor, max_len: Optional[int] = None) -> torch.Tensor:
    if max_len is None:
        max_len = int(lens.max())
    return torch.arange(int(max_len), device=lens.device, dtype=lens.dtype)[None, :] < lens[:, None]


class PositionalEmbedding(nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.register_buffer("inv_freq", 1 / (10000 ** (torch.arange(0.0, d, 2.0) / d)))

    def forward(self, pos_seq):  # (T,) -> (1, T, d): [sin | cos] concatenated
        ang = pos_seq[:, None].to(self.inv_freq.dtype) @ self.inv_freq[None, :]
        return torch.cat([ang.sin(), ang.cos()], dim=1)[None]


# --------------------------------------------------------------------------- #
# fs2/blocks.py:4-19, fs2/layers.py:11-48, fs2/variance_adaptor.py:18-81
# --------------------------------------------------------------------------- #
class _Model(nn.Module):  # gives the '.model.{0,1}' key segment of DepthwiseSeparableConv1d
    def __init__(self, cin, cout, k):
        super().__init__()
        self.model = nn.Sequential(
            nn.Conv1d(cin, cin, k, padding=(k - 1) // 2, groups=cin), nn.Conv1d(cin, cout, 1)
        )

    def forward(self, x):
        return self.model(x)


class _Transpose(nn.Module):  # '.module' key segment
    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, x):
        return self.module(x.transpose(1, 2)).transpose(1, 2)


class VarianceConvolutionLayer(nn.Module):
    def __init__(self, cin, cout, k, dropout, depthwise):
        super().__init__()
        conv = _Model(cin, cout, k) if depthwise else nn.Conv1d(cin, cout, k, padding=(k - 1) // 2)
        self.layers = nn.Sequential(_Transpose(conv), nn.ReLU(), nn.LayerNorm(cout), MaskedDropout(dropout))

    def forward(self, x):
        return self.layers(x)


class VariancePredictor(nn.Module):
    def __init__(self, input_dim, n_layers, n_channels, kernel_size, dropout, depthwise):
        super().__init__()
        self.conv = nn.ModuleList(
            VarianceConvolutionLayer(input_dim if i == 0 else n_channels, n_channels,
                                     kernel_size, dropout, depthwise)
            for i in range(n_layers)
        )
        self.linear = nn.Linear(n_channels, 1)

    def forward(self, x, mask=None):
        for m in self.conv:
            x = m(x)
        out = self.linear(x).squeeze(-1)
        return out * mask if mask is not None else out


def length_regulate(x, durations, max_length):
    """fs2/variance_adaptor.py:65-81 -- repeat each row durations[b, j] times."""
    rows = [torch.repeat_interleave(x[b], durations[b].long().clamp(min=0), dim=0)
            for b in range(x.shape[0])]
    lengths = torch.tensor([r.shape[0] for r in rows], dtype=torch.int32)
    max_length = min(int(lengths.max()), int(max_length))
    out = x.new_zeros(x.shape[0], max_length, x.shape[2])
    for b, r in enumerate(rows):
        n = min(r.shape[0], max_length)
        out[b, :n] = r[:n]
    mask = torch.arange(max_length)[None, :] < lengths[:, None]
    return out, mask.to(x.device)


# --------------------------------------------------------------------------- #
# Aligner: fs2/attn/attention.py:101-251, fs2/attn/alignment.py:48-74,
#          fs2/attn/attention_loss.py:22-73
# --------------------------------------------------------------------------- #
class _ConvNorm(nn.Module):  # '.conv' key segment (fs2/attn/attention.py:23-56, fs2/blocks.py:44-87)
    def __init__(self, cin, cout, k=1, gain="linear"):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, k, padding=(k - 1) // 2)
        nn.init.xavier_uniform_(self.conv.weight, gain=nn.init.calculate_gain(gain))

    def forward(self, x):
        return self.conv(x)


class ConvAttention(nn.Module):
    def __init__(self, n_mel=80, n_text=256, n_att=80):
        super().__init__()
        self.key_proj = nn.Sequential(
            _ConvNorm(n_text, 2 * n_text, 3, "relu"), nn.ReLU(), _ConvNorm(2 * n_text, n_att, 1))
        self.query_proj = nn.Sequential(
            _ConvNorm(n_mel, 2 * n_mel, 3, "relu"), nn.ReLU(), _ConvNorm(2 * n_mel, n_mel, 1),
            nn.ReLU(), _ConvNorm(n_mel, n_att, 1))

    def forward(self, queries, keys, mask, attn_prior):
        """queries (B, n_mel, T1), keys (B, n_text, T2), mask (B, T2, 1) True = pad,
        attn_prior (B, T1, T2).  Returns (attn, attn_logprob), each (B, 1, T1, T2)."""
        k = self.key_proj(keys)
        q = self.query_proj(queries)
        d = ((q[:, :, :, None] - k[:, :, None]) ** 2).sum(1, keepdim=True)
        attn = -0.0005 * d
        attn = F.log_softmax(attn, dim=3) + torch.log(attn_prior[:, None] + 1e-8)
        attn_logprob = attn.clone()
        attn = attn.masked_fill(mask.permute(0, 2, 1).unsqueeze(2), -float("inf"))
        return F.softmax(attn, dim=3), attn_logprob


def mas_width1(log_attn_map: np.ndarray) -> np.ndarray:
    """fs2/attn/alignment.py:48-74 in plain numpy fp32 (mel x text).
    Restated index for index, including the ``j - 1 == -1`` wrap-around read that
    numpy/numba negative indexing gives the reference when ``j == 0``."""
    neg_inf = log_attn_map.dtype.type(-np.inf)
    log_p = log_attn_map.copy()
    T1, T2 = log_p.shape
    log_p[0, 1:] = neg_inf
    for i in range(1, T1):
        prev = np.concatenate([np.array([neg_inf], dtype=log_p.dtype), log_p[i - 1, :-1]])
        log_p[i] += np.maximum(prev, log_p[i - 1])
    opt = np.zeros_like(log_p)
    j = T2 - 1
    for i in range(T1 - 1, 0, -1):
        opt[i, j] = 1
        if log_p[i - 1, j - 1] >= log_p[i - 1, j]:
            j -= 1
            if j == 0:
                opt[1:i, j] = 1
                break
    opt[0, j] = 1
    return opt


def binarize_attention(attn, in_lens, out_lens):
    """fs2/variance_adaptor.py:160-181."""
    out = np.zeros(tuple(attn.shape), dtype=np.float32)
    log_attn = torch.log(attn.detach()).to("cpu", torch.float32).numpy()
    for b in range(attn.shape[0]):
        t1, t2 = int(out_lens[b]), int(in_lens[b])
        out[b, 0, :t1, :t2] = mas_width1(log_attn[b, 0, :t1, :t2])
    return torch.tensor(out, device=attn.device, dtype=attn.dtype)


def average_variance(var, durs):
    """fs2/variance_adaptor.py:207-222 -- mean over each token's frames, counting
    only non-zero frames; 0 where a token has none."""
    ends = torch.cumsum(durs, dim=1).long()
    starts = F.pad(ends[:, :-1], (1, 0))
    nz = F.pad(torch.cumsum(var != 0.0, dim=1), (1, 0))
    cs = F.pad(torch.cumsum(var, dim=1), (1, 0))
    sums = (torch.gather(cs, 1, ends) - torch.gather(cs, 1, starts)).float()
    cnt = (torch.gather(nz, 1, ends) - torch.gather(nz, 1, starts)).float()
    return torch.where(cnt == 0.0, cnt, sums / cnt)


def attention_ctc_loss(attn_logprob, in_lens, out_lens, blank_logprob=-1.0):
    """fs2/attn/attention_loss.py:22-62."""
    max_key_len = attn_logprob.size(-1)
    x = attn_logprob.squeeze(1).permute(1, 0, 2)
    x = F.pad(x, (1, 0, 0, 0, 0, 0), value=blank_logprob)
    key_inds = torch.arange(max_key_len + 1, device=x.device, dtype=torch.long)
    x = x.masked_fill(key_inds.view(1, 1, -1) > in_lens.view(1, -1, 1).long(), -1e15)
    x = F.log_softmax(x, dim=-1)
    targets = key_inds[1:].unsqueeze(0).repeat(in_lens.numel(), 1)
    return F.ctc_loss(x, targets, input_lengths=out_lens.long(), target_lengths=in_lens.long(),
                      blank=0, reduction="mean", zero_infinity=True)


def attention_bin_loss(hard, soft, eps=1e-12):
    """fs2/attn/attention_loss.py:65-73."""
    return -torch.log(torch.clamp(soft[hard == 1], min=eps)).sum() / hard.sum()


# --------------------------------------------------------------------------- #
# GST style encoder: fs2/gst/model.py:14-280, fs2/gst/attn.py:48-194
# --------------------------------------------------------------------------- #
class _GstMHA(nn.Module):
    def __init__(self, q_dim, k_dim, v_dim, n_head, n_feat):
        super().__init__()
        self.d_k, self.h = n_feat // n_head, n_head
        self.linear_q, self.linear_k = nn.Linear(q_dim, n_feat), nn.Linear(k_dim, n_feat)
        self.linear_v, self.linear_out = nn.Linear(v_dim, n_feat), nn.Linear(n_feat, n_feat)

    def forward(self, query, key, value):
        B = query.size(0)
        q = self.linear_q(query).view(B, -1, self.h, self.d_k).transpose(1, 2)
        k = self.linear_k(key).view(B, -1, self.h, self.d_k).transpose(1, 2)
        v = self.linear_v(value).view(B, -1, self.h, self.d_k).transpose(1, 2)
        p = torch.softmax(q @ k.transpose(-2, -1) / math.sqrt(self.d_k), dim=-1)
        x = (p @ v).transpose(1, 2).contiguous().view(B, -1, self.h * self.d_k)
        return self.linear_out(x)


class _ReferenceEncoder(nn.Module):
    def __init__(self, idim=80, chans=(32, 32, 64, 64, 128, 128), gru_units=128):
        super().__init__()
        convs, cin, f = [], 1, idim
        for c in chans:
            convs += [nn.Conv2d(cin, c, 3, stride=2, padding=1, bias=False), nn.BatchNorm2d(c), nn.ReLU(inplace=True)]
            cin, f = c, (f - 3 + 2) // 2 + 1
        self.convs = nn.Sequential(*convs)
        self.gru = nn.GRU(f * chans[-1], gru_units, 1, batch_first=True)

    def forward(self, speech):  # (B, Lmax, idim)
        hs = self.convs(speech.unsqueeze(1)).transpose(1, 2)  # (B, L', C, F')
        hs = hs.contiguous().view(speech.size(0), hs.size(1), -1)
        _, h = self.gru(hs)
        return h[-1]


class _StyleTokenLayer(nn.Module):
    def __init__(self, ref_dim=128, tokens=10, token_dim=256, heads=4):
        super().__init__()
        self.gst_embs = nn.Parameter(torch.randn(tokens, token_dim // heads))
        self.mha = _GstMHA(ref_dim, token_dim // heads, token_dim // heads, heads, token_dim)

    def forward(self, ref):
        g = torch.tanh(self.gst_embs).unsqueeze(0).expand(ref.size(0), -1, -1)
        return self.mha(ref.unsqueeze(1), g, g).squeeze(1)


class StyleEncoder(nn.Module):
    def __init__(self, idim=80):
        super().__init__()
        self.ref_enc = _ReferenceEncoder(idim)
        self.stl = _StyleTokenLayer()

    def forward(self, speech):
        return self.stl(self.ref_enc(speech))

    def condition_on_gst_tokens(self, batch_size, index=0):  # fs2/gst/model.py:77-85
        gst = torch.tanh(self.stl.gst_embs)
        query = torch.zeros(batch_size, 1, gst.size(1) * 2)  # gst_token_dim // 2 = ref_dim
        keys = gst[index].unsqueeze(0).expand(batch_size, -1, -1)
        return self.stl.mha(query, keys, keys).squeeze(1)


# --------------------------------------------------------------------------- #
# fs2/layers.py:143-212
# --------------------------------------------------------------------------- #
class PostNet(nn.Module):
    def __init__(self, n_mel=80, dim=512, k=5, n=5):
        super().__init__()
        self.dropout_p = 0.5  # hard-coded in the reference; tests set 0 for determinism
        self.drop_factors = None  # injected masks, one (B, C, T) factor per layer (tests only)
        chans = [n_mel] + [dim] * (n - 1) + [n_mel]
        self.convolutions = nn.ModuleList(
            nn.Sequential(_ConvNorm(chans[i], chans[i + 1], k, "tanh" if i < n - 1 else "linear"),
                          nn.BatchNorm1d(chans[i + 1]))
            for i in range(n)
        )

    def forward(self, x):  # (B, T, n_mel)
        x = x.transpose(1, 2)
        for i, c in enumerate(self.convolutions):
            x = c(x)
            if i < len(self.convolutions) - 1:
                x = torch.tanh(x)
            if self.drop_factors is not None and self.training:
                x = x * self.drop_factors[i]
            else:
                x = F.dropout(x, self.dropout_p, self.training)
        return x.transpose(1, 2)


# --------------------------------------------------------------------------- #
# fs2/variance_adaptor.py:84-412
# --------------------------------------------------------------------------- #
class VarianceAdaptor(nn.Module):
    def __init__(self, config, stats):
        super().__init__()
        self.config, self.stats = config, stats
        vp, d = config.model.variance_predictors, config.model.encoder.input_dim

        def predictor(c):
            return VariancePredictor(d, c.n_layers, c.input_dim, c.kernel_size, c.dropout, c.depthwise)

        self.duration_predictor = predictor(vp.duration)
        self.pitch_predictor = predictor(vp.pitch)
        self.pitch_embedding = nn.Embedding(vp.pitch.n_bins, vp.pitch.input_dim)
        self.pitch_bins = nn.Parameter(
            torch.linspace(stats.pitch.norm_min, stats.pitch.norm_max, vp.pitch.n_bins - 1),
            requires_grad=False)
        self.energy_predictor = predictor(vp.energy)
        self.energy_embedding = nn.Embedding(vp.energy.n_bins, vp.energy.input_dim)
        self.energy_bins = nn.Parameter(
            torch.linspace(stats.energy.norm_min, stats.energy.norm_max, vp.energy.n_bins - 1),
            requires_grad=False)
        if config.model.learn_alignment:
            self.attention = ConvAttention(config.preprocessing.audio.n_mels, d, 80)

    @staticmethod
    def _embed(x, target, mask, predictor, embedding, bins, control, inference):
        prediction = predictor(x, mask)
        if not inference:
            return prediction, embedding(torch.bucketize(target, bins))
        prediction = prediction * control
        return prediction, embedding(torch.bucketize(prediction, bins))

    def forward(self, text_emb, encoder_output, batch, src_mask, control, inference=False,
                teacher_forcing=False):
        cfg = self.config.model
        x = encoder_output.clone()
        energy_t = batch["energy"] if not inference else None
        pitch_t = batch["pitch"] if not inference else None
        dur_t = batch["duration"] if batch.get("duration") is not None else None
        attn_logprob = attn_soft = attn_hard = None
        if (teacher_forcing or not inference) and cfg.learn_alignment:
            attn_soft, attn_logprob = self.attention(
                batch["mel"].transpose(1, 2), text_emb.transpose(1, 2),
                src_mask[..., None] == 0, batch["duration"])
            attn_hard = binarize_attention(attn_soft, batch["src_lens"], batch["mel_lens"])
            dur_t = attn_hard.sum(2)[:, 0, :].int()
            if energy_t is not None and cfg.variance_predictors.energy.level == "phone":
                energy_t = average_variance(energy_t, dur_t)
            if pitch_t is not None and cfg.variance_predictors.pitch.level == "phone":
                pitch_t = average_variance(pitch_t, dur_t)
            assert torch.all(dur_t.sum(dim=1) == batch["mel_lens"])
        pitch_p = energy_p = None
        if cfg.variance_predictors.energy.level == "phone":
            energy_p, e = self._embed(x, energy_t, src_mask, self.energy_predictor,
                                      self.energy_embedding, self.energy_bins, control.energy, inference)
            x = x + e
        if cfg.variance_predictors.pitch.level == "phone":
            pitch_p, e = self._embed(x, pitch_t, src_mask, self.pitch_predictor,
                                     self.pitch_embedding, self.pitch_bins, control.pitch, inference)
            x = x + e
        log_dur_p = self.duration_predictor(x, mask=src_mask)
        if teacher_forcing or not inference:
            dur_r = dur_t
        else:
            dur_r = torch.clamp(torch.round(torch.exp(log_dur_p) - 1) * control.duration, min=0).int()
        x, tgt_mask = length_regulate(x, dur_r, batch["max_mel_len"])
        if cfg.variance_predictors.energy.level == "frame":
            energy_p, e = self._embed(x, energy_t, tgt_mask, self.energy_predictor,
                                      self.energy_embedding, self.energy_bins, control.energy, inference)
            x = x + e
        if cfg.variance_predictors.pitch.level == "frame":
            pitch_p, e = self._embed(x, pitch_t, tgt_mask, self.pitch_predictor,
                                     self.pitch_embedding, self.pitch_bins, control.pitch, inference)
            x = x + e
        return dict(output=x, attn_logprob=attn_logprob, attn_soft=attn_soft, attn_hard=attn_hard,
                    duration_prediction=log_dur_p, duration_target=dur_t, pitch_prediction=pitch_p,
                    pitch_target=pitch_t, energy_prediction=energy_p, energy_target=energy_t,
                    duration_rounded=dur_r, target_mask=tgt_mask)


# --------------------------------------------------------------------------- #
# fs2/loss.py:19-126
# --------------------------------------------------------------------------- #
def fastspeech2_loss(config, output, batch, current_epoch: int) -> dict:
    fn = {"mse": F.mse_loss, "mae": F.l1_loss}
    m, t = config.model, config.training
    src_mask, tgt_mask = output["src_mask"], output["tgt_mask"]
    losses = {}
    for name, w in (("pitch", t.pitch_loss_weight), ("energy", t.energy_loss_weight)):
        tgt = output[f"{name}_target"]
        if tgt is None:
            continue
        c = getattr(m.variance_predictors, name)
        mask = src_mask if c.level.value == "phone" else tgt_mask
        losses[name] = fn[c.loss.value](output[f"{name}_prediction"] * mask, tgt * mask) * w
    log_d = torch.log(output["duration_target"].to(output["duration_prediction"].dtype) + 1) * src_mask
    losses["duration"] = fn[m.variance_predictors.duration.loss.value](
        output["duration_prediction"] * src_mask, log_d) * t.duration_loss_weight
    tm = tgt_mask.unsqueeze(2)
    spec_t = batch["mel"] * tm
    losses["spec"] = fn[m.mel_loss.value](output["output"] * tm, spec_t) * t.mel_loss_weight
    if m.use_postnet:
        losses["postnet"] = fn[m.mel_loss.value](output["postnet_output"] * tm, spec_t) * t.postnet_loss_weight
    if m.learn_alignment:
        losses["attn_ctc"] = attention_ctc_loss(
            output["attn_logprob"], batch["src_lens"], batch["mel_lens"]) * t.attn_ctc_loss_weight
        w = min(current_epoch / t.attn_bin_loss_warmup_epochs, 1.0) * t.attn_bin_loss_weight
        losses["attn_bin"] = attention_bin_loss(output["attn_hard"], output["attn_soft"]) * w
    losses["total"] = sum(losses.values())
    return losses


def noam_scale(step: int, warmup: int) -> float:
    """fs2/noam.py:20-26."""
    s = max(1, step)
    return warmup ** 0.5 * min(s ** (-0.5), s * warmup ** (-1.5))


# --------------------------------------------------------------------------- #
# fs2/model.py:38-268 (the train/inference step; Lightning plumbing omitted)
# --------------------------------------------------------------------------- #
class FastSpeech2Oracle(nn.Module):
    def __init__(self, config, stats, n_symbols: int, n_speakers: int = 0, n_langs: int = 0,
                 padding_idx: int = 0):
        super().__init__()
        from fastspeech2_lightning_amd.config import TargetTrainingTextRepresentationLevel as L
        from fastspeech2_lightning_amd.config import N_PHONOLOGICAL_FEATURES

        self.config, self.stats = config, stats
        m = config.model
        d = m.encoder.input_dim
        if m.target_text_representation_level == L.phonological_features:
            self.text_input_layer = nn.Linear(N_PHONOLOGICAL_FEATURES, d, bias=False)
        else:
            self.text_input_layer = nn.Embedding(n_symbols, d, padding_idx=padding_idx)
        self.position_embedding = PositionalEmbedding(d)
        if m.use_global_style_token_module:
            self.gst = StyleEncoder(idim=config.preprocessing.audio.n_mels)

        def conformer(c):
            return Conformer(c.input_dim, c.heads, c.feedforward_dim, c.layers, c.conv_kernel_size, c.dropout)

        self.encoder = conformer(m.encoder)
        self.variance_adaptor = VarianceAdaptor(config, stats)
        self.decoder = conformer(m.decoder)
        n_mels = config.preprocessing.audio.n_mels
        self.mel_linear = nn.Linear(m.decoder.input_dim, n_mels)
        if m.use_postnet:
            self.postnet = PostNet(n_mels)
        self.speaker_embedding = nn.Embedding(n_speakers, d) if m.multispeaker else None
        self.language_embedding = nn.Embedding(n_langs, d) if m.multilingual else None

    def forward(self, batch, control=None, inference=False):
        from fastspeech2_lightning_amd.config import InferenceControl
        from fastspeech2_lightning_amd.config import TargetTrainingTextRepresentationLevel as L

        control = control or InferenceControl()
        m = self.config.model
        teacher_forcing = bool(inference and batch.get("mel_lens") is not None)
        src_lens, mel_lens = batch["src_lens"], batch.get("mel_lens")
        max_src_len, max_mel_len = int(batch["max_src_len"]), batch["max_mel_len"]
        text = batch["pfs"] if m.target_text_representation_level == L.phonological_features else batch["text"]
        src_mask = mask_from_lens(src_lens, max_src_len)
        inputs = self.text_input_layer(text)
        pos = self.position_embedding(torch.arange(max_src_len).to(inputs.dtype)) * src_mask.unsqueeze(2)
        x, _ = self.encoder(inputs + pos, src_lens)
        if m.use_global_style_token_module:  # fs2/model.py:196-203 (training / teacher forcing: the target mel)
            ref_mel = batch.get("mel_style_reference")  # fs2/model.py:196-203
            if inference and torch.is_tensor(ref_mel):
                style = self.gst(ref_mel)
            elif inference and not teacher_forcing:
                style = self.gst.condition_on_gst_tokens(batch["src_lens"].size(0))
            else:
                style = self.gst(batch["mel"])
            x = x + style.unsqueeze(1)
        if self.speaker_embedding is not None:
            x = x + self.speaker_embedding(batch["speaker_id"]).unsqueeze(1)
        if self.language_embedding is not None:
            x = x + self.language_embedding(batch["language_id"]).unsqueeze(1)
        va = self.variance_adaptor(inputs, x, batch, src_mask, control, inference, teacher_forcing)
        if inference and not teacher_forcing:
            mel_lens = va["target_mask"].sum(1).int()
            max_mel_len = int(mel_lens.max())
        dpos = self.position_embedding(torch.arange(int(max_mel_len)).to(torch.float32))
        x, _ = self.decoder(va["output"] + dpos * va["target_mask"].unsqueeze(2), mel_lens)
        output = self.mel_linear(x)
        post = output + self.postnet(output) if m.use_postnet else None
        return dict(
            output=output, postnet_output=post, src_mask=src_mask, src_lens=src_lens,
            tgt_mask=va["target_mask"], tgt_lens=mel_lens, attn_logprob=va["attn_logprob"],
            attn_soft=va["attn_soft"], attn_hard=va["attn_hard"],
            duration_prediction=va["duration_prediction"], duration_target=va["duration_target"],
            energy_prediction=va["energy_prediction"], energy_target=va["energy_target"],
            pitch_prediction=va["pitch_prediction"], pitch_target=va["pitch_target"], text_input=text)

    def loss(self, output, batch, current_epoch: int = 0):
        return fastspeech2_loss(self.config, output, batch, current_epoch)


# synthetic LJSpeech-shaped batches are defined once, in the product package
from fastspeech2_lightning_amd.synthetic import beta_binomial_prior, synthetic_batch  # noqa: E402,F401


# --------------------------------------------------------------------------- #
# deterministic parameter values shared by the golden generator and the tests:
# fixtures then carry no weights (the reference hard-codes a 512-channel PostNet,
# which alone is 15 MB of fp32 weights)
# --------------------------------------------------------------------------- #
def seeded_state_dict(template: dict) -> dict:
    """Fill every tensor of a reference-layout state dict from a generator seeded
    by the CRC32 of its key.  Buffers that are functions of the config
    (``*_bins``, ``inv_freq``, ``num_batches_tracked``) are kept."""
    import zlib

    out = {}
    for k, v in template.items():
        if k.endswith(("_bins", "inv_freq", "num_batches_tracked")):
            out[k] = v.clone()
            continue
        g = torch.Generator().manual_seed(zlib.crc32(k.encode()))
        if k.endswith("running_var"):
            t = torch.rand(v.shape, generator=g) + 0.5
        elif k.endswith("running_mean"):
            t = 0.1 * torch.randn(v.shape, generator=g)
        elif v.dim() == 1 and k.endswith(".weight"):  # LayerNorm / BatchNorm scale
            t = 1 + 0.1 * torch.randn(v.shape, generator=g)
        elif v.dim() == 1:  # biases
            t = 0.1 * torch.randn(v.shape, generator=g)
        elif "embedding" in k or "text_input_layer" in k or "gst_embs" in k:
            t = 0.5 * torch.randn(v.shape, generator=g)
        else:
            fan_in = v[0].numel()
            t = torch.randn(v.shape, generator=g) / math.sqrt(fan_in)
        out[k] = t.to(v.dtype)
    pad = template.get("text_input_layer.weight")
    if pad is not None and pad.dim() == 2 and pad.shape[0] > 1 and "text_input_layer.weight" in out:
        out["text_input_layer.weight"][0].zero_()  # padding_idx row (fs2/model.py:83-89)
    return out


GRAD_SUBSAMPLE_THRESHOLD = 20000
GRAD_SUBSAMPLE_STRIDE = 101


def subsample(a: np.ndarray) -> np.ndarray:
    """Large gradients are stored as [l2-norm, sum, flat[::101]...]."""
    flat = a.reshape(-1).astype(np.float64)
    head = np.array([np.sqrt((flat ** 2).sum()), flat.sum()])
    return np.concatenate([head, flat[::GRAD_SUBSAMPLE_STRIDE]]).astype(np.float32)
