"""Batch producer for the hot path: per-utterance feature files -> collated
batch dict -> pinned, asynchronously prefetched device batches; spectrogram writer; validation loop.

Restates, without the parent toolkit, the parts of the reference that sit directly either side of the step:
  * on-disk layout ``<save_dir>/<kind>/<basename>--<speaker>--<language>--<suffix>.pt`` and the item dict of
    ``FastSpeechDataset.__getitem__`` (reference ``fs2/dataset.py:53-57``, ``:99-224``);
  * ``collate_method`` (reference ``fs2/dataset.py:257-293``) -- the input contract of ``FastSpeech2.forward``;
  * the ``.pt`` spectrogram output ``[n_mels, frames]`` trimmed by ``tgt_lens`` with chunk concatenation
    (reference ``fs2/prediction_writing_callback.py:257-277``);
  * ``validation_step`` + ``log_dict(sync_dist=True)`` (reference ``fs2/model.py:515-528``).
Host code only: the device work stays in ``FastSpeech2``.
"""
from __future__ import annotations

import threading
from pathlib import Path
from typing import Iterable, Iterator, Optional

import numpy as np
import torch

SEP = "--"


def feature_path(save_dir, kind: str, basename: str, speaker: str, language: str, suffix: str) -> Path:
    return Path(save_dir) / kind / SEP.join([basename, speaker, language, suffix])


class FeatureDataset(torch.utils.data.Dataset):
    """Training / teacher-forcing items of ``FastSpeechDataset`` (the text-processing inference branch belongs to
    the parent toolkit and is out of scope)."""

    def __init__(self, entries: list[dict], config, lang2id: dict, speaker2id: dict, text_processor=None):
        from .config import TextProcessor

        self.entries, self.config = entries, config
        self.lang2id, self.speaker2id = lang2id, speaker2id
        self.text_processor = text_processor or TextProcessor(config.text)
        self.save_dir = Path(config.preprocessing.save_dir)
        audio = config.preprocessing.audio
        self.sampling_rate = audio.input_sampling_rate
        self.spec_type = getattr(audio, "spec_type", "mel-librosa")

    def _load(self, bn, spk, lang, kind, fn):
        return torch.load(feature_path(self.save_dir, kind, bn, spk, lang, fn), weights_only=True)

    def __len__(self):
        return len(self.entries)

    def __getitem__(self, index):
        from .config import TargetTrainingTextRepresentationLevel as L

        item = self.entries[index]
        speaker, language = item.get("speaker", "default"), item.get("language", "default")
        bn = item["basename"]
        m = self.config.model
        mel = self._load(bn, speaker, language, "spec", f"spec-{self.sampling_rate}-{self.spec_type}.pt").transpose(0, 1)
        chars = m.target_text_representation_level == L.characters
        if m.learn_alignment:
            duration = self._load(bn, speaker, language, "attn", ("characters" if chars else "phones") + "-attn-prior.pt")
        else:
            try:
                duration = self._load(bn, speaker, language, "duration", "duration.pt")
            except FileNotFoundError as e:
                raise ValueError("model.learn_alignment = false requires text/audio alignments in "
                                 "'preprocessed/duration' (fs2/dataset.py:144-151)") from e
        tokens = item["character_tokens" if chars else "phone_tokens"]
        text = torch.IntTensor(self.text_processor.encode_escaped_string_sequence(tokens))
        pfs = None
        if m.target_text_representation_level == L.phonological_features:
            pfs = self._load(bn, speaker, language, "pfs", "pfs.pt")
        return {
            "mel": mel, "mel_style_reference": None, "duration": duration,
            "duration_control": item.get("duration_control", 1.0), "pfs": pfs, "text": text,
            "raw_text": item.get("characters", item.get("phones", "text")), "basename": bn,
            "speaker": speaker, "speaker_id": self.speaker2id[speaker], "language": language,
            "language_id": self.lang2id[language],
            "energy": self._load(bn, speaker, language, "energy", "energy.pt"),
            "pitch": self._load(bn, speaker, language, "pitch", "pitch.pt"),
            "is_last_input_chunk": None,
        }


def _padded(seqs: list, shape_tail_max: tuple, pin: bool) -> torch.Tensor:
    """One zero-initialised [B, *max extents] buffer (pinned when the prefetcher asked for it) with every sequence
    written into its top-left corner -- the batch tensor is built in the memory the H2D copy will read."""
    first = seqs[0]
    out = torch.zeros((len(seqs),) + shape_tail_max, dtype=first.dtype, pin_memory=pin)
    for row, seq in zip(out, seqs):
        row[tuple(slice(0, n) for n in seq.shape)] = seq
    return out


def collate(items: list[dict], learn_alignment: bool = True, pin_memory: bool = False) -> dict:
    """Batch contract of the reference's ``FastSpeech2DataModule.collate_method`` (fs2/dataset.py:257-293), produced
    column by column: every key of the items becomes a list; tensor / ndarray columns become one zero-padded tensor
    (ragged in time -- and, for the attention prior of a learned-alignment model, in tokens too: it is padded to
    (max_mel_len, max_src_len) even when no utterance reaches both); int columns become int32 vectors; anything else
    (names, raw text, None placeholders) stays a list.  ``src_lens`` / ``mel_lens`` are int32, the two maxima are
    0-dim int32 tensors, and a batch without mels (inference) gets ``mel_lens = None, max_mel_len = 1_000_000``.
    ``pin_memory``: allocate the padded tensors in pinned host memory (``DevicePrefetcher`` copies from them)."""
    columns = {key: [item[key] for item in items] for key in items[0]}
    src_lens = torch.tensor([len(t) for t in columns["text"]], dtype=torch.int32)
    has_mel = columns["mel"][0] is not None
    mel_lens = torch.tensor([len(m) for m in columns["mel"]], dtype=torch.int32) if has_mel else None
    max_src = src_lens.max()
    max_mel = mel_lens.max() if has_mel else 1_000_000
    batch = {}
    for key, col in columns.items():
        head = col[0]
        if isinstance(head, np.ndarray):
            col, head = [torch.from_numpy(np.ascontiguousarray(x)) for x in col], torch.from_numpy(head)
        if torch.is_tensor(head):
            if key == "duration" and learn_alignment:
                extents = (int(max_mel), int(max_src))
            else:
                extents = tuple(max(x.shape[d] for x in col) for d in range(head.dim()))
            batch[key] = _padded(col, extents, pin_memory)
        elif isinstance(head, int) and not isinstance(head, bool):
            batch[key] = torch.tensor(col, dtype=torch.int32)
        else:
            batch[key] = col
    batch.update(src_lens=src_lens, max_src_len=max_src, mel_lens=mel_lens, max_mel_len=max_mel)
    return batch


class DevicePrefetcher:
    """Keeps one batch ahead of the training step: the collated CPU batch is pinned and copied to the GPU on a
    side stream while the previous step computes; ``__next__`` only makes the compute stream wait for that copy's
    event.  ``prepare`` is ``FastSpeech2.prepare_batch`` (dtype conversion + placement)."""

    def __init__(self, batches: Iterable[dict], prepare, device):
        self.it: Iterator[dict] = iter(batches)
        self.prepare, self.device = prepare, torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self._next = None
        self._preload()

    @staticmethod
    def _pin(batch):
        out = {}
        for k, v in batch.items():
            out[k] = v.pin_memory() if torch.is_tensor(v) and v.dim() > 0 and not v.is_cuda else v
        return out

    def _preload(self):
        try:
            cpu = next(self.it)
        except StopIteration:
            self._next = None
            return
        with torch.cuda.stream(self.stream):
            dev = self.prepare(self._pin(cpu))
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._next = (dev, ev)

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        dev, ev = self._next
        torch.cuda.current_stream(self.device).wait_event(ev)
        for v in dev.values():  # the compute stream owns the tensors from here on
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(torch.cuda.current_stream(self.device))
        self._preload()
        return dev


class SpecWriter:
    """``.pt`` spectrogram writer of the reference's prediction callback: ``[n_mels, frames]`` trimmed by
    ``tgt_lens``; consecutive chunks of one utterance are concatenated until ``is_last_input_chunk``."""

    def __init__(self, out_dir, output_key: str, global_step: int = 0, sampling_rate: int = 22050,
                 spec_type: str = "mel-librosa"):
        self.dir = Path(out_dir) / "synthesized_spec"
        self.dir.mkdir(parents=True, exist_ok=True)
        self.output_key, self.global_step = output_key, global_step
        self.suffix = f"spec-pred-{sampling_rate}-{spec_type}.pt"
        self._spec, self._text = torch.tensor(()), ""

    def filename(self, basename, speaker, language) -> Path:
        return self.dir / SEP.join([basename, speaker, language, f"ckpt={self.global_step}", self.suffix])

    def write(self, outputs: dict, batch: dict) -> list[Path]:
        written = []
        lens = [int(n) for n in outputs["tgt_lens"]]
        last = batch.get("is_last_input_chunk") or [True] * len(lens)
        for i, data in enumerate(outputs[self.output_key]):
            self._spec = torch.cat((self._spec, data[: lens[i]].cpu().transpose(0, 1)), -1)
            self._text += batch["raw_text"][i]
            if last[i] is None or last[i]:
                path = self.filename(truncate_basename(slugify(self._text)), batch["speaker"][i], batch["language"][i])
                torch.save(self._spec, path)
                written.append(path)
                self._spec, self._text = torch.tensor(()), ""
        return written


def slugify(text: str) -> str:
    """File-name-safe form of a text (the parent toolkit's ``everyvoice.utils.slugify``: NFKC, lower case,
    non-word characters dropped, runs of blanks / dashes collapsed)."""
    import re
    import unicodedata
    text = unicodedata.normalize("NFKC", str(text))
    text = re.sub(r"[^\w\s-]", "", text.lower())
    return re.sub(r"[-\s]+", "-", text).strip("-_")


def truncate_basename(basename: str, max_length: int = 20) -> str:
    """reference ``fs2/utils/__init__.py:8-20``: slug cut to 20 characters + 8 hex digits of the sha1 of the
    full name, so that utterances sharing a prefix do not overwrite each other."""
    import hashlib
    cleaned = slugify(basename)
    if len(cleaned) <= max_length:
        return cleaned
    return cleaned[:max_length] + "-" + hashlib.sha1(bytes(basename, encoding="UTF-8")).hexdigest()[:8]


def validate(model, batches: Iterable[dict], process_group=None) -> dict:
    """Mean of every loss term over the validation batches and over the ranks (what
    ``log_dict(..., sync_dist=True)`` reports and checkpoint selection monitors as ``validation/total_loss``)."""
    import torch.distributed as dist

    total, n = None, 0
    keys = None
    for batch in batches:
        losses = model.validation_step(batch)
        keys = keys or list(losses)
        vec = torch.stack([losses[k].detach().float() for k in keys])
        total = vec if total is None else total + vec
        n += 1
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1
    if total is None and not distributed:
        return {}
    if distributed:
        # a rank whose shard is empty still takes part in the exchange (with zero sums) -- and learns the keys from it
        box = [keys]
        gathered = [None] * dist.get_world_size(process_group)
        dist.all_gather_object(gathered, box[0], group=process_group)
        keys = next((k for k in gathered if k), None)
        if keys is None:
            return {}
        if total is None:
            total = torch.zeros(len(keys), device=model.device_, dtype=torch.float32)
    stat = torch.cat([total, total.new_tensor([float(n)])])
    if distributed:
        dist.all_reduce(stat, group=process_group)
    mean = (stat[:-1] / stat[-1]).cpu()
    return {f"validation/{k}_loss": float(v) for k, v in zip(keys, mean)}
