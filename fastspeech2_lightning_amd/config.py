"""Configuration surface of the feature-prediction path.

Field names and defaults mirror the reference schema (reference
``fs2/config/__init__.py:31-317``) so that a reference YAML/JSON/checkpoint
``hyper_parameters.config`` validates here unchanged.  The reference builds its
schema on ``everyvoice.config.*`` base classes, which are not part of the
reference repository; only the handful of base fields the hot path reads
(``preprocessing.audio.n_mels``, ``preprocessing.save_dir``, ``text.symbols``,
the Noam optimizer block, ``training.batch_size`` ...) are re-declared.
Unknown keys are accepted and kept (``extra="allow"``) so that configs written
by the full toolkit still load.
"""
from __future__ import annotations

import json
from enum import Enum
from pathlib import Path
from typing import Any, Optional, Union

import yaml
from pydantic import BaseModel, ConfigDict, Field, model_validator

LATEST_VERSION: str = "1.1"  # reference fs2/config/__init__.py:28


class _Cfg(BaseModel):
    model_config = ConfigDict(extra="allow", use_enum_values=False)


class TargetTrainingTextRepresentationLevel(str, Enum):
    characters = "characters"
    ipa_phones = "ipa_phones"
    phonological_features = "phonological_features"


#: width of a phonological feature vector in the parent toolkit
N_PHONOLOGICAL_FEATURES = 38


class ConformerConfig(_Cfg):  # reference fs2/config/__init__.py:31-48
    layers: int = 4
    heads: int = 2
    input_dim: int = 256
    feedforward_dim: int = 1024
    conv_kernel_size: int = 9
    dropout: float = 0.2


class VarianceLevelEnum(str, Enum):
    phone = "phone"
    frame = "frame"


class VarianceLossEnum(str, Enum):
    mse = "mse"
    mae = "mae"


class VariancePredictorBase(_Cfg):  # reference fs2/config/__init__.py:67-94
    loss: VarianceLossEnum = VarianceLossEnum.mse
    n_layers: int = 5
    kernel_size: int = 3
    dropout: float = 0.5
    input_dim: int = 256
    n_bins: int = 256
    depthwise: bool = True


class VariancePredictorConfig(VariancePredictorBase):  # :97-105
    level: VarianceLevelEnum = VarianceLevelEnum.phone


class VariancePredictors(_Cfg):  # :108-120
    energy: VariancePredictorConfig = Field(default_factory=VariancePredictorConfig)
    duration: VariancePredictorBase = Field(default_factory=VariancePredictorBase)
    pitch: VariancePredictorConfig = Field(default_factory=VariancePredictorConfig)


class FastSpeech2ModelConfig(_Cfg):  # :123-175
    encoder: ConformerConfig = Field(default_factory=ConformerConfig)
    decoder: ConformerConfig = Field(default_factory=ConformerConfig)
    variance_predictors: VariancePredictors = Field(default_factory=VariancePredictors)
    target_text_representation_level: TargetTrainingTextRepresentationLevel = (
        TargetTrainingTextRepresentationLevel.characters
    )
    learn_alignment: bool = True
    use_global_style_token_module: bool = False
    max_length: int = 1000
    mel_loss: VarianceLossEnum = VarianceLossEnum.mse
    use_postnet: bool = True
    multilingual: bool = False
    multispeaker: bool = False


class NoamOptimizer(_Cfg):
    """The parent toolkit's Noam/AdamW block (fields read by
    reference ``fs2/model.py:530-549``)."""

    learning_rate: float = 1e-3
    eps: float = 1e-8
    weight_decay: float = 1e-6
    betas: tuple[float, float] = (0.9, 0.999)
    name: str = "noam"
    warmup_steps: int = 1000


class FastSpeech2TrainingConfig(_Cfg):  # :193-243
    batch_size: int = 16
    max_epochs: int = 1000
    max_steps: int = 100000
    use_weighted_sampler: bool = False
    optimizer: NoamOptimizer = Field(default_factory=NoamOptimizer)
    vocoder_path: Optional[str] = None
    mel_loss_weight: float = 1.0
    postnet_loss_weight: float = 1.0
    pitch_loss_weight: float = 0.1
    energy_loss_weight: float = 0.1
    duration_loss_weight: float = 0.1
    attn_ctc_loss_weight: float = 0.1
    attn_bin_loss_weight: float = 0.1
    attn_bin_loss_warmup_epochs: int = Field(default=100, ge=1)
    training_filelist: Optional[str] = None
    validation_filelist: Optional[str] = None
    train_data_workers: int = 4
    val_data_workers: int = 0


class AudioConfig(_Cfg):
    n_mels: int = 80
    spec_type: str = "mel-librosa"
    input_sampling_rate: int = 22050
    output_sampling_rate: int = 22050


class PreprocessingConfig(_Cfg):
    save_dir: str = "preprocessed"
    audio: AudioConfig = Field(default_factory=AudioConfig)


class TextConfig(_Cfg):
    """``symbols`` maps a category name to a list of symbols, as in the parent
    toolkit's text configuration."""

    symbols: dict[str, Any] = Field(default_factory=dict)


class TextProcessor:
    """Minimal symbol table: the hot path only needs ``len(symbols)`` and the
    index of the padding symbol (reference ``fs2/model.py:83-89``)."""

    _pad_symbol = "\x80"

    def __init__(self, text_config: TextConfig):
        seen: list[str] = []
        for value in text_config.symbols.values():
            if isinstance(value, str):
                value = [value]
            if not isinstance(value, (list, tuple)):
                continue
            for s in value:
                if isinstance(s, str) and s not in seen and s != self._pad_symbol:
                    seen.append(s)
        self.symbols = [self._pad_symbol] + sorted(seen)
        self._index = {s: i for i, s in enumerate(self.symbols)}

    def encode_text(self, text: str) -> list[int]:
        out, i = [], 0
        # greedy longest match over the symbol table
        longest = max((len(s) for s in self.symbols), default=1)
        while i < len(text):
            for width in range(min(longest, len(text) - i), 0, -1):
                idx = self._index.get(text[i : i + width])
                if idx is not None:
                    out.append(idx)
                    i += width
                    break
            else:
                i += 1  # unknown symbol: dropped
        return out


    def encode_escaped_string_sequence(self, tokens) -> list[int]:
        """Token sequence as stored by the preprocessor: symbols separated by '/' ('\\/' is a literal slash);
        a list of symbols is accepted as well.  Unknown symbols are dropped."""
        if isinstance(tokens, str):
            parts, cur, i = [], "", 0
            while i < len(tokens):
                if tokens[i] == "\\" and i + 1 < len(tokens):
                    cur += tokens[i + 1]
                    i += 2
                elif tokens[i] == "/":
                    parts.append(cur)
                    cur = ""
                    i += 1
                else:
                    cur += tokens[i]
                    i += 1
            parts.append(cur)
            tokens = parts
        return [self._index[t] for t in tokens if t in self._index]


class FastSpeech2Config(_Cfg):  # reference fs2/config/__init__.py:246-317
    VERSION: str = LATEST_VERSION
    model: FastSpeech2ModelConfig = Field(default_factory=FastSpeech2ModelConfig)
    path_to_model_config_file: Optional[str] = None
    training: FastSpeech2TrainingConfig = Field(default_factory=FastSpeech2TrainingConfig)
    path_to_training_config_file: Optional[str] = None
    preprocessing: PreprocessingConfig = Field(default_factory=PreprocessingConfig)
    path_to_preprocessing_config_file: Optional[str] = None
    text: TextConfig = Field(default_factory=TextConfig)
    path_to_text_config_file: Optional[str] = None

    @model_validator(mode="before")
    @classmethod
    def _load_partials_and_check_version(cls, data: Any, info) -> Any:
        """Partial files (reference :280-289) and the version gate (:299-317)."""
        if not isinstance(data, dict):
            return data
        data = dict(data)
        from packaging.version import Version

        version = Version(str(data.get("VERSION", "0.0")))
        if version > Version(LATEST_VERSION):
            raise ValueError(
                "Your config was created with a newer version of EveryVoice, "
                "please update your software."
            )
        if version < Version("1.0"):
            data["VERSION"] = "1.0"
        base = None
        if info.context is not None:
            base = info.context.get("config_path", None)
        for part in ("model", "training", "preprocessing", "text"):
            key = f"path_to_{part}_config_file"
            path = data.get(key)
            if path and part not in data:
                p = Path(path)
                if not p.is_absolute() and base is not None:
                    p = Path(base).parent / p
                if p.exists():
                    data[part] = _load_json_or_yaml(p)
        return data

    @staticmethod
    def load_config_from_path(path: Union[str, Path]) -> "FastSpeech2Config":
        """Reference ``fs2/config/__init__.py:291-297``."""
        path = Path(path)
        raw = _load_json_or_yaml(path)
        return FastSpeech2Config.model_validate(raw, context={"config_path": path})

    def model_checkpoint_dump(self) -> dict:
        return json.loads(self.model_dump_json())


def _load_json_or_yaml(path: Path) -> dict:
    text = Path(path).read_text(encoding="utf8")
    if str(path).endswith(".json"):
        return json.loads(text)
    return yaml.safe_load(text)


class InferenceControl(BaseModel):  # reference fs2/type_definitions_heavy.py:15-20
    model_config = ConfigDict(arbitrary_types_allowed=True)
    pitch: float = 1.0
    energy: float = 1.0
    duration: float = 1.0


class StatsInfo(BaseModel):  # :23-29
    min: float
    max: float
    std: float
    mean: float
    norm_min: float
    norm_max: float


class Stats(BaseModel):  # :32-37
    pitch: StatsInfo
    energy: StatsInfo
    character_length: Optional[StatsInfo] = None
    phone_length: Optional[StatsInfo] = None
    arpabet_length: Optional[StatsInfo] = None


class BadDataError(Exception):
    """Raised when aligner durations do not add up to the mel length
    (reference ``fs2/variance_adaptor.py:289-304``)."""
