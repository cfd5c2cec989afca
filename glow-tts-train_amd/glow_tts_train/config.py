"""Hyper-parameter containers read by `models.setup_model` (field names and defaults of the reference's
glow_tts_train/config.py:11-124), with the reference's JSON surface — `to_dict / from_dict / to_json / from_json`,
`save`, `load`, `load_and_merge`, `recursive_update` — on plain dataclasses (`dataclasses_json`, which the reference
mixes in, is not a dependency here).  Host-side only."""
from __future__ import annotations

import collections.abc
import json
import typing
from dataclasses import asdict, field, fields, is_dataclass, make_dataclass
from pathlib import Path


class _DictMixin:
    def to_dict(self) -> dict:
        return asdict(self)

    @classmethod
    def from_dict(cls, d: typing.Mapping[str, typing.Any]):
        kw = {}
        for f in fields(cls):
            if f.name not in d:
                continue
            v = d[f.name]
            sub = _NESTED.get((cls.__name__, f.name))
            if sub is not None and isinstance(v, collections.abc.Mapping):
                v = sub.from_dict(v)
            elif isinstance(v, list) and isinstance(getattr(cls, f.name, None), tuple):
                v = tuple(v)                       # JSON has no tuples (betas)
            kw[f.name] = v
        return cls(**kw)

    def to_json(self, **kw) -> str:
        return json.dumps(self.to_dict(), **kw)

    @classmethod
    def from_json(cls, text: str):
        return cls.from_dict(json.loads(text))


Opt = typing.Optional


def _table(name: str, spec: str, bases=(_DictMixin,), namespace=None):
    """A dataclass from a field table: one `name type default` triple per entry, `;`-separated."""
    rows = []
    for entry in spec.split(";"):
        entry = entry.strip()
        if entry:
            fname, ftype, default = entry.split(None, 2)
            rows.append((fname, eval(ftype), field(default=eval(default))))      # noqa: S307 - literals written below
    cls = make_dataclass(name, rows, bases=bases, namespace=namespace or {})
    cls.__module__ = __name__                       # picklable / importable by name
    return cls


# mel front end of the data the model is trained on (reference config.py:11-33)
AudioConfig = _table("AudioConfig", """
    filter_length int 1024; hop_length int 256; win_length int 1024; mel_channels int 80; sample_rate int 22050;
    sample_bytes int 2; channels int 1; mel_fmin float 0.0; mel_fmax Opt[float] 8000.0; ref_level_db float 20.0;
    spec_gain float 1.0; signal_norm bool True; min_level_db float -100.0; max_norm float 1.0; clip_norm bool True;
    symmetric_norm bool True; do_dynamic_range_compression bool True; convert_db_to_amp bool True""")

# network sizes (reference config.py:36-62)
ModelConfig = _table("ModelConfig", """
    num_symbols int 0; hidden_channels int 192; filter_channels int 768; filter_channels_dp int 256; kernel_size int 3;
    p_dropout float 0.1; n_blocks_dec int 12; n_layers_enc int 6; n_heads int 2; p_dropout_dec float 0.05;
    dilation_rate int 1; kernel_size_dec int 5; n_block_layers int 4; n_sqz int 2; prenet bool True; mean_only bool True;
    hidden_channels_enc int 192; hidden_channels_dec int 192; window_size int 4; n_speakers int 1; n_split int 4;
    sigmoid_scale bool False; block_length Opt[int] None; gin_channels int 0; n_frames_per_step int 1""")


class _TrainingMethods(_DictMixin):
    def save(self, config_file: typing.TextIO):
        """Write the configuration as JSON (reference config.py:83-85)."""
        json.dump(self.to_dict(), config_file, indent=4)

    @staticmethod
    def load(config_file: typing.TextIO) -> "TrainingConfig":
        """Read a configuration written by `save` (reference config.py:87-90)."""
        return TrainingConfig.from_json(config_file.read())

    @staticmethod
    def load_and_merge(config: "TrainingConfig",
                       config_files: typing.Iterable[typing.Union[str, Path, typing.TextIO]]) -> "TrainingConfig":
        """Overlay JSON files, in order, on `config` (reference config.py:92-112): later files win, nested objects
        (`audio`, `model`) are merged key by key, and keys no dataclass field carries are ignored."""
        merged = config.to_dict()
        for source in config_files:
            handle = open(source, "r") if isinstance(source, (str, Path)) else source
            with handle:
                TrainingConfig.recursive_update(merged, json.load(handle))
        return TrainingConfig.from_dict(merged)

    @staticmethod
    def recursive_update(base_dict: typing.Dict[typing.Any, typing.Any],
                         new_dict: typing.Mapping[typing.Any, typing.Any]) -> None:
        """In-place deep overwrite of `base_dict` by `new_dict` (reference config.py:114-124): a mapping is descended
        into when the base already holds something non-None under that key, anything else replaces the base value."""
        for key, value in new_dict.items():
            if isinstance(value, collections.abc.Mapping) and base_dict.get(key) is not None:
                TrainingConfig.recursive_update(base_dict[key], value)
            else:
                base_dict[key] = value


# optimisation schedule + the two tables above (reference config.py:65-81)
_training_rows = [(n, eval(t), field(default=eval(d))) for n, t, d in (e.strip().split(None, 2) for e in """
    seed int 1234; epochs int 10000; learning_rate float 1e0; betas typing.Tuple[float,float] (0.9,0.98); eps float 1e-9;
    grad_clip float 5.0; warmup_steps int 4000; scheduler str "noam"; batch_size int 32; fp16_run bool False;
    min_seq_length Opt[int] None; max_seq_length Opt[int] None""".split(";"))]
_training_rows += [("audio", AudioConfig, field(default_factory=AudioConfig)),
                   ("model", ModelConfig, field(default_factory=ModelConfig)),
                   ("version", int, field(default=1)), ("git_commit", str, field(default=""))]
TrainingConfig = make_dataclass("TrainingConfig", _training_rows, bases=(_TrainingMethods,))
TrainingConfig.__module__ = __name__

_NESTED = {("TrainingConfig", "audio"): AudioConfig, ("TrainingConfig", "model"): ModelConfig}
assert is_dataclass(TrainingConfig)
