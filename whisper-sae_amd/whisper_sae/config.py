"""Experiment configuration (pydantic v2), YAML-compatible with the reference's ``configs/*.yaml``.

Mirrors the schema of the reference's ``src/whisper_sae/config.py`` field for field (names, defaults,
bounds) so existing YAML files load unchanged.  Additions of this build, all optional so old
files stay valid:

* ``sae.sparsity_weight``  -- L1 weight handed to ``ReLUSAE`` (the reference's ``create_sae`` drops
  it and silently uses 0.01);
* ``training.resample_dead_every`` / ``training.resample_batch_size`` -- the trainer defaults the
  reference hard-codes in ``SAETrainer.__init__`` (training.py:41-42), exposed for YAML use.
"""

from __future__ import annotations

from pathlib import Path
from typing import Literal, Optional

import yaml
from pydantic import BaseModel, Field, model_validator

# (d_model, encoder layers, decoder layers) of the Whisper checkpoints on the HF hub
WHISPER_GEOMETRY = {
    "tiny": (384, 4, 4),
    "base": (512, 6, 6),
    "small": (768, 12, 12),
    "medium": (1024, 24, 24),
    "large": (1280, 32, 32),
    "large-v2": (1280, 32, 32),
    "large-v3": (1280, 32, 32),
}


class WhisperConfig(BaseModel):
    """Which Whisper checkpoint the activations come from; dimensions follow from the name."""

    model_name: str = Field("openai/whisper-tiny", description="HF hub id of the Whisper model")
    hidden_dim: int = Field(384, description="d_model of that checkpoint")
    num_encoder_layers: int = Field(4)
    num_decoder_layers: int = Field(4)

    @model_validator(mode="after")
    def _fill_geometry(self) -> "WhisperConfig":
        prefix = "openai/whisper-"
        if self.model_name.startswith(prefix):
            geo = WHISPER_GEOMETRY.get(self.model_name[len(prefix):])
            if geo is not None:
                self.hidden_dim, self.num_encoder_layers, self.num_decoder_layers = geo
        return self


class SAEConfig(BaseModel):
    """Sparse-autoencoder architecture."""

    expansion_factor: int = Field(8, ge=4, le=32, description="hidden_dim = input_dim * expansion_factor")
    activation: Literal["topk", "relu", "gelu"] = Field("topk")
    k: int = Field(32, ge=1, description="active features per token (TopK)")
    normalize_decoder: bool = Field(True, description="keep decoder columns at unit L2 norm")
    dead_feature_threshold: int = Field(10_000, description="steps without firing before a feature counts as dead")
    dead_feature_resample: bool = Field(True, description="re-initialise dead features from high-error inputs")
    sparsity_weight: float = Field(0.01, ge=0, description="L1 weight of the ReLU SAE (ignored by TopK)")

    def get_hidden_dim(self, input_dim: int) -> int:
        return self.expansion_factor * input_dim


class TrainingConfig(BaseModel):
    """Optimisation hyper-parameters."""

    batch_size: int = Field(128, ge=1)
    learning_rate: float = Field(1e-4, gt=0)
    weight_decay: float = Field(0.0, ge=0)
    epochs: int = Field(50, ge=1)
    warmup_steps: int = Field(1000, ge=0)
    gradient_clip: float = Field(1.0, gt=0)
    use_amp: bool = Field(True, description="bf16 MFMA contractions (fp32 accumulate) instead of fp32 MFMA")
    checkpoint_every: int = Field(10, description="epochs between checkpoints")
    seed: int = Field(42)
    num_workers: int = Field(4, ge=0)
    resample_dead_every: int = Field(5000, ge=1)
    resample_batch_size: int = Field(8192, ge=1)
    grad_exchange_dtype: str = Field("fp32", pattern="^(auto|fp32|bf16)$",
                                     description="data-parallel runs only (not in the reference, which is single-process): "
                                                 "dtype of the gradient all-reduce. fp32 (default) = the exact data-parallel "
                                                 "gradient; bf16 = half the bytes over xGMI, every rank's gradient rounded "
                                                 "once to bf16 and summed in bf16 (opt-in: it changes the step); auto = bf16 "
                                                 "when use_amp, fp32 otherwise")
    ddp_overlap_halves: bool = Field(False,
                                     description="data-parallel runs only: run the backward in two halves (decoder matrix, then "
                                                 "encoder matrix + biases) so that the first half's all-reduce runs under the second "
                                                 "half's contraction. Each half re-does the contraction's epilogue and doubles the "
                                                 "split-K slabs: +86 us of kernels at 384->3072 / B = 16384 on MI355X "
                                                 "(profiles/r03_ddp_structure.txt), so it pays only when one half's all-reduce takes "
                                                 "longer than that - not over xGMI at these sizes; default off = one contraction "
                                                 "launch, one collective")
    ddp_comm_reserve_cus: int = Field(24, ge=0, le=128,
                                      description="data-parallel runs only: compute units the encoder half of the backward leaves "
                                                  "free so that the all-reduce of the decoder half can run beside it (0 = none)")


class DataConfig(BaseModel):
    """Where the audio (and the activation cache) comes from."""

    dataset_name: str = Field("librispeech_asr")
    dataset_subset: str = Field("clean")
    dataset_split: str = Field("train.100")
    max_samples: int = Field(100_000, ge=1)
    cache_dir: Path = Field(Path("cache"))
    streaming: bool = Field(True)


class WandbConfig(BaseModel):
    """Weights & Biases logging (optional dependency)."""

    enabled: bool = Field(True)
    project: str = Field("whisper-sae")
    entity: Optional[str] = Field(None)
    name: Optional[str] = Field(None)
    tags: list[str] = Field(default_factory=list)
    log_every: int = Field(100)


class ExperimentConfig(BaseModel):
    """Everything ``scripts/train.py`` needs for one run."""

    whisper: WhisperConfig = Field(default_factory=WhisperConfig)
    sae: SAEConfig = Field(default_factory=SAEConfig)
    training: TrainingConfig = Field(default_factory=TrainingConfig)
    data: DataConfig = Field(default_factory=DataConfig)
    wandb: WandbConfig = Field(default_factory=WandbConfig)
    encoder_layers: list[int] = Field(default_factory=lambda: list(range(4)))
    decoder_layers: list[int] = Field(default_factory=lambda: list(range(4)))
    output_dir: Path = Field(Path("outputs"))
    experiment_name: str = Field("default")

    @classmethod
    def from_yaml(cls, path) -> "ExperimentConfig":
        with open(path, "r") as fh:
            raw = yaml.safe_load(fh) or {}
        return cls(**raw)

    def to_yaml(self, path) -> None:
        with open(path, "w") as fh:
            yaml.dump(self.model_dump(mode="json"), fh, default_flow_style=False)

    def get_run_dir(self) -> Path:
        target = self.output_dir / self.experiment_name
        target.mkdir(parents=True, exist_ok=True)
        return target


class LayerConfig(BaseModel):
    """One (component, layer) pair an SAE is trained on."""

    component: Literal["encoder", "decoder"]
    layer_idx: int = Field(ge=0)
    input_dim: int
    sae_config: SAEConfig = Field(default_factory=SAEConfig)
    training_config: TrainingConfig = Field(default_factory=TrainingConfig)

    @property
    def name(self) -> str:
        return f"{self.component}_layer{self.layer_idx}"

    @property
    def hidden_dim(self) -> int:
        return self.sae_config.get_hidden_dim(self.input_dim)
