#!/usr/bin/env python3
"""Train TopK SAEs on cached Whisper activations with the MI355X train step.

Same command line as the reference's ``scripts/train.py`` (``--config --layer --no-wandb
--extract-only --device --seed``, :40-81).  What differs:

* activations are loaded once into the on-device ring buffer and batches are drawn there;
* the Whisper model is only needed for extraction, which is outside this build's scope
  (``--extract-only`` and missing caches say so instead of downloading a model);
* ``--synthetic N`` trains on N synthetic activation rows (no cache needed; benchmarks, smoke runs);
* under ``torchrun`` every rank trains data-parallel on its shard of the rows (RCCL all-reduce).
"""

from __future__ import annotations

import argparse
import os
import random
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "whisper-sae_amd"))

from whisper_sae.config import ExperimentConfig  # noqa: E402
from whisper_sae.data.feature_cache import ActivationRing, FeatureCache, RingLoader  # noqa: E402
from whisper_sae.distributed import rank_and_world  # noqa: E402
from whisper_sae.sae.model import create_sae  # noqa: E402
from whisper_sae.sae.training import SAETrainer  # noqa: E402


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="Train SAEs on Whisper activations (MI355X)")
    ap.add_argument("--config", type=str, default=None, help="YAML experiment config")
    ap.add_argument("--layer", type=str, default=None, help="train one layer only, e.g. encoder:0")
    ap.add_argument("--no-wandb", action="store_true")
    ap.add_argument("--extract-only", action="store_true")
    ap.add_argument("--device", type=str, default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--synthetic", type=int, default=0, metavar="N", help="use N synthetic activation rows")
    ap.add_argument("--epochs", type=int, default=None, help="override training.epochs")
    return ap.parse_args(argv)


def seed_everything(seed: int) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def parse_layer(spec: str):
    kind, _, num = spec.partition(":")
    if kind not in ("encoder", "decoder") or not num.isdigit():
        raise ValueError(f"--layer expects encoder:N or decoder:N, got {spec!r}")
    return kind, int(num)


def train_layer(cfg: ExperimentConfig, component: str, layer_idx: int, cache: FeatureCache, device, synthetic: int,
                epochs: int | None) -> None:
    rank, world = rank_and_world()
    dtype = torch.bfloat16 if cfg.training.use_amp else torch.float32
    if synthetic:
        dim = cfg.whisper.hidden_dim
        ring = ActivationRing(synthetic, dim, device=device, dtype=dtype)
        ring.fill_synthetic(synthetic, seed=cfg.training.seed + rank)
        resample_rows = None
    else:
        if not cache.has_cache(component, layer_idx):
            print(f"no cached activations for {component} layer {layer_idx} under {cache.cache_dir}; extraction "
                  f"(Whisper forward hooks) is outside this build -- produce the cache with the reference's "
                  f"--extract-only, or pass --synthetic N")
            return
        feats, meta = cache.load(component, layer_idx)
        dim = feats.shape[1]
        print(f"loaded {feats.shape[0]:,} tokens, dim={dim} ({meta.model_name})")
        ring = cache.get_ring(component, layer_idx, device=device, dtype=dtype, features=feats)
        resample_rows = feats
    sae = create_sae(cfg.sae, dim)
    loader = RingLoader(ring, cfg.training.batch_size, shuffle=True, seed=cfg.training.seed, rank=rank, world_size=world)
    run_dir = cfg.output_dir / f"{cfg.experiment_name}_{component}_layer{layer_idx}"
    trainer = SAETrainer(sae, cfg.training, device=device, run_dir=run_dir,
                         resample_dead_every=cfg.training.resample_dead_every,
                         resample_batch_size=cfg.training.resample_batch_size,
                         resample_dead=cfg.sae.dead_feature_resample)
    if resample_rows is not None:
        trainer.set_resample_dataset(torch.utils.data.TensorDataset(resample_rows))
    if cfg.wandb.enabled and rank == 0:
        try:
            import wandb
            trainer.wandb_run = wandb.init(project=cfg.wandb.project, entity=cfg.wandb.entity,
                                           name=run_dir.name, tags=cfg.wandb.tags + [component, f"layer{layer_idx}"])
        except Exception as exc:  # optional dependency / offline
            print(f"W&B unavailable ({exc}); continuing without it")
    print(f"training {dim} -> {sae.hidden_dim} (k={cfg.sae.k}) for {epochs or cfg.training.epochs} epochs, "
          f"{len(loader)} steps/epoch, world={world}")
    trainer.train(loader, epochs=epochs or cfg.training.epochs)
    if rank == 0:
        torch.save({k: v.detach().clone() for k, v in sae.state_dict().items()}, run_dir / "sae_final.pt")
        trainer.save_metrics()
        print(f"saved {run_dir / 'sae_final.pt'} and metrics.json")
    if trainer.wandb_run is not None:
        trainer.wandb_run.finish()


def main(argv=None) -> None:
    args = parse_args(argv)
    cfg = ExperimentConfig.from_yaml(args.config) if args.config else ExperimentConfig()
    if args.seed is not None:
        cfg.training.seed = args.seed
    if args.no_wandb:
        cfg.wandb.enabled = False
    if args.extract_only:
        raise SystemExit("--extract-only (Whisper activation extraction) is outside this build's scope; run the "
                         "reference's extraction and point data.cache_dir at its cache")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device(args.device) if args.device else torch.device("cuda", local)
    if device.type != "cuda":
        raise SystemExit("this build trains on ROCm devices only (no CPU path)")
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    seed_everything(cfg.training.seed)
    cache = FeatureCache(cfg.data.cache_dir / "features", cfg.whisper, cfg.data)
    if args.layer:
        layers = [parse_layer(args.layer)]
    else:
        layers = [("encoder", i) for i in cfg.encoder_layers] + [("decoder", i) for i in cfg.decoder_layers]
    for component, idx in layers:
        train_layer(cfg, component, idx, cache, device, args.synthetic, args.epochs)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
