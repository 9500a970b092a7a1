#!/usr/bin/env python3
"""Train TopK SAEs on Whisper activations with the MI355X train step.

Same command line as the reference's ``scripts/train.py`` (``--config --layer --no-wandb
--extract-only --device --seed``, :40-81) and the same flow (:296-329): extract the activations of the
configured layers when their cache is missing (or ``--extract-only`` asks for it), then train one SAE per layer.
What differs:

* the Whisper model comes from a LOCAL directory (``--whisper-path``, loaded with ``local_files_only``) or is a
  seeded random-init model of the configured geometry (``--whisper-random-init``: smoke runs and tests) - never a
  hub name: this build downloads nothing.  ``run()`` also takes a model OBJECT;
* the mel features come from a tensor file (``--mel``: ``torch.save``d ``[N, n_mels, frames]``) or are synthetic
  (``--synthetic-mel N``); audio decoding and the LibriSpeech loader stay outside this build;
* ``--stream-to-ring``: extraction pushes the layer-normed activations straight into the on-device ring the trainer
  samples from (one kernel per hooked call) - no ``.cpu()``, no cache file, no reload;
* cached activations are loaded once into the on-device ring buffer and batches are drawn there;
* ``--synthetic N`` trains on N synthetic activation rows (no cache and no model needed; benchmarks, smoke runs);
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
from whisper_sae.data.feature_cache import ActivationRing, FeatureCache, RingLoader, extract_and_cache_features  # noqa: E402
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
    ap.add_argument("--whisper-path", type=str, default=None,
                    help="local directory of a Whisper checkpoint (HF format); loaded with local_files_only")
    ap.add_argument("--whisper-random-init", action="store_true",
                    help="seeded random-init Whisper of the configured geometry instead of a checkpoint (smoke runs)")
    ap.add_argument("--mel", type=str, default=None, help="torch.save'd mel features [N, n_mels, frames] to extract from")
    ap.add_argument("--synthetic-mel", type=int, default=0, metavar="N", help="extract from N synthetic mel clips")
    ap.add_argument("--mel-frames", type=int, default=3000, help="frames per synthetic clip (3000 = 30 s)")
    ap.add_argument("--stream-to-ring", action="store_true",
                    help="push extracted activations straight into the training ring (no cache files)")
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


def load_whisper(cfg: ExperimentConfig, args, device):
    """The Whisper model to extract from, or None when the command line names none."""
    if args.whisper_path:
        from transformers import WhisperForConditionalGeneration
        model = WhisperForConditionalGeneration.from_pretrained(args.whisper_path, local_files_only=True)
    elif args.whisper_random_init:
        from transformers import WhisperConfig, WhisperForConditionalGeneration
        w = cfg.whisper
        heads = max(1, w.hidden_dim // 64)
        wc = WhisperConfig(d_model=w.hidden_dim, encoder_layers=w.num_encoder_layers, decoder_layers=w.num_decoder_layers,
                           encoder_attention_heads=heads, decoder_attention_heads=heads, encoder_ffn_dim=4 * w.hidden_dim,
                           decoder_ffn_dim=4 * w.hidden_dim, max_source_positions=args.mel_frames // 2)
        torch.manual_seed(cfg.training.seed)
        model = WhisperForConditionalGeneration(wc)
    else:
        return None
    return model.to(device).eval()


def mel_batches(cfg: ExperimentConfig, args, model, batch_size: int = 16):
    """Batches of mel features ``[B, n_mels, frames]`` (reference scripts/train.py:300-307 uses batches of 16)."""
    if args.mel:
        mel = torch.load(args.mel, weights_only=True)
        if mel.ndim != 3:
            raise SystemExit(f"--mel: expected [N, n_mels, frames], got {tuple(mel.shape)}")
    elif args.synthetic_mel:
        g = torch.Generator().manual_seed(cfg.training.seed)
        mel = torch.randn(args.synthetic_mel, model.config.num_mel_bins, args.mel_frames, generator=g)
    else:
        raise SystemExit("extraction needs mel features: --mel FILE or --synthetic-mel N (audio decoding is outside this build)")
    return [mel[i:i + batch_size] for i in range(0, mel.shape[0], batch_size)]


def extract(cfg: ExperimentConfig, model, cache: FeatureCache, layers, device, batches, rings=None) -> dict:
    enc = [i for c, i in layers if c == "encoder"]
    dec = [i for c, i in layers if c == "decoder"]
    all_in_rings = bool(rings) and all(key in rings for key in layers)
    return extract_and_cache_features(model, None, batches, None if all_in_rings else cache, enc, dec, device=device,
                                      max_samples=cfg.data.max_samples, rings=rings)


def train_layer(cfg: ExperimentConfig, component: str, layer_idx: int, cache: FeatureCache, device, synthetic: int,
                epochs: int | None, ring: ActivationRing | None = None) -> None:
    rank, world = rank_and_world()
    dtype = torch.bfloat16 if cfg.training.use_amp else torch.float32
    resample_rows = None
    if ring is not None:  # filled by the extraction pass of this very run (--stream-to-ring)
        dim = ring.dim
        print(f"training from the ring the extraction filled: {len(ring):,} tokens, dim={dim}")
    elif synthetic:
        dim = cfg.whisper.hidden_dim
        ring = ActivationRing(synthetic, dim, device=device, dtype=dtype)
        ring.fill_synthetic(synthetic, seed=cfg.training.seed + rank)
    else:
        if not cache.has_cache(component, layer_idx):
            print(f"no cached activations for {component} layer {layer_idx} under {cache.cache_dir}: run with "
                  f"--whisper-path DIR (or --whisper-random-init) and --mel FILE (or --synthetic-mel N) to extract them, "
                  f"point data.cache_dir at a cache the reference's --extract-only wrote, or pass --synthetic N")
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


def run(cfg: ExperimentConfig, args, whisper_model=None) -> None:
    """Everything after argument parsing; ``whisper_model`` may be handed in as an object (tests, notebooks)."""
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

    # extraction (reference scripts/train.py:283-327): when a cache is missing, or on --extract-only
    rings = {}
    if not args.synthetic:
        missing = [(c, i) for c, i in layers if not cache.has_cache(c, i)]
        if missing or args.extract_only or args.stream_to_ring:
            model = whisper_model if whisper_model is not None else load_whisper(cfg, args, device)
            if model is None:
                if args.extract_only:
                    raise SystemExit("--extract-only needs a model: --whisper-path DIR or --whisper-random-init "
                                     "(this build never downloads one by hub name)")
            else:
                model = model.to(device).eval()
                batches = mel_batches(cfg, args, model)
                if args.stream_to_ring and not args.extract_only:
                    # one ring per layer, sized for what the extraction will push: whole batches while
                    # num_samples < max_samples; an encoder layer yields frames / 2 rows per clip, a decoder layer one
                    dtype = torch.bfloat16 if cfg.training.use_amp else torch.float32
                    clips, taken = 0, 0
                    for b in batches:
                        if taken >= cfg.data.max_samples:
                            break
                        taken += b.shape[0]
                        clips += b.shape[0]
                    frames = batches[0].shape[-1]
                    for c, i in layers:
                        rows = clips * (frames // 2 if c == "encoder" else 1)
                        rings[(c, i)] = ActivationRing(max(rows, 1), model.config.d_model, device=device, dtype=dtype)
                todo = layers if (args.extract_only or rings) else missing
                print(f"extracting {todo} with {type(model).__name__} (d_model {model.config.d_model})")
                extract(cfg, model, cache, todo, device, batches, rings=rings or None)
                del model
                torch.cuda.empty_cache()
    if args.extract_only:
        print("extract-only mode, skipping training")
    else:
        for component, idx in layers:
            train_layer(cfg, component, idx, cache, device, args.synthetic, args.epochs, ring=rings.get((component, idx)))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def main(argv=None, whisper_model=None) -> None:
    args = parse_args(argv)
    cfg = ExperimentConfig.from_yaml(args.config) if args.config else ExperimentConfig()
    if args.seed is not None:
        cfg.training.seed = args.seed
    if args.no_wandb:
        cfg.wandb.enabled = False
    run(cfg, args, whisper_model=whisper_model)


if __name__ == "__main__":
    main()
