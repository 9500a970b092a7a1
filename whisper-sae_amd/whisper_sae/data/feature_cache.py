"""Activation feed for SAE training on MI355X.

Replaces the feed half of the reference's ``src/whisper_sae/data/feature_cache.py`` (:169-197 --
``torch.load`` -> ``TensorDataset`` -> ``DataLoader(shuffle=True, pin_memory=True)`` -> per-step H2D
copy) with an **on-device ring buffer**: all activation rows of a layer sit in HBM (bf16 or f32),
an epoch is a seeded on-device permutation of the row indices, and a batch is just the index list --
the encode / decode kernels gather the rows from the ring themselves.

The on-disk cache format is the reference's (so ``--extract-only`` caches are interchangeable):
``<cache>/<model_short>_<component>_layer<N>.pt`` = ``torch.save(float32 [num_tokens, hidden_dim])``
plus ``..._meta.json`` with the ``CacheMetadata`` fields (feature_cache.py:23-57, :87-167).

``extract_and_cache_features`` (reference feature_cache.py:200-306) drives a Whisper model over batches of mel features
and leaves the tapped layers' activations in the reference's cache files and / or directly in ``ActivationRing`` objects
(row N2: producer -> ring -> trainer with no host round trip).  Audio IO and model download stay outside this build.
"""

from __future__ import annotations

import ctypes as C
import json
import math
from dataclasses import asdict, dataclass
from datetime import datetime
from pathlib import Path
from typing import Iterator, Literal, Optional

import torch
from torch import Tensor

from .. import _native as N
from ..config import DataConfig, WhisperConfig
from ..sae.engine import _dtype_code, require_device_tensor
from ..sae.training import RingBatch


@dataclass
class CacheMetadata:
    """Sidecar JSON of one cached layer (same fields as the reference's ``CacheMetadata``)."""

    model_name: str
    component: str
    layer_idx: int
    hidden_dim: int
    num_samples: int
    num_tokens: int
    created_at: str
    data_config: dict

    def to_json(self) -> str:
        def plain(v):
            if isinstance(v, Path):
                return str(v)
            if isinstance(v, dict):
                return {k: plain(x) for k, x in v.items()}
            return v
        return json.dumps({k: plain(v) for k, v in asdict(self).items()}, indent=2)

    @classmethod
    def from_json(cls, text: str) -> "CacheMetadata":
        return cls(**json.loads(text))


class ActivationRing:
    """Rows of activations resident in HBM (``wsae_ring``), drawn from as shuffled index batches."""

    def __init__(self, capacity_rows: int, dim: int, device="cuda", dtype: torch.dtype = torch.bfloat16):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise N.WsaeError("ActivationRing lives in GPU memory: pass a ROCm device")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if dtype not in (torch.bfloat16, torch.float32):
            raise TypeError("ring dtype must be torch.bfloat16 or torch.float32")
        self.dim, self.capacity, self.dtype = int(dim), int(capacity_rows), dtype
        self.lib = N.lib()
        handle = N._p()
        code = N.DT_BF16 if dtype == torch.bfloat16 else N.DT_F32
        N.check(self.lib.wsae_ring_create(self.device.index, self.capacity, self.dim, code, handle), "wsae_ring_create")
        self._h = handle.value
        self._storage = self._wrap()

    def _wrap(self) -> Tensor:
        """Zero-copy torch view of the ring's HBM storage (lifetime tied to this object)."""
        ptr = self.lib.wsae_ring_data(self._h)
        esz = 2 if self.dtype == torch.bfloat16 else 4
        nbytes = self.capacity * self.dim * esz
        iface = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 3}
        holder = type("_RingMem", (), {"__cuda_array_interface__": iface})()
        with torch.cuda.device(self.device):
            raw = torch.as_tensor(holder, device=self.device)
        self._holder = holder
        return raw.view(self.dtype).view(self.capacity, self.dim)

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    @property
    def data(self) -> Tensor:
        return self._storage

    def __len__(self) -> int:
        return int(self.lib.wsae_ring_size(self._h))

    def push(self, rows: Tensor) -> None:
        """Append activation rows ``[n, dim]`` (any device; converted to the ring dtype on the GPU)."""
        rows = rows.reshape(-1, self.dim)
        if rows.dtype not in (torch.float32, torch.bfloat16):
            rows = rows.float()
        step = 1 << 20
        for s in range(0, rows.shape[0], step):  # bounded staging for host-resident caches
            part = rows[s:s + step].to(self.device, non_blocking=True).contiguous()
            N.check(self.lib.wsae_ring_push(self._h, part.data_ptr(), _dtype_code(part), part.shape[0], self._stream()),
                    "wsae_ring_push")
            torch.cuda.current_stream(self.device).synchronize()  # `part` must outlive the copy kernel

    def push_layernorm(self, hidden: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> None:
        """``LayerNorm(hidden)`` rows straight into the ring (one kernel: normalise, cast, store; row N2).

        ``hidden``: device tensor ``[..., dim]`` (the hooked layer's output), ``weight`` / ``bias``: the LayerNorm's
        parameters (Whisper's final encoder / decoder norm, reference sae/hooks.py:86-88)."""
        rows = hidden.detach().reshape(-1, self.dim)
        if rows.device != self.device:
            raise N.WsaeError("push_layernorm takes hidden states that already live on the ring's device")
        if rows.dtype not in (torch.float32, torch.bfloat16):
            rows = rows.float()
        rows = rows.contiguous()
        w = weight.detach().to(device=self.device, dtype=torch.float32).contiguous()
        b = bias.detach().to(device=self.device, dtype=torch.float32).contiguous()
        N.check(self.lib.wsae_ring_push_layernorm(self._h, rows.data_ptr(), _dtype_code(rows), rows.shape[0], w.data_ptr(),
                                                  b.data_ptr(), float(eps), self._stream()), "wsae_ring_push_layernorm")
        rows.record_stream(torch.cuda.current_stream(self.device))

    def fill_synthetic(self, n_rows: int, seed: int = 42) -> None:
        N.check(self.lib.wsae_ring_fill_synthetic(self._h, C.c_uint64(seed), int(n_rows), self._stream()),
                "wsae_ring_fill_synthetic")

    def sample(self, n: int, seed: int, epoch: int, offset: int, out: Optional[Tensor] = None) -> Tensor:
        """Row indices ``perm_{seed,epoch}(offset .. offset+n)`` as int32 on the device."""
        if out is None:
            out = torch.empty(n, dtype=torch.int32, device=self.device)
        N.check(self.lib.wsae_ring_sample(self._h, C.c_uint64(seed), int(epoch), int(offset), int(n), out.data_ptr(),
                                          self._stream()), "wsae_ring_sample")
        return out

    def batch(self, n: int, seed: int, epoch: int, offset: int) -> RingBatch:
        return RingBatch(self._storage, self.sample(n, seed, epoch, offset))

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._storage = None
            self.lib.wsae_ring_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RingLoader:
    """Iterable of ``RingBatch`` with ``DataLoader`` semantics: ``len() = ceil(N / batch_size)``, last
    batch partial (no ``drop_last``, feature_cache.py:191-197), reshuffled every epoch.

    ``rank`` / ``world_size`` shard every epoch's permutation across data-parallel ranks: rank r takes
    positions ``r, r+world, ...`` of each global batch, so one global step consumes
    ``world_size * batch_size`` distinct rows and no two ranks see the same row in an epoch.
    """

    def __init__(self, ring: ActivationRing, batch_size: int, shuffle: bool = True, seed: int = 42, rank: int = 0,
                 world_size: int = 1):
        self.ring, self.batch_size, self.shuffle, self.seed = ring, int(batch_size), shuffle, int(seed)
        self.rank, self.world = int(rank), int(world_size)
        self.epoch = 0

    def __len__(self) -> int:
        return math.ceil(len(self.ring) / (self.batch_size * self.world))

    #: steps whose row indices are drawn by one ``wsae_ring_sample`` launch (a launch per step costs ~5 us
    #: of a ~330 us step; the indices of 16 steps are 1 MB)
    PREFETCH_STEPS = 16

    def __iter__(self) -> Iterator[RingBatch]:
        n = len(self.ring)
        per_step = self.batch_size * self.world
        epoch = self.epoch
        self.epoch += 1
        nsteps = len(self)
        step = 0
        while step < nsteps:
            lo = step * per_step + self.rank * self.batch_size
            if not self.shuffle:
                take = max(0, min(self.batch_size, n - lo))
                if take == 0:
                    lo, take = 0, min(self.batch_size, n)
                rows = torch.arange(lo, lo + take, dtype=torch.int32, device=self.ring.device)
                yield RingBatch(self.ring.data, rows)
                step += 1
                continue
            # full global steps ahead of us: positions [step * per_step, (step + g) * per_step) of the epoch's permutation
            g = min(self.PREFETCH_STEPS, nsteps - step, (n - step * per_step) // per_step)
            if g >= 2:
                block = self.ring.sample(g * per_step, self.seed, epoch, step * per_step).view(g, per_step)
                mine = slice(self.rank * self.batch_size, (self.rank + 1) * self.batch_size)
                for i in range(g):
                    yield RingBatch(self.ring.data, block[i, mine])
                step += g
                continue
            take = max(0, min(self.batch_size, n - lo))
            if take == 0:  # ragged tail: this rank re-reads rows from the start of the permutation
                lo, take = 0, min(self.batch_size, n)
            yield self.ring.batch(take, self.seed, epoch, lo)
            step += 1


class FeatureCache:
    """Per-layer activation cache on disk (reference ``FeatureCache``, feature_cache.py:60-197)."""

    def __init__(self, cache_dir: Path, whisper_config: WhisperConfig, data_config: DataConfig):
        self.cache_dir = Path(cache_dir)
        self.cache_dir.mkdir(parents=True, exist_ok=True)
        self.whisper_config = whisper_config
        self.data_config = data_config
        self.model_short = whisper_config.model_name.split("/")[-1]

    def _stem(self, component: str, layer_idx: int) -> str:
        return f"{self.model_short}_{component}_layer{layer_idx}"

    def _get_cache_path(self, component: str, layer_idx: int) -> Path:
        return self.cache_dir / f"{self._stem(component, layer_idx)}.pt"

    def _get_metadata_path(self, component: str, layer_idx: int) -> Path:
        return self.cache_dir / f"{self._stem(component, layer_idx)}_meta.json"

    def has_cache(self, component: str, layer_idx: int) -> bool:
        return self._get_cache_path(component, layer_idx).exists() and \
            self._get_metadata_path(component, layer_idx).exists()

    def load(self, component: str, layer_idx: int) -> tuple:
        feats = torch.load(self._get_cache_path(component, layer_idx), weights_only=True)
        meta = CacheMetadata.from_json(self._get_metadata_path(component, layer_idx).read_text())
        return feats, meta

    def save(self, features: Tensor, component: str, layer_idx: int, num_samples: int) -> None:
        torch.save(features, self._get_cache_path(component, layer_idx))
        meta = CacheMetadata(model_name=self.whisper_config.model_name, component=component, layer_idx=layer_idx,
                             hidden_dim=int(features.shape[-1]), num_samples=int(num_samples),
                             num_tokens=int(features.shape[0]), created_at=datetime.now().isoformat(),
                             data_config=self.data_config.model_dump())
        self._get_metadata_path(component, layer_idx).write_text(meta.to_json())

    def get_ring(self, component: str, layer_idx: int, device="cuda", dtype: torch.dtype = torch.bfloat16,
                 features: Optional[Tensor] = None) -> ActivationRing:
        """Load a cached layer into an on-device ring (one pass over the file, chunked H2D)."""
        if features is None:
            features, _ = self.load(component, layer_idx)
        ring = ActivationRing(features.shape[0], features.shape[1], device=device, dtype=dtype)
        ring.push(features)
        return ring

    def get_dataloader(self, component: Literal["encoder", "decoder"], layer_idx: int, batch_size: int,
                       shuffle: bool = True, num_workers: int = 0, device="cuda",
                       dtype: torch.dtype = torch.bfloat16, seed: int = 42, rank: int = 0, world_size: int = 1,
                       features: Optional[Tensor] = None) -> RingLoader:
        """Same call as the reference's ``get_dataloader`` (``num_workers`` is accepted and unused: there
        are no loader processes); returns a ``RingLoader`` over the HBM-resident rows."""
        del num_workers
        ring = self.get_ring(component, layer_idx, device=device, dtype=dtype, features=features)
        return RingLoader(ring, batch_size, shuffle=shuffle, seed=seed, rank=rank, world_size=world_size)


def extract_and_cache_features(whisper_model, processor, audio_dataloader, cache: Optional[FeatureCache], encoder_layers: list,
                               decoder_layers: list, device="cpu", max_samples: Optional[int] = None,
                               rings: Optional[dict] = None, show_progress: bool = True) -> dict:
    """Run ``whisper_model`` over ``audio_dataloader`` and keep the tapped layers' activations (reference
    feature_cache.py:200-306; same positional arguments - ``processor`` is accepted and, as in the reference, unused).

    Every batch (a tensor of mel features ``[B, n_mels, frames]``, or a tuple / list whose first item is one) goes
    through the encoder and, when decoder layers are tapped, one decoder step from the start token; the final LayerNorm
    is applied to each tapped output (``apply_layer_norm=True`` as the reference hard-codes).  Batches are taken while
    ``num_samples < max_samples`` - whole batches, so the count can overshoot exactly as the reference's does.

    Where the activations go:

    * layers named in ``rings`` (``{("encoder" | "decoder", layer): ActivationRing}``; not in the reference): LayerNorm +
      push into the ring in one kernel per hooked call, nothing cached, nothing written;
    * every other layer: flattened ``[tokens, D]`` chunks are collected on the host and written with ``cache.save`` - the
      reference's ``.pt`` + ``_meta.json`` pair (``cache`` may be ``None`` only when every layer has a ring).

    Returns ``{"num_samples": n, "tokens": {(component, layer): rows}}``.
    """
    from ..sae.hooks import WhisperActivationExtractor, flatten_activations, run_whisper_taps

    del processor  # (the reference's signature; its body never reads it either)
    rings = dict(rings or {})
    wanted = [("encoder", i) for i in encoder_layers] + [("decoder", i) for i in decoder_layers]
    unknown = [key for key in rings if key not in wanted]
    if unknown:
        raise ValueError(f"rings given for layers that are not extracted: {unknown}")
    to_disk = [key for key in wanted if key not in rings]
    if to_disk and cache is None:
        raise ValueError(f"no FeatureCache for the layers without a ring: {to_disk}")
    whisper_model = whisper_model.to(device)
    whisper_model.eval()
    extractor = WhisperActivationExtractor(model=whisper_model, encoder_layers=encoder_layers, decoder_layers=decoder_layers,
                                           apply_layer_norm=True)
    for (component, layer), ring in rings.items():
        extractor.attach_ring(component, layer, ring)
    chunks: dict = {key: [] for key in to_disk}
    before = {key: len(ring) for key, ring in rings.items()}
    num_samples = 0
    limit = max_samples if max_samples else float("inf")

    progress = task = None
    if show_progress:
        from rich.progress import BarColumn, Progress, SpinnerColumn, TaskProgressColumn, TextColumn
        progress = Progress(SpinnerColumn(), TextColumn("[progress.description]{task.description}"), BarColumn(),
                            TaskProgressColumn())
        progress.start()
        task = progress.add_task("[cyan]Extracting features...", total=max_samples if max_samples else None)
    try:
        with torch.no_grad(), extractor:
            for batch in audio_dataloader:
                if num_samples >= limit:
                    break
                if isinstance(batch, (list, tuple)):
                    batch = batch[0]
                batch_size = run_whisper_taps(whisper_model, extractor, batch, device)
                for component, layer in to_disk:
                    held = getattr(extractor.cache, component).get(layer)
                    if held:  # this batch's capture (the hook appended exactly one tensor)
                        chunks[(component, layer)].append(flatten_activations(held[-1], component))
                extractor.clear_cache()  # the chunks are kept flattened; the reference's cache grows without bound
                num_samples += batch_size
                if progress is not None:
                    progress.update(task, completed=min(num_samples, limit))
    finally:
        if progress is not None:
            progress.stop()

    tokens: dict = {}
    for component, layer in to_disk:
        if chunks[(component, layer)]:
            features = torch.cat(chunks[(component, layer)], dim=0)
            cache.save(features, component, layer, num_samples)
            tokens[(component, layer)] = int(features.shape[0])
            print(f"Cached {component} layer {layer}: {features.shape}")
    for key, ring in rings.items():
        tokens[key] = int(extractor.rows_delivered.get(key, 0))
        print(f"Ring {key[0]} layer {key[1]}: {tokens[key]} rows pushed, {len(ring)} resident (was {before[key]})")
    return {"num_samples": num_samples, "tokens": tokens}


__all__ = ["ActivationRing", "CacheMetadata", "FeatureCache", "RingBatch", "RingLoader", "extract_and_cache_features",
           "require_device_tensor"]
