"""Top-k activating examples per feature, tracked on the GPU.

Drop-in for ``FeatureActivation`` / ``TopKTracker`` / ``collect_top_activations`` of the reference
(/root/reference/src/whisper_sae/analysis/feature_viz.py:22-250, :425-484).  The reference walks the dense
``[batch, seq, features]`` tensor on the CPU, one Python heap per feature; here the per-feature lists live in HBM
([H][k] values + arrival ordinals) and one call of ``wsae_feature_topk_update`` (include/wsae.h) merges a batch
into them -- fed by the compact ``(values, indices)`` code of ``TopKSAE.encode_compact`` when the model offers it,
by the dense activations otherwise.  Transcriptions and metadata stay on the host and are joined to the device
lists when examples are read.

Ties: the reference keeps the earlier of two equal activations at the boundary (``>`` at feature_viz.py:153) but
raises ``TypeError`` when two equal values meet inside a heap (tuples fall through to comparing dataclasses); here
equal values are ordered by arrival (batch row, then position), always.

There is no CPU implementation: ``update`` raises ``WsaeError`` without the HIP library or a GPU.
"""

from __future__ import annotations

import bisect
import json
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Optional

import numpy as np
import torch
from torch import Tensor

from .. import _native as N

MAX_K = 64            # one list element per lane of a wavefront
_CHUNK_ENTRIES = 1 << 27  # entries per kernel call (8 B of workspace each)


@dataclass
class FeatureActivation:
    """One activation of one feature (reference feature_viz.py:22-57; same fields, same dict form)."""

    feature_idx: int
    activation_value: float
    sample_idx: int
    position_idx: int
    timestamp_ms: Optional[float] = None
    transcription: Optional[str] = None
    transcription_context: Optional[str] = None
    audio_path: Optional[str] = None
    metadata: dict = field(default_factory=dict)

    def to_dict(self) -> dict:
        return {"feature_idx": self.feature_idx, "activation_value": self.activation_value,
                "sample_idx": self.sample_idx, "position_idx": self.position_idx, "timestamp_ms": self.timestamp_ms,
                "transcription": self.transcription, "transcription_context": self.transcription_context,
                "audio_path": self.audio_path, "metadata": self.metadata}

    @classmethod
    def from_dict(cls, d: dict) -> "FeatureActivation":
        return cls(**d)


class _Segment:
    """Host record of one update: ordinals [base, base + rows) = (batch row b, position p) in row-major order."""

    __slots__ = ("base", "rows", "seq_len", "samples", "positions", "transcriptions", "metadata", "extras")

    def __init__(self, base, rows, seq_len, samples, positions=None, transcriptions=None, metadata=None, extras=None):
        self.base, self.rows, self.seq_len = base, rows, seq_len
        self.samples, self.positions = samples, positions  # positions: explicit per ordinal (loaded state) or None
        self.transcriptions, self.metadata, self.extras = transcriptions, metadata, extras


class TopKTracker:
    """Keeps the ``k`` strongest activations of every feature (reference feature_viz.py:59-250).

    ``update`` takes what the reference takes (dense activations ``[batch, features]`` or ``[batch, seq, features]``);
    ``update_compact`` takes the TopK code directly.  Reading (``get_top_examples`` ...) copies the lists to the host
    once per update.
    """

    def __init__(self, num_features: int, k: int = 20, device: Optional[torch.device | str] = None):
        if not 1 <= k <= MAX_K:
            raise ValueError(f"TopKTracker: k must be in [1, {MAX_K}] (one list element per wavefront lane), got {k}")
        self.num_features = int(num_features)
        self.k = int(k)
        self.device = torch.device(device) if device is not None else None
        self.samples_processed = 0
        self._next_ord = 0
        self._segments: list[_Segment] = []
        self._prune_at = 512  # segments kept before the host records are pruned (doubles while most stay live)
        self._bases: list[int] = []
        # device state (created by the first update) and its host copy
        self._vals: Optional[Tensor] = None
        self._ord: Optional[Tensor] = None
        self._cnt: Optional[Tensor] = None
        self._total: Optional[Tensor] = None
        self._ws: Optional[Tensor] = None
        self._host = (np.zeros((self.num_features, self.k), np.float32), np.zeros((self.num_features, self.k), np.int64),
                      np.zeros(self.num_features, np.int32))
        self._host_total = 0
        self._host_valid = True

    # ---- state ----------------------------------------------------------------------------------
    @property
    def total_activations(self) -> int:
        self._sync_host()
        return self._host_total

    @total_activations.setter
    def total_activations(self, v: int) -> None:
        self._sync_host()
        self._host_total = int(v)
        if self._total is not None:
            self._total.fill_(int(v))

    def _ensure_device(self, like: Optional[Tensor]) -> torch.device:
        if self._vals is not None:
            return self._vals.device
        dev = self.device or (like.device if like is not None and like.is_cuda else None)
        if dev is None:
            if not torch.cuda.is_available():
                raise N.WsaeError("TopKTracker.update needs a GPU: the per-feature lists live in device memory and "
                                  "there is no CPU implementation")
            dev = torch.device("cuda", torch.cuda.current_device())
        N.lib()  # fail loudly when the HIP library is not built
        hv, ho, hc = self._host
        self._vals = torch.from_numpy(hv).to(dev)
        self._ord = torch.from_numpy(ho).to(dev)
        self._cnt = torch.from_numpy(hc).to(dev)
        self._total = torch.tensor([self._host_total], dtype=torch.int64, device=dev)
        self.device = dev
        return dev

    def _sync_host(self) -> None:
        if self._host_valid or self._vals is None:
            return
        self._host = (self._vals.cpu().numpy(), self._ord.cpu().numpy(), self._cnt.cpu().numpy())
        self._host_total = int(self._total.item())
        self._host_valid = True

    # ---- updates --------------------------------------------------------------------------------
    def _record(self, rows: int, seq_len: int, sample_indices, transcriptions, metadata_list) -> int:
        if isinstance(sample_indices, Tensor):
            sample_indices = sample_indices.tolist()
        samples = [int(s) for s in sample_indices]
        if len(samples) * seq_len != rows:
            raise ValueError(f"TopKTracker.update: {len(samples)} sample indices for {rows // max(seq_len, 1)} samples")
        # prune BEFORE the new update is recorded: every earlier update has been merged into the device lists by then
        # (pruning after the append dropped the segment just added, whose ordinals no list holds yet).  The threshold
        # doubles whenever a prune leaves more than half of it alive, so a tracker whose segments all stay live does not
        # synchronise the host on every update.
        if len(self._segments) >= self._prune_at:
            self._prune()
            while len(self._segments) * 2 > self._prune_at:
                self._prune_at *= 2
        base = self._next_ord
        self._segments.append(_Segment(base, rows, seq_len, samples, None,
                                       list(transcriptions) if transcriptions else None,
                                       [dict(m) if m else {} for m in metadata_list] if metadata_list else None))
        self._bases.append(base)
        self._next_ord += rows
        self.samples_processed += len(samples)
        return base

    def _launch(self, vals: Tensor, idx: Optional[Tensor], rows: int, width: int, base: int) -> None:
        lib = N.lib()
        H = self.num_features
        step = max(1, _CHUNK_ENTRIES // width)
        for r0 in range(0, rows, step):
            r = min(step, rows - r0)
            need = int(lib.wsae_feature_topk_workspace_bytes(r * width, H))
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=vals.device)
            v = vals[r0:r0 + r]
            i = idx[r0:r0 + r] if idx is not None else None
            with torch.cuda.device(vals.device):
                N.check(lib.wsae_feature_topk_update(v.data_ptr(), N.ptr(i), r, width, H, self.k, base + r0,
                                                     self._vals.data_ptr(), self._ord.data_ptr(), self._cnt.data_ptr(),
                                                     self._total.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                                     torch.cuda.current_stream(vals.device).cuda_stream),
                        "wsae_feature_topk_update")
        self._host_valid = False

    def update(self, activations: Tensor, sample_indices, transcriptions: Optional[list] = None,
               metadata_list: Optional[list] = None) -> None:
        """Dense activations ``[batch, features]`` or ``[batch, seq, features]`` (reference feature_viz.py:94-158)."""
        a = activations.detach()
        if a.ndim == 2:
            a = a.unsqueeze(1)
        batch, seq_len, nf = a.shape
        assert nf == self.num_features
        dev = self._ensure_device(a)
        a = a.to(device=dev, dtype=torch.float32).reshape(batch * seq_len, nf).contiguous()
        base = self._record(batch * seq_len, seq_len, sample_indices, transcriptions, metadata_list)
        self._launch(a, None, batch * seq_len, nf, base)

    def update_compact(self, values: Tensor, indices: Tensor, sample_indices, seq_len: int = 1,
                       transcriptions: Optional[list] = None, metadata_list: Optional[list] = None) -> None:
        """The TopK code of ``batch * seq_len`` activation rows: ``values``/``indices`` ``[rows, k]`` as
        ``TopKSAE.encode_compact`` returns them (values <= 0 are not activations, as in the dense form)."""
        width = values.shape[-1]
        v = values.detach().reshape(-1, width)
        i = indices.detach().reshape(-1, width)
        rows = v.shape[0]
        dev = self._ensure_device(v)
        v = v.to(device=dev, dtype=torch.float32).contiguous()
        i = i.to(device=dev, dtype=torch.int32).contiguous()
        base = self._record(rows, seq_len, sample_indices, transcriptions, metadata_list)
        self._launch(v, i, rows, width, base)

    # ---- host join ------------------------------------------------------------------------------
    def _prune(self) -> None:
        """Drop the host records of updates no list refers to any more."""
        self._sync_host()
        _, ho, hc = self._host
        live = set()
        mask = np.arange(self.k)[None, :] < hc[:, None]
        for o in np.unique(ho[mask]):
            live.add(bisect.bisect_right(self._bases, int(o)) - 1)
        keep = [s for j, s in enumerate(self._segments) if j in live]
        self._segments = keep
        self._bases = [s.base for s in keep]

    def _example(self, f: int, value: float, ordinal: int) -> FeatureActivation:
        seg = self._segments[bisect.bisect_right(self._bases, ordinal) - 1]
        off = ordinal - seg.base
        if seg.positions is not None:  # loaded state: everything explicit
            extra = seg.extras[off]
            return FeatureActivation(feature_idx=f, activation_value=value, sample_idx=seg.samples[off],
                                     position_idx=seg.positions[off], **extra)
        b, pos = divmod(off, seg.seq_len)
        return FeatureActivation(
            feature_idx=f, activation_value=value, sample_idx=seg.samples[b], position_idx=pos,
            timestamp_ms=pos * 10.0,  # Whisper frames are 10 ms (feature_viz.py:140)
            transcription=seg.transcriptions[b] if seg.transcriptions else None,
            metadata=dict(seg.metadata[b]) if seg.metadata else {})

    def get_top_examples(self, feature_idx: int) -> list:
        """The feature's examples, strongest first (reference feature_viz.py:160-172)."""
        self._sync_host()
        hv, ho, hc = self._host
        n = int(hc[feature_idx])
        return [self._example(feature_idx, float(hv[feature_idx, j]), int(ho[feature_idx, j])) for j in range(n)]

    def get_all_top_examples(self) -> dict:
        return {i: self.get_top_examples(i) for i in range(self.num_features)}

    def get_feature_stats(self) -> dict:
        """num_examples / max / min / mean of the kept activations per feature (reference feature_viz.py:182-207)."""
        self._sync_host()
        hv, _, hc = self._host
        stats = {}
        for i in range(self.num_features):
            n = int(hc[i])
            if n:
                acts = [float(x) for x in hv[i, :n]]
                stats[i] = {"num_examples": n, "max_activation": max(acts), "min_activation": min(acts),
                            "mean_activation": sum(acts) / n}
            else:
                stats[i] = {"num_examples": 0, "max_activation": 0.0, "min_activation": 0.0, "mean_activation": 0.0}
        return stats

    # ---- JSON (same schema as the reference, feature_viz.py:209-250) -----------------------------
    def save(self, path) -> None:
        data = {"num_features": self.num_features, "k": self.k, "total_activations": self.total_activations,
                "samples_processed": self.samples_processed, "features": {}}
        for i in range(self.num_features):
            ex = self.get_top_examples(i)
            if ex:
                data["features"][str(i)] = [e.to_dict() for e in ex]
        with open(Path(path), "w") as fh:
            json.dump(data, fh, indent=2)

    @classmethod
    def load(cls, path, device=None) -> "TopKTracker":
        with open(Path(path)) as fh:
            data = json.load(fh)
        t = cls(num_features=data["num_features"], k=data["k"], device=device)
        t.samples_processed = data["samples_processed"]
        hv, ho, hc = t._host
        samples, positions, extras = [], [], []
        for fs, examples in data["features"].items():
            f = int(fs)
            acts = sorted((FeatureActivation.from_dict(e) for e in examples), key=lambda a: -a.activation_value)[:t.k]
            for j, a in enumerate(acts):
                hv[f, j], ho[f, j] = a.activation_value, len(samples)
                samples.append(a.sample_idx)
                positions.append(a.position_idx)
                extras.append({"timestamp_ms": a.timestamp_ms, "transcription": a.transcription,
                               "transcription_context": a.transcription_context, "audio_path": a.audio_path,
                               "metadata": a.metadata})
            hc[f] = len(acts)
        if samples:
            t._segments.append(_Segment(0, len(samples), 1, samples, positions, extras=extras))
            t._bases.append(0)
            t._next_ord = len(samples)
        t._host_total = int(data["total_activations"])
        return t


def collect_top_activations(model: torch.nn.Module, dataloader, num_features: int, k: int = 20,
                            device: str = "cuda") -> TopKTracker:
    """Top-k activating examples over a dataset (reference feature_viz.py:425-484).  A model with
    ``encode_compact`` (TopKSAE) hands its ``(values, indices)`` code straight to the tracker -- the dense
    ``[batch, features]`` matrix is never built; any other model goes through ``encode`` / ``forward`` as there."""
    tracker = TopKTracker(num_features=num_features, k=k, device=device)
    model.eval()
    sample_idx = 0
    with torch.no_grad():
        for batch in dataloader:
            if isinstance(batch, (tuple, list)):
                activations = batch[0]
                metadata = batch[1] if len(batch) > 1 else None
            else:
                activations, metadata = batch, None
            activations = activations.to(device)
            transcriptions = metadata.get("transcriptions") if isinstance(metadata, dict) else None
            if hasattr(model, "encode_compact"):
                lead = activations.shape[:-1]
                vals, idx = model.encode_compact(activations)
                batch_size = lead[0]
                seq_len = int(np.prod(lead[1:])) if len(lead) > 1 else 1
                tracker.update_compact(vals, idx, list(range(sample_idx, sample_idx + batch_size)), seq_len=seq_len,
                                       transcriptions=transcriptions)
            else:
                if hasattr(model, "encode"):
                    hidden = model.encode(activations)
                else:
                    output = model(activations)
                    hidden = output.hidden if hasattr(output, "hidden") else output[1]
                batch_size = hidden.shape[0]
                tracker.update(hidden, list(range(sample_idx, sample_idx + batch_size)), transcriptions=transcriptions)
            sample_idx += batch_size
    return tracker
