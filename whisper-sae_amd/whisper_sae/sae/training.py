"""``SAETrainer`` for MI355X -- the reference's trainer surface (src/whisper_sae/sae/training.py)
on top of the fused HIP train step.

One ``train_step`` = stage batch -> encode GEMM (MFMA) -> [TopK + sparse decode + MSE + dpre] (MFMA, one launch) ->
weight-gradient GEMMs (MFMA) -> [RCCL all-reduce under torch.distributed] -> clip + AdamW + decoder
renorm -> dead-feature scan, all enqueued on the current HIP stream without a host sync.  The
per-step scalars are written by the kernels into a device record and copied asynchronously to
pinned host memory; ``TrainingMetrics`` fields read them lazily, so the five ``.item()`` syncs of
the reference step (training.py:207-213) only happen if and when somebody looks at the numbers.

Documented divergences from the reference:
* autocast dtype is bf16 (MFMA) instead of fp16 + loss scaling; ``trainer.scaler`` is kept as a
  disabled ``GradScaler`` so attribute access keeps working;
* dead-feature resampling, which the reference defines but never calls, runs after the step
  counter advances when ``resample_dead=True`` is passed (``scripts/train.py`` passes
  ``config.sae.dead_feature_resample``);
* models without dead-feature tracking (``ReLUSAE``) report ``dead_feature_ratio = 0`` instead
  of raising ``AttributeError``.
"""

from __future__ import annotations

import bisect
import json
from dataclasses import dataclass, fields
from pathlib import Path
from typing import Optional

import torch
from torch import Tensor
from torch.optim.lr_scheduler import CosineAnnealingLR, LinearLR, SequentialLR

from .. import _native as N
from ..config import TrainingConfig
from ..distributed import WireExchange, barrier, rank_and_world, sync_gradients, world
from .engine import _dtype_code, require_device_tensor
from .model import relu_fp8_flag
from .optim import FusedAdamW


@dataclass
class TrainingMetrics:
    """Scalars of one training step (field set of the reference, training.py:19-29)."""

    loss: float
    reconstruction_loss: float
    sparsity_loss: float
    l0: float
    dead_feature_ratio: float
    learning_rate: float
    step: int


class _PendingMetrics(TrainingMetrics):
    """``TrainingMetrics`` whose device-produced fields are fetched on first access.

    The kernels of a step write its record straight into a slot of a device-side record ring (no
    per-step copy); reading any field synchronises the producing stream once, brings every finished
    record of that ring chunk to the host in ONE copy, and then caches plain floats.
    """

    _LAZY = ("loss", "reconstruction_loss", "sparsity_loss", "l0", "dead_feature_ratio")

    def __init__(self, chunk: "_RecordChunk", slot: int, stream, learning_rate: float, step: int,
                 sparsity_weight: float = 0.0):  # noqa: D401
        object.__setattr__(self, "_weight", sparsity_weight)
        object.__setattr__(self, "_chunk", chunk)
        object.__setattr__(self, "_slot", slot)
        object.__setattr__(self, "_stream", stream)
        object.__setattr__(self, "learning_rate", learning_rate)
        object.__setattr__(self, "step", step)

    def _resolve(self) -> None:
        chunk = self.__dict__.pop("_chunk", None)
        if chunk is None:
            return
        f = chunk.host_record(self.__dict__.pop("_slot"), self.__dict__.pop("_stream"))
        w = self.__dict__.pop("_weight", 0.0)
        sparsity = float(f[7]) if w else 0.0  # wsae_stats.reserved carries mean|hidden| on the ReLU path
        self.__dict__.update(loss=float(f[0]), reconstruction_loss=float(f[0]) - w * sparsity, sparsity_loss=sparsity,
                             l0=float(f[1]),
                             dead_feature_ratio=float(f[4]), grad_norm=float(f[2]), clip_coef=float(f[3]))

    def __getattr__(self, name):  # only reached for attributes not yet in __dict__
        if name in _PendingMetrics._LAZY or name in ("grad_norm", "clip_coef"):
            self._resolve()
            return self.__dict__[name]
        raise AttributeError(name)


class _RecordChunk:
    """``size`` zeroed step records on the device plus their lazily fetched host copy."""

    def __init__(self, size: int, device):
        self.dev = torch.zeros(size, N.STATS_WORDS, dtype=torch.int32, device=device)
        self.used = 0          # slots handed out
        self._host = None      # float32 view of the first _valid records
        self._valid = 0

    def host_record(self, slot: int, stream) -> Tensor:
        if slot >= self._valid:
            stream.synchronize()          # every record handed out so far is final after this
            n = self.used
            self._host = self.dev[:n].cpu().view(torch.float32)
            self._valid = n
        return self._host[slot]


class _MetricsRing:
    """Hands out one device record slot per step (a fresh zeroed chunk every ``chunk`` steps)."""

    def __init__(self, chunk: int = 4096):
        self.chunk = chunk
        self.cur: Optional[_RecordChunk] = None

    def next(self, device) -> tuple:
        if self.cur is None or self.cur.used == self.chunk:
            self.cur = _RecordChunk(self.chunk, device)
        slot = self.cur.used
        self.cur.used += 1
        return self.cur, slot


class SAETrainer:
    """Trains a sparse autoencoder on activation batches (reference training.py:32-379)."""

    def __init__(self, model, config: TrainingConfig, device="cpu", run_dir: Optional[Path] = None,
                 resample_dead_every: int = 5000, resample_batch_size: int = 8192, resample_dead: bool = False):
        self.model = model.to(device)
        self.config = config
        self.device = device
        self.run_dir = Path(run_dir) if run_dir is not None else Path("outputs")
        self.run_dir.mkdir(parents=True, exist_ok=True)
        self.resample_dead_every = resample_dead_every
        self.resample_batch_size = resample_batch_size
        self.resample_dead = resample_dead

        self.optimizer = FusedAdamW(model, lr=config.learning_rate, weight_decay=config.weight_decay)
        self.scheduler = None

        on_gpu = str(device).startswith("cuda")
        self.use_amp = bool(config.use_amp and on_gpu)  # bf16 MFMA contractions when set, fp32 MFMA otherwise
        exch = getattr(config, "grad_exchange_dtype", "auto")
        if exch == "auto":
            exch = "bf16" if self.use_amp else "fp32"
        self.grad_exchange = exch  # resolved: what sync_gradients puts on the wire under torch.distributed
        self._exchange_dtype = torch.bfloat16 if exch == "bf16" else torch.float32
        self.scaler = torch.amp.GradScaler("cuda", enabled=False)  # bf16 needs no loss scaling

        self.global_step = 0
        self.epoch = 0
        self.metrics_history: list[TrainingMetrics] = []
        self.num_resampled_total = 0
        self.wandb_run = None
        self._resample_dataset = None
        self._records = _MetricsRing()
        self._tracks_dead = hasattr(model, "get_dead_feature_ratio")
        self._clip = float(config.gradient_clip)
        self._sched_fast = all(hasattr(LinearLR, a) for a in ("get_lr", "_update_lr")) and \
            hasattr(SequentialLR, "get_last_lr")  # (the attributes the lean scheduler step touches exist in this torch)

    # -- resampling (reference training.py:89-134; see module docstring) ---------------------------
    def set_resample_dataset(self, dataset) -> None:
        self._resample_dataset = dataset

    def _maybe_resample_dead_features(self) -> int:
        ds = self._resample_dataset
        if ds is None or not hasattr(self.model, "resample_dead_features"):
            return 0
        if self.global_step == 0 or self.global_step % self.resample_dead_every != 0:
            return 0
        dist, nranks = world()
        batch = None
        if dist is None or dist.get_rank() == 0:
            n = len(ds)
            take = torch.randperm(n)[: self.resample_batch_size]
            tensors = getattr(ds, "tensors", None)
            if tensors is not None:  # TensorDataset: one indexed gather instead of per-item __getitem__
                batch = tensors[0][take.to(tensors[0].device)]
            else:
                items = [ds[int(i)] for i in take]
                batch = torch.stack([it[0] if isinstance(it, (tuple, list)) else it for it in items])
            batch = batch.to(self.device).float().contiguous()
        if dist is not None:
            # data parallel: parameters are replicated, so every rank must apply the IDENTICAL rewrite.  Each rank
            # holds a different shard of rows; rank 0 draws the resample batch and everybody receives it (rare:
            # every resample_dead_every steps, <= resample_batch_size x D floats).
            shape = torch.zeros(2, dtype=torch.int64, device=self.device)
            if batch is not None:
                shape[0], shape[1] = batch.shape[0], batch.shape[1]
            dist.broadcast(shape, src=0)
            if batch is None:
                batch = torch.empty(int(shape[0]), int(shape[1]), dtype=torch.float32, device=self.device)
            dist.broadcast(batch, src=0)
        count = self.model.resample_dead_features(batch)
        self.num_resampled_total += count
        if count > 0 and self.wandb_run is not None:
            self.wandb_run.log({"train/features_resampled": count}, step=self.global_step)
        return count

    # -- learning-rate schedule (reference training.py:136-159): torch's own scheduler classes --------
    def setup_scheduler(self, total_steps: int) -> None:
        warm = min(self.config.warmup_steps, total_steps // 10)
        ramp = LinearLR(self.optimizer, start_factor=0.01, end_factor=1.0, total_iters=warm)
        decay = CosineAnnealingLR(self.optimizer, T_max=total_steps - warm, eta_min=0.1 * self.config.learning_rate)
        self.scheduler = SequentialLR(self.optimizer, schedulers=[ramp, decay], milestones=[warm])

    def _scheduler_step(self) -> None:
        """``self.scheduler.step()`` without its per-call bookkeeping (warning filters, context managers, the counter
        wrapper: ~25 us of host time, a quarter of a small-batch step).  The objects are torch's own and stay exactly in
        sync - ``last_epoch``, ``_step_count``, ``_last_lr`` advance as ``step()`` advances them and the new rate comes from
        the active child's own ``get_lr()`` - so ``state_dict()`` / ``get_last_lr()`` / a later plain ``.step()`` see what they
        would have seen (``tests/test_host_logic.py`` holds the two paths bit-equal over every G5 schedule).  Anything that is
        not the ``SequentialLR(LinearLR, CosineAnnealingLR)`` this trainer builds goes through ``step()``."""
        sch = self.scheduler
        kids = getattr(sch, "_schedulers", None)
        if (type(sch) is not SequentialLR or kids is None or len(kids) != 2 or type(kids[0]) is not LinearLR
                or type(kids[1]) is not CosineAnnealingLR or not self._sched_fast):
            sch.step()
            return
        sch.last_epoch += 1
        idx = bisect.bisect_right(sch._milestones, sch.last_epoch)
        kid = kids[idx]
        if idx > 0 and sch._milestones[idx - 1] == sch.last_epoch:
            kid._update_lr(0)  # the hand-over step (once per run): torch's own path
        else:
            kid._step_count += 1
            kid.last_epoch += 1
            kid._get_lr_called_within_step = True
            try:
                lr = kid.get_lr()[0]
            finally:
                kid._get_lr_called_within_step = False
            self.optimizer.param_groups[0]["lr"] = lr
            kid._last_lr = [lr]
        sch._last_lr = kid._last_lr

    # -- the step ------------------------------------------------------------------------------------
    def train_step(self, batch) -> TrainingMetrics:
        """One optimisation step on ``batch`` ([B, D] tensor, or a tuple/list whose first item is one).

        ``batch`` may also be ``(ring_data, row_indices)`` produced by ``ActivationRing.batch``: then
        the kernels gather the rows straight from the on-device ring buffer.
        """
        model = self.model
        if not model.training:
            model.train()
        rows = None
        if isinstance(batch, RingBatch):
            x, rows = batch.data, batch.rows
        else:
            if isinstance(batch, (tuple, list)):
                batch = batch[0]
            x = batch.to(self.device, non_blocking=True)
        require_device_tensor(x, "batch")
        if not hasattr(model, "bind"):
            raise N.WsaeError(f"{type(model).__name__} has no fused MI355X train step in this build")
        eng = model.bind()
        lib, st = eng.lib, eng.stream()
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        x = x.contiguous()
        B = int(rows.numel()) if rows is not None else x.shape[0]
        prec = N.PREC_BF16 if self.use_amp else N.PREC_FP32
        token = model.param_token()
        if token != getattr(self, "_token", None):
            eng.invalidate()
        handle = eng.prepare(prec, B)
        opt = self.optimizer
        opt._ensure_state(eng)
        if getattr(model, "is_relu", False):
            return self._train_step_relu(model, eng, handle, opt, x, rows, B, prec)
        w = eng.work(B)
        pk, xd, rp = eng.pack.data_ptr(), _dtype_code(x), N.ptr(rows)
        chunk, slot = self._records.next(eng.device)
        stats = chunk.dev.data_ptr() + slot * N.STATS_WORDS * 4  # this step's record: written in place, never copied
        step_ptr = model.step_count.data_ptr()
        ddp = world()[1] > 1
        N.check(lib.wsae_ctx_set_fired(handle, opt.fired.data_ptr() if ddp else 0), "wsae_ctx_set_fired")
        # stage + encoder GEMM, then TopK + sparse decode + loss + dpre (one launch where the shape allows)
        N.check(lib.wsae_encode_decode(handle, pk, x.data_ptr(), xd, rp, B, w["vals"].data_ptr(), w["idx"].data_ptr(),
                                       step_ptr, 0, 1, w["dpre"].data_ptr(), model.feature_last_activated.data_ptr(),
                                       stats, st), "wsae_encode_decode")
        if not ddp:
            N.check(lib.wsae_weight_grads(handle, pk, x.data_ptr(), xd, rp, w["vals"].data_ptr(), w["idx"].data_ptr(),
                                          w["dpre"].data_ptr(), B, opt.grads.data_ptr(), st), "wsae_weight_grads")
            grad_scale = 1.0
        else:
            grad_scale = self._ddp_backward(eng, handle, opt, x, xd, rp, w, B, chunk, slot, stats, st)
        eng.generation += 1
        fused_norm = True  # the norm partials come with the gradients: from wsae_weight_grads, or from the wire unpack
        opt.apply_update(precision=prec, max_norm=self._clip, grad_scale=grad_scale,
                         normalize_decoder=True, batch=B, norm_from_wgrad=fused_norm, dead_scan=True,
                         stats_ptr=stats)
        self._token = model.param_token()
        if self.scheduler is not None:
            self._scheduler_step()
        self.global_step += 1
        metrics = _PendingMetrics(chunk, slot, torch.cuda.current_stream(eng.device), opt.param_groups[0]["lr"],
                                  self.global_step)
        if self.resample_dead:
            self._maybe_resample_dead_features()
        return metrics

    def _ddp_backward(self, eng, handle, opt, x, xd, rp, w, B, chunk, slot, stats, st) -> float:
        """Data-parallel backward (the reference is single-process).  Default: the single-process
        backward with its reduction kernel writing the exchange buffer ("wire", include/wsae.h) instead of the gradient pack,
        ONE all-reduce, one pass that turns the summed wire into the gradient pack + norm partials.  With
        ``TrainingConfig.ddp_overlap_halves`` the weight gradients are produced in two halves that land on the wire, and each
        half's all-reduce is started - asynchronously, on the process group's own stream - as soon as its half is there
        (measured: the split costs more kernel time than an all-reduce over xGMI can give back at these sizes):

            decoder contraction + reduction -> wire[0, HD)      | all-reduce A starts
            encoder contraction + reduction + biases + fired -> wire[HD, P+H)   (A runs underneath)   | all-reduce B starts
            wait A, B -> one pass turns the summed wire into the fp32 gradient pack + norm partials -> optimizer

        ONE collective per half; only B (and what is left of A) is exposed.  The two metric scalars (loss, l0: per-rank
        batch means) ride at the end of the wire as exact-summable digits (include/wsae.h) and their mean over the ranks
        replaces the local values in the step record.  No torch compute op: the wire is written
        by the reduction kernel in its dtype (fp32, or bf16 = half the bytes; ``TrainingConfig.grad_exchange_dtype``).
        Returns the factor for the summed gradients (1 / world)."""
        dist, nranks = world()
        lib = eng.lib
        wire_dt = N.DT_BF16 if self._exchange_dtype == torch.bfloat16 else N.DT_F32
        wire = opt.wire(self._exchange_dtype)
        hd = eng.H * eng.D
        pk = eng.pack.data_ptr()
        # metric scalars of this step (the decode launch has written them into the step's record): the reduction kernel
        # encodes them behind the fired indicators on the wire - digits whose sums are exact in bf16 as well - and the unpack
        # pass writes their rank means back into the record: no collective of their own
        digits = nranks <= 16  # (digit sums stay below 256 - exact in bf16 - for up to 16 ranks; beyond: a collective of their own)
        N.check(lib.wsae_ctx_set_wire_metrics(handle, stats if digits else 0), "wsae_ctx_set_wire_metrics")
        ex = WireExchange()
        met = None
        if not digits:
            met = chunk.dev[slot].view(torch.float32)[:2]  # summed in place in the step's record
            ex.start(met)
        args = (handle, pk, x.data_ptr(), xd, rp, w["vals"].data_ptr(), w["idx"].data_ptr(), w["dpre"].data_ptr(), B)
        if bool(getattr(self.config, "ddp_overlap_halves", False)) and lib.wsae_wgrad_parts_supported(handle):
            if getattr(self, "_reserve_set", None) != handle:  # once per ctx
                N.check(lib.wsae_ctx_set_comm_reserve(handle, int(getattr(self.config, "ddp_comm_reserve_cus", 0))),
                        "wsae_ctx_set_comm_reserve")
                self._reserve_set = handle
            N.check(lib.wsae_weight_grads_wire(*args, N.PART_DECODER, wire.data_ptr(), wire_dt, st), "wsae_weight_grads_wire")
            ex.start(wire[:hd])
            N.check(lib.wsae_weight_grads_wire(*args, N.PART_ENCODER, wire.data_ptr(), wire_dt, st), "wsae_weight_grads_wire")
            ex.start(wire[hd:])
        else:  # the default (and narrow inputs, where one launch holds both contractions): one collective over the whole wire
            N.check(lib.wsae_weight_grads_wire(*args, N.PART_ALL, wire.data_ptr(), wire_dt, st), "wsae_weight_grads_wire")
            ex.run(wire)  # (in stream order: nothing of this step could run beside it)
        scale = ex.finish()  # (RCCL: the compute stream waits for the asynchronous collectives; the host does not)
        N.check(lib.wsae_grads_unpack_wire(handle, wire.data_ptr(), wire_dt, opt.grads_ext.data_ptr(), N.ptr(met), nranks, stats, st),
                "wsae_grads_unpack_wire")
        return scale

    def _train_step_relu(self, model, eng, handle, opt, x, rows, B, prec) -> TrainingMetrics:
        """ReLU + L1 step (reference ReLUSAE under training.py:161-217; no dead-feature bookkeeping)."""
        lib, st = eng.lib, eng.stream()
        eng.reserve_relu(handle)
        fp8 = relu_fp8_flag(model)
        if fp8 and prec != N.PREC_BF16:
            raise N.WsaeError("ReLUSAE(precision='fp8') trains with TrainingConfig.use_amp = True (the fp8 forward belongs to the bf16 mode)")
        N.check(lib.wsae_ctx_set_relu_fp8(handle, fp8), "wsae_ctx_set_relu_fp8")
        w = eng.relu_work(B, handle)
        pk, xd, rp = eng.pack.data_ptr(), _dtype_code(x), N.ptr(rows)
        chunk, slot = self._records.next(eng.device)
        stats = chunk.dev.data_ptr() + slot * N.STATS_WORDS * 4
        weight = float(model.sparsity_weight)
        # NULL where the kernels keep the hidden code as bf16 in their own workspace and pass the residual gradient on themselves
        hid, rec = N.ptr(w["hidden"]), N.ptr(w["recon"])
        N.check(lib.wsae_relu_forward(handle, pk, x.data_ptr(), xd, rp, B, weight, hid, rec, stats, 0, st), "wsae_relu_forward")
        N.check(lib.wsae_relu_backward(handle, pk, x.data_ptr(), xd, rp, B, weight, hid, rec, opt.grads.data_ptr(), st),
                "wsae_relu_backward")
        eng.generation += 1
        ddp = world()[1] > 1
        grad_scale = sync_gradients(opt.grads, self._exchange_dtype) if ddp else 1.0
        # norm_from_wgrad = 2: take the global-norm partials the backward left when it did (single process only: an exchange
        # rewrites the gradients)
        opt.apply_update(precision=prec, max_norm=self._clip, grad_scale=grad_scale,
                         normalize_decoder=bool(model.normalize_decoder), batch=B, norm_from_wgrad=0 if ddp else 2,
                         dead_scan=False, stats_ptr=stats)
        self._token = model.param_token()
        if self.scheduler is not None:
            self._scheduler_step()
        self.global_step += 1
        return _PendingMetrics(chunk, slot, torch.cuda.current_stream(eng.device), opt.param_groups[0]["lr"],
                               self.global_step, sparsity_weight=weight)

    def train_epoch(self, dataloader, progress=None, task_id=None) -> list:
        """One pass over ``dataloader`` (reference training.py:219-259)."""
        seen = []
        for batch in dataloader:
            m = self.train_step(batch)
            seen.append(m)
            self.metrics_history.append(m)
            if progress is not None and task_id is not None:
                progress.update(task_id, advance=1)
            if self.wandb_run is not None and self.global_step % 100 == 0:
                self.wandb_run.log({"train/loss": m.loss, "train/reconstruction_loss": m.reconstruction_loss,
                                    "train/l0": m.l0, "train/dead_ratio": m.dead_feature_ratio,
                                    "train/lr": m.learning_rate}, step=self.global_step)
        self.epoch += 1
        return seen

    def train(self, dataloader, epochs: Optional[int] = None, checkpoint_every: Optional[int] = None) -> None:
        """Full loop with LR schedule, progress display and checkpoints (reference training.py:261-316)."""
        from rich.progress import BarColumn, Progress, SpinnerColumn, TaskProgressColumn, TextColumn

        epochs = epochs or self.config.epochs
        checkpoint_every = checkpoint_every or self.config.checkpoint_every
        steps_per_epoch = len(dataloader)
        self.setup_scheduler(steps_per_epoch * epochs)
        columns = (SpinnerColumn(), TextColumn("[progress.description]{task.description}"), BarColumn(),
                   TaskProgressColumn())
        # data parallel: every rank runs the loop, rank 0 alone draws the bars and (save_checkpoint) writes files
        with Progress(*columns, disable=rank_and_world()[0] != 0) as progress:
            outer = progress.add_task(f"[cyan]Training {epochs} epochs", total=epochs)
            for e in range(1, epochs + 1):
                inner = progress.add_task(f"[green]Epoch {e}/{epochs}", total=steps_per_epoch)
                ms = self.train_epoch(dataloader, progress, inner)
                mean_loss = sum(m.loss for m in ms) / len(ms)
                mean_l0 = sum(m.l0 for m in ms) / len(ms)
                progress.remove_task(inner)
                progress.update(outer, advance=1)
                progress.console.print(f"Epoch {e}: loss={mean_loss:.4f}, L0={mean_l0:.1f}, "
                                       f"dead={ms[-1].dead_feature_ratio:.1%}")
                if e % checkpoint_every == 0:
                    self.save_checkpoint(f"checkpoint_epoch{e}.pt")
        self.save_checkpoint("final.pt")

    # -- persistence (reference training.py:318-379: same keys, same file formats) --------------------
    def save_checkpoint(self, filename: str) -> Path:
        """Under torch.distributed only rank 0 writes (parameters and optimizer state are replicated, and every
        rank is handed the same ``run_dir``); the others wait at a barrier so that nobody reads a torn file."""
        target = self.run_dir / filename
        if rank_and_world()[0] != 0:
            barrier()
            return target
        payload = {
            "model_state_dict": {k: v.detach().clone() for k, v in self.model.state_dict().items()},
            "optimizer_state_dict": self.optimizer.state_dict(),
            "scheduler_state_dict": self.scheduler.state_dict() if self.scheduler else None,
            "global_step": self.global_step,
            "epoch": self.epoch,
            "config": self.config.model_dump(),
        }
        tmp = target.with_name(target.name + ".tmp")
        try:
            torch.save(payload, tmp)
            tmp.replace(target)  # atomic on POSIX: a reader sees the old file or the new one
        finally:
            barrier()  # reached even when the write raises (disk full, bad run_dir): the other ranks must not hang at theirs
        return target

    def load_checkpoint(self, path) -> None:
        ckpt = torch.load(path, map_location=self.device)
        self.model.load_state_dict(ckpt["model_state_dict"])
        self.optimizer.load_state_dict(ckpt["optimizer_state_dict"])
        if ckpt["scheduler_state_dict"] and self.scheduler:
            self.scheduler.load_state_dict(ckpt["scheduler_state_dict"])
        self.global_step = ckpt["global_step"]
        self.epoch = ckpt["epoch"]

    def save_metrics(self, filename: str = "metrics.json") -> Path:
        target = self.run_dir / filename
        if rank_and_world()[0] != 0:  # rank 0's history is the run's history (loss / l0 are already means over the ranks)
            return target
        keys = ("step", "loss", "reconstruction_loss", "sparsity_loss", "l0", "dead_feature_ratio", "learning_rate")
        rows = [{k: getattr(m, k) for k in keys} for m in self.metrics_history]
        target.write_text(json.dumps(rows, indent=2))
        return target


class RingBatch:
    """A batch that lives in the on-device ring: (ring storage, int32 row indices on the same device)."""

    __slots__ = ("data", "rows")

    def __init__(self, data: Tensor, rows: Tensor):
        self.data, self.rows = data, rows

    def __len__(self) -> int:
        return int(self.rows.numel())


assert {f.name for f in fields(TrainingMetrics)} == {"loss", "reconstruction_loss", "sparsity_loss", "l0",
                                                      "dead_feature_ratio", "learning_rate", "step"}
