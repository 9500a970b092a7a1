"""Data-parallel exchange for the SAE train step (absent from the reference, which is single-process).

Rows of the activation matrix are independent (the reference flattens ``[batch, seq, D]`` to
``[tokens, D]``, sae/hooks.py:213-230) and the loss is a mean over rows, so N ranks that each
average over their own B rows and then average the gradients reproduce exactly the gradient of the
concatenated N*B batch.  One process per GPU; ``torch.distributed`` backend ``"nccl"`` is RCCL on
ROCm (xGMI inside a node); ``"gloo"`` runs the same code on CPU tensors for tests.

The exchange buffer (the "wire", ``include/wsae.h``) is ``[dW_dT | dW_e | db_e | db_d | db_pre | fired]``, fp32 (the exact
data-parallel gradient; the default) or bf16 (``TrainingConfig.grad_exchange_dtype = "bf16"``: half the bytes over xGMI);
9.45 MB + 12 KB as fp32 at 384->3072.  The reduction kernel of the backward writes it directly - the decoder matrix first -
and the trainer all-reduces it in ONE collective issued in stream order (``SAETrainer._ddp_backward``; with
``TrainingConfig.ddp_overlap_halves`` in two: ``wire[:H D]`` as soon as the decoder half of the backward is done, under the
encoder half's contraction, and the rest after it); the 1/world factor is folded into the fused optimizer kernel as
``grad_scale``.  ``fired[f]`` is 1.0 on the ranks where
feature f fired in this step; its sum tells every rank which ``feature_last_activated`` entries to stamp with the current
step, which equals an ``all_reduce(MAX)`` of the clocks (clocks that agreed before the step either all advance to the step
or all stay) without another latency-bound collective.  The two metric scalars (loss, l0) ride behind the indicators as
base-16 digits of their fixed-point values (``encode_wire_metrics``): digit sums of up to 16 ranks stay below 256 and are
exact in bf16 as well, so they need no collective of their own.

``WireExchange`` / ``pack_to_wire`` / ``wire_to_pack`` restate the host side of that protocol on plain tensors (CPU tests
run it over gloo with oracle gradients; the trainer runs the same calls between its kernel launches).
"""

from __future__ import annotations

import torch


import torch.distributed as _dist


def world():
    """``(dist_module, world_size)``; ``(None, 1)`` when not running under torch.distributed."""
    if _dist.is_available() and _dist.is_initialized():
        n = _dist.get_world_size()
        if n > 1:
            return _dist, n
    return None, 1


def sync_gradients(flat: torch.Tensor, exchange_dtype: torch.dtype = torch.float32, unpack=None) -> float:
    """Sum ``flat`` (gradient pack, optionally followed by the fired indicators) over all ranks, in place.

    ``exchange_dtype=torch.bfloat16`` (``TrainingConfig.grad_exchange_dtype``: the default ``"auto"`` picks it in the
    bf16 mode, where the weight-gradient GEMMs already take bf16-rounded ``dpre`` / ``g`` / ``hidden``) sends the
    buffer as bf16 - half the bytes over xGMI, what PyTorch DDP's ``bf16_compress_hook`` does: every rank rounds its
    gradients to bf16, RCCL sums in bf16, the sum is widened back.  The fired indicators (sums of at most
    ``world_size`` ones) survive exactly; the gradients carry a relative error of about 2^-8 per addend.

    ``unpack`` (optional, bf16 wire only): called with the summed wire tensor instead of ``flat.copy_(wire)`` - the
    trainer passes ``wsae_grads_unpack_wire``, which widens into ``flat`` and leaves the norm partials in one pass.

    Returns the factor the caller must scale the summed gradients by (``1 / world_size``).
    """
    dist, n = world()
    if dist is None:
        return 1.0
    if exchange_dtype == torch.float32:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    else:
        wire = flat.to(exchange_dtype)
        dist.all_reduce(wire, op=dist.ReduceOp.SUM)
        if unpack is not None:
            unpack(wire)
        else:
            flat.copy_(wire)
    return 1.0 / n


WIRE_METRIC_SLOTS = 24  # include/wsae.h


def wire_offsets(input_dim: int, hidden_dim: int) -> dict:
    """Element offsets of the wire layout ``[dW_dT | dW_e | db_e | db_d | db_pre | fired | metric digits]``
    (``include/wsae.h``)."""
    hd = input_dim * hidden_dim
    fired = 2 * hd + hidden_dim + 2 * input_dim
    return {"W_dT": 0, "W_e": hd, "b_e": 2 * hd, "b_d": 2 * hd + hidden_dim, "b_pre": 2 * hd + hidden_dim + input_dim,
            "fired": fired, "metrics": fired + hidden_dim, "total": fired + hidden_dim + WIRE_METRIC_SLOTS, "split": hd}


def encode_wire_metrics(loss: float, l0: float) -> list:
    """Host restatement of the digits the reduction kernel writes behind the fired indicators: the loss as 40-bit fixed
    point (2^-24) in ten base-16 digits, l0 as 32-bit fixed point (2^-16) in eight, a non-finite flag, zeros.  Sums of a
    digit over up to 16 ranks stay below 256: exact in a bf16 all-reduce too."""
    import math
    d = [0.0] * WIRE_METRIC_SLOTS
    if not (math.isfinite(loss) and math.isfinite(l0)):
        d[18] = 1.0
        return d
    ql = min(int(max(loss, 0.0) * 16777216.0 + 0.5), (1 << 40) - 1)
    q0 = min(int(max(l0, 0.0) * 65536.0 + 0.5), (1 << 32) - 1)
    for i in range(10):
        d[i] = float((ql >> (4 * i)) & 15)
    for i in range(8):
        d[10 + i] = float((q0 >> (4 * i)) & 15)
    return d


def decode_wire_metrics(digit_sums, world: int) -> tuple:
    """``(mean loss, mean l0)`` from the summed digits (what ``wsae_grads_unpack_wire`` writes into the step record)."""
    if float(digit_sums[18]) > 0:
        return float("nan"), float("nan")
    ql = sum(float(digit_sums[i]) * 16.0 ** i for i in range(10))
    q0 = sum(float(digit_sums[10 + i]) * 16.0 ** i for i in range(8))
    return ql / 16777216.0 / world, q0 / 65536.0 / world


def pack_to_wire(grads_ext: torch.Tensor, input_dim: int, hidden_dim: int, dtype: torch.dtype = torch.float32,
                 metrics: tuple = None) -> torch.Tensor:
    """``[pack order: dW_e | dW_dT | biases | fired]`` -> wire order + the metric digits of ``metrics = (loss, l0)`` (zeros
    when None), in ``dtype`` (what the reduction kernel writes)."""
    hd = input_dim * hidden_dim
    tail = torch.tensor(encode_wire_metrics(*metrics) if metrics is not None else [0.0] * WIRE_METRIC_SLOTS,
                        dtype=grads_ext.dtype, device=grads_ext.device)
    return torch.cat([grads_ext[hd:2 * hd], grads_ext[:hd], grads_ext[2 * hd:], tail]).to(dtype)


def wire_to_pack(wire: torch.Tensor, input_dim: int, hidden_dim: int) -> torch.Tensor:
    """The gradient / fired part of the summed wire back in pack order, fp32 (what ``wsae_grads_unpack_wire`` produces; the
    metric digits are ``wire[wire_offsets(...)["metrics"]:]``, see ``decode_wire_metrics``)."""
    hd = input_dim * hidden_dim
    end = wire.numel() - WIRE_METRIC_SLOTS
    return torch.cat([wire[hd:2 * hd], wire[:hd], wire[2 * hd:end]]).float()


class WireExchange:
    """The collectives of one data-parallel step: ``start(view)`` issues an asynchronous ``all_reduce(SUM)`` of a slice
    of the wire (the tensor must stay alive until ``finish``), ``finish()`` makes the current stream (RCCL) or the host
    (gloo) wait for all of them.  Outside torch.distributed both are no-ops."""

    def __init__(self):
        self._works = []

    def start(self, view: torch.Tensor) -> None:
        dist, _ = world()
        if dist is not None:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))

    def run(self, view: torch.Tensor) -> None:
        """A collective with nothing to overlap: ``all_reduce(SUM)`` issued synchronously - the RCCL backend of this torch
        then runs it on the CURRENT stream, in order with the kernels around it, instead of handing the data to its own
        stream and back (two cross-queue waits: 20 us per step on MI355X, ``profiles/r03_ddp_structure.txt``); gloo blocks
        the host until it is done."""
        dist, _ = world()
        if dist is not None:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=False)

    def finish(self) -> float:
        for wk in self._works:
            wk.wait()
        self._works.clear()
        return 1.0 / world()[1]


def merge_clock(last_activated: torch.Tensor, fired_sum: torch.Tensor, step: int) -> torch.Tensor:
    """What ``wsae_adamw_step`` does with the summed indicators (host restatement for tests and CPU tools):
    ``last_activated[f] = step`` wherever any rank fired f in this step."""
    return torch.where(fired_sum > 0, torch.full_like(last_activated, step), last_activated)


def barrier() -> None:
    """All ranks meet (no-op outside torch.distributed)."""
    dist, _ = world()
    if dist is not None:
        dist.barrier()


def rank_and_world() -> tuple:
    dist, n = world()
    return (dist.get_rank(), n) if dist is not None else (0, 1)
