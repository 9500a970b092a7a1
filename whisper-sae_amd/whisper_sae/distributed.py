"""Data-parallel exchange for the SAE train step (absent from the reference, which is single-process).

Rows of the activation matrix are independent (the reference flattens ``[batch, seq, D]`` to
``[tokens, D]``, sae/hooks.py:213-230) and the loss is a mean over rows, so N ranks that each
average over their own B rows and then average the gradients reproduce exactly the gradient of the
concatenated N*B batch.  One process per GPU; ``torch.distributed`` backend ``"nccl"`` is RCCL on
ROCm (xGMI inside a node); ``"gloo"`` runs the same code on CPU tensors for tests.

Per step there is ONE collective: ``all_reduce(SUM)`` of the flat buffer ``[gradient pack | fired]``
(9.45 MB + 12 KB at 384->3072 as fp32 - the fp32 mode - or half of that as bf16 - the bf16 mode's default; the
1/world factor is folded into the fused optimizer kernel as ``grad_scale``).  ``fired[f]`` is 1.0 on the ranks where feature f fired in this step; its sum tells
every rank which ``feature_last_activated`` entries to stamp with the current step, which equals an
``all_reduce(MAX)`` of the clocks (clocks that agreed before the step either all advance to the step or
all stay) without a second, latency-bound collective.
"""

from __future__ import annotations

import torch


def world():
    """``(dist_module, world_size)``; ``(None, 1)`` when not running under torch.distributed."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist, dist.get_world_size()
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
