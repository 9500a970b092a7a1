"""Device-side state of one SAE: the flat parameter pack, native contexts and workspaces.

The five parameter tensors of a ``TopKSAE`` are views into one float32 buffer laid out as
``include/wsae.h`` describes (``W_e | W_dT | b_e | b_d | b_pre``); ``decoder.weight`` is exposed
as the transposed view of ``W_dT`` so the state-dict keeps the reference's ``[D, H]`` shape
(model.py:64) while the kernels read decoder columns as contiguous rows.
"""

from __future__ import annotations

import torch

from .. import _native as N

PARAM_ORDER = ("encoder.weight", "decoder.weight", "encoder.bias", "decoder.bias", "b_pre")


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return N.DT_F32
    if t.dtype == torch.bfloat16:
        return N.DT_BF16
    raise TypeError(f"activations must be float32 or bfloat16, got {t.dtype}")


def require_device_tensor(t: torch.Tensor, what: str) -> None:
    if t.device.type != "cuda":
        raise N.WsaeError(
            f"{what} is on '{t.device}': whisper_sae (MI355X build) runs its SAE math only as HIP kernels on a "
            f"ROCm device and has no CPU path.  Move the module and the data with .to('cuda').")


class SAEEngine:
    """Owns the pack, the per-precision ``wsae_ctx`` handles and scratch tensors on one device."""

    def __init__(self, device: torch.device, input_dim: int, hidden_dim: int, k: int):
        self.device = torch.device(device)
        self.D, self.H, self.k = int(input_dim), int(hidden_dim), int(k)
        self.P, self.off = N.pack_layout(self.D, self.H)
        self.lib = N.lib()  # raises WsaeError when the extension is not built
        self.pack = torch.zeros(self.P, dtype=torch.float32, device=self.device)
        self.generation = 0  # bumped by every call that leaves a batch's staged operands (xT / g / gT) in a ctx
        self._ctx: dict[int, tuple[int, int]] = {}  # precision -> (handle, max_batch)
        self._fresh: dict[int, bool] = {}  # precision -> derived shadows match the pack
        self.stats = torch.zeros(N.STATS_WORDS, dtype=torch.int32, device=self.device)
        self._work: dict = {}
        self._relu_reserved: set = set()

    # -- pack views ----------------------------------------------------------------------------
    def view(self, name: str, base: torch.Tensor | None = None) -> torch.Tensor:
        """View of parameter ``name`` inside ``base`` (default: the pack), reference shape."""
        buf = self.pack if base is None else base
        D, H, o = self.D, self.H, self.off
        if name == "encoder.weight":
            return buf[o[0]:o[1]].view(H, D)
        if name == "decoder.weight":
            return buf[o[1]:o[2]].view(H, D).t()
        if name == "encoder.bias":
            return buf[o[2]:o[3]]
        if name == "decoder.bias":
            return buf[o[3]:o[4]]
        if name == "b_pre":
            return buf[o[4]:o[4] + D]
        raise KeyError(name)

    # -- native contexts -----------------------------------------------------------------------
    def stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def ctx(self, precision: int, batch: int) -> int:
        have = self._ctx.get(precision)
        if have is not None and have[1] >= batch:
            return have[0]
        if have is not None:
            torch.cuda.synchronize(self.device)
            self.lib.wsae_ctx_destroy(have[0])
            self._relu_reserved.discard(have[0])
        cap = max(int(batch), 64)
        cfg = N.Config(self.D, self.H, self.k, cap, precision, self.device.index or 0)
        handle = N._p()
        with torch.cuda.device(self.device):
            N.check(self.lib.wsae_ctx_create(cfg, handle), "wsae_ctx_create")
        self._ctx[precision] = (handle.value, cap)
        self._fresh[precision] = False
        self.generation += 1  # a new ctx holds nobody's staged batch
        return handle.value

    def reserve_relu(self, handle: int) -> None:
        """Dense workspace of the ReLU path (allocated once per ctx, outside the launch functions)."""
        if handle not in self._relu_reserved:
            with torch.cuda.device(self.device):
                N.check(self.lib.wsae_ctx_reserve_relu(handle), "wsae_ctx_reserve_relu")
            self._relu_reserved.add(handle)

    def invalidate(self) -> None:
        """The pack changed outside ``wsae_adamw_step``: derived shadows must be rebuilt."""
        for p in self._fresh:
            self._fresh[p] = False

    def prepare(self, precision: int, batch: int, force: bool = False) -> int:
        handle = self.ctx(precision, batch)
        if force or not self._fresh.get(precision, False):
            N.check(self.lib.wsae_prepare(handle, self.pack.data_ptr(), self.stream()), "wsae_prepare")
            self._fresh[precision] = True
        return handle

    def mark_fresh(self, precision: int) -> None:
        """``wsae_adamw_step`` refreshed the shadows of this precision's ctx (only)."""
        self.invalidate()
        self._fresh[precision] = True

    def work(self, batch: int) -> dict:
        w = self._work.get(batch)
        if w is None:
            if len(self._work) > 4:
                self._work.clear()
            w = {
                "vals": torch.empty(batch, self.k, dtype=torch.float32, device=self.device),
                "idx": torch.empty(batch, self.k, dtype=torch.int32, device=self.device),
                "dpre": torch.empty(batch, self.k, dtype=torch.float32, device=self.device),
            }
            self._work[batch] = w
        return w

    def relu_work(self, batch: int, handle: int = 0) -> dict:
        """Scratch of one ReLU step.  ``hidden`` (dense fp32 [B, H]) and ``recon`` (fp32 [B, D]) are only allocated when the
        kernels need them (``wsae_relu_needs_hidden``: the bf16 row-major-GEMM flow keeps the code as bf16 in the ctx workspace
        and hands the residual gradient from the forward to the backward itself)."""
        w = self._work.get(("relu", batch))
        if w is None:
            if len(self._work) > 4:
                self._work.clear()
            need = True if not handle else bool(self.lib.wsae_relu_needs_hidden(handle, batch))
            w = {"hidden": torch.empty(batch, self.H, dtype=torch.float32, device=self.device) if need else None,
                 "recon": torch.empty(batch, self.D, dtype=torch.float32, device=self.device) if need else None}
            self._work[("relu", batch)] = w
        return w

    def stats_f32(self) -> torch.Tensor:
        return self.stats.view(torch.float32)

    def close(self) -> None:
        for handle, _ in self._ctx.values():
            try:
                self.lib.wsae_ctx_destroy(handle)
            except Exception:  # interpreter shutdown
                pass
        self._ctx.clear()

    def __del__(self):
        self.close()
