"""``FusedAdamW``: a ``torch.optim.Optimizer`` whose ``step`` is one pass of HIP kernels over the flat
parameter pack: global-L2 clip -> AdamW -> decoder column renorm -> bf16 shadow refresh
(reference: ``clip_grad_norm_`` + ``AdamW.step`` + ``normalize_decoder_weights``,
training.py:186-198; AdamW semantics of torch).

It keeps torch's optimizer surface so that the reference's LR schedulers
(``LinearLR``/``CosineAnnealingLR``/``SequentialLR``) drive ``param_groups[0]["lr"]`` unchanged and
``state_dict()`` has the layout of ``torch.optim.AdamW`` (per-parameter ``step`` / ``exp_avg`` /
``exp_avg_sq`` in ``module.parameters()`` order), i.e. checkpoints are interchangeable.
"""

from __future__ import annotations

import torch
from torch.optim import Optimizer

from .. import _native as N


class FusedAdamW(Optimizer):
    def __init__(self, module, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid AdamW hyper-parameter")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None,
                        decoupled_weight_decay=True)
        super().__init__(list(module.parameters()), defaults)
        self.module = module
        self._t = 0  # number of updates applied
        self._m = None  # flat exp_avg / exp_avg_sq in pack layout
        self._v = None
        self.grads = None  # flat gradient pack the backward kernels fill

    # -- flat state ------------------------------------------------------------------------------
    def _ensure_state(self, eng) -> None:
        if self._m is not None and self._m.device == eng.device and self._m.numel() == eng.P:
            return
        self._m = torch.zeros(eng.P, dtype=torch.float32, device=eng.device)
        self._v = torch.zeros(eng.P, dtype=torch.float32, device=eng.device)
        # [gradient pack (P) | fired indicators (H)]: under DDP the whole buffer is ONE all-reduce, the tail
        # carrying the dead-feature clock merge (include/wsae.h, wsae_ctx_set_fired)
        self.grads_ext = torch.zeros(eng.P + eng.H, dtype=torch.float32, device=eng.device)
        self.grads = self.grads_ext[:eng.P]
        self.fired = self.grads_ext[eng.P:]
        self._wire = None
        self._alias_state(eng)

    def wire(self, dtype: torch.dtype) -> torch.Tensor:
        """The data-parallel exchange buffer ``[dW_dT | dW_e | db_e | db_d | db_pre | fired | metric digits]``
        (include/wsae.h), P + H + WSAE_WIRE_METRIC_SLOTS elements of ``dtype``; allocated on first use, persistent so that
        asynchronous collectives may hold views of it."""
        n = self.grads_ext.numel() + N.WIRE_METRIC_SLOTS
        if self._wire is None or self._wire.dtype != dtype or self._wire.numel() != n:
            self._wire = torch.zeros(n, dtype=dtype, device=self.grads_ext.device)
        return self._wire

    def _alias_state(self, eng) -> None:
        """Point every parameter's exp_avg / exp_avg_sq at its slice of the flat buffers, carrying
        over values that were loaded (``load_state_dict``) or lived on another device."""
        named = self.module._named_core_params()
        by_id = {id(p): name for name, p in named.items()}
        for p in self.param_groups[0]["params"]:
            name = by_id.get(id(p))
            if name is None:
                continue
            st = self.state[p]
            for key, flat in (("exp_avg", self._m), ("exp_avg_sq", self._v)):
                view = eng.view(name, flat)
                old = st.get(key)
                if old is not None and old.data_ptr() != view.data_ptr():
                    view.copy_(old.to(device=eng.device, dtype=torch.float32))
                st[key] = view
            st["step"] = torch.tensor(float(self._t))

    def grad_view(self, name: str) -> torch.Tensor:
        return self.module._engine.view(name, self.grads)

    # -- Optimizer API ---------------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = True) -> None:  # gradients live in self.grads, overwritten each step
        super().zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None, **kw):
        """Apply one update from ``self.grads`` (filled by ``wsae_weight_grads``); keyword arguments as ``apply_update``.
        (This is torch's hooked / profiled ``Optimizer.step`` entry; the trainer calls ``apply_update`` directly - the
        wrapper torch puts around ``step`` costs ~10 us of host time per call, which is a third of a small-batch step.)"""
        if closure is not None:
            raise NotImplementedError("FusedAdamW does not take a closure")
        return self.apply_update(**kw)

    def apply_update(self, *, precision: int = N.PREC_FP32, max_norm: float = 0.0, grad_scale: float = 1.0,
                     normalize_decoder: bool = False, batch: int = 64, norm_from_wgrad: bool = False, dead_scan: bool = False,
                     stats_ptr: int = 0):
        """One fused update (clip -> AdamW -> renorm -> shadows -> dead scan) enqueued on the current stream.

        ``stats_ptr``: device address of the step record to fill (default: the engine's own record).
        """
        eng = self.module.bind()
        self._ensure_state(eng)
        g = self.param_groups[0]
        mod = self.module
        self._t += 1
        handle = eng.ctx(precision, batch)
        N.check(eng.lib.wsae_adamw_step(handle, eng.pack.data_ptr(), self.grads.data_ptr(), self._m.data_ptr(),
                                        self._v.data_ptr(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                        float(g["eps"]), float(g["weight_decay"]), self._t, float(max_norm),
                                        float(grad_scale), 1 if normalize_decoder else 0, int(norm_from_wgrad),
                                        mod.feature_last_activated.data_ptr() if dead_scan else 0,
                                        mod.step_count.data_ptr() if dead_scan else 0,
                                        int(mod.dead_feature_threshold) if dead_scan else 0,
                                        stats_ptr or eng.stats.data_ptr(), eng.stream()), "wsae_adamw_step")
        eng.mark_fresh(precision)
        self._opt_called = True  # (what torch's wrapper around step() records for the scheduler's step-order check)
        self._steps_dirty = True  # the per-parameter "step" tensors of the torch state are refreshed when somebody looks
        return None

    def _sync_steps(self) -> None:
        if getattr(self, "_steps_dirty", False):
            for st in self.state.values():
                if "step" in st:
                    st["step"].fill_(float(self._t))
            self._steps_dirty = False

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        steps = [float(st["step"]) for st in self.state.values() if "step" in st]
        self._t = int(max(steps)) if steps else 0
        eng = getattr(self.module, "_engine", None)
        if eng is not None:  # otherwise the first step() binds and carries the loaded moments over
            if self._m is None or self._m.device != eng.device:
                self._ensure_state(eng)
            else:
                self._alias_state(eng)
