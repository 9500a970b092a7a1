"""SAE modules for MI355X: same classes, constructor arguments, attributes and state-dict keys as
the reference's ``src/whisper_sae/sae/model.py``, with the arithmetic in ``libwsae_hip.so``.

* ``TopKSAE``  (reference model.py:26-257)  encode -> TopK -> sparse decode -> MSE, dead-feature
  tracking and resampling, all as HIP kernels working on a compact ``(values, indices)[B, k]`` code;
  the dense ``hidden [B, H]`` tensor of ``SAEOutput`` is materialised only for API callers.
* ``ReLUSAE``  (reference model.py:260-322).
* ``create_sae`` (reference model.py:325-354), additionally forwarding ``config.sparsity_weight``.

Host-side torch is used for tensor storage, initial random initialisation (same RNG draw order as
the reference, so ``torch.manual_seed(s)`` gives the same initial weights) and autograd plumbing.
There is no CPU compute path: tensors must live on a ROCm device or ``WsaeError`` is raised.
"""

from __future__ import annotations

from typing import NamedTuple, Optional

import torch
from torch import Tensor, nn

from .. import _native as N
from ..config import SAEConfig
from .engine import PARAM_ORDER, SAEEngine, _dtype_code, require_device_tensor


class SAEOutput(NamedTuple):
    """What ``forward`` returns (field order of the reference, model.py:15-23)."""

    reconstructed: Tensor
    hidden: Tensor
    loss: Tensor
    reconstruction_loss: Tensor
    sparsity_loss: Tensor
    l0: Tensor


def _precision_code(precision: Optional[str]) -> int:
    if precision is None:
        precision = "bf16" if torch.is_autocast_enabled() else "fp32"
    if precision in ("bf16", "amp", "fp8"):  # "fp8" (ReLUSAE only): BF16 mode with e4m3 operands in the two forward GEMMs
        return N.PREC_BF16
    if precision == "fp32":
        return N.PREC_FP32
    raise ValueError(f"precision must be 'bf16', 'fp32', 'fp8' (ReLUSAE) or None, got {precision!r}")


def relu_fp8_flag(module) -> int:
    """1 when the module asks for the fp8 forward (``ReLUSAE(precision="fp8")``, BASELINE.json configs[4])."""
    return 1 if getattr(module, "precision", None) == "fp8" else 0


class _TopKForward(torch.autograd.Function):
    """encode_topk -> decode_loss (-> weight_grads in backward) as one autograd node.

    Gradients are defined for ``loss`` (= ``reconstruction_loss``) with respect to the five
    parameters and the input; ``reconstructed`` / ``hidden`` / ``l0`` are returned detached.
    """

    @staticmethod
    def forward(ctx, x, w_e, b_e, w_d, b_d, b_pre, module, prec):
        eng: SAEEngine = module._engine
        lib = eng.lib
        lead = x.shape[:-1]
        x2 = x.reshape(-1, eng.D)
        if x2.dtype not in (torch.float32, torch.bfloat16):
            x2 = x2.float()
        x2 = x2.contiguous()
        B = x2.shape[0]
        handle = eng.prepare(prec, B, force=True)
        st = eng.stream()
        training = module.training
        need_bwd = any(ctx.needs_input_grad[:6])
        vals = torch.empty(B, eng.k, dtype=torch.float32, device=eng.device)
        idx = torch.empty(B, eng.k, dtype=torch.int32, device=eng.device)
        recon = torch.empty(B, eng.D, dtype=torch.float32, device=eng.device)
        dpre = torch.empty(B, eng.k, dtype=torch.float32, device=eng.device) if need_bwd else None
        step_ptr = module.step_count.data_ptr() if training else 0
        last_ptr = module.feature_last_activated.data_ptr() if training else 0
        pk, xd = eng.pack.data_ptr(), _dtype_code(x2)
        N.check(lib.wsae_ctx_set_fired(handle, 0), "wsae_ctx_set_fired")  # the trainer's DDP clock exchange is per step
        # want_bwd: bit 0 = keep g / dpre for the weight gradients, bit 1 = also the fp32 g that dL/dx reads
        want = (1 if need_bwd else 0) | (2 if (need_bwd and ctx.needs_input_grad[0]) else 0)
        N.check(lib.wsae_encode_decode(handle, pk, x2.data_ptr(), xd, 0, B, vals.data_ptr(), idx.data_ptr(), step_ptr,
                                       recon.data_ptr(), want, N.ptr(dpre), last_ptr, eng.stats.data_ptr(), st),
                "wsae_encode_decode")
        hidden = torch.empty(B, eng.H, dtype=torch.float32, device=eng.device)
        N.check(lib.wsae_densify(handle, vals.data_ptr(), idx.data_ptr(), B, hidden.data_ptr(), st), "wsae_densify")
        sf = eng.stats_f32()
        loss = sf[0].clone()
        l0 = sf[1].clone()
        eng.generation += 1
        ctx.module, ctx.prec, ctx.gen, ctx.B = module, prec, eng.generation, B
        ctx.x_shape = x.shape
        ctx.save_for_backward(x2, vals, idx, dpre if dpre is not None else vals)
        ctx.has_dpre = dpre is not None
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(recon, hidden, l0)
        module._last_code = (vals, idx)
        return recon.reshape(*lead, eng.D), hidden.reshape(*lead, eng.H), loss, l0

    @staticmethod
    def backward(ctx, g_recon, g_hidden, g_loss, g_l0):
        if g_loss is None:
            return (None,) * 8
        module, prec, B = ctx.module, ctx.prec, ctx.B
        eng: SAEEngine = module._engine
        lib = eng.lib
        x2, vals, idx, dpre = ctx.saved_tensors
        st = eng.stream()
        handle = eng.prepare(prec, B, force=True)
        pk, xd = eng.pack.data_ptr(), _dtype_code(x2)
        if eng.generation != ctx.gen or not ctx.has_dpre:
            # another forward has reused the ctx workspace since: rebuild xT / g / gT for this batch
            tmp_v = torch.empty_like(vals)
            tmp_i = torch.empty_like(idx)
            dpre = torch.empty_like(vals)
            N.check(lib.wsae_encode_topk(handle, pk, x2.data_ptr(), xd, 0, B, tmp_v.data_ptr(), tmp_i.data_ptr(), 0,
                                         eng.stats.data_ptr(), st), "wsae_encode_topk")
            scratch = torch.zeros(N.STATS_WORDS, dtype=torch.int32, device=eng.device)
            N.check(lib.wsae_decode_loss(handle, pk, x2.data_ptr(), xd, 0, vals.data_ptr(), idx.data_ptr(), B, 0,
                                         3 if ctx.needs_input_grad[0] else 1, dpre.data_ptr(), 0, 0, scratch.data_ptr(), st),
                    "wsae_decode_loss")
            eng.generation += 1
        grads = torch.empty(eng.P, dtype=torch.float32, device=eng.device)
        N.check(lib.wsae_weight_grads(handle, pk, x2.data_ptr(), xd, 0, vals.data_ptr(), idx.data_ptr(),
                                      dpre.data_ptr(), B, grads.data_ptr(), st), "wsae_weight_grads")
        grads.mul_(g_loss)
        need = ctx.needs_input_grad
        dx = None
        if need[0]:
            dx = torch.empty(B, eng.D, dtype=torch.float32, device=eng.device)
            N.check(lib.wsae_input_grad(handle, pk, idx.data_ptr(), dpre.data_ptr(), B, dx.data_ptr(), 1, st),
                    "wsae_input_grad")
            dx = (dx * g_loss).reshape(ctx.x_shape)
        gv = lambda name, on: eng.view(name, grads) if on else None  # noqa: E731
        return (dx, gv("encoder.weight", need[1]), gv("encoder.bias", need[2]), gv("decoder.weight", need[3]),
                gv("decoder.bias", need[4]), gv("b_pre", need[5]), None, None)


class TopKSAE(nn.Module):
    """TopK sparse autoencoder (reference model.py:26-257).

    ``precision``: ``"bf16"`` (bf16 MFMA contractions, fp32 accumulate -- what ``use_amp`` selects in
    the trainer), ``"fp32"`` (fp32 MFMA, the reference's CPU semantics) or ``None`` = follow
    ``torch.autocast`` (bf16 inside an autocast region, fp32 otherwise).
    """

    def __init__(self, input_dim: int, hidden_dim: int, k: int = 32, normalize_decoder: bool = True,
                 dead_feature_threshold: int = 10_000, precision: Optional[str] = None):
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.k = k
        self.normalize_decoder = normalize_decoder
        self.dead_feature_threshold = dead_feature_threshold
        self.precision = precision
        # same construction (and RNG draw) order as the reference: encoder, decoder, xavier(decoder.weight)
        self.encoder = nn.Linear(input_dim, hidden_dim, bias=True)
        self.decoder = nn.Linear(hidden_dim, input_dim, bias=True)
        self.b_pre = nn.Parameter(torch.zeros(input_dim))
        with torch.no_grad():  # reference model.py:79-89: xavier -> unit-norm columns -> x0.1
            nn.init.xavier_uniform_(self.decoder.weight)
            w = self.decoder.weight.data
            self.decoder.weight.data = w / w.norm(dim=0, keepdim=True).clamp_min(1e-12) * 0.1
        self.register_buffer("feature_last_activated", torch.zeros(hidden_dim, dtype=torch.long))
        self.register_buffer("step_count", torch.tensor(0, dtype=torch.long))
        self._engine: Optional[SAEEngine] = None
        self._bound_ptrs = None
        self._last_code = None

    # -- device binding --------------------------------------------------------------------------
    def _named_core_params(self):
        return {"encoder.weight": self.encoder.weight, "decoder.weight": self.decoder.weight,
                "encoder.bias": self.encoder.bias, "decoder.bias": self.decoder.bias, "b_pre": self.b_pre}

    def bind(self) -> SAEEngine:
        """Make the five parameters views of one device pack (idempotent; re-binds after ``.to()``,
        ``param.data = ...`` or anything else that re-pointed a parameter)."""
        dev = self.b_pre.device
        require_device_tensor(self.b_pre, "TopKSAE")
        if self.k > self.hidden_dim:
            raise ValueError(f"k={self.k} exceeds hidden_dim={self.hidden_dim}")
        eng = self._engine
        if eng is not None and eng.device == dev and eng.k == self.k:
            # fast path (every train step comes through here twice): nothing was re-pointed since the last full check
            ptrs = (self.encoder.weight.data_ptr(), self.decoder.weight.data_ptr(), self.encoder.bias.data_ptr(),
                    self.decoder.bias.data_ptr(), self.b_pre.data_ptr(), self.feature_last_activated.data_ptr(),
                    self.step_count.data_ptr())
            if ptrs == self._bound_ptrs:
                return eng
        if eng is None or eng.device != dev or eng.k != self.k:
            if eng is not None:
                eng.close()
            eng = SAEEngine(dev, self.input_dim, self.hidden_dim, self.k)
            self._engine = eng
        with torch.no_grad():
            for name, p in self._named_core_params().items():
                v = eng.view(name)
                if p.data_ptr() != v.data_ptr() or p.shape != v.shape or p.stride() != v.stride():
                    v.copy_(p.detach().to(device=dev, dtype=torch.float32))
                    p.data = v
                    eng.invalidate()
        for buf in (self.feature_last_activated, self.step_count):
            if buf.device != dev:
                raise N.WsaeError("module buffers and parameters are on different devices; use module.to(device)")
        # (the views are slices of the engine's pack: equal pointers = same device, shape and stride as checked above)
        self._bound_ptrs = (self.encoder.weight.data_ptr(), self.decoder.weight.data_ptr(), self.encoder.bias.data_ptr(),
                            self.decoder.bias.data_ptr(), self.b_pre.data_ptr(), self.feature_last_activated.data_ptr(),
                            self.step_count.data_ptr())
        return eng

    def param_token(self) -> tuple:
        """Changes whenever a parameter was re-pointed or modified in place through autograd-visible ops."""
        return tuple((p.data_ptr(), p._version) for p in self._named_core_params().values())

    # -- reference API -----------------------------------------------------------------------------
    def normalize_decoder_weights(self) -> None:
        """Unit-norm decoder columns (reference model.py:91-96; always normalises, like the reference)."""
        eng = self.bind()
        handle = eng.ctx(_precision_code(self.precision), 64)
        N.check(eng.lib.wsae_normalize_decoder(handle, eng.pack.data_ptr(), eng.stream()), "wsae_normalize_decoder")
        eng.invalidate()

    def _code(self, x: Tensor, training: bool):
        eng = self.bind()
        require_device_tensor(x, "input")
        x2 = x.reshape(-1, eng.D)
        if x2.dtype not in (torch.float32, torch.bfloat16):
            x2 = x2.float()
        x2 = x2.contiguous()
        B = x2.shape[0]
        prec = _precision_code(self.precision)
        handle = eng.prepare(prec, B, force=True)
        vals = torch.empty(B, eng.k, dtype=torch.float32, device=eng.device)
        idx = torch.empty(B, eng.k, dtype=torch.int32, device=eng.device)
        N.check(eng.lib.wsae_encode_topk(handle, eng.pack.data_ptr(), x2.data_ptr(), _dtype_code(x2), 0, B,
                                         vals.data_ptr(), idx.data_ptr(), 0, eng.stats.data_ptr(), eng.stream()),
                "wsae_encode_topk")
        eng.generation += 1  # the ctx now holds THIS batch's staged operands: an earlier forward must restage
        return eng, handle, x2, vals, idx

    @torch.no_grad()
    def encode_compact(self, x: Tensor):
        """TopK code without the dense scatter: ``(values [B,k] f32 pre-activations, indices [B,k] i32)``."""
        _, _, _, vals, idx = self._code(x, False)
        return vals, idx

    @torch.no_grad()
    def pre_activation(self, x: Tensor) -> Tensor:
        """Dense ``encoder(x - b_pre)`` [B, H] (reference model.py:108-111)."""
        eng = self.bind()
        require_device_tensor(x, "input")
        x2 = x.reshape(-1, eng.D)
        x2 = (x2 if x2.dtype in (torch.float32, torch.bfloat16) else x2.float()).contiguous()
        B = x2.shape[0]
        handle = eng.prepare(_precision_code(self.precision), B, force=True)
        pre = torch.empty(B, eng.H, dtype=torch.float32, device=eng.device)
        N.check(eng.lib.wsae_encode_dense(handle, eng.pack.data_ptr(), x2.data_ptr(), _dtype_code(x2), 0, B,
                                          pre.data_ptr(), eng.stream()), "wsae_encode_dense")
        eng.generation += 1
        return pre.reshape(*x.shape[:-1], eng.H)

    @torch.no_grad()
    def encode(self, x: Tensor) -> Tensor:
        """Dense sparse code ``[.., H]`` with at most ``k`` non-zeros per row (reference model.py:98-118)."""
        eng, handle, x2, vals, idx = self._code(x, False)
        hidden = torch.empty(x2.shape[0], eng.H, dtype=torch.float32, device=eng.device)
        N.check(eng.lib.wsae_densify(handle, vals.data_ptr(), idx.data_ptr(), x2.shape[0], hidden.data_ptr(),
                                     eng.stream()), "wsae_densify")
        return hidden.reshape(*x.shape[:-1], eng.H)

    @torch.no_grad()
    def decode(self, hidden: Tensor) -> Tensor:
        """``decoder(hidden) + b_pre`` for any dense code (reference model.py:120-129)."""
        eng = self.bind()
        require_device_tensor(hidden, "hidden")
        h2 = hidden.reshape(-1, eng.H).float().contiguous()
        handle = eng.ctx(_precision_code(self.precision), 64)
        recon = torch.empty(h2.shape[0], eng.D, dtype=torch.float32, device=eng.device)
        N.check(eng.lib.wsae_decode_dense(handle, eng.pack.data_ptr(), h2.data_ptr(), h2.shape[0], recon.data_ptr(),
                                          eng.stream()), "wsae_decode_dense")
        return recon.reshape(*hidden.shape[:-1], eng.D)

    def forward(self, x: Tensor) -> SAEOutput:
        """Reference model.py:131-166.  In training mode also advances the dead-feature clock."""
        self.bind()
        require_device_tensor(x, "input")
        prec = _precision_code(self.precision)
        recon, hidden, loss, l0 = _TopKForward.apply(x, self.encoder.weight, self.encoder.bias, self.decoder.weight,
                                                     self.decoder.bias, self.b_pre, self, prec)
        return SAEOutput(reconstructed=recon, hidden=hidden, loss=loss, reconstruction_loss=loss,
                         sparsity_loss=torch.zeros((), device=x.device), l0=l0)

    # -- dead features (reference model.py:183-257) --------------------------------------------------
    def get_dead_features(self) -> Tensor:
        eng = self.bind()
        mask = torch.empty(self.hidden_dim, dtype=torch.uint8, device=eng.device)
        handle = eng.ctx(_precision_code(self.precision), 64)
        N.check(eng.lib.wsae_dead_scan(handle, self.feature_last_activated.data_ptr(), self.step_count.data_ptr(),
                                       int(self.dead_feature_threshold), mask.data_ptr(), eng.stats.data_ptr(),
                                       eng.stream()), "wsae_dead_scan")
        return mask.bool()

    def get_dead_feature_ratio(self) -> float:
        self.get_dead_features()
        return float(self._engine.stats_f32()[4].item())

    @torch.no_grad()
    def resample_dead_features(self, inputs: Tensor, num_resample: Optional[int] = None) -> int:
        """Reference model.py:197-257 including its quirks: the forward on ``inputs`` advances the
        dead-feature clock in train mode, and the returned count is the capped number of dead
        features even when fewer rows than that were available to rewrite them."""
        eng = self.bind()
        require_device_tensor(inputs, "inputs")
        lib, st = eng.lib, eng.stream()
        x2 = inputs.reshape(-1, eng.D)
        x2 = (x2 if x2.dtype in (torch.float32, torch.bfloat16) else x2.float()).contiguous()
        Br = x2.shape[0]
        prec = _precision_code(self.precision)
        handle = eng.prepare(prec, Br, force=True)
        mask = torch.empty(self.hidden_dim, dtype=torch.uint8, device=eng.device)
        N.check(lib.wsae_dead_scan(handle, self.feature_last_activated.data_ptr(), self.step_count.data_ptr(),
                                   int(self.dead_feature_threshold), mask.data_ptr(), eng.stats.data_ptr(), st),
                "wsae_dead_scan")
        if int(eng.stats[5].item()) == 0:  # host decision, as in the reference (model.py:219-220)
            return 0
        training = self.training
        # (data parallel: this forward runs identically on every rank, so its clock stamps need no exchange - keep
        # them out of the indicator buffer that rides on the next gradient all-reduce)
        N.check(lib.wsae_ctx_set_fired(handle, 0), "wsae_ctx_set_fired")
        vals = torch.empty(Br, eng.k, dtype=torch.float32, device=eng.device)
        idx = torch.empty(Br, eng.k, dtype=torch.int32, device=eng.device)
        recon = torch.empty(Br, eng.D, dtype=torch.float32, device=eng.device)
        pk, xd = eng.pack.data_ptr(), _dtype_code(x2)
        N.check(lib.wsae_encode_topk(handle, pk, x2.data_ptr(), xd, 0, Br, vals.data_ptr(), idx.data_ptr(),
                                     self.step_count.data_ptr() if training else 0, eng.stats.data_ptr(), st),
                "wsae_encode_topk")
        N.check(lib.wsae_decode_loss(handle, pk, x2.data_ptr(), xd, 0, vals.data_ptr(), idx.data_ptr(), Br,
                                     recon.data_ptr(), 0, 0,
                                     self.feature_last_activated.data_ptr() if training else 0,
                                     self.step_count.data_ptr() if training else 0, eng.stats.data_ptr(), st),
                "wsae_decode_loss")
        row_err = torch.empty(Br, dtype=torch.float32, device=eng.device)
        eng.generation += 1
        N.check(lib.wsae_row_errors(handle, x2.data_ptr(), xd, 0, recon.data_ptr(), Br, row_err.data_ptr(), 0, st),
                "wsae_row_errors")
        n_out = torch.zeros(1, dtype=torch.int32, device=eng.device)
        cap = -1 if num_resample is None else int(num_resample)
        N.check(lib.wsae_resample_dead(handle, pk, x2.data_ptr(), xd, 0, Br, row_err.data_ptr(), mask.data_ptr(),
                                       self.feature_last_activated.data_ptr(), self.step_count.data_ptr(), cap,
                                       n_out.data_ptr(), 0, st), "wsae_resample_dead")
        eng.invalidate()
        return int(n_out.item())

    def extra_repr(self) -> str:
        return f"input_dim={self.input_dim}, hidden_dim={self.hidden_dim}, k={self.k}"


class _ReLUForward(torch.autograd.Function):
    """wsae_relu_forward (-> wsae_relu_backward) as one autograd node; gradients are defined for ``loss``
    with respect to the four parameters (the reference computes no ``dL/dx`` either)."""

    @staticmethod
    def forward(ctx, x, w_e, b_e, w_d, b_d, module, prec):
        eng: SAEEngine = module._engine
        lead = x.shape[:-1]
        x2 = x.reshape(-1, eng.D)
        if x2.dtype not in (torch.float32, torch.bfloat16):
            x2 = x2.float()
        x2 = x2.contiguous()
        B = x2.shape[0]
        handle = eng.prepare(prec, B, force=True)
        eng.reserve_relu(handle)
        N.check(eng.lib.wsae_ctx_set_relu_fp8(handle, relu_fp8_flag(module)), "wsae_ctx_set_relu_fp8")
        hidden = torch.empty(B, eng.H, dtype=torch.float32, device=eng.device)
        recon = torch.empty(B, eng.D, dtype=torch.float32, device=eng.device)
        sparsity = torch.empty((), dtype=torch.float32, device=eng.device)
        N.check(eng.lib.wsae_relu_forward(handle, eng.pack.data_ptr(), x2.data_ptr(), _dtype_code(x2), 0, B,
                                          float(module.sparsity_weight), hidden.data_ptr(), recon.data_ptr(),
                                          eng.stats.data_ptr(), sparsity.data_ptr(), eng.stream()), "wsae_relu_forward")
        sf = eng.stats_f32()
        loss, l0 = sf[0].clone(), sf[1].clone()
        eng.generation += 1
        ctx.module, ctx.prec, ctx.gen, ctx.B = module, prec, eng.generation, B
        ctx.save_for_backward(x2, hidden, recon)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(recon, hidden, sparsity, l0)
        return recon.reshape(*lead, eng.D), hidden.reshape(*lead, eng.H), loss, sparsity, l0

    @staticmethod
    def backward(ctx, g_recon, g_hidden, g_loss, g_sparsity, g_l0):
        if g_loss is None:
            return (None,) * 7
        module, prec, B = ctx.module, ctx.prec, ctx.B
        eng: SAEEngine = module._engine
        x2, hidden, recon = ctx.saved_tensors
        handle = eng.prepare(prec, B, force=True)
        eng.reserve_relu(handle)
        N.check(eng.lib.wsae_ctx_set_relu_fp8(handle, relu_fp8_flag(module)), "wsae_ctx_set_relu_fp8")
        pk, xd, st = eng.pack.data_ptr(), _dtype_code(x2), eng.stream()
        w = float(module.sparsity_weight)
        if eng.generation != ctx.gen:  # another forward reused the ctx workspace since: rebuild xT / hidden^T for this batch
            h2, r2 = torch.empty_like(hidden), torch.empty_like(recon)
            N.check(eng.lib.wsae_relu_forward(handle, pk, x2.data_ptr(), xd, 0, B, w, h2.data_ptr(), r2.data_ptr(), 0, 0,
                                              st), "wsae_relu_forward")
            eng.generation += 1
        grads = torch.empty(eng.P, dtype=torch.float32, device=eng.device)
        N.check(eng.lib.wsae_relu_backward(handle, pk, x2.data_ptr(), xd, 0, B, w, hidden.data_ptr(), recon.data_ptr(),
                                           grads.data_ptr(), st), "wsae_relu_backward")
        grads.mul_(g_loss)
        need = ctx.needs_input_grad
        gv = lambda name, on: eng.view(name, grads) if on else None  # noqa: E731
        return (None, gv("encoder.weight", need[1]), gv("encoder.bias", need[2]), gv("decoder.weight", need[3]),
                gv("decoder.bias", need[4]), None, None)


class ReLUSAE(nn.Module):
    """ReLU + L1 sparse autoencoder (reference model.py:260-322).

    No ``b_pre``, default ``nn.Linear`` initialisation with (optionally) unit-norm decoder columns; the
    loss is ``mse + sparsity_weight * mean|hidden|``.  The kernels (``wsae_relu_forward/backward``) run on
    the TopK parameter pack with the pre-bias slot held at zero.  Unlike the reference (whose trainer
    crashes on it) it can be trained by ``SAETrainer``: there is no dead-feature
    bookkeeping for this module, as in the reference.
    """

    is_relu = True

    def __init__(self, input_dim: int, hidden_dim: int, sparsity_weight: float = 0.01,
                 normalize_decoder: bool = True, precision: Optional[str] = None):
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.sparsity_weight = sparsity_weight
        self.normalize_decoder = normalize_decoder
        self.precision = precision
        self.encoder = nn.Linear(input_dim, hidden_dim)
        self.decoder = nn.Linear(hidden_dim, input_dim)
        if normalize_decoder:
            with torch.no_grad():  # reference model.py:285-286 (F.normalize(dim=0))
                w = self.decoder.weight.data
                self.decoder.weight.data = w / w.norm(dim=0, keepdim=True).clamp_min(1e-12)
        self._engine: Optional[SAEEngine] = None

    def _named_core_params(self):
        return {"encoder.weight": self.encoder.weight, "decoder.weight": self.decoder.weight,
                "encoder.bias": self.encoder.bias, "decoder.bias": self.decoder.bias}

    def bind(self) -> SAEEngine:
        """Make the four parameters views of one device pack (the pre-bias slot of the pack stays zero)."""
        dev = self.encoder.weight.device
        require_device_tensor(self.encoder.weight, "ReLUSAE")
        eng = self._engine
        if eng is None or eng.device != dev:
            if eng is not None:
                eng.close()
            eng = SAEEngine(dev, self.input_dim, self.hidden_dim, 1)
            self._engine = eng
        with torch.no_grad():
            for name, p in self._named_core_params().items():
                v = eng.view(name)
                if p.data_ptr() != v.data_ptr() or p.shape != v.shape or p.stride() != v.stride():
                    v.copy_(p.detach().to(device=dev, dtype=torch.float32))
                    p.data = v
                    eng.invalidate()
        return eng

    def param_token(self) -> tuple:
        return tuple((p.data_ptr(), p._version) for p in self._named_core_params().values())

    def normalize_decoder_weights(self) -> None:
        """Unit-norm decoder columns when ``normalize_decoder`` is set (reference model.py:296-302)."""
        if not self.normalize_decoder:
            return
        eng = self.bind()
        handle = eng.ctx(_precision_code(self.precision), 64)
        N.check(eng.lib.wsae_normalize_decoder(handle, eng.pack.data_ptr(), eng.stream()), "wsae_normalize_decoder")
        eng.invalidate()

    def forward(self, x: Tensor) -> SAEOutput:
        """Reference model.py:304-322."""
        self.bind()
        require_device_tensor(x, "input")
        prec = _precision_code(self.precision)
        recon, hidden, loss, sparsity, l0 = _ReLUForward.apply(x, self.encoder.weight, self.encoder.bias,
                                                               self.decoder.weight, self.decoder.bias, self, prec)
        return SAEOutput(reconstructed=recon, hidden=hidden, loss=loss,
                         reconstruction_loss=(loss - self.sparsity_weight * sparsity).detach(),
                         sparsity_loss=sparsity, l0=l0)

    def extra_repr(self) -> str:
        return f"input_dim={self.input_dim}, hidden_dim={self.hidden_dim}, sparsity_weight={self.sparsity_weight}"


def create_sae(config: SAEConfig, input_dim: int) -> nn.Module:
    """Build the SAE a config describes (reference model.py:325-354)."""
    hidden_dim = config.get_hidden_dim(input_dim)
    if config.activation == "topk":
        return TopKSAE(input_dim=input_dim, hidden_dim=hidden_dim, k=config.k,
                       normalize_decoder=config.normalize_decoder,
                       dead_feature_threshold=config.dead_feature_threshold)
    # "relu" and (as in the reference) "gelu" both map to the ReLU SAE
    return ReLUSAE(input_dim=input_dim, hidden_dim=hidden_dim, sparsity_weight=config.sparsity_weight,
                   normalize_decoder=config.normalize_decoder)
