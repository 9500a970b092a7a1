"""Transcoders for MI355X: same classes, constructor arguments, attributes and state-dict keys as the reference's
``src/whisper_sae/sae/transcoder.py`` (``TopKTranscoder`` :32-252, ``SkipTranscoder`` :255-422, ``create_transcoder``
:425-460), on the TopK-SAE kernels of ``libwsae_hip.so``.

A transcoder is the TopK SAE with three differences: the target of the MSE is a second tensor
(``mlp_output``), there is no pre-encoder bias, and input and output widths may differ.  The kernels take all three
without a new code path:

* ``wsae_encode_topk`` runs on the input, ``wsae_decode_loss`` gets the TARGET as its ``x``, ``wsae_weight_grads`` the
  input again (its ``dW_e`` contraction pairs ``dpre`` with the input rows);
* the pre-bias slot of the parameter pack stays zero and is not a parameter (as for ``ReLUSAE``);
* the engine works at ``D = max(input_dim, output_dim)`` rounded up to 32: narrower tensors are zero-padded on the way
  in, the padded weight columns start at zero and receive exactly zero gradient, and ``wsae_ctx_set_loss_cols`` keeps
  the MSE a mean over the real ``output_dim`` columns.

``SkipTranscoder`` adds the dense affine skip path ``skip(x)``: a plain library GEMM (``torch.nn.functional.linear``)
whose output is subtracted from the target before the sparse path sees it, so ``predicted = decoder(hidden) + skip(x)``
and the gradient of the loss reaches the skip parameters through ordinary autograd.

Like the reference, there is no trainer route for these modules: they are trained by the caller's own optimizer loop
(``loss.backward()`` fills ``.grad`` of every parameter through the HIP backward kernels).
"""

from __future__ import annotations

from typing import NamedTuple, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .. import _native as N
from .engine import SAEEngine, _dtype_code, require_device_tensor
from .model import _precision_code


class TranscoderOutput(NamedTuple):
    """What ``forward`` returns (field order of the reference, transcoder.py:21-29)."""

    predicted: Tensor
    hidden: Tensor
    loss: Tensor
    reconstruction_loss: Tensor
    sparsity_loss: Tensor
    l0: Tensor


def _as_rows(t: Tensor, width: int, padded: int) -> Tensor:
    """``[.., width]`` -> contiguous ``[rows, padded]`` float32 / bfloat16 (zero columns beyond ``width``)."""
    t2 = t.reshape(-1, width)
    if t2.dtype not in (torch.float32, torch.bfloat16):
        t2 = t2.float()
    if padded != width:
        t2 = F.pad(t2, (0, padded - width))
    return t2.contiguous()


class _SparsePath(torch.autograd.Function):
    """encode_topk(input) -> decode_loss(target) (-> weight_grads in backward) as one autograd node.

    Gradients are defined for ``loss`` with respect to the four parameters, the input and the target."""

    @staticmethod
    def forward(ctx, x, target, w_e, b_e, w_d, b_d, module, prec):
        eng: SAEEngine = module._engine
        lib = eng.lib
        din, dout, dp = module.input_dim, module.output_dim, eng.D
        x2, t2 = _as_rows(x, din, dp), _as_rows(target, dout, dp)
        B = x2.shape[0]
        if t2.shape[0] != B:
            raise ValueError(f"mlp_input has {B} rows, mlp_output {t2.shape[0]}")
        handle = eng.prepare(prec, B, force=True)
        st = eng.stream()
        training = module.training
        need_bwd = any(ctx.needs_input_grad[:6])
        vals = torch.empty(B, eng.k, dtype=torch.float32, device=eng.device)
        idx = torch.empty(B, eng.k, dtype=torch.int32, device=eng.device)
        pred = torch.empty(B, dp, dtype=torch.float32, device=eng.device)
        dpre = torch.empty(B, eng.k, dtype=torch.float32, device=eng.device) if need_bwd else None
        step_ptr = module.step_count.data_ptr() if training else 0
        last_ptr = module.feature_last_activated.data_ptr() if training else 0
        pk = eng.pack.data_ptr()
        N.check(lib.wsae_ctx_set_loss_cols(handle, module._mse_cols()), "wsae_ctx_set_loss_cols")
        N.check(lib.wsae_ctx_set_fired(handle, 0), "wsae_ctx_set_fired")
        N.check(lib.wsae_encode_topk(handle, pk, x2.data_ptr(), _dtype_code(x2), 0, B, vals.data_ptr(), idx.data_ptr(),
                                     step_ptr, eng.stats.data_ptr(), st), "wsae_encode_topk")
        # bit 1: keep the fp32 g when the target needs a gradient (the skip path trains through it)
        want = (1 if need_bwd else 0) | (2 if (need_bwd and ctx.needs_input_grad[1]) else 0)
        N.check(lib.wsae_decode_loss(handle, pk, t2.data_ptr(), _dtype_code(t2), 0, vals.data_ptr(), idx.data_ptr(), B,
                                     pred.data_ptr(), want, N.ptr(dpre), last_ptr, step_ptr, eng.stats.data_ptr(), st),
                "wsae_decode_loss")
        hidden = torch.empty(B, eng.H, dtype=torch.float32, device=eng.device)
        N.check(lib.wsae_densify(handle, vals.data_ptr(), idx.data_ptr(), B, hidden.data_ptr(), st), "wsae_densify")
        sf = eng.stats_f32()
        loss, l0 = sf[0].clone(), sf[1].clone()
        eng.generation += 1
        ctx.module, ctx.prec, ctx.gen, ctx.B, ctx.want = module, prec, eng.generation, B, want
        ctx.x_shape, ctx.t_shape = x.shape, target.shape
        ctx.save_for_backward(x2, t2, vals, idx, dpre if dpre is not None else vals)
        ctx.has_dpre = dpre is not None
        ctx.set_materialize_grads(False)
        pred_out = pred[:, :dout].reshape(*x.shape[:-1], dout)
        hidden_out = hidden.reshape(*x.shape[:-1], eng.H)
        ctx.mark_non_differentiable(pred_out, hidden_out, l0)
        module._last_code = (vals, idx)
        return pred_out, hidden_out, loss, l0

    @staticmethod
    def backward(ctx, g_pred, g_hidden, g_loss, g_l0):
        if g_loss is None:
            return (None,) * 8
        module, prec, B = ctx.module, ctx.prec, ctx.B
        eng: SAEEngine = module._engine
        lib = eng.lib
        din, dout, dp = module.input_dim, module.output_dim, eng.D
        x2, t2, vals, idx, dpre = ctx.saved_tensors
        st = eng.stream()
        handle = eng.prepare(prec, B, force=True)
        pk = eng.pack.data_ptr()
        need = ctx.needs_input_grad
        N.check(lib.wsae_ctx_set_loss_cols(handle, module._mse_cols()), "wsae_ctx_set_loss_cols")
        if eng.generation != ctx.gen or not ctx.has_dpre:
            # another call has reused the ctx workspace since: restage this batch (input, then g / dpre from the target)
            tmp_v, tmp_i = torch.empty_like(vals), torch.empty_like(idx)
            dpre = torch.empty_like(vals)
            N.check(lib.wsae_encode_topk(handle, pk, x2.data_ptr(), _dtype_code(x2), 0, B, tmp_v.data_ptr(), tmp_i.data_ptr(),
                                         0, eng.stats.data_ptr(), st), "wsae_encode_topk")
            scratch = torch.zeros(N.STATS_WORDS, dtype=torch.int32, device=eng.device)
            N.check(lib.wsae_decode_loss(handle, pk, t2.data_ptr(), _dtype_code(t2), 0, vals.data_ptr(), idx.data_ptr(), B, 0,
                                         3 if need[1] else 1, dpre.data_ptr(), 0, 0, scratch.data_ptr(), st),
                    "wsae_decode_loss")
            eng.generation += 1
        grads = torch.empty(eng.P, dtype=torch.float32, device=eng.device)
        N.check(lib.wsae_weight_grads(handle, pk, x2.data_ptr(), _dtype_code(x2), 0, vals.data_ptr(), idx.data_ptr(),
                                      dpre.data_ptr(), B, grads.data_ptr(), st), "wsae_weight_grads")
        grads.mul_(g_loss)
        dx = dt = None
        if need[0]:
            dx = torch.empty(B, dp, dtype=torch.float32, device=eng.device)
            N.check(lib.wsae_input_grad(handle, pk, idx.data_ptr(), dpre.data_ptr(), B, dx.data_ptr(), 0, st),
                    "wsae_input_grad")
            dx = (dx[:, :din] * g_loss).reshape(ctx.x_shape)
        if need[1]:  # d loss / d target = -g   (g = 2 (predicted - target) / (B out): the fp32 copy kept by decode)
            g32 = torch.empty(B, dp, dtype=torch.float32, device=eng.device)
            N.check(lib.wsae_last_residual_grad(handle, B, g32.data_ptr(), st), "wsae_last_residual_grad")
            dt = (g32[:, :dout] * (-g_loss)).reshape(ctx.t_shape)
        gv = lambda name, on: module._sliced(name, grads) if on else None  # noqa: E731
        return (dx, dt, gv("encoder.weight", need[2]), gv("encoder.bias", need[3]), gv("decoder.weight", need[4]),
                gv("decoder.bias", need[5]), None, None)


class _TranscoderBase(nn.Module):
    """What both transcoders share: the encoder / decoder pair bound to one padded parameter pack, encode / decode,
    the sparse path and the dead-feature clock."""

    def _setup(self, input_dim, output_dim, hidden_dim, k, normalize_decoder, dead_feature_threshold, precision):
        self.input_dim, self.output_dim, self.hidden_dim, self.k = input_dim, output_dim, hidden_dim, k
        self.normalize_decoder = normalize_decoder
        self.dead_feature_threshold = dead_feature_threshold
        self.precision = precision
        self.encoder = nn.Linear(input_dim, hidden_dim, bias=True)
        self.decoder = nn.Linear(hidden_dim, output_dim, bias=True)
        self._engine: Optional[SAEEngine] = None
        self._last_code = None

    def _register_clock(self):
        self.register_buffer("feature_last_activated", torch.zeros(self.hidden_dim, dtype=torch.long))
        self.register_buffer("step_count", torch.tensor(0, dtype=torch.long))

    # -- device binding ------------------------------------------------------------------------------
    def _named_core_params(self):
        return {"encoder.weight": self.encoder.weight, "decoder.weight": self.decoder.weight,
                "encoder.bias": self.encoder.bias, "decoder.bias": self.decoder.bias}

    def _sliced(self, name: str, base: Optional[Tensor] = None) -> Tensor:
        """The reference-shaped view of parameter ``name`` inside the (padded) pack layout of ``base``."""
        v = self._engine.view(name, base)
        if name == "encoder.weight":
            return v[:, :self.input_dim]
        if name == "decoder.weight":
            return v[:self.output_dim, :]
        if name == "decoder.bias":
            return v[:self.output_dim]
        return v

    def _mse_cols(self) -> int:
        """Columns the reconstruction MSE averages over (``wsae_ctx_set_loss_cols``)."""
        return self.output_dim

    def bind(self) -> SAEEngine:
        anchor = self._named_core_params()["encoder.weight"]
        dev = anchor.device
        require_device_tensor(anchor, type(self).__name__)
        if self.k > self.hidden_dim:
            raise ValueError(f"k={self.k} exceeds hidden_dim={self.hidden_dim}")
        eng = self._engine
        if eng is None or eng.device != dev or eng.k != self.k:
            if eng is not None:
                eng.close()
            dp = (max(self.input_dim, self.output_dim) + 31) // 32 * 32
            eng = SAEEngine(dev, dp, self.hidden_dim, self.k)  # (the pack is zero-initialised: padding, pre-bias slot)
            self._engine = eng
        with torch.no_grad():
            for name, p in self._named_core_params().items():
                v = self._sliced(name)
                if p.data_ptr() != v.data_ptr() or p.shape != v.shape or p.stride() != v.stride():
                    v.copy_(p.detach().to(device=dev, dtype=torch.float32))
                    p.data = v
                    eng.invalidate()
        for buf in (self.feature_last_activated, self.step_count):
            if buf.device != dev:
                raise N.WsaeError("module buffers and parameters are on different devices; use module.to(device)")
        return eng

    # -- reference API ---------------------------------------------------------------------------------
    def normalize_decoder_weights(self) -> None:
        """Unit-norm decoder columns (reference transcoder.py:105-110)."""
        eng = self.bind()
        handle = eng.ctx(_precision_code(self.precision), 64)
        N.check(eng.lib.wsae_normalize_decoder(handle, eng.pack.data_ptr(), eng.stream()), "wsae_normalize_decoder")
        eng.invalidate()

    def _code(self, x: Tensor):
        eng = self.bind()
        require_device_tensor(x, "input")
        x2 = _as_rows(x, self.input_dim, eng.D)
        B = x2.shape[0]
        handle = eng.prepare(_precision_code(self.precision), B, force=True)
        vals = torch.empty(B, eng.k, dtype=torch.float32, device=eng.device)
        idx = torch.empty(B, eng.k, dtype=torch.int32, device=eng.device)
        N.check(eng.lib.wsae_encode_topk(handle, eng.pack.data_ptr(), x2.data_ptr(), _dtype_code(x2), 0, B, vals.data_ptr(),
                                         idx.data_ptr(), 0, eng.stats.data_ptr(), eng.stream()), "wsae_encode_topk")
        eng.generation += 1
        return eng, handle, B, vals, idx

    @torch.no_grad()
    def encode(self, x: Tensor) -> Tensor:
        """Sparse latent ``[.., hidden_dim]`` with at most ``k`` non-zeros per row (reference transcoder.py:112-129)."""
        eng, handle, B, vals, idx = self._code(x)
        hidden = torch.empty(B, eng.H, dtype=torch.float32, device=eng.device)
        N.check(eng.lib.wsae_densify(handle, vals.data_ptr(), idx.data_ptr(), B, hidden.data_ptr(), eng.stream()),
                "wsae_densify")
        return hidden.reshape(*x.shape[:-1], eng.H)

    @torch.no_grad()
    def decode(self, hidden: Tensor) -> Tensor:
        """``decoder(hidden)`` for any dense code (reference transcoder.py:131-140)."""
        eng = self.bind()
        require_device_tensor(hidden, "hidden")
        h2 = hidden.reshape(-1, eng.H).float().contiguous()
        handle = eng.ctx(_precision_code(self.precision), 64)
        out = torch.empty(h2.shape[0], eng.D, dtype=torch.float32, device=eng.device)
        N.check(eng.lib.wsae_decode_dense(handle, eng.pack.data_ptr(), h2.data_ptr(), h2.shape[0], out.data_ptr(),
                                          eng.stream()), "wsae_decode_dense")
        return out[:, :self.output_dim].reshape(*hidden.shape[:-1], self.output_dim)

    def _sparse(self, mlp_input: Tensor, target: Tensor):
        self.bind()
        require_device_tensor(mlp_input, "mlp_input")
        require_device_tensor(target, "mlp_output")
        return _SparsePath.apply(mlp_input, target, self.encoder.weight, self.encoder.bias, self.decoder.weight,
                                 self.decoder.bias, self, _precision_code(self.precision))

    # -- dead features (reference transcoder.py:178-196) -------------------------------------------------
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

    def extra_repr(self) -> str:
        return f"input_dim={self.input_dim}, output_dim={self.output_dim}, hidden_dim={self.hidden_dim}, k={self.k}"


class TopKTranscoder(_TranscoderBase):
    """TopK transcoder: predicts the MLP output from the MLP input through a k-sparse code (reference
    transcoder.py:32-252).  ``precision`` as for ``TopKSAE``."""

    def __init__(self, input_dim: int, output_dim: int, hidden_dim: int, k: int = 32, normalize_decoder: bool = True,
                 dead_feature_threshold: int = 10_000, precision: Optional[str] = None):
        super().__init__()
        # same construction (and RNG draw) order as the reference: encoder, decoder, then the decoder initialisation
        self._setup(input_dim, output_dim, hidden_dim, k, normalize_decoder, dead_feature_threshold, precision)
        with torch.no_grad():  # reference transcoder.py:96-103: xavier -> unit-norm columns -> x0.1
            nn.init.xavier_uniform_(self.decoder.weight)
            self.decoder.weight.data = F.normalize(self.decoder.weight.data, dim=0)
            self.decoder.weight.data *= 0.1
        self._register_clock()

    def forward(self, mlp_input: Tensor, mlp_output: Tensor) -> TranscoderOutput:
        """Reference transcoder.py:142-176.  In training mode also advances the dead-feature clock."""
        pred, hidden, loss, l0 = self._sparse(mlp_input, mlp_output)
        return TranscoderOutput(predicted=pred, hidden=hidden, loss=loss, reconstruction_loss=loss,
                                sparsity_loss=torch.zeros((), device=mlp_input.device), l0=l0)

    @torch.no_grad()
    def resample_dead_features(self, mlp_inputs: Tensor, mlp_outputs: Tensor, num_resample: Optional[int] = None) -> int:
        """Reference transcoder.py:198-252: dead features (ascending, capped) are rewritten from the highest-error
        rows - encoder row = the L2-normalised INPUT row, decoder column = the L2-normalised RESIDUAL row - with the
        reference's quirks (the forward advances the clock in train mode; the capped dead count is returned)."""
        eng = self.bind()
        require_device_tensor(mlp_inputs, "mlp_inputs")
        lib, st = eng.lib, eng.stream()
        x2, t2 = _as_rows(mlp_inputs, self.input_dim, eng.D), _as_rows(mlp_outputs, self.output_dim, eng.D)
        Br = x2.shape[0]
        handle = eng.prepare(_precision_code(self.precision), Br, force=True)
        mask = torch.empty(self.hidden_dim, dtype=torch.uint8, device=eng.device)
        N.check(lib.wsae_dead_scan(handle, self.feature_last_activated.data_ptr(), self.step_count.data_ptr(),
                                   int(self.dead_feature_threshold), mask.data_ptr(), eng.stats.data_ptr(), st),
                "wsae_dead_scan")
        if int(eng.stats[5].item()) == 0:
            return 0
        training = self.training
        N.check(lib.wsae_ctx_set_loss_cols(handle, self._mse_cols()), "wsae_ctx_set_loss_cols")
        N.check(lib.wsae_ctx_set_fired(handle, 0), "wsae_ctx_set_fired")
        vals = torch.empty(Br, eng.k, dtype=torch.float32, device=eng.device)
        idx = torch.empty(Br, eng.k, dtype=torch.int32, device=eng.device)
        pred = torch.empty(Br, eng.D, dtype=torch.float32, device=eng.device)
        pk = eng.pack.data_ptr()
        step_ptr = self.step_count.data_ptr() if training else 0
        N.check(lib.wsae_encode_topk(handle, pk, x2.data_ptr(), _dtype_code(x2), 0, Br, vals.data_ptr(), idx.data_ptr(),
                                     step_ptr, eng.stats.data_ptr(), st), "wsae_encode_topk")
        N.check(lib.wsae_decode_loss(handle, pk, t2.data_ptr(), _dtype_code(t2), 0, vals.data_ptr(), idx.data_ptr(), Br,
                                     pred.data_ptr(), 0, 0, self.feature_last_activated.data_ptr() if training else 0,
                                     step_ptr, eng.stats.data_ptr(), st), "wsae_decode_loss")
        eng.generation += 1
        row_err = torch.empty(Br, dtype=torch.float32, device=eng.device)
        resid = torch.empty(Br, eng.D, dtype=torch.float32, device=eng.device)
        N.check(lib.wsae_row_errors(handle, t2.data_ptr(), _dtype_code(t2), 0, pred.data_ptr(), Br, row_err.data_ptr(),
                                    resid.data_ptr(), st), "wsae_row_errors")
        n_out = torch.zeros(1, dtype=torch.int32, device=eng.device)
        cap = -1 if num_resample is None else int(num_resample)
        N.check(lib.wsae_resample_dead(handle, pk, x2.data_ptr(), _dtype_code(x2), 0, Br, row_err.data_ptr(), mask.data_ptr(),
                                       self.feature_last_activated.data_ptr(), self.step_count.data_ptr(), cap,
                                       n_out.data_ptr(), resid.data_ptr(), st), "wsae_resample_dead")
        eng.invalidate()
        return int(n_out.item())


class SkipTranscoder(_TranscoderBase):
    """Transcoder with an affine skip connection: ``predicted = decoder(hidden) + skip(x)`` (reference
    transcoder.py:255-422).  Decoder and skip start at zero (the reference's "paper" initialisation)."""

    def __init__(self, input_dim: int, output_dim: int, hidden_dim: int, k: int = 32, normalize_decoder: bool = True,
                 dead_feature_threshold: int = 10_000, precision: Optional[str] = None):
        super().__init__()
        # construction order of the reference: encoder, decoder, skip (then zeros for decoder and skip)
        self._setup(input_dim, output_dim, hidden_dim, k, normalize_decoder, dead_feature_threshold, precision)
        self.skip = nn.Linear(input_dim, output_dim, bias=True)
        with torch.no_grad():  # reference transcoder.py:314-330
            for p in (self.decoder.weight, self.decoder.bias, self.skip.weight, self.skip.bias):
                p.zero_()
        self._register_clock()

    def set_output_bias(self, mean_output: Tensor) -> None:
        """Decoder bias <- empirical mean of the MLP outputs (reference transcoder.py:332-344)."""
        with torch.no_grad():
            self.decoder.bias.copy_(mean_output.to(self.decoder.bias.device))
        if self._engine is not None:
            self._engine.invalidate()

    def forward(self, mlp_input: Tensor, mlp_output: Tensor) -> TranscoderOutput:
        """Reference transcoder.py:365-403."""
        skip_out = self.skip(mlp_input.float())              # dense affine path: a plain library GEMM
        pred_sparse, hidden, loss, l0 = self._sparse(mlp_input, mlp_output.float() - skip_out)
        return TranscoderOutput(predicted=pred_sparse + skip_out.detach(), hidden=hidden, loss=loss,
                                reconstruction_loss=loss, sparsity_loss=torch.zeros((), device=mlp_input.device), l0=l0)

    @torch.no_grad()
    def get_skip_contribution(self, mlp_input: Tensor, mlp_output: Tensor) -> float:
        """Fraction of the output variance the skip path alone explains (reference transcoder.py:405-422)."""
        skip_pred = self.skip(mlp_input.float())
        skip_var = ((skip_pred - mlp_output) ** 2).mean()
        total_var = ((mlp_output - mlp_output.mean(dim=0)) ** 2).mean()
        return float((1 - skip_var / (total_var + 1e-8)).item())


def create_transcoder(input_dim: int, output_dim: int, hidden_dim: int, k: int = 32, use_skip: bool = True,
                      **kwargs) -> nn.Module:
    """Reference transcoder.py:425-460."""
    cls = SkipTranscoder if use_skip else TopKTranscoder
    return cls(input_dim=input_dim, output_dim=output_dim, hidden_dim=hidden_dim, k=k, **kwargs)
