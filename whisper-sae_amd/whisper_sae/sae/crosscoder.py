"""Cross-layer crosscoders for MI355X: same classes, constructor arguments, parameter names and shapes, buffers and
output tuple as the reference's ``src/whisper_sae/sae/crosscoder.py`` (``CrosscoderOutput`` :26-35,
``CrossLayerCrosscoder`` :38-283, ``TopKCrossLayerCrosscoder`` :286-379, ``create_crosscoder`` :382-417), with the
TopK variant running on the TopK-SAE kernels of ``libwsae_hip.so`` ("a further sibling").

A TopK crosscoder IS a TopK SAE on the concatenated layers.  With ``x = [acts_l0 | acts_l1 | ...]`` of width
``n_layers * d_model``:

* the sum over layers of ``acts_l @ W_enc[l]`` (crosscoder.py:331-336) is one GEMM against the ``[d_sae, n_layers *
  d_model]`` matrix whose row ``s`` holds ``W_enc[:, :, s]`` flattened - exactly the ``W_e`` slot of the parameter pack,
  so ``W_enc`` is the strided view ``W_e.view(d_sae, n_layers, d_model).permute(1, 2, 0)`` of it;
* ``W_dec [d_sae, n_layers, d_model]`` flattened over its last two axes is the ``W_dT [H][D]`` slot as it stands, and
  the reference normalises it over that same flattened row (:117-121), which is what ``wsae_normalize_decoder`` does;
* ``b_dec [n_layers, d_model]`` flattened is ``b_d``; there is no pre-encoder bias (the slot stays zero);
* the loss is the SUM over layers of per-layer means (:352-358) = the squared error summed over all columns divided by
  ``B * d_model``: ``wsae_ctx_set_loss_cols(ctx, d_model)`` makes ``wsae_decode_loss`` (and its gradient seed
  ``g = 2 r / (B d_model)``) compute exactly that.

So ``forward`` is ``wsae_encode_topk`` -> ``wsae_decode_loss`` (-> ``wsae_weight_grads`` / ``wsae_input_grad`` in
backward) on one ``[B, n_layers * d_model]`` tensor, through the same autograd node as the transcoders.  The engine
limits apply to the concatenated width: ``n_layers * d_model <= 2048`` (Whisper-tiny x 4 layers = 1536, -base x 4 =
2048), ``k <= 128``.

The ReLU variant (``CrossLayerCrosscoder``, ``activation="relu"``) runs on the ReLU-SAE kernels the same way
(``wsae_relu_forward`` / ``wsae_relu_backward`` on the concatenated layers, ``loss_cols = d_model``).  Its sparsity term is
the decoder-norm-weighted L1 of the crosscoder paper (:213-217), ``mean_b sum_s |h_bs| n_s`` with ``n_s`` the norm of
feature ``s``'s flattened decoder row: the kernels take ``n`` as per-feature weights of their L1 term
(``wsae_ctx_set_relu_l1_weights``) with ``sparsity_weight * d_sae`` as the coefficient (their term is a mean over
``B * d_sae``).  ``n`` also depends on ``W_dec``; that part of the gradient, ``sparsity_weight * mean_b|h_bs| * W_dec[s] /
n_s``, is a ``[d_sae, n_layers * d_model]`` element-wise expression added in the backward (torch).
"""

from __future__ import annotations

from typing import Dict, List, NamedTuple, Optional

import torch
from torch import Tensor, nn

from .. import _native as N
from .engine import SAEEngine, require_device_tensor
from .model import _precision_code
from .engine import _dtype_code
from .transcoder import _SparsePath, _TranscoderBase, _as_rows


class CrosscoderOutput(NamedTuple):
    """What ``forward`` returns (field order of the reference, crosscoder.py:26-35)."""

    reconstructed: Dict[int, Tensor]
    hidden: Tensor
    loss: Tensor
    reconstruction_loss: Tensor
    sparsity_loss: Tensor
    l0: Tensor
    per_layer_loss: Dict[int, Tensor]


class _ReLUCrossPath(torch.autograd.Function):
    """wsae_relu_forward (-> wsae_relu_backward) on the concatenated layers with the decoder norms as per-feature L1
    weights; gradients of ``loss`` with respect to the four parameters (the norm term of ``W_dec`` included)."""

    @staticmethod
    def forward(ctx, xc, w_e, b_e, w_d, b_d, module, prec):
        eng: SAEEngine = module._engine
        x2 = _as_rows(xc, module.input_dim, eng.D)
        B = x2.shape[0]
        handle = eng.prepare(prec, B, force=True)
        eng.reserve_relu(handle)
        lib, st = eng.lib, eng.stream()
        norms = module.get_decoder_norms().detach().float().contiguous()
        lam = float(module.sparsity_weight)
        N.check(lib.wsae_ctx_set_relu_fp8(handle, 0), "wsae_ctx_set_relu_fp8")
        N.check(lib.wsae_ctx_set_loss_cols(handle, module.d_model), "wsae_ctx_set_loss_cols")
        N.check(lib.wsae_ctx_set_relu_l1_weights(handle, norms.data_ptr()), "wsae_ctx_set_relu_l1_weights")
        hidden = torch.empty(B, eng.H, dtype=torch.float32, device=eng.device)
        recon = torch.empty(B, eng.D, dtype=torch.float32, device=eng.device)
        wmean = torch.empty((), dtype=torch.float32, device=eng.device)  # sum_b sum_s n_s |h| / (B H)
        try:
            N.check(lib.wsae_relu_forward(handle, eng.pack.data_ptr(), x2.data_ptr(), _dtype_code(x2), 0, B, lam * eng.H,
                                          hidden.data_ptr(), recon.data_ptr(), eng.stats.data_ptr(), wmean.data_ptr(), st),
                    "wsae_relu_forward")
        finally:  # the weights are this call's: leave the ctx as the ReLU SAE expects it
            lib.wsae_ctx_set_relu_l1_weights(handle, 0)
            lib.wsae_ctx_set_loss_cols(handle, eng.D)
        sf = eng.stats_f32()
        loss, l0 = sf[0].clone(), sf[1].clone()
        sparsity = wmean * float(eng.H)  # mean_b sum_s n_s |h_bs|  (reference crosscoder.py:217)
        eng.generation += 1
        ctx.module, ctx.prec, ctx.gen, ctx.B, ctx.lam = module, prec, eng.generation, B, lam
        ctx.save_for_backward(x2, hidden, recon, norms)
        ctx.set_materialize_grads(False)
        recon_out = recon[:, :module.input_dim]
        ctx.mark_non_differentiable(recon_out, hidden, sparsity, l0)
        return recon_out, hidden, loss, sparsity, l0

    @staticmethod
    def backward(ctx, g_recon, g_hidden, g_loss, g_sparsity, g_l0):
        if g_loss is None:
            return (None,) * 7
        module, prec, B, lam = ctx.module, ctx.prec, ctx.B, ctx.lam
        eng: SAEEngine = module._engine
        x2, hidden, recon, norms = ctx.saved_tensors
        handle = eng.prepare(prec, B, force=True)
        eng.reserve_relu(handle)
        lib, st, pk, xd = eng.lib, eng.stream(), eng.pack.data_ptr(), _dtype_code(x2)
        N.check(lib.wsae_ctx_set_relu_fp8(handle, 0), "wsae_ctx_set_relu_fp8")
        N.check(lib.wsae_ctx_set_loss_cols(handle, module.d_model), "wsae_ctx_set_loss_cols")
        N.check(lib.wsae_ctx_set_relu_l1_weights(handle, norms.data_ptr()), "wsae_ctx_set_relu_l1_weights")
        grads = torch.empty(eng.P, dtype=torch.float32, device=eng.device)
        try:
            if eng.generation != ctx.gen:  # another call reused the ctx workspace since: restage this batch
                h2, r2 = torch.empty_like(hidden), torch.empty_like(recon)
                N.check(lib.wsae_relu_forward(handle, pk, x2.data_ptr(), xd, 0, B, lam * eng.H, h2.data_ptr(), r2.data_ptr(), 0, 0,
                                              st), "wsae_relu_forward")
                eng.generation += 1
            N.check(lib.wsae_relu_backward(handle, pk, x2.data_ptr(), xd, 0, B, lam * eng.H, hidden.data_ptr(),
                                           recon.data_ptr(), grads.data_ptr(), st), "wsae_relu_backward")
        finally:
            lib.wsae_ctx_set_relu_l1_weights(handle, 0)
            lib.wsae_ctx_set_loss_cols(handle, eng.D)
        need = ctx.needs_input_grad
        if need[3] and lam != 0.0:
            # d/dW_dec of lam * mean_b sum_s |h_bs| n_s through n_s = ||W_dec[s]||:  lam * mean_b|h_bs| * W_dec[s] / n_s
            col = hidden.sum(dim=0) * (lam / B)                       # hidden >= 0
            wd = module._sliced("decoder.weight").detach().reshape(module.d_sae, -1)
            gview = eng.view("decoder.weight", grads)                # [Dp, H] view of the pack layout
            gview[:module.input_dim, :].add_((wd * (col / norms.clamp_min(1e-30)).unsqueeze(1)).t())
        grads.mul_(g_loss)
        gv = lambda name, on: module._sliced(name, grads) if on else None  # noqa: E731
        return (None, gv("encoder.weight", need[1]), gv("encoder.bias", need[2]), gv("decoder.weight", need[3]),
                gv("decoder.bias", need[4]), None, None)


class CrossLayerCrosscoder(_TranscoderBase):
    """Shared sparse code over several layers (reference crosscoder.py:38-283): parameters ``W_enc [n_layers, d_model,
    d_sae]``, ``b_enc [d_sae]``, ``W_dec [d_sae, n_layers, d_model]``, ``b_dec [n_layers, d_model]``."""

    def __init__(self, d_model: int, n_layers: int, d_sae: int, layer_indices: Optional[List[int]] = None,
                 activation: str = "relu", sparsity_weight: float = 0.01, normalize_decoder: bool = True,
                 dead_feature_threshold: int = 10_000, precision: Optional[str] = None):
        nn.Module.__init__(self)
        self.d_model, self.n_layers, self.d_sae = d_model, n_layers, d_sae
        self.layer_indices = layer_indices or list(range(n_layers))
        self.activation = activation
        self.sparsity_weight = sparsity_weight
        self.normalize_decoder = normalize_decoder
        self.dead_feature_threshold = dead_feature_threshold
        self.precision = precision
        # the names the shared transcoder machinery works with
        self.input_dim = self.output_dim = n_layers * d_model
        self.hidden_dim = d_sae
        self.k = min(32, d_sae)  # engine shape for decode(); the TopK subclass sets the real k
        self._engine: Optional[SAEEngine] = None
        self._last_code = None
        # same creation order (and RNG draw) as the reference (:88-99)
        self.W_enc = nn.Parameter(torch.empty(n_layers, d_model, d_sae))
        self.b_enc = nn.Parameter(torch.zeros(d_sae))
        self.W_dec = nn.Parameter(torch.empty(d_sae, n_layers, d_model))
        self.b_dec = nn.Parameter(torch.zeros(n_layers, d_model))
        self._init_weights()
        self._register_clock()

    def _init_weights(self) -> None:
        """Reference crosscoder.py:107-122: xavier decoder -> unit rows over (layers x d_model) x 0.1; each layer's
        encoder starts as the transpose of its decoder block."""
        with torch.no_grad():
            nn.init.xavier_uniform_(self.W_dec)
            if self.normalize_decoder:
                flat = nn.functional.normalize(self.W_dec.view(self.d_sae, -1), dim=1)
                self.W_dec.data = flat.view(self.d_sae, self.n_layers, self.d_model)
                self.W_dec.data *= 0.1
            self.W_enc.data = self.W_dec.data.permute(1, 2, 0).contiguous()

    # -- binding: the four parameters are views of the engine's pack --------------------------------------
    def _named_core_params(self):
        return {"encoder.weight": self.W_enc, "decoder.weight": self.W_dec, "encoder.bias": self.b_enc,
                "decoder.bias": self.b_dec}

    def _sliced(self, name: str, base: Optional[Tensor] = None) -> Tensor:
        v = self._engine.view(name, base)
        L, d, w = self.n_layers, self.d_model, self.input_dim
        if name == "encoder.weight":   # pack [H][Dp] -> [L, d, S]
            return v[:, :w].unflatten(1, (L, d)).permute(1, 2, 0)
        if name == "decoder.weight":   # engine view is W_dT^T [Dp, H] -> [S, L, d]
            return v[:w, :].t().unflatten(1, (L, d))
        if name == "decoder.bias":
            return v[:w].view(L, d)
        return v

    def _mse_cols(self) -> int:
        return self.d_model

    def _check_width(self) -> None:
        if self.input_dim > 2048:
            raise N.WsaeError(f"n_layers * d_model = {self.input_dim} exceeds the engine's row width limit of 2048")

    def _gather(self, layer_activations: Dict[int, Tensor], need_all: bool) -> Tensor:
        """``[B, n_layers * d_model]`` with layer ``layer_indices[i]`` in columns ``i*d_model ..``; layers missing
        from the dict contribute zeros (encode sums over the layers it is given, crosscoder.py:156-163)."""
        if not layer_activations:
            raise ValueError("layer_activations is empty")
        for key in layer_activations:
            if key not in self.layer_indices:
                raise ValueError(f"{key} is not in list")  # what list.index raises in the reference
        first = next(iter(layer_activations.values()))
        require_device_tensor(first, "layer_activations")
        parts = []
        for li in self.layer_indices:
            a = layer_activations.get(li)
            if a is None:
                if need_all:
                    raise KeyError(li)
                a = torch.zeros(first.shape[0], self.d_model, dtype=first.dtype, device=first.device)
            elif a.dim() != 2 or a.shape[1] != self.d_model or a.shape[0] != first.shape[0]:
                raise ValueError(f"layer {li}: expected [{first.shape[0]}, {self.d_model}], got {tuple(a.shape)}")
            parts.append(a)
        return torch.cat(parts, dim=1)

    def _split(self, flat: Tensor) -> Dict[int, Tensor]:
        d = self.d_model
        return {li: flat[:, i * d:(i + 1) * d] for i, li in enumerate(self.layer_indices)}

    # -- reference API -----------------------------------------------------------------------------------
    def get_decoder_norms(self) -> Tensor:
        """L2 norm of each feature's decoder row over all layers, ``[d_sae]`` (reference crosscoder.py:124-131)."""
        return torch.norm(self.W_dec.reshape(self.d_sae, -1), dim=1)

    def _relu(self, xc: Tensor):
        if self.activation != "relu":
            raise ValueError(f"Unknown activation: {self.activation}")  # as the reference (crosscoder.py:167)
        self._check_width()
        self.bind()
        return _ReLUCrossPath.apply(xc, self.W_enc, self.b_enc, self.W_dec, self.b_dec, self,
                                    _precision_code(self.precision))

    @torch.no_grad()
    def encode(self, layer_activations: Dict[int, Tensor]) -> Tensor:
        """``relu(sum_l acts_l @ W_enc[l] + b_enc)`` of the layers given (reference crosscoder.py:142-169)."""
        return self._relu(self._gather(layer_activations, need_all=False))[1]

    @torch.no_grad()
    def decode(self, hidden: Tensor) -> Dict[int, Tensor]:
        """Per-layer reconstructions of any dense code (reference crosscoder.py:171-186)."""
        self._check_width()
        return self._split(_TranscoderBase.decode(self, hidden))

    def forward(self, layer_activations: Dict[int, Tensor]) -> CrosscoderOutput:
        """Reference crosscoder.py:188-235: sum of per-layer MSEs + ``sparsity_weight`` x decoder-norm-weighted L1."""
        xc = self._gather(layer_activations, need_all=True)
        recon, hidden, loss, sparsity, l0 = self._relu(xc)
        with torch.no_grad():
            err = (recon - xc.detach().float()).square_().view(-1, self.n_layers, self.d_model).mean(dim=(0, 2))
            if self.training:  # the dead-feature clock (crosscoder.py:237-242; the ReLU kernels keep none)
                self.step_count += 1
                self.feature_last_activated[(hidden > 0).any(dim=0)] = self.step_count
        per_layer = {li: err[i] for i, li in enumerate(self.layer_indices)}
        return CrosscoderOutput(reconstructed=self._split(recon), hidden=hidden, loss=loss, reconstruction_loss=err.sum(),
                                sparsity_loss=sparsity, l0=l0, per_layer_loss=per_layer)

    def get_feature_layer_norms(self) -> Tensor:
        """``[d_sae, n_layers]`` decoder norm of every feature in every layer (reference crosscoder.py:251-261)."""
        return torch.norm(self.W_dec, dim=2)

    def get_cross_layer_features(self, threshold: float = 0.1) -> Tensor:
        """Features whose decoder has more than ``threshold`` of its largest per-layer norm in at least two layers
        (reference crosscoder.py:263-283)."""
        layer_norms = self.get_feature_layer_norms()
        relative = layer_norms / (layer_norms.max(dim=1, keepdim=True).values + 1e-8)
        return (relative > threshold).sum(dim=1) >= 2

    def extra_repr(self) -> str:
        return f"d_model={self.d_model}, n_layers={self.n_layers}, d_sae={self.d_sae}, layers={self.layer_indices}"


class TopKCrossLayerCrosscoder(CrossLayerCrosscoder):
    """Crosscoder whose shared code keeps the ``k`` largest pre-activations per row (reference crosscoder.py:286-379),
    on the TopK-SAE kernels."""

    def __init__(self, d_model: int, n_layers: int, d_sae: int, k: int = 32, layer_indices: Optional[List[int]] = None,
                 normalize_decoder: bool = True, dead_feature_threshold: int = 10_000, precision: Optional[str] = None):
        super().__init__(d_model=d_model, n_layers=n_layers, d_sae=d_sae, layer_indices=layer_indices, activation="relu",
                         sparsity_weight=0.0, normalize_decoder=normalize_decoder,
                         dead_feature_threshold=dead_feature_threshold, precision=precision)
        self.k = k

    @torch.no_grad()
    def encode(self, layer_activations: Dict[int, Tensor]) -> Tensor:
        """Shared sparse code ``[B, d_sae]`` of the layers given (reference crosscoder.py:323-345)."""
        self._check_width()
        return _TranscoderBase.encode(self, self._gather(layer_activations, need_all=False))

    def forward(self, layer_activations: Dict[int, Tensor]) -> CrosscoderOutput:
        """Reference crosscoder.py:347-379.  ``loss`` carries the gradient (HIP backward kernels); the per-layer losses
        and reconstructions are read-outs.  In training mode also advances the dead-feature clock."""
        self._check_width()
        xc = self._gather(layer_activations, need_all=True)
        self.bind()
        recon, hidden, loss, l0 = _SparsePath.apply(xc, xc, self.W_enc, self.b_enc, self.W_dec, self.b_dec, self,
                                                    _precision_code(self.precision))
        with torch.no_grad():
            err = (recon - xc.detach().float()).square_().view(-1, self.n_layers, self.d_model).mean(dim=(0, 2))
        per_layer = {li: err[i] for i, li in enumerate(self.layer_indices)}
        return CrosscoderOutput(reconstructed=self._split(recon), hidden=hidden, loss=loss, reconstruction_loss=loss,
                                sparsity_loss=torch.zeros((), device=xc.device), l0=l0, per_layer_loss=per_layer)


def create_crosscoder(d_model: int, n_layers: int, d_sae: int, k: Optional[int] = None, use_topk: bool = True,
                      **kwargs) -> nn.Module:
    """Reference crosscoder.py:382-417."""
    if use_topk:
        return TopKCrossLayerCrosscoder(d_model=d_model, n_layers=n_layers, d_sae=d_sae, k=k or 32, **kwargs)
    return CrossLayerCrosscoder(d_model=d_model, n_layers=n_layers, d_sae=d_sae, **kwargs)
