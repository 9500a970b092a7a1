"""Whisper activation extraction with forward hooks, feeding the on-device ring.

Drop-in for the reference's ``whisper_sae.sae.hooks`` (/root/reference/src/whisper_sae/sae/hooks.py:15-230): the same four
names - ``ActivationCache``, ``WhisperActivationExtractor``, ``extract_features_batch``, ``flatten_activations`` - with the
same signatures and the same tensors.  The structure is this build's own:

* a **tap** is one ``(component, layer)`` whose output is wanted; the extractor is a table of taps, every tap has a *sink*,
  and ONE forward hook (``_Tap.__call__``) serves encoder and decoder layers alike;
* sinks: ``_ListSink`` keeps the (layer-normed) tensor, on the host as the reference does (hooks.py:92, :106) or - with
  ``keep_on_device=True`` - where it was produced; ``_RingSink`` sends the layer output through the final LayerNorm into the
  ``ActivationRing`` the trainer samples from in ONE kernel (``wsae_ring_push_layernorm``) and keeps nothing: extraction and
  SAE training share the GPU with no host round trip;
* ``run_whisper_taps`` is the one place that drives the model (encoder pass, one decoder step from the start token): both
  ``extract_features_batch`` here and ``extract_and_cache_features`` (data/feature_cache.py) go through it.

The model is whatever ``transformers`` provides (``WhisperForConditionalGeneration``); this module only hooks it.
"""

from __future__ import annotations

from typing import Literal, Optional

import torch
from torch import Tensor

COMPONENTS = ("encoder", "decoder")


class ActivationCache:
    """Per-layer lists of captured activations, ``cache.encoder[layer]`` / ``cache.decoder[layer]`` (reference
    hooks.py:15-37: same attributes and accessors)."""

    def __init__(self, encoder: Optional[dict] = None, decoder: Optional[dict] = None):
        self.encoder: dict = {} if encoder is None else encoder
        self.decoder: dict = {} if decoder is None else decoder

    def _joined(self, component: str, layer: int) -> Optional[Tensor]:
        chunks = getattr(self, component).get(layer)
        return torch.cat(chunks, dim=0) if chunks else None

    def get_encoder_activations(self, layer: int) -> Optional[Tensor]:
        return self._joined("encoder", layer)

    def get_decoder_activations(self, layer: int) -> Optional[Tensor]:
        return self._joined("decoder", layer)

    def clear(self) -> None:
        for component in COMPONENTS:
            getattr(self, component).clear()

    def __eq__(self, other) -> bool:
        return isinstance(other, ActivationCache) and self.encoder == other.encoder and self.decoder == other.decoder

    def __repr__(self) -> str:
        return f"ActivationCache(encoder={self.encoder!r}, decoder={self.decoder!r})"


class _ListSink:
    """Append the activation to ``store[layer]``; on the host unless told to stay on the device."""

    def __init__(self, store: dict, layer: int, to_host: bool):
        self.store, self.layer, self.to_host = store, layer, to_host

    def take(self, hidden: Tensor, norm) -> None:
        activation = hidden if norm is None else norm(hidden)
        self.store.setdefault(self.layer, []).append(activation.cpu() if self.to_host else activation)


class _RingSink:
    """LayerNorm + push into an ``ActivationRing`` in one kernel; nothing is kept."""

    def __init__(self, ring):
        self.ring = ring

    def take(self, hidden: Tensor, norm) -> None:
        if norm is None:
            self.ring.push(hidden)
        else:
            self.ring.push_layernorm(hidden, norm.weight, norm.bias, norm.eps)


class _Tap:
    """The forward hook of one tapped layer: hidden states -> (final LayerNorm) -> sink."""

    def __init__(self, owner: "WhisperActivationExtractor", component: str, layer: int):
        self.owner, self.component, self.layer = owner, component, layer

    def __call__(self, module, inputs, output) -> None:
        # an encoder layer returns its hidden states alone or first in a tuple (hooks.py:79-84); for a decoder layer the
        # reference takes output[0] WHATEVER the layer returns (hooks.py:99-101) - with a transformers build whose decoder
        # layers return a bare tensor that is batch element 0, and golden set G14 pins exactly that
        hidden = output[0] if (self.component == "decoder" or isinstance(output, (tuple, list))) else output
        owner = self.owner
        norm = owner._final_norm[self.component] if owner.apply_layer_norm else None
        owner._sink(self.component, self.layer).take(hidden.detach(), norm)
        key = (self.component, self.layer)
        owner.rows_delivered[key] = owner.rows_delivered.get(key, 0) + hidden.numel() // hidden.shape[-1]


class WhisperActivationExtractor:
    """Capture encoder / decoder layer outputs of a Whisper model (reference hooks.py:40-144).

    ``keep_on_device`` (not in the reference): cached activations stay where they were produced instead of going to the
    host - for callers that consume them on the GPU right away.  The default is the reference's behaviour, which also bounds
    device memory on long extraction runs; layers with a ring attached are never cached at all.
    """

    def __init__(self, model, encoder_layers: Optional[list] = None, decoder_layers: Optional[list] = None,
                 apply_layer_norm: bool = True, keep_on_device: bool = False):
        self.model = model
        self.encoder_layers = encoder_layers or []
        self.decoder_layers = decoder_layers or []
        self.apply_layer_norm = apply_layer_norm
        self.keep_on_device = keep_on_device
        self.cache = ActivationCache()
        self._hooks: list = []
        self._rings: dict = {}
        self.rows_delivered: dict = {}  # (component, layer) -> activation rows handed to a sink so far
        self._final_norm = {"encoder": model.model.encoder.layer_norm, "decoder": model.model.decoder.layer_norm}

    # the reference exposes the two norms under these names (hooks.py:71-72)
    @property
    def _encoder_layer_norm(self):
        return self._final_norm["encoder"]

    @property
    def _decoder_layer_norm(self):
        return self._final_norm["decoder"]

    def taps(self) -> list:
        """``[(component, layer), ...]`` in registration order: encoder layers first (hooks.py:113-123)."""
        return [("encoder", i) for i in self.encoder_layers] + [("decoder", i) for i in self.decoder_layers]

    def attach_ring(self, component: Literal["encoder", "decoder"], layer: int, ring) -> None:
        """Send this layer's (layer-normed) activations into ``ring`` instead of the cache."""
        if component not in COMPONENTS:
            raise ValueError(f"component must be one of {COMPONENTS}, got {component!r}")
        self._rings[(component, layer)] = ring

    def _sink(self, component: str, layer: int):
        ring = self._rings.get((component, layer))
        if ring is not None:
            return _RingSink(ring)
        return _ListSink(getattr(self.cache, component), layer, to_host=not self.keep_on_device)

    def register_hooks(self) -> None:
        self.remove_hooks()
        for component, layer in self.taps():
            block = getattr(self.model.model, component).layers[layer]
            self._hooks.append(block.register_forward_hook(_Tap(self, component, layer)))

    def remove_hooks(self) -> None:
        while self._hooks:
            self._hooks.pop().remove()

    def clear_cache(self) -> None:
        self.cache.clear()

    def __enter__(self) -> "WhisperActivationExtractor":
        self.register_hooks()
        return self

    def __exit__(self, *exc) -> None:
        self.remove_hooks()


def run_whisper_taps(model, extractor: WhisperActivationExtractor, input_features: Tensor, device) -> int:
    """One batch of mel features through the encoder and - when decoder layers are tapped - one decoder step from the
    start token (what both reference drivers do: hooks.py:179-196, feature_cache.py:263-281).  The extractor's hooks must
    be registered.  Returns the batch size."""
    input_features = input_features.to(device)
    batch_size = input_features.size(0)
    encoder_hidden = model.model.encoder(input_features).last_hidden_state
    if extractor.decoder_layers:
        start = torch.full((batch_size, 1), model.config.decoder_start_token_id, dtype=torch.long, device=device)
        model.model.decoder(input_ids=start, encoder_hidden_states=encoder_hidden)
    return batch_size


def extract_features_batch(model, input_features: Tensor, encoder_layers: list, decoder_layers: list,
                           apply_layer_norm: bool = True, device="cpu", rings: Optional[dict] = None,
                           keep_on_device: bool = False) -> dict:
    """Activations of one batch, ``{"encoder": {layer: [B, T, D]}, "decoder": {layer: [B, 1, D]}}`` (reference
    hooks.py:147-210).

    ``rings``: optional ``{("encoder" | "decoder", layer): ActivationRing}``; those layers are pushed into their ring and do
    not appear in the returned dict.  ``keep_on_device``: see ``WhisperActivationExtractor``."""
    model.eval()
    extractor = WhisperActivationExtractor(model=model, encoder_layers=encoder_layers, decoder_layers=decoder_layers,
                                           apply_layer_norm=apply_layer_norm, keep_on_device=keep_on_device)
    for (component, layer), ring in (rings or {}).items():
        extractor.attach_ring(component, layer, ring)
    with torch.no_grad(), extractor:
        run_whisper_taps(model, extractor, input_features, device)
    results: dict = {component: {} for component in COMPONENTS}
    for component, layer in extractor.taps():
        joined = extractor.cache._joined(component, layer)
        if joined is not None:
            results[component][layer] = joined
    return results


def flatten_activations(activations: Tensor, component: Literal["encoder", "decoder"]) -> Tensor:
    """``[batch, seq_len, hidden] -> [batch * seq_len, hidden]`` for either component (reference hooks.py:213-230)."""
    if activations.ndim != 3:
        raise ValueError(f"expected [batch, seq_len, hidden], got {tuple(activations.shape)}")
    return activations.reshape(-1, activations.shape[-1])
