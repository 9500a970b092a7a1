"""Whisper activation extraction with forward hooks, feeding the on-device ring.

Drop-in for the reference's ``whisper_sae.sae.hooks`` (/root/reference/src/whisper_sae/sae/hooks.py:15-230):
``ActivationCache``, ``WhisperActivationExtractor``, ``extract_features_batch``, ``flatten_activations`` with the same
signatures and the same tensors.  Two things differ, both about where the bytes go:

* cached activations stay on the device they were produced on (the reference moves every hooked output to the host,
  hooks.py:92, :106, to be concatenated, written to disk and loaded again by the trainer);
* a hook can have a **ring** attached (``attach_ring``): then the layer output goes through the final LayerNorm and
  into the ``ActivationRing`` the trainer samples from in ONE kernel (``wsae_ring_push_layernorm``), and nothing is
  cached at all - extraction and SAE training share the GPU with no host round trip.

The model is whatever ``transformers`` provides (``WhisperForConditionalGeneration``); this module only hooks it.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Literal, Optional

import torch
from torch import Tensor, nn


@dataclass
class ActivationCache:
    """Activations of several layers (reference hooks.py:15-37)."""

    encoder: dict = field(default_factory=dict)
    decoder: dict = field(default_factory=dict)

    def clear(self) -> None:
        self.encoder.clear()
        self.decoder.clear()

    def get_encoder_activations(self, layer: int) -> Optional[Tensor]:
        if layer not in self.encoder or not self.encoder[layer]:
            return None
        return torch.cat(self.encoder[layer], dim=0)

    def get_decoder_activations(self, layer: int) -> Optional[Tensor]:
        if layer not in self.decoder or not self.decoder[layer]:
            return None
        return torch.cat(self.decoder[layer], dim=0)


class WhisperActivationExtractor:
    """Capture encoder / decoder layer outputs of a Whisper model (reference hooks.py:40-144)."""

    def __init__(self, model, encoder_layers: Optional[list] = None, decoder_layers: Optional[list] = None,
                 apply_layer_norm: bool = True):
        self.model = model
        self.encoder_layers = encoder_layers or []
        self.decoder_layers = decoder_layers or []
        self.apply_layer_norm = apply_layer_norm
        self.cache = ActivationCache()
        self._hooks: list = []
        self._rings: dict = {}
        self._encoder_layer_norm = model.model.encoder.layer_norm
        self._decoder_layer_norm = model.model.decoder.layer_norm

    def attach_ring(self, component: Literal["encoder", "decoder"], layer: int, ring) -> None:
        """Send this layer's (layer-normed) activations into ``ring`` instead of the cache."""
        self._rings[(component, layer)] = ring

    def _deliver(self, component: str, layer_idx: int, hidden: Tensor, norm: nn.Module, store: dict) -> None:
        ring = self._rings.get((component, layer_idx))
        if ring is not None:
            if self.apply_layer_norm:
                ring.push_layernorm(hidden, norm.weight, norm.bias, norm.eps)
            else:
                ring.push(hidden)
            return
        activation = hidden
        if self.apply_layer_norm:
            activation = norm(activation)
        store.setdefault(layer_idx, []).append(activation)  # stays on its device

    def _make_encoder_hook(self, layer_idx: int) -> Callable:
        def hook(module: nn.Module, input: tuple, output) -> None:
            # encoder layers return (hidden_states, attention_weights) or the hidden states alone (hooks.py:79-84)
            hidden_states = output[0] if isinstance(output, tuple) else output
            self._deliver("encoder", layer_idx, hidden_states.detach(), self._encoder_layer_norm, self.cache.encoder)

        return hook

    def _make_decoder_hook(self, layer_idx: int) -> Callable:
        def hook(module: nn.Module, input: tuple, output) -> None:
            # hooks.py:99-101 takes output[0] whatever the layer returns
            self._deliver("decoder", layer_idx, output[0].detach(), self._decoder_layer_norm, self.cache.decoder)

        return hook

    def register_hooks(self) -> None:
        self.remove_hooks()
        for layer_idx in self.encoder_layers:
            layer = self.model.model.encoder.layers[layer_idx]
            self._hooks.append(layer.register_forward_hook(self._make_encoder_hook(layer_idx)))
        for layer_idx in self.decoder_layers:
            layer = self.model.model.decoder.layers[layer_idx]
            self._hooks.append(layer.register_forward_hook(self._make_decoder_hook(layer_idx)))

    def remove_hooks(self) -> None:
        for hook in self._hooks:
            hook.remove()
        self._hooks.clear()

    def clear_cache(self) -> None:
        self.cache.clear()

    def __enter__(self) -> "WhisperActivationExtractor":
        self.register_hooks()
        return self

    def __exit__(self, *args) -> None:
        self.remove_hooks()


def extract_features_batch(model, input_features: Tensor, encoder_layers: list, decoder_layers: list,
                           apply_layer_norm: bool = True, device="cpu", rings: Optional[dict] = None) -> dict:
    """One batch through the encoder (and one decoder step from the start token), reference hooks.py:147-210.

    ``rings``: optional ``{("encoder" | "decoder", layer): ActivationRing}``; those layers are pushed into their
    ring and do not appear in the returned dict."""
    model.eval()
    input_features = input_features.to(device)
    extractor = WhisperActivationExtractor(model=model, encoder_layers=encoder_layers, decoder_layers=decoder_layers,
                                           apply_layer_norm=apply_layer_norm)
    for (component, layer), ring in (rings or {}).items():
        extractor.attach_ring(component, layer, ring)
    with torch.no_grad(), extractor:
        encoder_outputs = model.model.encoder(input_features)
        encoder_hidden = encoder_outputs.last_hidden_state
        if decoder_layers:
            batch_size = input_features.size(0)
            decoder_input_ids = torch.full((batch_size, 1), model.config.decoder_start_token_id, dtype=torch.long,
                                           device=device)
            _ = model.model.decoder(input_ids=decoder_input_ids, encoder_hidden_states=encoder_hidden)
    results: dict = {"encoder": {}, "decoder": {}}
    for layer_idx in encoder_layers:
        activations = extractor.cache.get_encoder_activations(layer_idx)
        if activations is not None:
            results["encoder"][layer_idx] = activations
    for layer_idx in decoder_layers:
        activations = extractor.cache.get_decoder_activations(layer_idx)
        if activations is not None:
            results["decoder"][layer_idx] = activations
    return results


def flatten_activations(activations: Tensor, component: Literal["encoder", "decoder"]) -> Tensor:
    """``[batch, seq_len, hidden] -> [batch * seq_len, hidden]`` (reference hooks.py:213-230)."""
    batch_size, seq_len, hidden_dim = activations.shape
    return activations.view(-1, hidden_dim)
