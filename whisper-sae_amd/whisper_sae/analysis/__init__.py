"""Feature analysis right after the SAE path: per-feature top activations kept on the
device.  Mirrors the names of the reference's ``whisper_sae.analysis.feature_viz`` that sit on that path."""

from .feature_viz import FeatureActivation, TopKTracker, collect_top_activations

__all__ = ["FeatureActivation", "TopKTracker", "collect_top_activations"]
