"""Activation feed: on-device ring buffer + the reference's on-disk cache format + the extraction driver."""

from .feature_cache import ActivationRing, CacheMetadata, FeatureCache, RingLoader, extract_and_cache_features

__all__ = ["ActivationRing", "CacheMetadata", "FeatureCache", "RingLoader", "extract_and_cache_features"]
