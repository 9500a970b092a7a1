"""Activation feed: on-device ring buffer + the reference's on-disk cache format."""

from .feature_cache import ActivationRing, CacheMetadata, FeatureCache, RingLoader

__all__ = ["ActivationRing", "CacheMetadata", "FeatureCache", "RingLoader"]
