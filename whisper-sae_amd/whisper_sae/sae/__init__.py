"""Sparse autoencoders, transcoders and their trainer (MI355X build)."""

from .crosscoder import CrosscoderOutput, CrossLayerCrosscoder, TopKCrossLayerCrosscoder, create_crosscoder
from .model import ReLUSAE, SAEOutput, TopKSAE, create_sae
from .training import RingBatch, SAETrainer, TrainingMetrics
from .transcoder import SkipTranscoder, TopKTranscoder, TranscoderOutput, create_transcoder

__all__ = ["ReLUSAE", "SAEOutput", "TopKSAE", "create_sae", "RingBatch", "SAETrainer", "TrainingMetrics",
           "SkipTranscoder", "TopKTranscoder", "TranscoderOutput", "create_transcoder", "CrosscoderOutput",
           "CrossLayerCrosscoder", "TopKCrossLayerCrosscoder", "create_crosscoder"]
