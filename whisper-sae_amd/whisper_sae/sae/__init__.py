"""Sparse autoencoders and their trainer (MI355X build)."""

from .model import ReLUSAE, SAEOutput, TopKSAE, create_sae
from .training import RingBatch, SAETrainer, TrainingMetrics

__all__ = ["ReLUSAE", "SAEOutput", "TopKSAE", "create_sae", "RingBatch", "SAETrainer", "TrainingMetrics"]
