"""Host-side trainers mirroring experiments/fitting/trainers (the nef phase of the meta-learning trainer)."""
from .pde_trainer import MetaSGDPDETrainer, TrainState, meta_gradients

__all__ = ["MetaSGDPDETrainer", "TrainState", "meta_gradients"]
