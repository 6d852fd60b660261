"""Host-side trainers mirroring experiments/fitting/trainers (the nef phase of the meta-learning trainer)."""
from .pde_trainer import MetaSGDPDETrainer, TrainState, meta_gradients
from .nonmaml_pde_trainer import NonMetaPDETrainer, NonMetaTrainState

__all__ = ["MetaSGDPDETrainer", "TrainState", "meta_gradients", "NonMetaPDETrainer", "NonMetaTrainState"]
