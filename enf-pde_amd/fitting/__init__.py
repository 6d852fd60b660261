"""Host-side callers of the decoder, mirroring experiments/fitting (the "harness" rows of SURVEY.md 8):

  get_model_pde(cfg)   experiments/fitting/__init__.py:14-65   (the nef half; ODE models are out of scope)
  inner_loop(...)      trainers/pde_trainer.py:122-235         MAML inner loop: per-signal latent SGD
  decode(...)          trainers/pde_trainer.py:389-405         full-grid decode (chunking optional)
  shard_signals / allreduce_mean_   SURVEY.md 8e               meta-batch data parallelism over RCCL
  MetaSGDPDETrainer    trainers/pde_trainer.py:60-67,237-288   outer (meta) step: meta-gradient + optax-rule optimisers
  NonMetaPDETrainer    trainers/nonmaml_pde_trainer.py:56-171  auto-decoder training step (first-order, exact)
"""
from .model import get_model_pde
from .inner_loop import inner_loop, decode, make_masks, default_meta_sgd_lrs
from .parallel import shard_range, allreduce_mean_, init_distributed
from .trainers import MetaSGDPDETrainer, TrainState, meta_gradients, NonMetaPDETrainer, NonMetaTrainState

__all__ = ["get_model_pde", "inner_loop", "decode", "make_masks", "default_meta_sgd_lrs", "shard_range",
           "allreduce_mean_", "init_distributed", "MetaSGDPDETrainer", "TrainState", "meta_gradients", "NonMetaPDETrainer", "NonMetaTrainState"]
