"""Host-side callers of the decoder, mirroring experiments/fitting (the "harness" rows of SURVEY.md 8):

  get_model_pde(cfg)   experiments/fitting/__init__.py:14-65   (nef, ode_model)
  inner_loop(...)      trainers/pde_trainer.py:122-235         MAML inner loop: per-signal latent SGD
  decode(...)          trainers/pde_trainer.py:389-405         full-grid decode (chunking optional)
  shard_signals / allreduce_mean_   SURVEY.md 8e               meta-batch data parallelism over RCCL
  MetaSGDPDETrainer    trainers/pde_trainer.py:60-67,237-500   outer steps: nef (meta-gradient), ode, dual; val_step
  ode_models           ode_models/ponita_ode_g.py, mlp_ode.py  PonitaODEGen (fused SepGconv HIP kernels), MLPODE
  solve_latent_ode     trainers/trainer_utils/solvers.py:69-162 Euler / RK4 over the latent tuple
  NonMetaPDETrainer    trainers/nonmaml_pde_trainer.py:56-171  auto-decoder training step (first-order, exact)
"""
from .model import get_model_pde
from .inner_loop import inner_loop, decode, make_masks, default_meta_sgd_lrs
from .parallel import shard_range, allreduce_mean_, init_distributed
from .trainers import MetaSGDPDETrainer, TrainState, meta_gradients, NonMetaPDETrainer, NonMetaTrainState
from .trainers.trainer_utils import solve_latent_ode
from .ode_models import PonitaODEGen, MLPODE

__all__ = ["get_model_pde", "inner_loop", "decode", "make_masks", "default_meta_sgd_lrs", "shard_range",
           "allreduce_mean_", "init_distributed", "MetaSGDPDETrainer", "TrainState", "meta_gradients", "NonMetaPDETrainer", "NonMetaTrainState",
           "solve_latent_ode", "PonitaODEGen", "MLPODE"]
