from .solvers import solve_latent_ode, euler_step, rk4_step

__all__ = ["solve_latent_ode", "euler_step", "rk4_step"]
