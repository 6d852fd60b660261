"""Fixed-step solvers over the latent tuple (p, a, window), mirroring
experiments/fitting/trainers/trainer_utils/solvers.py (_euler_step_treemapped :69-83, _rk4_step_treemapped :86-105,
_solve_latent_ode :108-162).  Differentiable: the trajectory is a stack of the steps, not an in-place buffer."""
import torch


def _axpy(x, h, k):
    return tuple(None if xi is None else xi + h * ki for xi, ki in zip(x, k))


def euler_step(f, x, t, h):
    return _axpy(x, h, f(x, t))


def rk4_step(f, x, t, h):
    k1 = f(x, t)
    k2 = f(_axpy(x, 0.5 * h, k1), t + 0.5 * h)
    k3 = f(_axpy(x, 0.5 * h, k2), t + 0.5 * h)
    k4 = f(_axpy(x, h, k3), t + h)
    return tuple(None if xi is None else xi + (h / 6.0) * (a + 2 * b + 2 * c + d) for xi, a, b, c, d in zip(x, k1, k2, k3, k4))


def solve_latent_ode(f, latents, t0, tf, h, method="rk4", stop_gradient=False):
    """Returns (p_traj, a_traj, window_traj), each (batch, num_steps + 1, ...); num_steps = int((tf - t0) / h).
    ``stop_gradient``: every step starts from a detached state (solvers.py:141-150)."""
    if method not in ("rk4", "euler"):
        raise ValueError(f"Unknown method: {method}")
    num_steps = int((tf - t0) / h)
    step = rk4_step if method == "rk4" else euler_step
    traj, t = [tuple(latents)], t0
    for _ in range(num_steps):
        x = traj[-1]
        if stop_gradient:
            x = tuple(None if xi is None else xi.detach() for xi in x)
        traj.append(step(f, x, t, h))
        t += h
    return tuple(None if traj[0][i] is None else torch.stack([s[i] for s in traj], 1) for i in range(3))
