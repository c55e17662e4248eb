"""Closed-form adjoints of the two differentiable models (the arithmetic of the HIP backward sweep) against
torch autograd through the oracle's restatement of the reference models."""
import numpy as np
import torch

from oracle import mpg_oracle as O


def test_path_tracking_model_step_vjp_matches_autograd():
    rng = np.random.Generator(np.random.PCG64(0))
    N = 257
    obs = np.stack([rng.uniform(-19.5, 15.2, N), rng.normal(0, 2, N), rng.normal(0, .5, N), rng.normal(0, 1, N),
                    rng.normal(0, .8, N), rng.uniform(0, 1200, N)], 1)
    obs[0, 0], obs[1, 0] = 14.99, -18.99          # next v_x beyond the [1, 35] clip in some rows
    a = rng.uniform(-1, 1, (N, 2))
    lam, rho = rng.standard_normal((N, 6)), rng.standard_normal(N)
    eps = rng.standard_normal(N)
    ot = torch.tensor(obs, dtype=torch.float64, requires_grad=True)
    at = torch.tensor(a, dtype=torch.float64, requires_grad=True)
    m = O.PathTrackingModelOracle()
    m.reset(ot)
    o2, rew = m.rollout_out(at, torch.tensor(eps))
    # adjoint is taken w.r.t. the veh state; obs differs from it by a constant shift of v_x
    loss = (o2 * torch.tensor(lam)).sum() + (rew * torch.tensor(rho)).sum()
    g_o, g_a = torch.autograd.grad(loss, [ot, at])
    s = obs.copy()
    s[:, 0] += 20.0
    gs, ga = O.pt_model_step_vjp(s, a, lam, rho)
    np.testing.assert_allclose(gs, g_o.numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ga, g_a.numpy(), rtol=1e-9, atol=1e-9)


def test_pendulum_model_step_vjp_matches_autograd():
    rng = np.random.Generator(np.random.PCG64(1))
    N = 129
    s = rng.standard_normal((N, 4)) * np.array([0.5, 0.3, 0.5, 0.8])
    a = rng.uniform(-3, 3, (N, 1))
    lam_new, rho = rng.standard_normal((N, 4)), rng.standard_normal(N)
    eps = rng.standard_normal(N)
    st = torch.tensor(s, requires_grad=True)
    at = torch.tensor(a, requires_grad=True)
    m = O.InvertedPendulumModelOracle()
    m.reset(st)
    s2, rew = m.rollout_out(at, torch.tensor(eps))
    loss = (s2 * torch.tensor(lam_new)).sum() + (rew * torch.tensor(rho)).sum()
    g_s, g_a = torch.autograd.grad(loss, [st, at])
    # fold the reward (taken on the new noisy state) into the adjoint of the new state
    s2n = s2.detach().numpy()
    lam = lam_new + rho[:, None] * np.stack([-0.02 * s2n[:, 0], -2 * s2n[:, 1], -2e-3 * s2n[:, 2], -2e-3 * s2n[:, 3]], 1)
    gs, ga = O.pd_model_step_vjp(s, a, lam, rho)
    np.testing.assert_allclose(gs, g_s.numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ga, g_a.numpy(), rtol=1e-9, atol=1e-9)
