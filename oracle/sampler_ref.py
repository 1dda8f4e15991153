"""ORACLE (test infrastructure, never shipped, never the thing measured).

PyTorch-CPU fp32 restatement of the two ODE samplers NS2VC drives the denoiser
with, following the reference's order of floating-point operations:

* VP noise schedule with piecewise-linear log-alpha (``sampler/dpm_solver.py:6-167``,
  ``interpolate_fn`` ``:1253-1292``),
* the ``model_type="x_start"`` wrapper: x0 -> eps -> x0 round trip
  (``sampler/dpm_solver.py:271-292, 433-442``),
* DPM-Solver++(2M), multistep, time_uniform (``sampler/dpm_solver.py:1171-1213,
  547-580, 796-831``),
* UniPC-bh2 order 2, multistep (``sampler/uni_pc.py:606-658, 471-567``) — with the
  reference's batch>1 broadcast bug (``uni_pc.py:190-191``) fixed, i.e. equal to
  running the reference once per batch item.

Pinned against the imported reference by ``tests/golden/make_golden.py``.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline may import it.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch


def linear_betas(n: int = 1000) -> torch.Tensor:
    """model.py:426-433 + the float32 buffer cast at :471-473."""
    scale = 1000.0 / n
    return torch.linspace(scale * 1e-4, scale * 0.02, n, dtype=torch.float64).to(torch.float32)


class VPSchedule:
    def __init__(self, betas: torch.Tensor):
        betas = betas.to(torch.float32)
        self.log_alpha = 0.5 * torch.log(1 - betas).cumsum(dim=0)          # (N,)
        self.N = self.log_alpha.numel()
        self.knots = torch.linspace(0.0, 1.0, self.N + 1)[1:].to(torch.float32)

    def log_alpha_at(self, t: torch.Tensor) -> torch.Tensor:
        """Piecewise-linear interpolation with linear extrapolation outside the knots."""
        t = t.reshape(-1).to(torch.float32).contiguous()
        K = self.N
        # segment [i, i+1] with knots[i] <= t (ties resolve to the left segment,
        # like the reference's sort-based search); clamp to the outermost segments
        i = torch.searchsorted(self.knots, t, right=False) - 1
        i = i.clamp(0, K - 2)
        x0, x1 = self.knots[i], self.knots[i + 1]
        y0, y1 = self.log_alpha[i], self.log_alpha[i + 1]
        return y0 + (t - x0) * (y1 - y0) / (x1 - x0)

    def alpha(self, t):
        return torch.exp(self.log_alpha_at(t))

    def sigma(self, t):
        return torch.sqrt(1.0 - torch.exp(2.0 * self.log_alpha_at(t)))

    def lam(self, t):
        la = self.log_alpha_at(t)
        return la - 0.5 * torch.log(1.0 - torch.exp(2.0 * la))

    def timesteps(self, steps: int) -> torch.Tensor:
        return torch.linspace(1.0, 1.0 / self.N, steps + 1, dtype=torch.float32)

    def model_time(self, t: torch.Tensor) -> torch.Tensor:
        """Continuous label -> the UNet's (fractional) discrete timestep, dpm_solver.py:278."""
        return (t - 1.0 / self.N) * self.N


X0Fn = Callable[[torch.Tensor, torch.Tensor], torch.Tensor]      # (x, t_model[B]) -> x0_pred


def _data_prediction(sched: VPSchedule, x0_fn: X0Fn, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """x0 -> eps -> x0, as the reference wrapper + data_prediction_fn do."""
    B = x.shape[0]
    tv = t.reshape(-1).expand(B)
    x0 = x0_fn(x, sched.model_time(tv))
    a = sched.alpha(tv)[:, None, None]
    s = sched.sigma(tv)[:, None, None]
    eps = (x - a * x0) / s
    return (x - s * eps) / a


@torch.no_grad()
def dpm_solver_pp_2m(x0_fn: X0Fn, betas: torch.Tensor, x: torch.Tensor, steps: int, order: int = 2,
                     trace: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    sched = VPSchedule(betas)
    assert steps >= order
    ts = sched.timesteps(steps)
    t_prev = [ts[0:1]]
    m_prev = [_data_prediction(sched, x0_fn, x, ts[0:1])]

    def first(x, s, t, m0):
        lam_s, lam_t = sched.lam(s), sched.lam(t)
        h = lam_t - lam_s
        phi = torch.expm1(-h)
        return sched.sigma(t) / sched.sigma(s) * x - sched.alpha(t) * phi * m0

    def second(x, t):
        m1, m0 = m_prev[-2], m_prev[-1]
        t1, t0 = t_prev[-2], t_prev[-1]
        lam1, lam0, lam_t = sched.lam(t1), sched.lam(t0), sched.lam(t)
        h0 = lam0 - lam1
        h = lam_t - lam0
        r0 = h0 / h
        D1 = (1.0 / r0) * (m0 - m1)
        phi = torch.expm1(-h)
        a_t = sched.alpha(t)
        return sched.sigma(t) / sched.sigma(t0) * x - a_t * phi * m0 - 0.5 * (a_t * phi) * D1

    for step in range(1, order):
        t = ts[step:step + 1]
        x = first(x, t_prev[-1], t, m_prev[-1])
        if trace is not None:
            trace.append(x.clone())
        t_prev.append(t)
        m_prev.append(_data_prediction(sched, x0_fn, x, t))
    for step in range(order, steps + 1):
        t = ts[step:step + 1]
        step_order = min(order, steps + 1 - step) if steps < 10 else order
        if step_order == 1:
            x = first(x, t_prev[-1], t, m_prev[-1])
        else:
            x = second(x, t)
        if trace is not None:
            trace.append(x.clone())
        for i in range(order - 1):
            t_prev[i], m_prev[i] = t_prev[i + 1], m_prev[i + 1]
        t_prev[-1] = t
        if step < steps:
            m_prev[-1] = _data_prediction(sched, x0_fn, x, t)
    return x


@torch.no_grad()
def unipc_bh2(x0_fn: X0Fn, betas: torch.Tensor, x: torch.Tensor, steps: int, order: int = 2,
              trace: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    assert order == 2, "NS2VC uses order 2 (model.py:683)"
    sched = VPSchedule(betas)
    assert steps >= order
    ts = sched.timesteps(steps)

    def update(x, m_list, t_list, t, ord_, use_corrector):
        t0 = t_list[-1]
        m0 = m_list[-1]
        lam0, lam_t = sched.lam(t0), sched.lam(t)
        a_t = sched.alpha(t)
        h = lam_t - lam0
        hh = -h
        h_phi_1 = torch.expm1(hh)
        B_h = torch.expm1(hh)
        D1 = None
        rk = None
        if ord_ == 2:
            rk = (sched.lam(t_list[-2]) - lam0) / h
            D1 = (m_list[-2] - m0) / rk
        x_bar = sched.sigma(t) / sched.sigma(t0) * x - a_t * h_phi_1 * m0
        x_t = x_bar if D1 is None else x_bar - a_t * B_h * (0.5 * D1)
        m_t = None
        if use_corrector:
            m_t = _data_prediction(sched, x0_fn, x_t, t)
            if ord_ == 1:
                x_t = x_bar - a_t * B_h * (0.5 * (m_t - m0))
            else:
                # R = [[1, 1], [rk, 1]], b from uni_pc.py:504-520
                h_phi_k = h_phi_1 / hh - 1.0
                b0 = h_phi_k * 1.0 / B_h
                h_phi_k2 = h_phi_k / hh - 1.0 / 2.0
                b1 = h_phi_k2 * 2.0 / B_h
                one = torch.ones((), dtype=torch.float32)
                R = torch.stack([torch.stack([one, one]), torch.stack([rk.reshape(()), one])])
                rho = torch.linalg.solve(R, torch.stack([b0.reshape(()), b1.reshape(())]))
                x_t = x_bar - a_t * B_h * (rho[0] * D1 + rho[1] * (m_t - m0))
        return x_t, m_t

    t_list = [ts[0:1]]
    m_list = [_data_prediction(sched, x0_fn, x, ts[0:1])]
    for step in range(1, order):
        t = ts[step:step + 1]
        x, m = update(x, m_list, t_list, t, step, True)
        if trace is not None:
            trace.append(x.clone())
        t_list.append(t)
        m_list.append(m)
    for step in range(order, steps + 1):
        t = ts[step:step + 1]
        step_order = min(order, steps + 1 - step)
        x, m = update(x, m_list, t_list, t, step_order, step != steps)
        if trace is not None:
            trace.append(x.clone())
        for i in range(order - 1):
            t_list[i], m_list[i] = t_list[i + 1], m_list[i + 1]
        t_list[-1] = t
        if step < steps:
            m_list[-1] = m
    return x
