"""Host-side solver tables for the captured sampling loop.

All DPM-Solver++ / UniPC coefficients are data-independent scalars (SURVEY
fact 9), so they are computed ONCE per (betas, solver, steps) on the host in
float64 and handed to the engine as a flat float32 table; the device loop
indexes the table by step and contains no host synchronisation.

Reference math restated here:
  schedule      sampler/dpm_solver.py:100-154 (discrete VP, piecewise-linear log alpha)
  time grid     sampler/dpm_solver.py:474,1159-1160 (time_uniform), model time :278
  DPM-Solver++  sampler/dpm_solver.py:547-580 (1st order), :796-831 (2M), loop :1171-1213
  UniPC-bh2     sampler/uni_pc.py:471-567 (update), loop :606-658

Unified recurrence executed by the engine after the i-th denoiser evaluation
(i = 0 .. steps-1), all tensors elementwise, scalars from row i of the table:

    eps   = (xe - alpha*x0) / sigma ;  m = (xe - sigma*eps) / alpha      # x_start wrapper round trip
    x     = xbar - g0*d1 - g1*(m - m_prev)                               # (corrected) state at t_i
    xbar' = A*x - Bc*m                                                   # first-order part towards t_{i+1}
    d1'   = d1c*(m_prev - m)                                             # scaled backward difference
    xe'   = xbar' - pc*d1'                                               # point where the next evaluation happens
    m_prev' = m

with xbar = xe = x_T, d1 = m_prev = 0 before the first evaluation.  After the
last evaluation ``xe'`` is the sample.  For DPM-Solver++(2M): g0 = previous pc,
g1 = 0.  For UniPC: g0 = alpha_t*B_h*rho_0, g1 = alpha_t*B_h*rho_1, pc = alpha*B_h/2.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

# column layout of the per-step coefficient table (float32, NCOEF per row)
COEF_COLUMNS = ("t_model", "alpha", "sigma", "g0", "g1", "A", "Bc", "d1c", "pc", "_pad0", "_pad1", "_pad2")
NCOEF = len(COEF_COLUMNS)
SOLVERS = ("dpmsolver++", "unipc")


def linear_betas(n: int = 1000) -> np.ndarray:
    """reference model.py:426-433: linspace in float64, stored as a float32 buffer (:471-473)."""
    scale = 1000.0 / n
    return np.linspace(scale * 1e-4, scale * 0.02, n, dtype=np.float64).astype(np.float32)


class VPSchedule:
    """Discrete VP schedule in float64 (from the float32 betas the reference stores)."""

    def __init__(self, betas: np.ndarray):
        b32 = np.asarray(betas, dtype=np.float32)
        # the reference evaluates log(1-beta).cumsum() in float32; keep that so the
        # knots agree to float32 rounding, then continue in float64
        self.log_alpha = (0.5 * np.cumsum(np.log((1.0 - b32).astype(np.float32)).astype(np.float32), dtype=np.float32)).astype(np.float64)
        self.N = int(self.log_alpha.shape[0])
        self.knots = np.linspace(0.0, 1.0, self.N + 1, dtype=np.float32)[1:].astype(np.float64)

    def log_alpha_at(self, t: float) -> float:
        i = int(np.searchsorted(self.knots, t, side="left")) - 1
        i = min(max(i, 0), self.N - 2)
        x0, x1 = self.knots[i], self.knots[i + 1]
        y0, y1 = self.log_alpha[i], self.log_alpha[i + 1]
        return float(y0 + (t - x0) * (y1 - y0) / (x1 - x0))

    def alpha(self, t: float) -> float:
        return float(np.exp(self.log_alpha_at(t)))

    def sigma(self, t: float) -> float:
        return float(np.sqrt(1.0 - np.exp(2.0 * self.log_alpha_at(t))))

    def lam(self, t: float) -> float:
        la = self.log_alpha_at(t)
        return float(la - 0.5 * np.log(1.0 - np.exp(2.0 * la)))

    def timesteps(self, steps: int) -> np.ndarray:
        # the reference builds the grid with torch.linspace in float32
        return np.linspace(1.0, 1.0 / self.N, steps + 1, dtype=np.float32).astype(np.float64)

    def model_time(self, t: float) -> float:
        return float(np.float32((np.float32(t) - np.float32(1.0 / self.N)) * np.float32(self.N)))


@dataclass
class SolverTable:
    solver: str
    steps: int
    coef: np.ndarray          # (steps, NCOEF) float32
    timesteps: np.ndarray     # (steps+1,) continuous labels
    detail: Dict[str, np.ndarray]

    @property
    def t_model(self) -> np.ndarray:
        return self.coef[:, 0]


def build_table(solver: str, steps: int, betas: Optional[np.ndarray] = None, order: int = 2) -> SolverTable:
    """Coefficient table for ``steps`` denoiser evaluations (NFE == steps)."""
    if solver not in SOLVERS:
        raise ValueError(f"solver must be one of {SOLVERS}, got {solver!r}")
    if order not in (1, 2):
        raise ValueError("order must be 1 or 2")
    if steps < order:
        raise ValueError(f"steps ({steps}) must be >= order ({order})")   # dpm_solver.py:1172 / uni_pc.py:607
    sched = VPSchedule(linear_betas() if betas is None else betas)
    ts = sched.timesteps(steps)
    lam = np.array([sched.lam(t) for t in ts])
    alpha = np.array([sched.alpha(t) for t in ts])
    sigma = np.array([sched.sigma(t) for t in ts])
    coef = np.zeros((steps, NCOEF), dtype=np.float64)
    det = {k: np.zeros(steps + 1) for k in ("h", "rk", "B_h", "rho0", "rho1", "order")}

    def step_order(k: int) -> int:
        """order of the update that lands on ts[k], k = 1..steps"""
        if k < order:
            return k
        if solver == "unipc":
            return min(order, steps + 1 - k)                  # lower_order_final always, uni_pc.py:636-637
        return min(order, steps + 1 - k) if steps < 10 else order   # dpm_solver.py:1198-1201

    prev_pc = 0.0
    for i in range(steps):
        row = coef[i]
        row[0] = sched.model_time(ts[i])
        row[1], row[2] = alpha[i], sigma[i]
        # ---- correction of the state at t_i (uses the update that landed on ts[i])
        if i == 0:
            row[3] = row[4] = 0.0
        elif solver == "dpmsolver++":
            row[3], row[4] = prev_pc, 0.0
        else:
            k = i
            h = lam[k] - lam[k - 1]
            hh = -h
            h_phi_1 = np.expm1(hh)
            B_h = np.expm1(hh)
            ab = alpha[k] * B_h
            if step_order(k) == 1:
                rho0, rho1 = 0.0, 0.5                            # uni_pc.py:541-542
            else:
                rk = (lam[k - 2] - lam[k - 1]) / h
                h_phi_k = h_phi_1 / hh - 1.0
                b0 = h_phi_k / B_h
                b1 = (h_phi_k / hh - 0.5) * 2.0 / B_h
                rho0, rho1 = np.linalg.solve(np.array([[1.0, 1.0], [rk, 1.0]]), np.array([b0, b1]))   # :544
            row[3], row[4] = ab * rho0, ab * rho1
            det["rho0"][k], det["rho1"][k] = rho0, rho1
        # ---- move towards t_{i+1}
        k = i + 1
        h = lam[k] - lam[k - 1]
        phi = np.expm1(-h)                                       # = h_phi_1 = B_h (bh2) with hh = -h
        row[5] = sigma[k] / sigma[k - 1]
        row[6] = alpha[k] * phi
        det["h"][k], det["B_h"][k], det["order"][k] = h, phi, step_order(k)
        if step_order(k) == 2:
            rk = (lam[k - 2] - lam[k - 1]) / h                   # = -r0 of dpm_solver.py:822
            row[7] = 1.0 / rk
            row[8] = 0.5 * alpha[k] * phi
            det["rk"][k] = rk
        else:
            row[7] = row[8] = 0.0
        prev_pc = row[8]
    return SolverTable(solver, steps, coef.astype(np.float32), ts, det)


def run_table_numpy(table: SolverTable, x0_fn: Callable[[np.ndarray, np.ndarray], np.ndarray], x_T: np.ndarray,
                    trace: Optional[List[np.ndarray]] = None) -> np.ndarray:
    """Host executor of the unified recurrence (float32), used by CPU tests to pin
    the tables against the oracle samplers.  ``x0_fn(x, t_model[B]) -> x0``."""
    f = np.float32
    xbar = x_T.astype(f).copy()
    xe = xbar.copy()
    d1 = np.zeros_like(xbar)
    m_prev = np.zeros_like(xbar)
    B = x_T.shape[0]
    for i in range(table.steps):
        t_model, alpha, sigma, g0, g1, A, Bc, d1c, pc = (f(v) for v in table.coef[i, :9])
        x0 = x0_fn(xe, np.full((B,), t_model, dtype=f)).astype(f)
        eps = (xe - alpha * x0) / sigma
        m = (xe - sigma * eps) / alpha
        x = xbar - g0 * d1 - g1 * (m - m_prev)
        xbar = A * x - Bc * m
        d1 = d1c * (m_prev - m)
        xe = xbar - pc * d1
        m_prev = m
        if trace is not None:
            trace.append(xe.copy())
    return xe
