"""PID-controlled Lagrange multiplier for CPPOPID / TRPOPID (replaces omnisafe/common/pid_lagrange.py:26-146).

The controller is a pure function of (gains, state, measured episode cost): `pid_step` maps one state to the
next, once per epoch on the host.  It evaluates, in double precision and in this order (the order fixes the
rounding, and tests/test_oracle_vs_reference.py compares trajectories with the reference to the last bit):

    e      = cost - limit
    I'     = clip0(I + e * ki)                         (clipped to [0, 1] as well when diff_norm)
    P'     = P * alpha_p + (1 - alpha_p) * e           exponential moving average of the error
    D'     = D * alpha_d + (1 - alpha_d) * cost        exponential moving average of the cost
    d      = max(0, D' - D_delayed)                    rise of the smoothed cost over `d_delay` epochs
    lambda = clip(kp * P' + I' + kd * d)               to [0, 1] (diff_norm), [0, inf) (sum_norm) or
                                                       [0, penalty_max] (neither)

`PIDLagrangian` keeps the reference's constructor keywords (they arrive as `**lagrange_cfgs` from the YAML
files) and its two public members, `pid_update(ep_cost_avg)` and `lagrangian_multiplier`; a float32 device
copy feeds the surrogate-advantage computation inside the actor kernels.
"""
from __future__ import annotations

from dataclasses import dataclass, replace

import torch


@dataclass(frozen=True)
class PidGains:
    kp: float
    ki: float
    kd: float
    alpha_p: float       # smoothing of the proportional error
    alpha_d: float       # smoothing of the cost fed to the derivative term
    limit: float         # cost limit the controller regulates to
    ceiling: float       # penalty_max (used only when neither normalisation is on)
    unit_interval: bool  # diff_norm: integral and output live in [0, 1]
    unbounded: bool      # sum_norm: no upper bound on the output


@dataclass(frozen=True)
class PidState:
    integral: float
    err_ema: float
    cost_ema: float
    history: tuple[float, ...]  # smoothed costs of the last `d_delay` epochs, oldest first
    output: float


def pid_step(k: PidGains, s: PidState, cost: float, depth: int) -> PidState:
    """One controller update (see the module docstring for the formula and its evaluation order)."""
    err = float(cost - k.limit)
    integral = max(0.0, s.integral + err * k.ki)
    if k.unit_interval:
        integral = max(0.0, min(1.0, integral))
    err_ema = s.err_ema * k.alpha_p + (1 - k.alpha_p) * err
    cost_ema = s.cost_ema * k.alpha_d + (1 - k.alpha_d) * float(cost)
    rise = max(0.0, cost_ema - s.history[0])
    out = max(0.0, k.kp * err_ema + integral + k.kd * rise)
    if k.unit_interval:
        out = min(1.0, out)
    elif not k.unbounded:
        out = min(out, k.ceiling)
    history = (s.history + (cost_ema,))[-depth:]
    return replace(s, integral=integral, err_ema=err_ema, cost_ema=cost_ema, history=history, output=out)


class PIDLagrangian:
    def __init__(self, pid_kp: float, pid_ki: float, pid_kd: float, pid_d_delay: int,
                 pid_delta_p_ema_alpha: float, pid_delta_d_ema_alpha: float, sum_norm: bool,
                 diff_norm: bool, penalty_max: int, lagrangian_multiplier_init: float,
                 cost_limit: float, device=None) -> None:
        self.gains = PidGains(kp=pid_kp, ki=pid_ki, kd=pid_kd, alpha_p=pid_delta_p_ema_alpha,
                              alpha_d=pid_delta_d_ema_alpha, limit=cost_limit, ceiling=penalty_max,
                              unit_interval=bool(diff_norm), unbounded=bool(sum_norm))
        self._depth = int(pid_d_delay)
        self.state = PidState(integral=lagrangian_multiplier_init, err_ema=0.0, cost_ema=0.0,
                              history=(0.0,), output=0.0)
        self._device_copy = (torch.zeros(1, dtype=torch.float32, device=device)
                             if device is not None else None)

    @property
    def lagrangian_multiplier(self) -> float:
        return self.state.output

    @property
    def device_multiplier(self) -> torch.Tensor:
        assert self._device_copy is not None, 'constructed without a device'
        return self._device_copy

    def pid_update(self, ep_cost_avg: float) -> None:
        self.state = pid_step(self.gains, self.state, ep_cost_avg, self._depth)
        if self._device_copy is not None:
            self._device_copy.fill_(self.state.output)
