"""PID Lagrange multiplier -- mirror of omnisafe/common/pid_lagrange.py:26-146.

One scalar controller updated once per epoch from the mean episode cost (host float arithmetic, exactly
the reference's sequence of operations); a float32 device copy feeds the surrogate-advantage computation
inside the actor kernels."""
from __future__ import annotations

from collections import deque

import torch


class PIDLagrangian:  # pylint: disable=too-many-instance-attributes
    def __init__(self, pid_kp: float, pid_ki: float, pid_kd: float, pid_d_delay: int,
                 pid_delta_p_ema_alpha: float, pid_delta_d_ema_alpha: float, sum_norm: bool,
                 diff_norm: bool, penalty_max: int, lagrangian_multiplier_init: float,
                 cost_limit: float, device=None) -> None:
        self._pid_kp, self._pid_ki, self._pid_kd = pid_kp, pid_ki, pid_kd
        self._pid_d_delay = pid_d_delay
        self._pid_delta_p_ema_alpha = pid_delta_p_ema_alpha
        self._pid_delta_d_ema_alpha = pid_delta_d_ema_alpha
        self._penalty_max = penalty_max
        self._sum_norm, self._diff_norm = sum_norm, diff_norm
        self._pid_i: float = lagrangian_multiplier_init
        self._cost_ds: deque = deque(maxlen=self._pid_d_delay)
        self._cost_ds.append(0.0)
        self._delta_p: float = 0.0
        self._cost_d: float = 0.0
        self._cost_limit: float = cost_limit
        self._cost_penalty: float = 0.0
        self._device_copy = None
        if device is not None:
            self._device_copy = torch.zeros(1, dtype=torch.float32, device=device)

    @property
    def lagrangian_multiplier(self) -> float:
        return self._cost_penalty

    @property
    def device_multiplier(self) -> torch.Tensor:
        assert self._device_copy is not None, 'constructed without a device'
        return self._device_copy

    def pid_update(self, ep_cost_avg: float) -> None:
        """pid_lagrange.py:101-146."""
        delta = float(ep_cost_avg - self._cost_limit)
        self._pid_i = max(0.0, self._pid_i + delta * self._pid_ki)
        if self._diff_norm:
            self._pid_i = max(0.0, min(1.0, self._pid_i))
        a_p = self._pid_delta_p_ema_alpha
        self._delta_p *= a_p
        self._delta_p += (1 - a_p) * delta
        a_d = self._pid_delta_d_ema_alpha
        self._cost_d *= a_d
        self._cost_d += (1 - a_d) * float(ep_cost_avg)
        pid_d = max(0.0, self._cost_d - self._cost_ds[0])
        pid_o = self._pid_kp * self._delta_p + self._pid_i + self._pid_kd * pid_d
        self._cost_penalty = max(0.0, pid_o)
        if self._diff_norm:
            self._cost_penalty = min(1.0, self._cost_penalty)
        if not (self._diff_norm or self._sum_norm):
            self._cost_penalty = min(self._cost_penalty, self._penalty_max)
        self._cost_ds.append(self._cost_d)
        if self._device_copy is not None:
            self._device_copy.fill_(self._cost_penalty)
