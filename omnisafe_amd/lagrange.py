"""Lagrange multiplier -- mirror of omnisafe/common/lagrange.py:27-136.

lambda is ONE scalar updated once per epoch (K7: negligible host work in the reference too).  The dual
ascent step is torch.optim's optimiser (default Adam) on loss -lambda * (Jc - cost_limit), clamped to
[0, upper bound]; the arithmetic is delegated to the very same torch CPU scalar ops the reference uses,
so the multiplier trajectory is bit-identical.  A device-resident copy feeds the surrogate-advantage
computation inside the actor kernel, so no per-minibatch ``.item()`` is needed."""
from __future__ import annotations

import torch


class Lagrange:
    def __init__(self, cost_limit: float, lagrangian_multiplier_init: float, lambda_lr: float,
                 lambda_optimizer: str = 'Adam', lagrangian_upper_bound: float | None = None,
                 device=None) -> None:
        self.cost_limit = cost_limit
        self.lambda_lr = lambda_lr
        self.lagrangian_upper_bound = lagrangian_upper_bound
        assert hasattr(torch.optim, lambda_optimizer), f'Optimizer={lambda_optimizer} not found in torch.'
        self._param = torch.nn.Parameter(torch.as_tensor(max(lagrangian_multiplier_init, 0.0)),
                                         requires_grad=True)
        self._opt = getattr(torch.optim, lambda_optimizer)([self._param], lr=lambda_lr)
        self._device_copy = None
        if device is not None:
            self._device_copy = torch.empty(1, dtype=torch.float32, device=device)
            self._device_copy.fill_(self.lagrangian_multiplier)

    @property
    def lagrangian_multiplier(self) -> float:
        return float(self._param.detach())

    @property
    def device_multiplier(self) -> torch.Tensor:
        """float32 device scalar read by osa_ppo_minibatch (surrogate (A_r - l A_c)/(1 + l))."""
        assert self._device_copy is not None, 'constructed without a device'
        return self._device_copy

    def compute_lambda_loss(self, mean_ep_cost: float) -> torch.Tensor:
        return -self._param * (mean_ep_cost - self.cost_limit)

    def update_lagrange_multiplier(self, Jc: float) -> None:
        # d/d(lambda) of -lambda * (Jc - limit) is -(Jc - limit) rounded to float32 -- exactly what autograd's
        # mul / neg backward produce for the python-scalar operand -- so the gradient is set directly: no autograd
        # graph, no backward pass on the host between the rollout's one synchronisation and the update's first launch
        # (half of the ~0.2 ms this step kept the device idle per epoch; same bits: tests/test_host_logic.py::
        # test_lagrange_vs_reference_golden, test_oracle_vs_reference.py::test_lagrange_and_pid_live)
        g = torch.tensor(-(Jc - self.cost_limit), dtype=torch.float32)
        if self._param.grad is None:
            self._param.grad = g
        else:
            self._param.grad.copy_(g)
        self._opt.step()
        self._param.data.clamp_(0.0, self.lagrangian_upper_bound)
        if self._device_copy is not None:
            self._device_copy.fill_(self.lagrangian_multiplier)
