"""Sibling on-policy algorithms that only override the surrogate / multiplier logic.

Each class mirrors its reference counterpart's hooks and logged keys; the per-sample arithmetic is the
same set of libomnisafe_amd kernels as PPOLag / TRPOLag / CPO -- the combined advantage
(A_r - p A_c) / (1 + p) is formed inside the actor kernels from a device scalar p:

  PDO       naive_lagrange/pdo.py:25-100        PolicyGradient + Lagrange
  RCPO      naive_lagrange/rcpo.py:25-103       NaturalPG + Lagrange
  IPO       penalty_function/ipo.py:24-74       PPO + interior-point penalty kappa / (limit - Jc)
  OnCRPO    primal/crpo.py:25-80                TRPO on A_r, or on -A_c while the cost exceeds the limit
  CPPOPID   pid_lagrange/cppo_pid.py:25-103     PPO + PID-controlled multiplier
  TRPOPID   pid_lagrange/trpo_pid.py:25-98      TRPO + PID-controlled multiplier
  PCPO      second_order/pcpo.py:31-152         CPO's machinery with the projection step
  FOCOPS    first_order/focops.py:31-230        PolicyGradient + Lagrange, KL-regularised masked surrogate
  CUP       first_order/cup.py:30-200           PPO step, then a KL-regularised cost-projection stage
  P3O       penalty_function/p3o.py:27-132      PPO + exact penalty kappa * relu(cost surrogate + Jc - limit)
(the last three use osa_ppo_minibatch_ext: the per-step kernels with the extended actor loss)

  PPOSaute / TRPOSaute          saute/{ppo,trpo}_saute.py        PPO / TRPO behind the SauteAdapter
  PPOSimmerPID / TRPOSimmerPID  simmer/{ppo,trpo}_simmer_pid.py  ... behind the SimmerAdapter (PID budget)
"""
from __future__ import annotations

import numpy as np
import torch

from .. import distributed as dist
from ..adapter import EarlyTerminatedAdapter, SauteAdapter, SimmerAdapter
from ..lagrange import Lagrange
from ..models import SurrogateExt
from ..pid_lagrange import PIDLagrangian
from ..update import PPOUpdater
from .policy_gradient import PPO, PolicyGradient
from .registry import register
from .trust_region_algos import CPO, TRPO, NaturalPG


def _cfg_dict(c) -> dict:
    return c.todict() if hasattr(c, 'todict') else dict(c)


class _LagrangeMixin:
    """_init / _init_log / _update of the naive-Lagrange family (pdo.py:36-80, rcpo.py:36-80)."""
    _lagrange_min_max = True

    def _init(self) -> None:
        super()._init()
        self._lagrange = Lagrange(**_cfg_dict(self._cfgs.lagrange_cfgs), device=self._device)

    def _init_log(self) -> None:
        super()._init_log()
        self._logger.register_key('Metrics/LagrangeMultiplier', min_and_max=self._lagrange_min_max)

    def _lagrange_tensor(self) -> torch.Tensor:
        return self._lagrange.device_multiplier

    def _update(self) -> None:
        Jc = self._logger.get_stats('Metrics/EpCost')[0]
        self._lagrange.update_lagrange_multiplier(Jc)
        super()._update()
        self._logger.store({'Metrics/LagrangeMultiplier': self._lagrange.lagrangian_multiplier})


class _PIDMixin:
    """cppo_pid.py:36-82 / trpo_pid.py:36-78."""

    def _init(self) -> None:
        super()._init()
        self._lagrange = PIDLagrangian(**_cfg_dict(self._cfgs.lagrange_cfgs), device=self._device)

    def _init_log(self) -> None:
        super()._init_log()
        self._logger.register_key('Metrics/LagrangeMultiplier')

    def _lagrange_tensor(self) -> torch.Tensor:
        return self._lagrange.device_multiplier

    def _update(self) -> None:
        Jc = self._logger.get_stats('Metrics/EpCost')[0]
        self._lagrange.pid_update(Jc)
        super()._update()
        self._logger.store({'Metrics/LagrangeMultiplier': self._lagrange.lagrangian_multiplier})


@register
class PDO(_LagrangeMixin, PolicyGradient):
    pass


@register
class RCPO(_LagrangeMixin, NaturalPG):
    pass


@register
class CPPOPID(_PIDMixin, PPO):
    pass


@register
class TRPOPID(_PIDMixin, TRPO):
    pass


@register
class IPO(PPO):
    def _init(self) -> None:
        super()._init()
        self._penalty = torch.zeros(1, dtype=torch.float32, device=self._device)

    def _init_log(self) -> None:
        super()._init_log()
        self._logger.register_key('Misc/Penalty')

    def _lagrange_tensor(self) -> torch.Tensor:
        return self._penalty

    def _update(self) -> None:
        """ipo.py:44-74: the penalty is a function of the epoch's mean episode cost only (the reference
        recomputes the same value in every minibatch)."""
        a = self._cfgs.algo_cfgs
        Jc = self._logger.get_stats('Metrics/EpCost')[0]
        penalty = a.kappa / (a.cost_limit - Jc + 1e-8)
        if penalty < 0 or penalty > a.penalty_max:
            penalty = a.penalty_max
        self._penalty.fill_(float(penalty))
        super()._update()
        self._logger.store({'Misc/Penalty': float(penalty)})


@register
class OnCRPO(TRPO):
    def __init__(self, env_id: str, cfgs) -> None:
        super().__init__(env_id, cfgs)
        self._rew_update = 0
        self._cost_update = 0

    def _init_log(self) -> None:
        super()._init_log()
        self._logger.register_key('Misc/RewUpdate')
        self._logger.register_key('Misc/CostUpdate')

    def _update_actor(self, data: dict) -> None:
        """crpo.py:58-80: optimise the reward advantage while the cost is within limit + distance,
        otherwise minimise the cost advantage (surrogate advantage -A_c)."""
        a = self._cfgs.algo_cfgs
        Jc = self._logger.get_stats('Metrics/EpCost')[0]
        if Jc <= a.cost_limit + a.distance:
            self._rew_update += 1
            self._adv_key_r = 'adv_r'
        else:
            self._cost_update += 1
            self._logger.store({'Misc/RewUpdate': self._rew_update, 'Misc/CostUpdate': self._cost_update})
            data = dict(data)
            data['neg_adv_c'] = -data['adv_c']
            self._adv_key_r = 'neg_adv_c'
        super()._update_actor(data)


@register
class PCPO(CPO):
    def _update_actor(self, data: dict) -> None:
        """pcpo.py:41-152: reward step sqrt(2 delta / q) F x followed by the projection onto the cost
        constraint along p = F^-1 b."""
        ac, s, a = self._actor_critic, self._solver, self._cfgs.algo_cfgs
        theta_old = ac.params[0].clone()
        s.begin(data['obs'])
        loss_r, grad_r = s.actor_loss_grad(data, 'adv_r', 'adv_c', self._lambda_zero)
        loss_reward_before = float(loss_r)
        g = s.lincomb(-1.0, grad_r)
        x = s.conjugate_gradients(g)
        assert torch.isfinite(x).all(), 'x is not finite'
        H_inv_g = s.fvp(x)  # (sic) pcpo.py:78: named H_inv_g, computed as F x
        s.fvp_calls += 1  # the reference evaluates (and logs the KL of) F x twice: pcpo.py:79 and :80
        xHx = float(s.dot(x, H_inv_g))
        assert xHx >= 0, 'xHx is negative'
        f = np.float32
        alpha = float(np.sqrt(f(2 * a.target_kl) / (f(xHx) + f(1e-8))))
        loss_c, grad_c = s.actor_loss_grad(data, 'adv_c', 'adv_c', self._lambda_zero)
        loss_cost_before = -float(loss_c)
        b = s.lincomb(-1.0, grad_c)
        ep_costs = float(self._logger.get_stats('Metrics/EpCost')[0] - a.cost_limit)
        p = s.conjugate_gradients(b)
        q = xHx
        r = float(s.dot(g, p))
        sc = float(s.dot(b, p))
        c1 = np.sqrt(f(2 * a.target_kl) / (f(q) + f(1e-8)))
        c2 = max((np.sqrt(f(2 * a.target_kl) / f(q)) * f(r) + f(ep_costs)) / f(sc), f(0.0))
        step_direction = s.lincomb(float(c1), H_inv_g, -float(c2), p)
        before = self._eval_at(data, theta_old, torch.zeros_like(step_direction), 'adv_r', self._lambda_zero)
        step, accept_step = self._cpo_search_step(data, theta_old, step_direction, g, loss_reward_before,
                                                  loss_cost_before, total_steps=200, violation_c=ep_costs,
                                                  optim_case=0)
        final = s.evaluate_candidates(data, theta_old, step, [1.0], 'adv_r', self._lambda_zero).numpy()
        # the reference's `_loss_pi` calls in order (pcpo.py:69, cpo.py:125 per tried candidate, pcpo.py:129)
        self._store_loss_pi_call(before[0], before[3], theta_old)
        for frac, row in self._tried:
            self._store_loss_pi_call(row[0], row[3], s.lincomb(1.0, theta_old, frac, step_direction))
        s.lincomb(1.0, theta_old, 1.0, step, out=ac.params[0])
        self._store_loss_pi_call(final[0, 0], final[0, 3], ac.params[0])
        self._last_actor_update = dict(g=g, x=x, b=b, p=p, xHx=xHx, alpha=alpha, q=q, r=r, s=sc,
                                       step_direction=step_direction, final_step=step, accept_step=accept_step,
                                       loss_reward_before=loss_reward_before, loss_cost_before=loss_cost_before)
        self._logger.store({
            'Loss/Loss_pi': float(final[0, 0] + final[0, 1]), 'Misc/AcceptanceStep': accept_step,
            'Misc/Alpha': alpha, 'Misc/FinalStepNorm': float(step.norm()), 'Misc/xHx': xHx,
            'Misc/H_inv_g': float(x.norm()), 'Misc/gradient_norm': float(g.norm()),
            'Misc/cost_gradient_norm': float(b.norm()), 'Misc/Lambda_star': 1.0, 'Misc/Nu_star': 1.0,
            'Misc/OptimCase': 1, 'Misc/A': 1.0, 'Misc/B': 1.0, 'Misc/q': q, 'Misc/r': r, 'Misc/s': sc})


@register
class FOCOPS(_LagrangeMixin, PolicyGradient):
    """focops.py: per-sample loss (KL(pi_theta || pi_old) - ratio * adv / focops_lam) * 1[KL <= eta] with the
    Lagrangian advantage; critics and KL early stop as in PolicyGradient._update."""
    _lagrange_min_max = False

    def _make_updater(self) -> PPOUpdater:
        up = super()._make_updater()
        a = self._cfgs.algo_cfgs
        up.ext = SurrogateExt(kl_coef=1.0, kl_mask_eta=float(a.focops_eta), ratio_scale=1.0 / float(a.focops_lam))
        return up


@register
class P3O(PPO):
    def _init_log(self) -> None:
        super()._init_log()
        self._logger.register_key('Loss/Loss_pi_cost', delta=True)

    def _make_updater(self) -> PPOUpdater:
        up = super()._make_updater()
        up.ext = SurrogateExt(cost_kappa=float(self._cfgs.algo_cfgs.kappa))
        return up

    def _update(self) -> None:
        """p3o.py:62-68: the penalty offset Jc - cost_limit is a constant of the epoch."""
        a = self._cfgs.algo_cfgs
        self._updater.ext.cost_excess = float(self._logger.get_stats('Metrics/EpCost')[0] - a.cost_limit)
        super()._update()
        st = self._updater._stats[:self._last_update_steps, 10].double().cpu()  # noqa: SLF001
        self._logger.store({'Loss/Loss_pi_cost': float(st.mean())})


@register
class CUP(_LagrangeMixin, PPO):
    """cup.py: stage 1 = PPO on the reward advantage; stage 2 = actor-only minibatch passes on
    lambda * coef * ratio * A_c + KL(pi_theta || pi_stage1) with their own KL early stop."""
    _lagrange_min_max = False

    def _lagrange_tensor(self) -> torch.Tensor:
        return self._lambda_zero  # CUP does not override _compute_adv_surrogate: stage 1 sees A_r only

    def _init(self) -> None:
        super()._init()
        a = self._cfgs.algo_cfgs
        self._updater2 = PPOUpdater(
            self._actor_critic, batch_size=a.batch_size, update_iters=a.update_iters, target_kl=a.target_kl,
            kl_early_stop=a.kl_early_stop, entropy_coef=0.0, use_critic_norm=a.use_critic_norm,
            critic_norm_coef=a.critic_norm_coef, use_max_grad_norm=True,  # cup.py:160 clips whenever set
            max_grad_norm=a.max_grad_norm, use_cost=a.use_cost, loss_kind=1, seed=int(self._cfgs.seed) + 1,
            update_critics=False, ext=SurrogateExt(kl_coef=1.0))

    def _init_log(self) -> None:
        super()._init_log()
        self._logger.register_key('Loss/Loss_pi_c', delta=True)
        self._logger.register_key('Train/SecondStepStopIter')
        self._logger.register_key('Train/SecondStepEntropy')
        self._logger.register_key('Train/SecondStepPolicyRatio', min_and_max=True)

    def _update(self) -> None:
        a = self._cfgs.algo_cfgs
        super()._update()  # lambda step, then the PPO stage (which consumes buf.get())
        data = self._last_update_data
        lam = self._lagrange.lagrangian_multiplier
        coef = (1 - a.gamma * a.lam) / (1 - a.gamma)
        stage2 = dict(data)
        # -(ratio * adv') with adv' = -(lambda * coef) * A_c is the reference's lambda * coef * ratio * A_c
        stage2['adv_r'] = data['adv_c'] * (-(lam * coef))
        perms = getattr(self, '_perms_override', None)
        if perms is not None:
            perms = perms[self._updater.update_iters:]
        out = self._updater2.run(stage2, self._lambda_zero, perms=perms, actor_lr=self._current_actor_lr(),
                                 critic_lr=0.0)
        st = out['stats'].double().cpu()
        lg = self._logger
        lg.extend('Train/SecondStepPolicyRatio', st[:, 3].tolist())
        lg.store({'Loss/Loss_pi_c': float(st[:, 2].mean()), 'Train/SecondStepEntropy': float(st[:, 4].mean()),
                  'Train/SecondStepStopIter': out['stop_iter']})


class _AugmentedEnvMixin:
    """_init_env / _init_log of the Saute and Simmer algorithm classes (ppo_saute.py:37-73)."""
    _adapter_cls = SauteAdapter

    def _init_env(self) -> None:
        c = self._cfgs
        self._env = self._adapter_cls(self._env_id, c.train_cfgs.vector_env_nums, self._seed, c)
        assert c.algo_cfgs.steps_per_epoch % (dist.world_size() * c.train_cfgs.vector_env_nums) == 0, (
            'The number of steps per epoch is not divisible by the number of environments.')
        self._steps_per_epoch = (c.algo_cfgs.steps_per_epoch // dist.world_size()
                                 // c.train_cfgs.vector_env_nums)

    def _init_log(self) -> None:
        super()._init_log()
        self._logger.register_key('Metrics/EpBudget')


class _SimmerMixin(_AugmentedEnvMixin):
    _adapter_cls = SimmerAdapter

    def _update(self) -> None:
        """ppo_simmer_pid.py:75-79: steer the safety budget with the epoch's mean episode cost first."""
        Jc = self._logger.get_stats('Metrics/EpCost')[0]
        self._env.control_budget(Jc)
        super()._update()


@register
class PPOSaute(_AugmentedEnvMixin, PPO):
    pass


@register
class TRPOSaute(_AugmentedEnvMixin, TRPO):
    pass


@register
class PPOSimmerPID(_SimmerMixin, PPO):
    pass


@register
class TRPOSimmerPID(_SimmerMixin, TRPO):
    pass


class _EarlyTerminatedMixin:
    """_init_env of ppo_early_terminated.py:40-66 / trpo_early_terminated.py."""

    def _init_env(self) -> None:
        c = self._cfgs
        self._env = EarlyTerminatedAdapter(self._env_id, c.train_cfgs.vector_env_nums, self._seed, c)
        assert c.algo_cfgs.steps_per_epoch % (dist.world_size() * c.train_cfgs.vector_env_nums) == 0, (
            'The number of steps per epoch is not divisible by the number of environments.')
        self._steps_per_epoch = (c.algo_cfgs.steps_per_epoch // dist.world_size()
                                 // c.train_cfgs.vector_env_nums)


@register
class PPOEarlyTerminated(_EarlyTerminatedMixin, PPO):
    pass


@register
class TRPOEarlyTerminated(_EarlyTerminatedMixin, TRPO):
    pass

