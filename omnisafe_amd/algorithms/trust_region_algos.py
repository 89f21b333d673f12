"""NaturalPG / TRPO / TRPOLag / CPO on the device.

Mirrors omnisafe/algorithms/on_policy/base/natural_pg.py:34-230, base/trpo.py:34-222,
naive_lagrange/trpo_lag.py:28-100 and second_order/cpo.py:36-462: same hooks, same logged keys, same
acceptance rules.  The per-sample arithmetic runs in libomnisafe_amd (trust_region.py); the scalar
LQCLP case analysis of CPO (6 reduced scalars) is host float32 algebra exactly as in the reference.
"""
from __future__ import annotations

import numpy as np
import torch

from ..lagrange import Lagrange
from ..trust_region import TrustRegionSolver
from ..update import PPOUpdater
from .policy_gradient import PolicyGradient
from .registry import register


@register
class NaturalPG(PolicyGradient):
    _loss_kind = 1
    _adv_key_r = 'adv_r'

    def _init(self) -> None:
        super()._init()
        a = self._cfgs.algo_cfgs
        self._solver = TrustRegionSolver(self._actor_critic, a.cg_iters, a.cg_damping,
                                         getattr(a, 'fvp_sample_freq', 1))

    def _make_updater(self) -> PPOUpdater:
        up = super()._make_updater()
        up.update_actor = False  # critics only in the minibatch loop (natural_pg.py:209-222)
        return up

    def _init_log(self) -> None:
        super()._init_log()
        for k in ('Misc/Alpha', 'Misc/FinalStepNorm', 'Misc/gradient_norm', 'Misc/xHx', 'Misc/H_inv_g'):
            self._logger.register_key(k)

    # ---- shared pieces -------------------------------------------------------------------------
    def _store_loss_pi_call(self, loss: float, ratio: float, theta: torch.Tensor) -> None:
        """What ONE call of the reference's `_loss_pi` leaves in the logger (policy_gradient.py:574-588):
        Loss/Loss_pi, the mean ratio, and entropy / std of the policy at `theta` (padded actor vector).  The
        reference calls it at theta_old, at every line-search candidate it tries and at the final parameters, and
        the epoch's csv value is the MEAN over those calls -- reproduced call for call."""
        lay = self._actor_critic.layout
        log_std = theta[lay.oLS:lay.oLS + lay.act_dim]
        self._logger.store({'Loss/Loss_pi': float(loss), 'Train/PolicyRatio': float(ratio),
                            'Train/Entropy': float(1.4189385332046727 + log_std.mean()),
                            'Train/PolicyStd': float(log_std.exp().mean())})

    def _store_fvp_kl_calls(self) -> None:
        """`_fvp` logs kl_divergence(p, q).mean() of every call (natural_pg.py:91-119); both distributions come
        from the current parameters there, so each stored value is 0 up to float32 noise (~1e-9)."""
        n = self._solver.fvp_calls + self._solver.cg_solves  # + the F(0) evaluation that opens every CG solve there
        for _ in range(n - getattr(self, '_fvp_logged', 0)):
            self._logger.store({'Train/KL': 0.0})
        self._fvp_logged = n

    def _eval_at(self, data: dict, theta_old, step, adv_key=None, lagrange=None):
        """[loss_pi, loss_cost, kl, mean ratio] at theta_old + step."""
        return self._solver.evaluate_candidates(data, theta_old, step, [1.0], adv_key or self._adv_key_r,
                                                self._lagrange_tensor() if lagrange is None else lagrange).numpy()[0]

    def _policy_gradient(self, data: dict):
        """loss, g = -flat_grad of the surrogate loss at theta_old, p_dist snapshot (trpo.py:176-186)."""
        s = self._solver
        s.begin(data['obs'])
        loss, grad = s.actor_loss_grad(data, self._adv_key_r, 'adv_c', self._lagrange_tensor())
        return loss, s.lincomb(-1.0, grad)

    def _natural_direction(self, g: torch.Tensor):
        s = self._solver
        x = s.conjugate_gradients(g)
        assert torch.isfinite(x).all(), 'x is not finite'
        xHx = float(s.dot(x, s.fvp(x)))
        assert xHx >= 0, 'xHx is negative'
        alpha = float(np.sqrt(np.float32(2 * self._cfgs.algo_cfgs.target_kl) / (np.float32(xHx) + np.float32(1e-8))))
        return x, xHx, alpha

    def _update_actor(self, data: dict) -> None:
        """natural_pg.py:121-182: theta <- theta_old + alpha x (no line search)."""
        ac = self._actor_critic
        theta_old = ac.params[0].clone()
        _, g = self._policy_gradient(data)
        x, xHx, alpha = self._natural_direction(g)
        step = self._solver.lincomb(alpha, x)
        assert torch.isfinite(step).all(), 'step_direction is not finite'
        before = self._eval_at(data, theta_old, torch.zeros_like(step))
        after = self._eval_at(data, theta_old, step)
        self._solver.lincomb(1.0, theta_old, 1.0, step, out=ac.params[0])
        self._store_fvp_kl_calls()
        self._store_loss_pi_call(before[0], before[3], theta_old)     # natural_pg.py:154
        self._store_loss_pi_call(after[0], after[3], ac.params[0])    # natural_pg.py:169
        self._logger.store({'Misc/Alpha': alpha, 'Misc/FinalStepNorm': float(step.norm()), 'Misc/xHx': xHx,
                            'Misc/gradient_norm': float(g.norm()), 'Misc/H_inv_g': float(x.norm())})

    def _update(self) -> None:
        """natural_pg.py:184-230: one full-batch actor step, then critic minibatch passes."""
        data = self._buf.get()
        self._update_actor(data)
        out = self._updater.run(data, self._lambda_zero, actor_lr=0.0,
                                critic_lr=float(self._cfgs.model_cfgs.critic.lr),
                                perms=getattr(self, '_perms_override', None))
        a = self._cfgs.algo_cfgs
        summ = PPOUpdater.summarize(out, a.critic_norm_coef, a.use_critic_norm)
        self._buf.check_gae_sync()  # (the stream is drained here: sticky time-out word of the chained GAE scan)
        lg = self._logger
        lg.store({'Loss/Loss_reward_critic': summ['Loss/Loss_reward_critic']})
        if a.use_cost:
            lg.store({'Loss/Loss_cost_critic': summ['Loss/Loss_cost_critic']})
        lg.store({'Train/StopIter': a.update_iters, 'Value/Adv': float(data['adv_r'].mean())})


@register
class TRPO(NaturalPG):
    def _init_log(self) -> None:
        super()._init_log()
        self._logger.register_key('Misc/AcceptanceStep')

    def _search_step_size(self, data: dict, theta_old, step_direction, g, loss_before: float,
                          total_steps: int = 15, decay: float = 0.8):
        """trpo.py:56-148 on batched candidate evaluations."""
        target_kl = self._cfgs.algo_cfgs.target_kl
        fracs = [1.0]
        for _ in range(total_steps - 1):
            fracs.append(fracs[-1] * decay)
        res = self._solver.evaluate_candidates(data, theta_old, step_direction, fracs, self._adv_key_r,
                                               self._lagrange_tensor()).numpy()
        final_kl, acceptance_step, step_frac = 0.0, 0, fracs[-1] * decay
        self._tried = []  # (frac, row) of every candidate the reference would have evaluated
        for j, frac in enumerate(fracs):
            loss, kl = float(res[j, 0]), float(res[j, 2])
            self._tried.append((frac, res[j]))
            loss_improve = loss_before - loss
            if not np.isfinite(loss):
                self._logger.log('WARNING: loss_pi not finite')
            elif loss_improve < 0:
                self._logger.log('INFO: did not improve improve <0')
            elif kl > target_kl:
                self._logger.log('INFO: violated KL constraint.')
            else:
                acceptance_step, final_kl, step_frac = j + 1, kl, frac
                self._logger.log(f'Accept step at i={acceptance_step}')
                break
        else:
            self._logger.log('INFO: no suitable step found...')
            step_direction = torch.zeros_like(step_direction)
        self._store_fvp_kl_calls()
        self._logger.store({'Train/KL': final_kl})
        return self._solver.lincomb(step_frac, step_direction), acceptance_step

    def _update_actor(self, data: dict) -> None:
        """trpo.py:150-222."""
        ac = self._actor_critic
        theta_old = ac.params[0].clone()
        loss, g = self._policy_gradient(data)
        loss_before = float(loss)
        x, xHx, alpha = self._natural_direction(g)
        step_direction = self._solver.lincomb(alpha, x)
        assert torch.isfinite(step_direction).all(), 'step_direction is not finite'
        before = self._eval_at(data, theta_old, torch.zeros_like(step_direction))
        step, accept_step = self._search_step_size(data, theta_old, step_direction, g, loss_before)
        final = self._eval_at(data, theta_old, step)
        # the reference's `_loss_pi` calls in order: theta_old (trpo.py:178), every tried candidate
        # (trpo.py:104-110), the final parameters (trpo.py:207)
        self._store_loss_pi_call(before[0], before[3], theta_old)
        for frac, row in self._tried:
            self._store_loss_pi_call(row[0], row[3], self._solver.lincomb(1.0, theta_old, frac, step_direction))
        self._solver.lincomb(1.0, theta_old, 1.0, step, out=ac.params[0])
        self._store_loss_pi_call(final[0], final[3], ac.params[0])
        self._last_actor_update = dict(g=g, x=x, xHx=xHx, alpha=alpha, step_direction=step_direction,
                                       final_step=step, accept_step=accept_step, loss_before=loss_before)
        self._logger.store({'Misc/Alpha': alpha, 'Misc/FinalStepNorm': float(step.norm()), 'Misc/xHx': xHx,
                            'Misc/gradient_norm': float(g.norm()), 'Misc/H_inv_g': float(x.norm()),
                            'Misc/AcceptanceStep': accept_step})


@register
class TRPOLag(TRPO):
    """trpo_lag.py:28-100: TRPO on the surrogate (A_r - lambda A_c)/(1 + lambda) + dual ascent."""

    def _init(self) -> None:
        super()._init()
        lc = self._cfgs.lagrange_cfgs
        self._lagrange = Lagrange(**(lc.todict() if hasattr(lc, 'todict') else dict(lc)), device=self._device)

    def _init_log(self) -> None:
        super()._init_log()
        self._logger.register_key('Metrics/LagrangeMultiplier', min_and_max=True)

    def _lagrange_tensor(self) -> torch.Tensor:
        return self._lagrange.device_multiplier

    def _update(self) -> None:
        Jc = self._logger.get_stats('Metrics/EpCost')[0]
        assert not np.isnan(Jc), 'cost for updating lagrange multiplier is nan'
        self._lagrange.update_lagrange_multiplier(Jc)
        super()._update()
        self._logger.store({'Metrics/LagrangeMultiplier': self._lagrange.lagrangian_multiplier})


def cpo_determine_case(bb: float, ep_costs: float, q, r, s, target_kl: float):
    """CPO._determine_case (cpo.py:237-268) in float32 scalar algebra."""
    f = np.float32
    q, r, s = f(q), f(r), f(s)
    if bb <= 1e-6 and ep_costs < 0:
        return 4, f(0.0), f(0.0)
    assert np.isfinite(r), 'r is not finite'
    assert np.isfinite(s), 's is not finite'
    A = q - r ** 2 / (s + f(1e-8))
    B = f(2 * target_kl) - f(ep_costs) ** 2 / (s + f(1e-8))
    if ep_costs < 0 and B < 0:
        case = 3
    elif ep_costs < 0 <= B:
        case = 2
    elif ep_costs >= 0 and B >= 0:
        case = 1
    else:
        case = 0
    return case, A, B


def cpo_step_coefficients(case: int, xHx, A, B, q, r, s, ep_costs: float, target_kl: float):
    """CPO._step_direction (cpo.py:271-337): returns (cx, cp, lambda*, nu*) with
    step_direction = cx * x + cp * p."""
    f = np.float32
    xHx, q, r, s, c = f(xHx), f(q), f(r), f(s), f(ep_costs)
    if case in (3, 4):
        alpha = np.sqrt(f(2 * target_kl) / (xHx + f(1e-8)))
        return f(alpha), f(0.0), f(1) / (alpha + f(1e-8)), f(0.0)
    if case in (1, 2):
        with np.errstate(all='ignore'):
            lambda_a = np.sqrt(f(A) / f(B))
            lambda_b = np.sqrt(q / f(2 * target_kl))
        eps_cost = c + f(1e-8)
        bound = f(float(r) / float(eps_cost))
        if ep_costs < 0:
            la = np.clip(lambda_a, f(0.0), bound)
            lb = np.clip(lambda_b, bound, f(np.inf))
        else:
            la = np.clip(lambda_a, bound, f(np.inf))
            lb = np.clip(lambda_b, f(0.0), bound)

        def f_a(lam):
            return f(-0.5) * (f(A) / (lam + f(1e-8)) + f(B) * lam) - r * c / (s + f(1e-8))

        def f_b(lam):
            return f(-0.5) * (q / (lam + f(1e-8)) + f(2 * target_kl) * lam)

        lam = la if f_a(la) >= f_b(lb) else lb
        nu = max(lam * c - r, f(0.0)) / (s + f(1e-8))
        inv = f(1.0) / (lam + f(1e-8))
        return f(inv), f(-inv * nu), f(lam), f(nu)
    nu = np.sqrt(f(2 * target_kl) / (s + f(1e-8)))
    return f(0.0), f(-nu), f(0.0), f(nu)


@register
class CPO(TRPO):
    """second_order/cpo.py:36-462."""

    def _init_log(self) -> None:
        super()._init_log()
        for k in ('Misc/cost_gradient_norm', 'Misc/A', 'Misc/B', 'Misc/q', 'Misc/r', 'Misc/s',
                  'Misc/Lambda_star', 'Misc/Nu_star', 'Misc/OptimCase'):
            self._logger.register_key(k)

    def _cpo_search_step(self, data, theta_old, step_direction, g, loss_reward_before: float,
                         loss_cost_before: float, total_steps: int = 15, decay: float = 0.8,
                         violation_c: float = 0.0, optim_case: int = 0):
        """cpo.py:57-180 on batched candidate evaluations (same accept / continue rules, including the
        reference's non-decaying `continue` on a non-finite KL)."""
        target_kl = self._cfgs.algo_cfgs.target_kl
        fracs = [1.0]
        for _ in range(total_steps - 1):
            fracs.append(fracs[-1] * decay)
        chunk, cache = 20, {}

        def evaluated(k: int):
            """Candidate k of the geometric sequence; evaluated lazily, 20 candidates per batch (PCPO
            asks for up to 200 steps and accepts within the first few)."""
            c0 = (k // chunk) * chunk
            if c0 not in cache:
                cache[c0] = self._solver.evaluate_candidates(data, theta_old, step_direction,
                                                             fracs[c0:c0 + chunk], 'adv_r',
                                                             self._lambda_zero).numpy()
            return cache[c0][k - c0]

        step_frac, k, kl, acceptance_step, accepted = 1.0, 0, 0.0, 0, False
        self._tried = []
        for step in range(total_steps):
            acceptance_step = step + 1
            row = evaluated(k)
            self._tried.append((step_frac, row))
            loss_reward, loss_cost, kl = float(row[0]), float(row[1]), float(row[2])
            loss_reward_improve = loss_reward_before - loss_reward
            loss_cost_diff = loss_cost - loss_cost_before
            if not np.isfinite(kl):
                self._logger.log('WARNING: KL not finite')
                continue  # cpo.py:150-152: no decay of step_frac
            if loss_reward_improve < 0 if optim_case > 1 else False:
                self._logger.log('INFO: did not improve improve <0')
            elif loss_cost_diff > max(-violation_c, 0):
                self._logger.log(f'INFO: no improve {loss_cost_diff} > {max(-violation_c, 0)}')
            elif kl > target_kl:
                self._logger.log(f'INFO: violated KL constraint {kl} at step {step + 1}.')
            else:
                self._logger.log(f'Accept step at i={step + 1}')
                accepted = True
                break
            step_frac *= decay
            k += 1
        if not accepted:
            self._logger.log('INFO: no suitable step found...')
            step_direction = torch.zeros_like(step_direction)
            acceptance_step = 0
        self._store_fvp_kl_calls()
        self._logger.store({'Train/KL': kl})
        return self._solver.lincomb(step_frac, step_direction), acceptance_step

    def _update_actor(self, data: dict) -> None:
        """cpo.py:340-462."""
        ac, s, a = self._actor_critic, self._solver, self._cfgs.algo_cfgs
        theta_old = ac.params[0].clone()
        s.begin(data['obs'])
        loss_r, grad_r = s.actor_loss_grad(data, 'adv_r', 'adv_c', self._lambda_zero)
        loss_reward_before = float(loss_r)
        g = s.lincomb(-1.0, grad_r)
        x, xHx, alpha = self._natural_direction(g)
        # cost surrogate mean(ratio * adv_c): the kernel differentiates -mean(ratio * adv)
        loss_c, grad_c = s.actor_loss_grad(data, 'adv_c', 'adv_c', self._lambda_zero)
        loss_cost_before = -float(loss_c)
        b = s.lincomb(-1.0, grad_c)
        ep_costs = float(self._logger.get_stats('Metrics/EpCost')[0] - a.cost_limit)
        p = s.conjugate_gradients(b)
        q = xHx
        r = float(s.dot(g, p))
        sc = float(s.dot(b, p))
        bb = float(s.dot(b, b))
        case, A, B = cpo_determine_case(bb, ep_costs, q, r, sc, a.target_kl)
        cx, cp, lambda_star, nu_star = cpo_step_coefficients(case, xHx, A, B, q, r, sc, ep_costs, a.target_kl)
        step_direction = s.lincomb(float(cx), x, float(cp), p)
        before = self._eval_at(data, theta_old, torch.zeros_like(step_direction), 'adv_r', self._lambda_zero)
        step, accept_step = self._cpo_search_step(data, theta_old, step_direction, g, loss_reward_before,
                                                  loss_cost_before, total_steps=20, violation_c=ep_costs,
                                                  optim_case=case)
        final = s.evaluate_candidates(data, theta_old, step, [1.0], 'adv_r', self._lambda_zero).numpy()
        # `_loss_pi` calls of the reference in order: cpo.py:369 (theta_old), :125 (every tried candidate),
        # :439 (final parameters); the combined reward + cost loss is stored once more at :445
        self._store_loss_pi_call(before[0], before[3], theta_old)
        for frac, row in self._tried:
            self._store_loss_pi_call(row[0], row[3], s.lincomb(1.0, theta_old, frac, step_direction))
        s.lincomb(1.0, theta_old, 1.0, step, out=ac.params[0])
        self._store_loss_pi_call(final[0, 0], final[0, 3], ac.params[0])
        self._last_actor_update = dict(g=g, x=x, b=b, p=p, xHx=xHx, alpha=alpha, q=q, r=r, s=sc, case=case,
                                       A=float(A), B=float(B), lambda_star=float(lambda_star),
                                       nu_star=float(nu_star), step_direction=step_direction, final_step=step,
                                       accept_step=accept_step, loss_reward_before=loss_reward_before,
                                       loss_cost_before=loss_cost_before)
        self._logger.store({
            'Loss/Loss_pi': float(final[0, 0] + final[0, 1]), 'Misc/AcceptanceStep': accept_step,
            'Misc/Alpha': alpha, 'Misc/FinalStepNorm': float(step.norm()), 'Misc/xHx': xHx,
            'Misc/H_inv_g': float(x.norm()), 'Misc/gradient_norm': float(g.norm()),
            'Misc/cost_gradient_norm': float(b.norm()), 'Misc/Lambda_star': float(lambda_star),
            'Misc/Nu_star': float(nu_star), 'Misc/OptimCase': int(case), 'Misc/A': float(A),
            'Misc/B': float(B), 'Misc/q': q, 'Misc/r': r, 'Misc/s': sc})
