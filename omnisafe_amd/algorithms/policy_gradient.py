"""PolicyGradient / PPO / PPOLag on the device.

Mirrors omnisafe/algorithms/on_policy/base/policy_gradient.py:39-588, base/ppo.py:27-87 and
naive_lagrange/ppo_lag.py:28-102: same hook structure (_init_env/_init_model/_init/_init_log/learn/
_update), same logged keys, same return value of ``learn``.  All per-sample arithmetic runs in
libomnisafe_amd kernels (see omnisafe_amd/update.py, adapter.py, buffer.py).
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from .. import distributed as dist
from ..adapter import OnPolicyAdapter
from ..buffer import VectorOnPolicyBuffer
from ..lagrange import Lagrange
from ..logger import Logger
from ..models import ConstraintActorCritic
from ..update import PPOUpdater
from .base_algo import BaseAlgo
from .registry import register


@register
class PolicyGradient(BaseAlgo):  # pylint: disable=too-many-instance-attributes
    _loss_kind = 1  # plain ratio * adv (policy_gradient.py:574-578)

    # ------------------------------------------------------------------ init hooks
    def _init_env(self) -> None:
        """policy_gradient.py:48-77."""
        c = self._cfgs
        self._env = OnPolicyAdapter(self._env_id, c.train_cfgs.vector_env_nums, self._seed, c)
        assert c.algo_cfgs.steps_per_epoch % (dist.world_size() * c.train_cfgs.vector_env_nums) == 0, (
            'The number of steps per epoch is not divisible by the number of environments.')
        self._steps_per_epoch = (c.algo_cfgs.steps_per_epoch // dist.world_size()
                                 // c.train_cfgs.vector_env_nums)

    def _init_model(self) -> None:
        """policy_gradient.py:79-105."""
        c = self._cfgs
        self._actor_critic = ConstraintActorCritic(self._env.observation_space, self._env.action_space,
                                                   c.model_cfgs, c.train_cfgs.epochs, device=self._device)
        self._actor_critic.set_seed(self._seed)
        if dist.collectives_active():
            self._actor_critic.sync_params()
        if getattr(c.model_cfgs, 'exploration_noise_anneal', False):  # policy_gradient.py:101-105
            self._actor_critic.set_annealing(epochs=[0, c.train_cfgs.epochs], std=list(c.model_cfgs.std_range))

    def _init(self) -> None:
        """policy_gradient.py:107-131."""
        a = self._cfgs.algo_cfgs
        self._buf = VectorOnPolicyBuffer(
            obs_space=self._env.observation_space, act_space=self._env.action_space,
            size=self._steps_per_epoch, gamma=a.gamma, lam=a.lam, lam_c=a.lam_c,
            advantage_estimator=a.adv_estimation_method, standardized_adv_r=a.standardized_rew_adv,
            standardized_adv_c=a.standardized_cost_adv, penalty_coefficient=a.penalty_coef,
            num_envs=self._cfgs.train_cfgs.vector_env_nums, device=self._device)
        self._lambda_zero = torch.zeros(1, dtype=torch.float32, device=self._device)
        self._updater = self._make_updater()

    def _make_updater(self) -> PPOUpdater:
        a = self._cfgs.algo_cfgs
        return PPOUpdater(
            self._actor_critic, batch_size=a.batch_size, update_iters=a.update_iters,
            target_kl=a.target_kl, kl_early_stop=a.kl_early_stop, clip=getattr(a, 'clip', 0.2),
            entropy_coef=a.entropy_coef, use_critic_norm=a.use_critic_norm,
            critic_norm_coef=a.critic_norm_coef, use_max_grad_norm=a.use_max_grad_norm,
            max_grad_norm=a.max_grad_norm, use_cost=a.use_cost, loss_kind=self._loss_kind,
            seed=int(self._cfgs.seed), dp_mode=os.environ.get('OSA_DP_MODE', 'replicated'))

    def _init_log(self) -> None:
        """policy_gradient.py:133-236: same keys, same order."""
        c = self._cfgs
        self._logger = Logger(output_dir=c.logger_cfgs.log_dir, exp_name=c.exp_name, seed=c.seed,
                              use_tensorboard=c.logger_cfgs.use_tensorboard,
                              use_wandb=c.logger_cfgs.use_wandb, config=c,
                              verbose=getattr(c.logger_cfgs, 'verbose', True))
        what_to_save = {'pi': self._actor_critic.actor}
        if c.algo_cfgs.obs_normalize:
            what_to_save['obs_normalizer'] = self._env.save()['obs_normalizer']
        self._logger.setup_torch_saver(what_to_save)
        self._logger.torch_save()
        lg, w = self._logger, c.logger_cfgs.window_lens
        lg.register_key('Metrics/EpRet', window_length=w)
        lg.register_key('Metrics/EpCost', window_length=w)
        lg.register_key('Metrics/EpLen', window_length=w)
        lg.register_key('Train/Epoch')
        lg.register_key('Train/Entropy')
        lg.register_key('Train/KL')
        lg.register_key('Train/StopIter')
        lg.register_key('Train/PolicyRatio', min_and_max=True)
        lg.register_key('Train/LR')
        if c.model_cfgs.actor_type == 'gaussian_learning':
            lg.register_key('Train/PolicyStd')
        lg.register_key('TotalEnvSteps')
        lg.register_key('Loss/Loss_pi', delta=True)
        lg.register_key('Value/Adv')
        lg.register_key('Loss/Loss_reward_critic', delta=True)
        lg.register_key('Value/reward')
        if c.algo_cfgs.use_cost:
            lg.register_key('Loss/Loss_cost_critic', delta=True)
            lg.register_key('Value/cost')
        lg.register_key('Time/Total')
        lg.register_key('Time/Rollout')
        lg.register_key('Time/Update')
        lg.register_key('Time/Epoch')
        lg.register_key('Time/FPS')
        for key in self._env.env_spec_keys:
            lg.register_key(key)

    # ------------------------------------------------------------------ learn
    def learn(self) -> tuple[float, float, float]:
        """policy_gradient.py:238-306.  Wall-clock keys bracket device work with a stream
        synchronisation, so Time/FPS is the metric of BASELINE.json (env-steps/s, rollout + update)."""
        c = self._cfgs
        start_time = time.time()
        self._logger.log('INFO: Start training')
        try:
            self._learn_epochs(c, start_time)
        finally:
            self._logger.flush()  # a deferred csv row (logger.py) reaches the disk even when an epoch raises
        ep_ret = self._logger.get_stats('Metrics/EpRet')[0]
        ep_cost = self._logger.get_stats('Metrics/EpCost')[0]
        ep_len = self._logger.get_stats('Metrics/EpLen')[0]
        self._logger.close()
        self._env.close()
        return ep_ret, ep_cost, ep_len

    def _learn_epochs(self, c, start_time: float) -> None:
        for epoch in range(c.train_cfgs.epochs):
            epoch_time = time.time()
            rollout_time = time.time()
            self._env.rollout(steps_per_epoch=self._steps_per_epoch, agent=self._actor_critic,
                              buffer=self._buf, logger=self._logger)
            torch.cuda.synchronize(self._device)
            self._logger.store({'Time/Rollout': time.time() - rollout_time})
            update_time = time.time()
            self._update()
            torch.cuda.synchronize(self._device)
            self._logger.store({'Time/Update': time.time() - update_time})
            if getattr(c.model_cfgs, 'exploration_noise_anneal', False):  # policy_gradient.py:271-272
                self._actor_critic.annealing(epoch)
            if c.model_cfgs.actor.lr is not None:
                self._actor_critic.actor_scheduler.step()
            self._logger.store({
                'TotalEnvSteps': (epoch + 1) * c.algo_cfgs.steps_per_epoch,
                'Time/FPS': c.algo_cfgs.steps_per_epoch / (time.time() - epoch_time),
                'Time/Total': time.time() - start_time,
                'Time/Epoch': time.time() - epoch_time,
                'Train/Epoch': epoch,
                'Train/LR': (0.0 if c.model_cfgs.actor.lr is None
                             else self._actor_critic.actor_scheduler.get_last_lr()[0]),
            })
            self._logger.dump_tabular()
            if (epoch + 1) % c.logger_cfgs.save_model_freq == 0 or (epoch + 1) == c.train_cfgs.epochs:
                self._logger.torch_save()

    # ------------------------------------------------------------------ update
    def _lagrange_tensor(self) -> torch.Tensor:
        """Device scalar lambda for _compute_adv_surrogate; PolicyGradient/PPO use adv_r only
        (policy_gradient.py:526-542) which is the lambda = 0 case of (adv_r - l*adv_c)/(1 + l)."""
        return self._lambda_zero

    def _current_actor_lr(self) -> float:
        if self._cfgs.model_cfgs.actor.lr is None:
            return 0.0
        return float(self._actor_critic.actor_scheduler.get_last_lr()[0])

    def _update(self) -> None:
        """policy_gradient.py:308-405."""
        data = self._buf.get()
        out = self._updater.run(data, self._lagrange_tensor(), actor_lr=self._current_actor_lr(),
                                critic_lr=float(self._cfgs.model_cfgs.critic.lr),
                                perms=getattr(self, '_perms_override', None))  # parity tests inject the order
        self._last_update_data, self._last_update_steps = data, out['steps']
        a = self._cfgs.algo_cfgs
        summ = PPOUpdater.summarize(out, a.critic_norm_coef, a.use_critic_norm)
        self._buf.check_gae_sync()  # (the stream is drained here: sticky time-out word of the chained GAE scan)
        lg = self._logger
        ratio = summ['per_step']['ratio_mean']
        # min/max/std are over minibatch means, as in the reference (logger.py:277); one bulk append instead
        # of 40 960 logger.store calls per epoch
        lg.extend('Train/PolicyRatio', ratio.tolist())
        lg.store({'Train/Entropy': summ['Train/Entropy'], 'Loss/Loss_pi': summ['Loss/Loss_pi'],
                  'Loss/Loss_reward_critic': summ['Loss/Loss_reward_critic'],
                  'Train/PolicyStd': self._actor_critic.actor.std})
        if a.use_cost:
            lg.store({'Loss/Loss_cost_critic': summ['Loss/Loss_cost_critic']})
        # the reference logs the LAST minibatch's adv_r.mean() here (the dataloader's loop variables shadow the
        # full batch: policy_gradient.py:369-377, 402) -- same here
        lg.store({'Train/StopIter': out['stop_iter'], 'Value/Adv': self._gather_mean(data['adv_r'], out['last_minibatch']),
                  'Train/KL': out['kl']})

    def _gather_mean(self, x: torch.Tensor, idx: torch.Tensor) -> float:
        """mean(x[idx]) in one launch (osa_gather_mean) instead of an index kernel, a reduction and a division."""
        from .. import _lib

        lib = _lib.load(require_gpu=True)
        out = self.__dict__.setdefault('_gm_out', torch.empty(1, dtype=torch.float32, device=x.device))
        idx = idx.contiguous()
        _lib.check(lib.osa_gather_mean(_lib.ptr(x), _lib.ptr(idx), idx.numel(), _lib.ptr(out), _lib.stream_ptr()),
                   'osa_gather_mean')
        return float(out)


@register
class PPO(PolicyGradient):
    """base/ppo.py:27-87: clipped surrogate + entropy bonus."""
    _loss_kind = 0


@register
class PPOLag(PPO):
    """naive_lagrange/ppo_lag.py:28-102."""

    def _init(self) -> None:
        super()._init()
        lc = self._cfgs.lagrange_cfgs
        lc = lc.todict() if hasattr(lc, 'todict') else dict(lc)
        self._lagrange = Lagrange(**lc, device=self._device)

    def _init_log(self) -> None:
        super()._init_log()
        self._logger.register_key('Metrics/LagrangeMultiplier', min_and_max=True)

    def _lagrange_tensor(self) -> torch.Tensor:
        return self._lagrange.device_multiplier

    def _update(self) -> None:
        """ppo_lag.py:52-80: dual ascent on lambda with the epoch's mean episode cost, then the
        policy / critic update with the surrogate (adv_r - lambda adv_c)/(1 + lambda)."""
        Jc = self._logger.get_stats('Metrics/EpCost')[0]
        assert not np.isnan(Jc), 'cost for updating lagrange multiplier is nan'
        self._lagrange.update_lagrange_multiplier(Jc)
        super()._update()
        self._logger.store({'Metrics/LagrangeMultiplier': self._lagrange.lagrangian_multiplier})
