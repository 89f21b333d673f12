"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy + CPU torch) of the reference's on-policy hot path
(PKU-Alignment/omnisafe v0.5.0).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module -- and only as the checker / the timed CPU baseline.  The
product (``omnisafe_amd``) never imports it and fails loudly when its HIP library is missing.

Parity status: PINNED.  Every function here is checked
  * against the live reference in the build container (tests/test_oracle_vs_reference.py, skipped when
    /root/reference is absent), and
  * against the committed golden vectors in tests/golden/*.npz that were produced by the unmodified
    reference via oracle/make_golden.py (these travel to the GPU box).
The reference's own known-answer tests for this path (tests/test_utils.py:95-115 discount_cumsum;
tests/test_policy.py:55-74 CPO case ids) are restated in tests/test_oracle_golden.py.

Third-party arithmetic: the reference's numerics below the algorithm level are PyTorch's
(torch >= 1.10, here 2.10.0): Normal.log_prob/entropy/kl_divergence, optim.Adam, clip_grad_norm_,
mse_loss, autograd.  The oracle calls the same CPU torch ops, so for those pieces the oracle *is* the
reference arithmetic; what is restated is the reference's own algorithmic code, cited per function
(paths relative to /root/reference/).
"""
from __future__ import annotations

import math
from typing import Callable

import numpy as np
import torch

# --------------------------------------------------------------------------------------------------
# K5: discounted cumulative sum / GAE  (omnisafe/utils/math.py:59-82,
#     omnisafe/common/buffer/onpolicy_buffer.py:148-203, 299-331)
# --------------------------------------------------------------------------------------------------


def discount_cumsum(x: np.ndarray, discount: float) -> np.ndarray:
    """y[L-1] = x[L-1]; y[i] = x[i] + discount * y[i+1], all in float64 with a separately rounded
    multiply and add (omnisafe/utils/math.py:76-82).  Returns float64."""
    y = np.asarray(x).astype(np.float64).copy()
    cumsum = y[-1]
    d = np.float64(discount)
    for idx in range(len(y) - 2, -1, -1):
        cumsum = y[idx] + d * cumsum
        y[idx] = cumsum
    return y


def finish_path(reward, cost, value_r, value_c, last_value_r, last_value_c, gamma, lam, lam_c,
                penalty_coef=0.0, estimator='gae'):
    """One path of one env, literal restatement of OnPolicyBuffer.finish_path
    (omnisafe/common/buffer/onpolicy_buffer.py:170-203) + _calculate_adv_and_value_targets
    ('gae' :299-303, 'gae-rtg' :305-310, 'plain' :328-331).  Inputs float32 1-D arrays of length L,
    bootstraps float32 scalars.  Returns float32 (adv_r, tgt_r, adv_c, tgt_c, disc_ret) exactly as
    they land in the float32 buffer.  NB: gamma multiplies a float32 tensor, so the *float32*
    rounding of gamma is what enters delta; the recurrences run in float64 with the python-float
    discount (math.py:77-81)."""
    f32 = np.float32
    rewards = np.concatenate([np.asarray(reward, f32), np.asarray([last_value_r], f32)])
    values_r = np.concatenate([np.asarray(value_r, f32), np.asarray([last_value_r], f32)])
    costs = np.concatenate([np.asarray(cost, f32), np.asarray([last_value_c], f32)])
    values_c = np.concatenate([np.asarray(value_c, f32), np.asarray([last_value_c], f32)])
    disc_ret = discount_cumsum(rewards, gamma)[:-1].astype(f32)
    rewards = (rewards - f32(penalty_coef) * costs).astype(f32)

    def adv_and_target(values, rews, lam_):
        g32 = f32(gamma)
        if estimator in ('gae', 'gae-rtg'):
            deltas = ((rews[:-1] + (g32 * values[1:]).astype(f32)).astype(f32) - values[:-1]).astype(f32)
            adv = discount_cumsum(deltas, gamma * lam_)
            if estimator == 'gae':
                target = adv + values[:-1].astype(np.float64)
            else:
                target = discount_cumsum(rews, gamma)[:-1]
        elif estimator == 'plain':
            adv = ((rews[:-1] + (g32 * values[1:]).astype(f32)).astype(f32) - values[:-1]).astype(f32)
            target = discount_cumsum(rews, gamma)[:-1]
        elif estimator == 'vtrace':
            # onpolicy_buffer.py:312-326 -> _calculate_v_trace :380-405 with policy == behaviour
            # probabilities (rho = c = 1): an all-float32 backward recursion
            L = len(values) - 1
            v_s = values[:-1].copy()
            last = values[-1]
            for idx in range(L - 1, -1, -1):
                delta = f32(f32(rews[idx] + f32(g32 * values[idx + 1])) - values[idx])
                v_s[idx] = f32(v_s[idx] + f32(delta + f32(g32 * f32(last - values[idx + 1]))))
                last = v_s[idx]
            v_s_plus_1 = np.concatenate([v_s[1:], values[-1:]])
            adv = ((rews[:-1] + (g32 * v_s_plus_1).astype(f32)).astype(f32) - values[:-1]).astype(f32)
            target = v_s
        else:
            raise NotImplementedError(estimator)
        return adv.astype(f32), target.astype(f32)

    adv_r, tgt_r = adv_and_target(values_r, rewards, lam)
    adv_c, tgt_c = adv_and_target(values_c, costs, lam_c)
    return adv_r, tgt_r, adv_c, tgt_c, disc_ret


def gae_time_major(reward, cost, value_r, value_c, path_end, boot_r, boot_c, gamma, lam, lam_c,
                   penalty_coef=0.0, estimator='gae'):
    """Same arithmetic as ``finish_path`` for a whole (T, N) time-major buffer in one backward sweep
    vectorised over envs.  ``path_end[t, n] != 0`` marks the last step of a path whose bootstrap values
    are ``boot_r[t, n]``, ``boot_c[t, n]`` (the arguments VectorOnPolicyBuffer.finish_path(idx=n) got,
    omnisafe/common/buffer/vector_onpolicy_buffer.py:101-111).  Every env's last stored step must be
    a path end (the reference always finishes paths at epoch end, onpolicy_adapter.py:114-136).
    Bit-identical to the per-path version: the float64 recurrence of one path never mixes with
    another's because the carry is re-seeded at every path end."""
    f32, f64 = np.float32, np.float64
    reward = np.asarray(reward, f32)
    cost = np.asarray(cost, f32)
    value_r = np.asarray(value_r, f32)
    value_c = np.asarray(value_c, f32)
    T, N = reward.shape
    g32 = f32(gamma)
    pc = f32(penalty_coef)
    out = {k: np.zeros((T, N), f32) for k in ('adv_r', 'tgt_r', 'adv_c', 'tgt_c', 'disc_ret')}
    nv_r = np.zeros(N, f32)
    nv_c = np.zeros(N, f32)
    a_r = np.zeros(N, f64)
    a_c = np.zeros(N, f64)
    ret_r = np.zeros(N, f64)  # discounted return of raw reward
    rtg_r = np.zeros(N, f64)  # rewards-to-go of penalised reward (gae-rtg / plain targets)
    rtg_c = np.zeros(N, f64)
    vs_r = np.zeros(N, f32)  # v-trace carries (float32 recursion)
    vs_c = np.zeros(N, f32)
    d_r, d_c, d_g = f64(gamma * lam), f64(gamma * lam_c), f64(gamma)
    for t in range(T - 1, -1, -1):
        end = np.asarray(path_end[t]) != 0
        br = np.asarray(boot_r[t], f32)
        bc = np.asarray(boot_c[t], f32)
        nv_r = np.where(end, br, nv_r)
        nv_c = np.where(end, bc, nv_c)
        a_r = np.where(end, 0.0, a_r)
        a_c = np.where(end, 0.0, a_c)
        ret_r = np.where(end, br.astype(f64), ret_r)
        rtg_r = np.where(end, (br - pc * bc).astype(f32).astype(f64), rtg_r)
        rtg_c = np.where(end, bc.astype(f64), rtg_c)
        vs_r = np.where(end, br, vs_r)
        vs_c = np.where(end, bc, vs_c)
        r_pen = (reward[t] - pc * cost[t]).astype(f32)
        delta_r = ((r_pen + (g32 * nv_r).astype(f32)).astype(f32) - value_r[t]).astype(f32)
        delta_c = ((cost[t] + (g32 * nv_c).astype(f32)).astype(f32) - value_c[t]).astype(f32)
        ret_r = reward[t].astype(f64) + d_g * ret_r
        rtg_r = r_pen.astype(f64) + d_g * rtg_r
        rtg_c = cost[t].astype(f64) + d_g * rtg_c
        if estimator == 'vtrace':
            adv_vr = ((r_pen + (g32 * vs_r).astype(f32)).astype(f32) - value_r[t]).astype(f32)
            adv_vc = ((cost[t] + (g32 * vs_c).astype(f32)).astype(f32) - value_c[t]).astype(f32)
            vs_r = (value_r[t] + (delta_r + (g32 * (vs_r - nv_r).astype(f32)).astype(f32)).astype(f32)).astype(f32)
            vs_c = (value_c[t] + (delta_c + (g32 * (vs_c - nv_c).astype(f32)).astype(f32)).astype(f32)).astype(f32)
            out['adv_r'][t], out['adv_c'][t] = adv_vr, adv_vc
            out['tgt_r'][t], out['tgt_c'][t] = vs_r, vs_c
            out['disc_ret'][t] = ret_r.astype(f32)
            nv_r, nv_c = value_r[t], value_c[t]
            continue
        if estimator == 'plain':
            adv_r64, adv_c64 = delta_r.astype(f64), delta_c.astype(f64)
        else:
            a_r = delta_r.astype(f64) + d_r * a_r
            a_c = delta_c.astype(f64) + d_c * a_c
            adv_r64, adv_c64 = a_r, a_c
        out['adv_r'][t] = adv_r64.astype(f32)
        out['adv_c'][t] = adv_c64.astype(f32)
        if estimator == 'gae':
            out['tgt_r'][t] = (adv_r64 + value_r[t].astype(f64)).astype(f32)
            out['tgt_c'][t] = (adv_c64 + value_c[t].astype(f64)).astype(f32)
        else:
            out['tgt_r'][t] = rtg_r.astype(f32)
            out['tgt_c'][t] = rtg_c.astype(f32)
        out['disc_ret'][t] = ret_r.astype(f32)
        nv_r, nv_c = value_r[t], value_c[t]
    return out


def gae_per_path(reward, cost, value_r, value_c, path_end, boot_r, boot_c, gamma, lam, lam_c,
                 penalty_coef=0.0, estimator='gae'):
    """(T, N) buffer processed the way the reference does it: env by env, path by path, through
    ``finish_path``.  Slow; used to pin ``gae_time_major`` and on small parity cases."""
    T, N = np.asarray(reward).shape
    out = {k: np.zeros((T, N), np.float32) for k in ('adv_r', 'tgt_r', 'adv_c', 'tgt_c', 'disc_ret')}
    for n in range(N):
        start = 0
        for t in range(T):
            if path_end[t][n]:
                sl = slice(start, t + 1)
                a_r, t_r, a_c, t_c, dr = finish_path(
                    reward[sl, n], cost[sl, n], value_r[sl, n], value_c[sl, n],
                    boot_r[t][n], boot_c[t][n], gamma, lam, lam_c, penalty_coef, estimator)
                out['adv_r'][sl, n], out['tgt_r'][sl, n] = a_r, t_r
                out['adv_c'][sl, n], out['tgt_c'][sl, n] = a_c, t_c
                out['disc_ret'][sl, n] = dr
                start = t + 1
    return out


# --------------------------------------------------------------------------------------------------
# K6: VectorOnPolicyBuffer.get  (omnisafe/common/buffer/vector_onpolicy_buffer.py:113-138,
#     omnisafe/utils/distributed.py:361-393)
# --------------------------------------------------------------------------------------------------


def dist_statistics_scalar(value: torch.Tensor):
    """world_size == 1 restatement of distributed.py:382-392: float32 torch.sum, population std."""
    global_sum = torch.sum(value)
    global_n = torch.tensor(len(value))
    mean = global_sum / global_n
    global_sum_sq = torch.sum((value - mean) ** 2)
    std = torch.sqrt(global_sum_sq / global_n)
    return mean, std


def env_major(x_tm: np.ndarray) -> np.ndarray:
    """(T, N, ...) time-major -> (N*T, ...) env-major, the order VectorOnPolicyBuffer.get
    concatenates sub-buffers in (vector_onpolicy_buffer.py:125-129)."""
    x = np.asarray(x_tm)
    return np.ascontiguousarray(np.swapaxes(x, 0, 1)).reshape((-1,) + x.shape[2:])


def buffer_get(adv_r_tm, adv_c_tm, standardized_adv_r=True, standardized_adv_c=True):
    """Standardisation of get(): adv_r <- (adv_r - mean)/(std + 1e-8) with population std;
    adv_c <- adv_c - mean_c (vector_onpolicy_buffer.py:131-136).  Returns env-major float32 arrays and
    the float32 statistics (mean_r, std_r, mean_c)."""
    a_r = torch.from_numpy(env_major(adv_r_tm).copy())
    a_c = torch.from_numpy(env_major(adv_c_tm).copy())
    mean_r, std_r = dist_statistics_scalar(a_r)
    mean_c, _ = dist_statistics_scalar(a_c)
    if standardized_adv_r:
        a_r = (a_r - mean_r) / (std_r + 1e-8)
    if standardized_adv_c:
        a_c = a_c - mean_c
    return a_r.numpy(), a_c.numpy(), (float(mean_r), float(std_r), float(mean_c))


# --------------------------------------------------------------------------------------------------
# K2: running observation normaliser (omnisafe/common/normalizer.py:88-139, envs/wrapper.py:231-241)
# --------------------------------------------------------------------------------------------------


class Normalizer:
    """Chan/Golub/LeVeque batch merge exactly as normalizer.py:109-139 (float32 state, int64 count),
    normalise = clamp((x - mean)/std, -clip, clip) with std floored at 1e-2; raw data returned
    untouched while count <= 1 (normalizer.py:102-107)."""

    def __init__(self, shape, clip=5.0):
        self.shape = tuple(shape)
        self.mean = torch.zeros(self.shape)
        self.sumsq = torch.zeros(self.shape)
        self.var = torch.zeros(self.shape)
        self.std = torch.zeros(self.shape)
        self.count = 0
        self.clip = clip * torch.ones(self.shape)
        self.first = True

    def push(self, raw: torch.Tensor) -> None:
        if raw.shape == self.shape:
            raw = raw.unsqueeze(0)
        if self.first:
            self.mean = torch.mean(raw, dim=0)
            self.sumsq = torch.sum((raw - self.mean) ** 2, dim=0)
            self.count = raw.shape[0]
            self.first = False
        else:
            count_raw = raw.shape[0]
            count = self.count + count_raw
            mean_raw = torch.mean(raw, dim=0)
            delta = mean_raw - self.mean
            self.mean = self.mean + delta * count_raw / count
            sumq_raw = torch.sum((raw - mean_raw) ** 2, dim=0)
            self.sumsq = self.sumsq + (sumq_raw + delta ** 2 * self.count * count_raw / count)
            self.count = count
        self.var = self.sumsq / (self.count - 1)
        self.std = torch.sqrt(self.var)
        self.std = torch.max(self.std, 1e-2 * torch.ones_like(self.std))

    def normalize(self, data: torch.Tensor) -> torch.Tensor:
        self.push(data)
        if self.count <= 1:
            return data
        out = (data - self.mean) / self.std
        return torch.clamp(out, -self.clip, self.clip)


# --------------------------------------------------------------------------------------------------
# K1: actor-critic MLPs (omnisafe/utils/model.py:73-111, models/actor/gaussian_learning_actor.py:29-139,
#     models/critic/v_critic.py:75-92, models/actor_critic/constraint_actor_critic.py:84-109)
# --------------------------------------------------------------------------------------------------

_ACT = {'identity': torch.nn.Identity, 'relu': torch.nn.ReLU, 'sigmoid': torch.nn.Sigmoid,
        'softplus': torch.nn.Softplus, 'tanh': torch.nn.Tanh}


def build_mlp(sizes, activation='tanh'):
    """Linear/activation stack with identity output and kaiming_uniform(a=sqrt(5)) weights
    (utils/model.py:36,103-111).  Parameter names are '0.weight','0.bias','2.weight',..."""
    layers = []
    for j in range(len(sizes) - 1):
        lin = torch.nn.Linear(sizes[j], sizes[j + 1])
        torch.nn.init.kaiming_uniform_(lin.weight, a=math.sqrt(5))
        layers += [lin, (_ACT[activation] if j < len(sizes) - 2 else torch.nn.Identity)()]
    return torch.nn.Sequential(*layers)


class Actor(torch.nn.Module):
    """Gaussian policy with state-independent log_std; parameter order log_std, mean.0.weight, ...
    (gaussian_learning_actor.py:55-60 -- nn.Module yields own parameters before sub-modules)."""

    def __init__(self, obs_dim, act_dim, hidden=(64, 64), activation='tanh'):
        super().__init__()
        self.mean = build_mlp([obs_dim, *hidden, act_dim], activation)
        self.log_std = torch.nn.Parameter(torch.zeros(act_dim))

    def dist(self, obs):
        return torch.distributions.Normal(self.mean(obs), torch.exp(self.log_std))


class Critic(torch.nn.Module):
    def __init__(self, obs_dim, hidden=(64, 64), activation='tanh'):
        super().__init__()
        self.critic_0 = build_mlp([obs_dim, *hidden, 1], activation)

    def forward(self, obs):
        return torch.squeeze(self.critic_0(obs), -1)


class ActorCritic:
    """ConstraintActorCritic restatement: three nets + three Adam optimisers
    (models/actor_critic/actor_critic.py:91-113, constraint_actor_critic.py:77-82)."""

    def __init__(self, obs_dim, act_dim, hidden=(64, 64), activation='tanh', actor_lr=3e-4,
                 critic_lr=3e-4):
        self.actor = Actor(obs_dim, act_dim, hidden, activation)
        self.reward_critic = Critic(obs_dim, hidden, activation)
        self.cost_critic = Critic(obs_dim, hidden, activation)
        self.actor_optimizer = (torch.optim.Adam(self.actor.parameters(), lr=actor_lr)
                                if actor_lr is not None else None)
        self.reward_critic_optimizer = torch.optim.Adam(self.reward_critic.parameters(), lr=critic_lr)
        self.cost_critic_optimizer = torch.optim.Adam(self.cost_critic.parameters(), lr=critic_lr)

    def step(self, obs, eps=None, deterministic=False):
        """constraint_actor_critic.py:102-109.  ``eps`` (standard normal, same shape as the action)
        replaces Normal.rsample's generator draw so a device kernel can be fed the same noise:
        rsample is mean + eps*std (torch.distributions.Normal.rsample)."""
        with torch.no_grad():
            value_r = self.reward_critic(obs)
            value_c = self.cost_critic(obs)
            d = self.actor.dist(obs)
            if deterministic:
                act = d.mean
            elif eps is None:
                act = d.rsample()
            else:
                act = d.mean + eps * d.stddev
            logp = d.log_prob(act).sum(axis=-1)
        return act, value_r, value_c, logp


def flat_params(module) -> torch.Tensor:
    """utils/tools.py:35-65 get_flat_params_from."""
    return torch.cat([p.data.view(-1) for p in module.parameters() if p.requires_grad])


def flat_grads(module) -> torch.Tensor:
    """utils/tools.py:68-91 get_flat_gradients_from."""
    return torch.cat([p.grad.view(-1) for p in module.parameters()
                      if p.requires_grad and p.grad is not None])


def set_flat_params(module, vals: torch.Tensor) -> None:
    """utils/tools.py:94-129 set_param_values_to_model."""
    i = 0
    for p in module.parameters():
        if p.requires_grad:
            n = p.numel()
            p.data = vals[i:i + n].view(p.shape).clone()
            i += n
    assert i == len(vals)


# --------------------------------------------------------------------------------------------------
# K7: Lagrange multiplier (omnisafe/common/lagrange.py:67-136)
# --------------------------------------------------------------------------------------------------


class Lagrange:
    def __init__(self, cost_limit, lagrangian_multiplier_init, lambda_lr, lambda_optimizer='Adam',
                 lagrangian_upper_bound=None):
        self.cost_limit = cost_limit
        self.lagrangian_upper_bound = lagrangian_upper_bound
        self.lagrangian_multiplier = torch.nn.Parameter(
            torch.as_tensor(max(lagrangian_multiplier_init, 0.0)), requires_grad=True)
        self.opt = getattr(torch.optim, lambda_optimizer)([self.lagrangian_multiplier], lr=lambda_lr)

    def update_lagrange_multiplier(self, Jc: float) -> None:
        self.opt.zero_grad()
        loss = -self.lagrangian_multiplier * (Jc - self.cost_limit)
        loss.backward()
        self.opt.step()
        self.lagrangian_multiplier.data.clamp_(0.0, self.lagrangian_upper_bound)


# --------------------------------------------------------------------------------------------------
# K9/K10/K11: PPO-Lag minibatch update (algorithms/on_policy/base/policy_gradient.py:345-524,
#     base/ppo.py:66-87, naive_lagrange/ppo_lag.py:82-102)
# --------------------------------------------------------------------------------------------------


def critic_step(critic, opt, obs, target, critic_norm_coef=0.001, max_grad_norm=40.0,
                use_critic_norm=True, use_max_grad_norm=True):
    """policy_gradient.py:428-445 (reward) / :468-485 (cost).  Returns the logged loss."""
    opt.zero_grad()
    loss = torch.nn.functional.mse_loss(critic(obs), target)
    if use_critic_norm:
        for p in critic.parameters():
            loss = loss + p.pow(2).sum() * critic_norm_coef
    loss.backward()
    if use_max_grad_norm:
        torch.nn.utils.clip_grad_norm_(critic.parameters(), max_grad_norm)
    opt.step()
    return float(loss.detach())


def ppo_loss_pi(actor, obs, act, logp, adv, clip=0.2, entropy_coef=0.0):
    """base/ppo.py:66-87.  Returns (loss, entropy, ratio mean)."""
    d = actor.dist(obs)
    logp_ = d.log_prob(act).sum(axis=-1)
    ratio = torch.exp(logp_ - logp)
    ratio_cliped = torch.clamp(ratio, 1 - clip, 1 + clip)
    loss = -torch.min(ratio * adv, ratio_cliped * adv).mean()
    loss = loss - entropy_coef * d.entropy().mean()
    return loss, float(d.entropy().mean().detach()), ratio.detach()


def pg_loss_pi(actor, obs, act, logp, adv):
    """policy_gradient.py:574-578 (TRPO / CPO surrogate: unclipped ratio * adv)."""
    d = actor.dist(obs)
    logp_ = d.log_prob(act).sum(axis=-1)
    ratio = torch.exp(logp_ - logp)
    return -(ratio * adv).mean(), float(d.entropy().mean().detach()), ratio.detach()


def lag_adv_surrogate(adv_r, adv_c, lam):
    """ppo_lag.py:101-102 / trpo_lag.py:99-100."""
    return (adv_r - lam * adv_c) / (1 + lam)


def actor_step(actor, opt, obs, act, logp, adv_r, adv_c, lam, clip=0.2, entropy_coef=0.0,
               max_grad_norm=40.0, use_max_grad_norm=True):
    """policy_gradient.py:514-524 with PPOLag's surrogate.  Returns (loss, entropy, ratio)."""
    adv = lag_adv_surrogate(adv_r, adv_c, lam)
    loss, ent, ratio = ppo_loss_pi(actor, obs, act, logp, adv, clip, entropy_coef)
    opt.zero_grad()
    loss.backward()
    if use_max_grad_norm:
        torch.nn.utils.clip_grad_norm_(actor.parameters(), max_grad_norm)
    opt.step()
    return float(loss.detach()), ent, ratio


def kl_old_new(actor, obs, old_mean, old_std):
    """policy_gradient.py:383-389: kl_divergence(old, new).sum(-1, keepdim=True).mean()."""
    with torch.no_grad():
        new = actor.dist(obs)
        old = torch.distributions.Normal(old_mean, old_std)
        return float(torch.distributions.kl.kl_divergence(old, new).sum(-1, keepdim=True).mean())


def ppolag_update(ac: ActorCritic, data: dict, lam: float, perms, batch_size=64, update_iters=40,
                  target_kl=0.02, kl_early_stop=True, clip=0.2, entropy_coef=0.0,
                  critic_norm_coef=0.001, max_grad_norm=40.0, use_critic_norm=True,
                  use_max_grad_norm=True, use_cost=True, max_minibatches=None):
    """PolicyGradient._update (policy_gradient.py:345-405) with the minibatch order injected:
    ``perms[i]`` is the sample permutation of iteration i (the reference draws it with
    torch.randperm on the CPU default generator through DataLoader's RandomSampler; the last partial
    minibatch is kept, drop_last=False).  ``data`` holds env-major float32 tensors obs, act, logp,
    target_value_r, target_value_c, adv_r, adv_c.  Returns a stats dict."""
    obs, act, logp = data['obs'], data['act'], data['logp']
    tgt_r, tgt_c, adv_r, adv_c = (data['target_value_r'], data['target_value_c'], data['adv_r'],
                                  data['adv_c'])
    with torch.no_grad():
        old = ac.actor.dist(obs)
        old_mean, old_std = old.mean.clone(), old.stddev.clone()
    M = obs.shape[0]
    stats = {'loss_r': [], 'loss_c': [], 'loss_pi': [], 'ratio_mean': [], 'entropy': []}
    update_counts, final_kl = 0, 0.0
    for i in range(update_iters):
        perm = torch.as_tensor(perms[i], dtype=torch.long)
        nmb = 0
        for s in range(0, M, batch_size):
            idx = perm[s:s + batch_size]
            stats['loss_r'].append(critic_step(ac.reward_critic, ac.reward_critic_optimizer, obs[idx],
                                               tgt_r[idx], critic_norm_coef, max_grad_norm,
                                               use_critic_norm, use_max_grad_norm))
            if use_cost:
                stats['loss_c'].append(critic_step(ac.cost_critic, ac.cost_critic_optimizer, obs[idx],
                                                   tgt_c[idx], critic_norm_coef, max_grad_norm,
                                                   use_critic_norm, use_max_grad_norm))
            lp, ent, ratio = actor_step(ac.actor, ac.actor_optimizer, obs[idx], act[idx], logp[idx],
                                        adv_r[idx], adv_c[idx], lam, clip, entropy_coef,
                                        max_grad_norm, use_max_grad_norm)
            stats['loss_pi'].append(lp)
            stats['entropy'].append(ent)
            stats['ratio_mean'].append(float(ratio.mean()))
            nmb += 1
            if max_minibatches is not None and nmb >= max_minibatches:
                break
        final_kl = kl_old_new(ac.actor, obs, old_mean, old_std)
        update_counts += 1
        if kl_early_stop and final_kl > target_kl:
            break
    stats['stop_iter'] = update_counts
    stats['kl'] = final_kl
    return stats


def _dp_step(module, opt, losses, max_grad_norm, use_max_grad_norm):
    """One data-parallel optimiser step of one network as the reference runs it on `world` ranks
    (policy_gradient.py:434-443 / 474-483 / 517-524 with utils/distributed.py:167-198): every rank back-propagates
    ITS loss, clips ITS gradient by ITS norm (`clip_grad_norm_` runs before `avg_grads`), then each parameter's
    gradient becomes all_reduce(SUM) / world and every rank takes the same Adam step.  `losses` = one closure per
    rank, each returning that rank's loss on its own minibatch.  The replicas are identical, so one copy of the
    network stands for all of them.  Rank order of the sum = rank 0 first (for two ranks a float sum is
    commutative: bit-exact against gloo)."""
    world = len(losses)
    grads, vals = [], []
    params = list(module.parameters())
    for fn in losses:
        opt.zero_grad()
        loss = fn()
        loss.backward()
        if use_max_grad_norm:
            torch.nn.utils.clip_grad_norm_(params, max_grad_norm)
        grads.append([p.grad.detach().clone() for p in params])
        vals.append(float(loss.detach()))
    for i, p in enumerate(params):
        acc = grads[0][i].clone()
        for r in range(1, world):
            acc += grads[r][i]
        p.grad = acc / world  # dist_avg: dist_sum(value) / world_size()
    opt.step()
    return vals


def ppolag_update_dp(ac: ActorCritic, datas: list, lam: float, perms: list, batch_size=64, update_iters=40,
                     target_kl=0.02, kl_early_stop=True, clip=0.2, entropy_coef=0.0, critic_norm_coef=0.001,
                     max_grad_norm=40.0, use_critic_norm=True, use_max_grad_norm=True, use_cost=True):
    """`ppolag_update` under data parallelism: `datas[r]` is rank r's (already globally standardised) batch,
    `perms[r][i]` rank r's permutation of pass i.  Per-rank statistics are returned as lists over ranks; the KL is
    the rank average (policy_gradient.py:390)."""
    world = len(datas)
    M = datas[0]['obs'].shape[0]
    olds = []
    with torch.no_grad():
        for d in datas:
            o = ac.actor.dist(d['obs'])
            olds.append((o.mean.clone(), o.stddev.clone()))
    stats = {'loss_r': [], 'loss_c': [], 'loss_pi': []}
    update_counts, final_kl = 0, 0.0
    for i in range(update_iters):
        pm = [torch.as_tensor(perms[r][i], dtype=torch.long) for r in range(world)]
        for s in range(0, M, batch_size):
            idx = [pm[r][s:s + batch_size] for r in range(world)]

            def critic_loss(critic, key, r):
                def fn():
                    d = datas[r]
                    loss = torch.nn.functional.mse_loss(critic(d['obs'][idx[r]]), d[key][idx[r]])
                    if use_critic_norm:
                        for p in critic.parameters():
                            loss = loss + p.pow(2).sum() * critic_norm_coef
                    return loss
                return fn

            def actor_loss(r):
                def fn():
                    d = datas[r]
                    adv = lag_adv_surrogate(d['adv_r'][idx[r]], d['adv_c'][idx[r]], lam)
                    return ppo_loss_pi(ac.actor, d['obs'][idx[r]], d['act'][idx[r]], d['logp'][idx[r]], adv, clip,
                                       entropy_coef)[0]
                return fn

            stats['loss_r'].append(_dp_step(ac.reward_critic, ac.reward_critic_optimizer,
                                            [critic_loss(ac.reward_critic, 'target_value_r', r) for r in range(world)],
                                            max_grad_norm, use_max_grad_norm))
            if use_cost:
                stats['loss_c'].append(_dp_step(ac.cost_critic, ac.cost_critic_optimizer,
                                                [critic_loss(ac.cost_critic, 'target_value_c', r) for r in range(world)],
                                                max_grad_norm, use_max_grad_norm))
            stats['loss_pi'].append(_dp_step(ac.actor, ac.actor_optimizer, [actor_loss(r) for r in range(world)],
                                             max_grad_norm, use_max_grad_norm))
        kls = [kl_old_new(ac.actor, datas[r]['obs'], *olds[r]) for r in range(world)]
        final_kl = float(np.float32(sum(np.float32(k) for k in kls)) / np.float32(world))
        update_counts += 1
        if kl_early_stop and final_kl > target_kl:
            break
    stats['stop_iter'], stats['kl'] = update_counts, final_kl
    return stats


def dp_standardise(raw_adv_r: list, raw_adv_c: list):
    """VectorOnPolicyBuffer.get's statistics under `world` ranks (vector_onpolicy_buffer.py:131-136 ->
    utils/distributed.py:382-392): global mean = all-reduced sum / all-reduced count; global (population) std from
    the all-reduced sum of squared deviations from that mean; adv_r <- (adv_r - mean) / (std + 1e-8), adv_c <- adv_c -
    mean_c.  Inputs: per-rank raw advantages, env-major float32 tensors.  float32 throughout, as the reference."""
    def stats(parts):
        gsum = sum(torch.sum(p) for p in parts)
        gn = torch.tensor(float(sum(len(p) for p in parts)))
        mean = gsum / gn
        gss = sum(torch.sum((p - mean) ** 2) for p in parts)
        return mean, torch.sqrt(gss / gn)

    mr, sr = stats(raw_adv_r)
    mc, _ = stats(raw_adv_c)
    return [(p - mr) / (sr + 1e-8) for p in raw_adv_r], [p - mc for p in raw_adv_c], (mr, sr, mc)


# --------------------------------------------------------------------------------------------------
# K12-K16: natural gradient machinery (utils/math.py:86-132, base/natural_pg.py:74-182,
#     base/trpo.py:56-222, second_order/cpo.py:57-462)
# --------------------------------------------------------------------------------------------------


def conjugate_gradients(fvp: Callable[[torch.Tensor], torch.Tensor], b: torch.Tensor, num_steps=10,
                        residual_tol=1e-10, eps=1e-6) -> torch.Tensor:
    """utils/math.py:116-132."""
    x = torch.zeros_like(b)
    r = b - fvp(x)
    p = r.clone()
    rdotr = torch.dot(r, r)
    for _ in range(num_steps):
        z = fvp(p)
        alpha = rdotr / (torch.dot(p, z) + eps)
        x = x + alpha * p
        r = r - alpha * z
        new_rdotr = torch.dot(r, r)
        if torch.sqrt(new_rdotr) < residual_tol:
            break
        mu = new_rdotr / (rdotr + eps)
        p = r + mu * p
        rdotr = new_rdotr
    return x


def fvp(actor: Actor, fvp_obs: torch.Tensor, v: torch.Tensor, cg_damping=0.1) -> torch.Tensor:
    """natural_pg.py:91-119: Hessian-vector product of mean KL(pi_old || pi_theta) -- NB ``.mean()``
    over all (M x D_a) elements -- by double backward, plus damping * v."""
    actor.zero_grad()
    q = actor.dist(fvp_obs)
    with torch.no_grad():
        p = actor.dist(fvp_obs)
    kl = torch.distributions.kl.kl_divergence(p, q).mean()
    params = tuple(actor.parameters())
    grads = torch.autograd.grad(kl, params, create_graph=True)
    flat_grad_kl = torch.cat([g.view(-1) for g in grads])
    kl_p = (flat_grad_kl * v).sum()
    grads = torch.autograd.grad(kl_p, params, retain_graph=False)
    flat = torch.cat([g.contiguous().view(-1) for g in grads])
    return flat + v * cg_damping


def trpo_search_step(actor, obs, act, logp, adv, step_direction, grads, loss_before, old_mean,
                     old_std, target_kl=0.01, total_steps=15, decay=0.8):
    """base/trpo.py:93-148.  Returns (accepted step vector, acceptance index (1-based; 0 = rejected))."""
    step_frac = 1.0
    theta_old = flat_params(actor)
    expected_improve = grads.dot(step_direction)
    final_kl = 0.0
    acceptance_step = 0
    for step in range(total_steps):
        new_theta = theta_old + step_frac * step_direction
        set_flat_params(actor, new_theta)
        with torch.no_grad():
            loss, _, _ = pg_loss_pi(actor, obs, act, logp, adv)
            q = actor.dist(obs)
            p = torch.distributions.Normal(old_mean, old_std)
            kl = torch.distributions.kl.kl_divergence(p, q).mean()
        loss_improve = loss_before - loss
        if not torch.isfinite(loss):
            pass
        elif loss_improve < 0:
            pass
        elif kl > target_kl:
            pass
        else:
            acceptance_step = step + 1
            final_kl = float(kl)
            break
        step_frac *= decay
    else:
        step_direction = torch.zeros_like(step_direction)
        acceptance_step = 0
    set_flat_params(actor, theta_old)
    return step_frac * step_direction, acceptance_step, float(expected_improve), final_kl


def _fvp_raw(actor: Actor, fvp_obs: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """natural_pg.py:91-111 up to (not including) the averaging across ranks and the damping term."""
    actor.zero_grad()
    q = actor.dist(fvp_obs)
    with torch.no_grad():
        p = actor.dist(fvp_obs)
    kl = torch.distributions.kl.kl_divergence(p, q).mean()
    params = tuple(actor.parameters())
    grads = torch.autograd.grad(kl, params, create_graph=True)
    flat_grad_kl = torch.cat([g.view(-1) for g in grads])
    kl_p = (flat_grad_kl * v).sum()
    grads = torch.autograd.grad(kl_p, params, retain_graph=False)
    return torch.cat([g.contiguous().view(-1) for g in grads])


def trpolag_update_dp(ac: ActorCritic, datas: list, lam: float, perms: list, batch_size=128, update_iters=10,
                      target_kl=0.01, cg_iters=15, cg_damping=0.1, fvp_sample_freq=1, critic_norm_coef=0.001,
                      max_grad_norm=40.0, use_critic_norm=True, use_max_grad_norm=True, use_cost=True,
                      total_steps=15, decay=0.8):
    """One TRPOLag `_update()` as the reference runs it on `world` ranks (natural_pg.py:185-240 -> base/trpo.py:150-222
    with utils/distributed.py:167-198, 259-275): the actor first, on every rank's FULL batch -- policy gradient
    averaged over the ranks (avg_grads), every Fisher-vector product averaged over the ranks before the damping term
    (natural_pg.py:113-117), the line search on the rank averages of loss improvement and KL (trpo.py:118-124) with the
    finiteness test on the rank's OWN loss -- then `update_iters` passes of critic steps, each clipped per rank and
    averaged (`_dp_step`).  The replicas are identical, so one copy of every network stands for all ranks; the
    decisions that use rank-local values are taken as rank 0 takes them (ranks only differ there if a loss is not
    finite).  `perms[r][i]`: rank r's minibatch order of critic pass i."""
    world = len(datas)
    actor = ac.actor
    w32 = np.float32(world)
    theta_old = flat_params(actor)
    advs = [lag_adv_surrogate(d['adv_r'], d['adv_c'], lam) for d in datas]
    # ---- policy gradient, averaged (dist_sum / world per parameter tensor)
    losses, gsum = [], None
    olds = []
    for r, d in enumerate(datas):
        actor.zero_grad()
        loss = pg_loss_pi(actor, d['obs'], d['act'], d['logp'], advs[r])[0]
        with torch.no_grad():
            o = actor.dist(d['obs'])
            olds.append((o.mean.clone(), o.stddev.clone()))
        loss.backward()
        g = flat_grads(actor)
        gsum = g.clone() if gsum is None else gsum + g
        losses.append(loss.detach().clone())
    grads = -(gsum / world)
    loss_before = sum(losses[1:], losses[0].clone()) / world  # dist_avg(loss): a 0-dim float32 tensor

    def fvp_avg(v):
        acc = None
        for d in datas:
            f = _fvp_raw(actor, d['obs'][::fvp_sample_freq], v)
            acc = f.clone() if acc is None else acc + f
        return acc / world + v * cg_damping

    x = conjugate_gradients(fvp_avg, grads, cg_iters)
    xHx = torch.dot(x, fvp_avg(x))
    alpha = torch.sqrt(2 * target_kl / (xHx + 1e-8))
    step_direction = x * alpha
    # ---- line search on the rank averages (trpo.py:93-148)
    step_frac, acceptance_step, final_kl = 1.0, 0, 0.0
    expected_improve = grads.dot(step_direction)
    for step in range(total_steps):
        set_flat_params(actor, theta_old + step_frac * step_direction)
        with torch.no_grad():
            ls, kls = [], []
            for r, d in enumerate(datas):
                ls.append(pg_loss_pi(actor, d['obs'], d['act'], d['logp'], advs[r])[0])
                qd = actor.dist(d['obs'])
                pd = torch.distributions.Normal(*olds[r])
                kls.append(torch.distributions.kl.kl_divergence(pd, qd).mean())
            kl = float((sum(kls[1:], kls[0].clone()) / world).mean())
            imps = [loss_before - l for l in ls]
            loss_improve = sum(imps[1:], imps[0].clone()) / world
        if not torch.isfinite(ls[0]):
            pass
        elif loss_improve.item() < 0:
            pass
        elif kl > target_kl:
            pass
        else:
            acceptance_step, final_kl = step + 1, kl
            break
        step_frac *= decay
    else:
        step_direction = torch.zeros_like(step_direction)
    final_step = step_frac * step_direction
    set_flat_params(actor, theta_old + final_step)
    stats = {'xHx': float(xHx), 'alpha': float(alpha), 'acceptance_step': acceptance_step, 'kl': final_kl,
             'final_step_norm': float(torch.norm(final_step)), 'gradient_norm': float(torch.norm(grads)),
             'expected_improve': float(expected_improve), 'loss_r': [], 'loss_c': []}
    # ---- critics (natural_pg.py:205-223): every step clipped per rank, then averaged
    M = datas[0]['obs'].shape[0]
    for i in range(update_iters):
        pm = [torch.as_tensor(perms[r][i], dtype=torch.long) for r in range(world)]
        for s in range(0, M, batch_size):
            idx = [pm[r][s:s + batch_size] for r in range(world)]

            def critic_loss(critic, key, r):
                def fn():
                    d = datas[r]
                    loss = torch.nn.functional.mse_loss(critic(d['obs'][idx[r]]), d[key][idx[r]])
                    if use_critic_norm:
                        for p in critic.parameters():
                            loss = loss + p.pow(2).sum() * critic_norm_coef
                    return loss
                return fn

            stats['loss_r'].append(_dp_step(ac.reward_critic, ac.reward_critic_optimizer,
                                            [critic_loss(ac.reward_critic, 'target_value_r', r) for r in range(world)],
                                            max_grad_norm, use_max_grad_norm))
            if use_cost:
                stats['loss_c'].append(_dp_step(ac.cost_critic, ac.cost_critic_optimizer,
                                                [critic_loss(ac.cost_critic, 'target_value_c', r) for r in range(world)],
                                                max_grad_norm, use_max_grad_norm))
    del w32
    return stats


def cpo_determine_case(b_grads, ep_costs, q, r, s, target_kl=0.01):
    """second_order/cpo.py:237-268.  Returns (optim_case, A, B)."""
    if b_grads.dot(b_grads) <= 1e-6 and ep_costs < 0:
        A = torch.zeros(1)
        B = torch.zeros(1)
        optim_case = 4
    else:
        assert torch.isfinite(r).all() and torch.isfinite(s).all()
        A = q - r ** 2 / (s + 1e-8)
        B = 2 * target_kl - ep_costs ** 2 / (s + 1e-8)
        if ep_costs < 0 and B < 0:
            optim_case = 3
        elif ep_costs < 0 <= B:
            optim_case = 2
        elif ep_costs >= 0 and B >= 0:
            optim_case = 1
        else:
            optim_case = 0
    return optim_case, A, B


def cpo_step_direction(optim_case, xHx, x, A, B, q, p, r, s, ep_costs, target_kl=0.01):
    """second_order/cpo.py:284-337.  Returns (step_direction, lambda_star, nu_star)."""
    if optim_case in (3, 4):
        alpha = torch.sqrt(2 * target_kl / (xHx + 1e-8))
        nu_star = torch.zeros(1)
        lambda_star = 1 / (alpha + 1e-8)
        step_direction = alpha * x
    elif optim_case in (1, 2):
        def project(data, low, high):
            return torch.clamp(data, low, high)

        lambda_a = torch.sqrt(A / B)
        lambda_b = torch.sqrt(q / (2 * target_kl))
        r_num = r.item()
        eps_cost = ep_costs + 1e-8
        if ep_costs < 0:
            lambda_a_star = project(lambda_a, torch.as_tensor(0.0), r_num / eps_cost)
            lambda_b_star = project(lambda_b, r_num / eps_cost, torch.as_tensor(torch.inf))
        else:
            lambda_a_star = project(lambda_a, r_num / eps_cost, torch.as_tensor(torch.inf))
            lambda_b_star = project(lambda_b, torch.as_tensor(0.0), r_num / eps_cost)

        def f_a(lam):
            return -0.5 * (A / (lam + 1e-8) + B * lam) - r * ep_costs / (s + 1e-8)

        def f_b(lam):
            return -0.5 * (q / (lam + 1e-8) + 2 * target_kl * lam)

        lambda_star = (lambda_a_star if f_a(lambda_a_star) >= f_b(lambda_b_star) else lambda_b_star)
        nu_star = torch.clamp(lambda_star * ep_costs - r, min=0) / (s + 1e-8)
        step_direction = 1.0 / (lambda_star + 1e-8) * (x - nu_star * p)
    else:
        lambda_star = torch.zeros(1)
        nu_star = torch.sqrt(2 * target_kl / (s + 1e-8))
        step_direction = -nu_star * p
    return step_direction, lambda_star, nu_star


def cpo_update_dp(ac: ActorCritic, datas: list, ep_costs: float, perms: list, batch_size=128, update_iters=10,
                  target_kl=0.01, cg_iters=15, cg_damping=0.1, fvp_sample_freq=1, critic_norm_coef=0.001,
                  max_grad_norm=40.0, use_critic_norm=True, use_max_grad_norm=True, total_steps=20, decay=0.8):
    """One CPO `_update()` as the reference runs it on `world` ranks (second_order/cpo.py:57-462 under
    utils/distributed.py:167-198, 259-275): reward AND cost policy gradients averaged over the ranks, every
    Fisher-vector product of both conjugate-gradient solves averaged before the damping term, the case analysis on the
    averaged quantities with `ep_costs` = cross-rank mean episode cost - cost limit, the two-constraint line search on
    the rank averages of reward improvement, cost difference and KL (cpo.py:106-178), then the critic passes of
    `trpolag_update_dp`.  One copy of every network stands for all (identical) replicas."""
    world = len(datas)
    actor = ac.actor
    theta_old = flat_params(actor)

    def avg(ts):
        return sum(ts[1:], ts[0].clone()) / world

    def loss_cost_of(d):
        dist = actor.dist(d['obs'])
        ratio = torch.exp(dist.log_prob(d['act']).sum(axis=-1) - d['logp'])
        return (ratio * d['adv_c']).mean()

    def averaged_gradient(loss_fn):
        ls, gsum = [], None
        for d in datas:
            actor.zero_grad()
            loss = loss_fn(d)
            loss.backward()
            g = flat_grads(actor)
            gsum = g.clone() if gsum is None else gsum + g
            ls.append(loss.detach().clone())
        return avg(ls), gsum / world

    olds = []
    with torch.no_grad():
        for d in datas:
            o = actor.dist(d['obs'])
            olds.append((o.mean.clone(), o.stddev.clone()))
    loss_reward_before, g_r = averaged_gradient(lambda d: pg_loss_pi(actor, d['obs'], d['act'], d['logp'], d['adv_r'])[0])
    grads = -g_r

    def fvp_avg(v):
        acc = None
        for d in datas:
            f = _fvp_raw(actor, d['obs'][::fvp_sample_freq], v)
            acc = f.clone() if acc is None else acc + f
        return acc / world + v * cg_damping

    x = conjugate_gradients(fvp_avg, grads, cg_iters)
    xHx = x.dot(fvp_avg(x))
    alpha = torch.sqrt(2 * target_kl / (xHx + 1e-8))
    loss_cost_before, b_grads = averaged_gradient(loss_cost_of)
    p = conjugate_gradients(fvp_avg, b_grads, cg_iters)
    q = xHx
    r = grads.dot(p)
    s = b_grads.dot(p)
    optim_case, A, B = cpo_determine_case(b_grads, ep_costs, q, r, s, target_kl)
    step_direction, lambda_star, nu_star = cpo_step_direction(optim_case, xHx, x, A, B, q, p, r, s, ep_costs, target_kl)
    # ---- line search (cpo.py:106-178)
    step_frac, acceptance_step = 1.0, 0
    kl = torch.zeros(1)
    for step in range(total_steps):
        set_flat_params(actor, theta_old + step_frac * step_direction)
        acceptance_step = step + 1
        with torch.no_grad():
            lr_, lc_, kls = [], [], []
            for i, d in enumerate(datas):
                lr_.append(pg_loss_pi(actor, d['obs'], d['act'], d['logp'], d['adv_r'])[0])
                lc_.append(loss_cost_of(d))
                kls.append(torch.distributions.kl.kl_divergence(torch.distributions.Normal(*olds[i]),
                                                                actor.dist(d['obs'])).mean())
            kl = avg(kls)
            loss_reward_improve = avg([loss_reward_before - l for l in lr_])
            loss_cost_diff = avg([l - loss_cost_before for l in lc_])
        if not torch.isfinite(kl):
            continue
        if loss_reward_improve < 0 if optim_case > 1 else False:
            pass
        elif loss_cost_diff > max(-ep_costs, 0):
            pass
        elif kl > target_kl:
            pass
        else:
            break
        step_frac *= decay
    else:
        step_direction = torch.zeros_like(step_direction)
        acceptance_step = 0
    final_step = step_frac * step_direction
    set_flat_params(actor, theta_old + final_step)
    stats = {'optim_case': int(optim_case), 'xHx': float(xHx), 'alpha': float(alpha), 'q': float(q), 'r': float(r),
             's': float(s), 'A': float(A), 'B': float(B), 'lambda_star': float(lambda_star), 'nu_star': float(nu_star),
             'acceptance_step': acceptance_step, 'kl': float(kl), 'final_step_norm': float(final_step.norm()),
             'gradient_norm': float(torch.norm(grads)), 'cost_gradient_norm': float(torch.norm(b_grads)),
             'loss_r': [], 'loss_c': []}
    M = datas[0]['obs'].shape[0]
    for i in range(update_iters):
        pm = [torch.as_tensor(perms[k][i], dtype=torch.long) for k in range(world)]
        for s0 in range(0, M, batch_size):
            idx = [pm[k][s0:s0 + batch_size] for k in range(world)]

            def critic_loss(critic, key, k):
                def fn():
                    d = datas[k]
                    loss = torch.nn.functional.mse_loss(critic(d['obs'][idx[k]]), d[key][idx[k]])
                    if use_critic_norm:
                        for w in critic.parameters():
                            loss = loss + w.pow(2).sum() * critic_norm_coef
                    return loss
                return fn

            stats['loss_r'].append(_dp_step(ac.reward_critic, ac.reward_critic_optimizer,
                                            [critic_loss(ac.reward_critic, 'target_value_r', k) for k in range(world)],
                                            max_grad_norm, use_max_grad_norm))
            stats['loss_c'].append(_dp_step(ac.cost_critic, ac.cost_critic_optimizer,
                                            [critic_loss(ac.cost_critic, 'target_value_c', k) for k in range(world)],
                                            max_grad_norm, use_max_grad_norm))
    return stats


# --------------------------------------------------------------------------------------------------
# a1: OnPolicyAdapter.rollout (omnisafe/adapter/onpolicy_adapter.py:58-136) on a *recorded* env trace
# --------------------------------------------------------------------------------------------------


def rollout_on_trace(ac: ActorCritic, norm: Normalizer, trace: dict, gamma=0.99, lam=0.95, lam_c=0.95,
                     penalty_coef=0.0, estimator='gae'):
    """Replays what the reference adapter does for one epoch, with the environment replaced by a
    recorded trace (raw, un-normalised env outputs) and the policy noise injected:

      trace['reset_obs'] (N, D_o); per step t: trace['obs'][t] (N, D_o) raw next obs (post auto-reset),
      'reward','cost' (N,), 'terminated','truncated' (N,) bool, 'final_obs'[t] (N, D_o) raw (rows valid
      where the env finished), 'eps'[t] (N, D_a) standard-normal draws of the vector policy step.

    Semantics restated: ObsNormalize pushes the finished rows of final_observation first, then the
    whole next-obs batch (envs/wrapper.py:231-241); bootstrap = 0 if terminated, V(final_obs) if
    truncated, V(next_obs) at epoch end (onpolicy_adapter.py:114-126); episode metrics are logged in
    (step, env) order (:128-134).  Returns time-major buffer arrays, GAE outputs and episode logs."""
    T = len(trace['reward'])
    N = trace['reset_obs'].shape[0]
    t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    obs = norm.normalize(t_(trace['reset_obs']).float())
    buf = {k: [] for k in ('obs', 'act', 'reward', 'cost', 'value_r', 'value_c', 'logp')}
    path_end = np.zeros((T, N), np.uint8)
    boot_r = np.zeros((T, N), np.float32)
    boot_c = np.zeros((T, N), np.float32)
    ep_ret, ep_cost, ep_len = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros(N, np.float32)
    episodes = []
    for t in range(T):
        act, v_r, v_c, logp = ac.step(obs, eps=t_(trace['eps'][t]))
        reward, cost = t_(trace['reward'][t]), t_(trace['cost'][t])
        term = np.asarray(trace['terminated'][t]).astype(bool)
        trunc = np.asarray(trace['truncated'][t]).astype(bool)
        fin = term | trunc
        final_norm = t_(trace['final_obs'][t]).float().clone()
        if fin.any():
            m = torch.from_numpy(fin)
            final_norm[m] = norm.normalize(final_norm[m])
        next_obs = norm.normalize(t_(trace['obs'][t]).float())
        ep_ret += trace['reward'][t]
        ep_cost += trace['cost'][t]
        ep_len += 1
        for k, v in (('obs', obs), ('act', act), ('reward', reward), ('cost', cost), ('value_r', v_r),
                     ('value_c', v_c), ('logp', logp)):
            buf[k].append(v.numpy().copy())
        obs = next_obs
        epoch_end = t >= T - 1
        for n in range(N):
            if epoch_end or term[n] or trunc[n]:
                lr_, lc_ = 0.0, 0.0
                if not term[n]:
                    if epoch_end:
                        _, a, b, _ = ac.step(obs[n], deterministic=True)
                        lr_, lc_ = float(a), float(b)
                    if trunc[n]:
                        _, a, b, _ = ac.step(final_norm[n], deterministic=True)
                        lr_, lc_ = float(a), float(b)
                if term[n] or trunc[n]:
                    episodes.append((float(ep_ret[n]), float(ep_cost[n]), float(ep_len[n])))
                    ep_ret[n] = ep_cost[n] = ep_len[n] = 0.0
                path_end[t, n] = 1
                boot_r[t, n], boot_c[t, n] = lr_, lc_
    buf = {k: np.stack(v) for k, v in buf.items()}
    gae = gae_time_major(buf['reward'], buf['cost'], buf['value_r'], buf['value_c'], path_end, boot_r,
                         boot_c, gamma, lam, lam_c, penalty_coef, estimator)
    return buf, gae, dict(path_end=path_end, boot_r=boot_r, boot_c=boot_c,
                          episodes=np.asarray(episodes, np.float32).reshape(-1, 3))


# --------------------------------------------------------------------------------------------------
# Learnable synthetic CMDP ("SynthReach-v0", obs 60 / act 2 like SafetyPointGoal1): NOT part of the
# reference -- Safety-Gymnasium is not installable here, so learning parity (north_star: "episode
# return/cost within +-1 sigma over 3 seeds") is checked on this stand-in, whose dynamics are stated
# once here and implemented by omnisafe_amd's device env (osa_reach_env_step) and by the CPU env the
# unmodified reference trains on (oracle/ref_harness.py: register_reach_env).
# --------------------------------------------------------------------------------------------------
REACH_STEP = np.float32(0.1)     # displacement per unit action
REACH_BOUND = np.float32(1.5)    # |position| clip
REACH_GOAL_R = np.float32(0.15)  # goal reached inside this radius: +1 reward, goal resampled
REACH_HAZARD_R = np.float32(0.3)  # cost 1 inside this radius of the hazard


def reach_env_step(state: np.ndarray, action: np.ndarray):
    """One transition of the point-reach task in float32, no fused multiply-adds.

    state (N, 6) = [p_x, p_y, g_x, g_y, h_x, h_y]; action (N, >=2).
    Returns (new_p (N,2), reward (N,), cost (N,), reached (N,) bool); the caller resamples the goal of
    the ``reached`` envs uniformly in [-1,1]^2 and builds obs = [p, g-p, h-p, 0...]."""
    s = np.asarray(state, dtype=np.float32)
    a = np.clip(np.asarray(action, dtype=np.float32)[:, :2], np.float32(-1), np.float32(1))
    p, g, h = s[:, 0:2], s[:, 2:4], s[:, 4:6]
    q = np.clip(p + REACH_STEP * a, -REACH_BOUND, REACH_BOUND).astype(np.float32)

    def dist(u, v):
        d = (u - v).astype(np.float32)
        return np.sqrt((d[:, 0] * d[:, 0]).astype(np.float32) + (d[:, 1] * d[:, 1]).astype(np.float32),
                       dtype=np.float32)

    d0, d1 = dist(p, g), dist(q, g)
    reached = d1 < REACH_GOAL_R
    reward = (d0 - d1).astype(np.float32) + reached.astype(np.float32)
    cost = (dist(q, h) < REACH_HAZARD_R).astype(np.float32)
    return q, reward, cost, reached


def reach_env_obs(state: np.ndarray, obs_dim: int) -> np.ndarray:
    s = np.asarray(state, dtype=np.float32)
    obs = np.zeros((s.shape[0], obs_dim), np.float32)
    obs[:, 0:2] = s[:, 0:2]
    obs[:, 2:4] = s[:, 2:4] - s[:, 0:2]
    obs[:, 4:6] = s[:, 4:6] - s[:, 0:2]
    return obs


# ---------------------------------------------------------------------------------------------------------------
# Minibatch shuffles.  The reference draws one torch.randperm per pass (DataLoader(shuffle=True),
# policy_gradient.py:357-377); the product evaluates a keyed bijection instead of sorting random keys
# (omnisafe_amd/csrc/shuffle_kernels.hip).  This is the numpy twin of that construction -- the checker of the index
# arithmetic (bit-exact), not a restatement of torch's generator.
SHUF_ROUNDS = 24
_M64 = (1 << 64) - 1


def _splitmix64(s: int):
    s = (s + 0x9E3779B97F4A7C15) & _M64
    z = s
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return s, z ^ (z >> 31)


def _shuf_f(x: np.ndarray, k0: int, k1: int) -> np.ndarray:
    x = x ^ np.uint32(k0)
    x = x ^ (x >> np.uint32(16))
    x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
    x = (x + np.uint32(k1)).astype(np.uint32)
    x = x ^ (x >> np.uint32(15))
    x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
    x = x ^ (x >> np.uint32(16))
    return x


def shuffle_rows(row_seeds, M: int) -> np.ndarray:
    """perm[row][i] = the cycle-walked image of i under the row's 24-step alternating Feistel bijection."""
    bits = 1
    while (1 << bits) < M:
        bits += 1
    lo_bits = bits >> 1
    hi_bits = bits - lo_bits
    mask_lo = np.uint32((1 << lo_bits) - 1)
    mask_hi = np.uint32((1 << hi_bits) - 1)
    out = np.empty((len(row_seeds), M), np.int64)
    with np.errstate(over='ignore'):
        for row, seed in enumerate(row_seeds):
            keys = []
            for t in range(SHUF_ROUNDS):
                s = (int(seed) & _M64) ^ ((0xD1B54A32D192ED03 * (t + 1)) & _M64)
                _, z = _splitmix64(s)
                keys += [z & 0xFFFFFFFF, z >> 32]
            v = np.arange(M, dtype=np.uint64)
            todo = np.ones(M, bool)
            while todo.any():
                w = v[todo]
                hi = ((w >> np.uint64(lo_bits)).astype(np.uint32)) & mask_hi
                lo = w.astype(np.uint32) & mask_lo
                for r in range(0, SHUF_ROUNDS, 2):
                    hi = hi ^ (_shuf_f(lo, keys[2 * r], keys[2 * r + 1]) & mask_hi)
                    lo = lo ^ (_shuf_f(hi, keys[2 * r + 2], keys[2 * r + 3]) & mask_lo)
                w = (hi.astype(np.uint64) << np.uint64(lo_bits)) | lo.astype(np.uint64)
                v[todo] = w
                idx = np.flatnonzero(todo)
                todo[idx[w < M]] = False
            out[row] = v.astype(np.int64)
    return out
