"""Device-resident OnPolicyAdapter.

Mirror of omnisafe/adapter/onpolicy_adapter.py:30-190 (+ the wrapper chain of online_adapter.py:85-140):
``OnPolicyAdapter(env_id, num_envs, seed, cfgs)`` and ``rollout(steps_per_epoch, agent, buffer,
logger)``.  The reference walks all N envs in Python at every step (2N tensor->bool host syncs, N x 7
index assignments into N buffers, a single-row ``agent.step`` per finished env); here one vector step
is a short, sync-free sequence of kernels:

    osa_policy_step          pi / V_r / V_c forward + sample + logp, written straight into buffer row t
    osa_action_scale         ActionScale wrapper (envs/wrapper.py:510-514)
    env.step                 (synthetic env: osa_synth_env_step)
    osa_normalizer_push/apply   ObsNormalize wrapper (final_observation rows first, then next obs,
                             envs/wrapper.py:231-241), result written into buffer row t+1
    osa_policy_step (critics)   bootstrap values V(final_obs) / V(next_obs), batched
    osa_rollout_post_step    episode accounting + bootstrap selection -> path_end / boot rows

Episode metrics are extracted from the device once per epoch and appended to the logger in the
reference's (step, env) order, so the window-100 statistics and the Lagrange update see the same data.
"""
from __future__ import annotations

import torch

from . import _lib
from . import envs as envs_mod
from .buffer import VectorOnPolicyBuffer
from .models import ConstraintActorCritic
from .normalizer import Normalizer


# host-side wrappers (TimeLimit / AutoReset) and the host-env bridge live in host_env.py; re-exported here because
# this is where the reference keeps its wrapper chain (adapter/online_adapter.py:120-140)
from .host_env import AutoReset, HostEnvBridge, TimeLimit, _EnvWrapper  # noqa: E402,F401


class OnPolicyAdapter:  # pylint: disable=too-many-instance-attributes
    def __init__(self, env_id: str, num_envs: int, seed: int, cfgs, env=None) -> None:
        self._lib = _lib.load(require_gpu=True)
        self._cfgs = cfgs
        self._env_id = env_id
        self._device = torch.device(cfgs.train_cfgs.device)
        env_cfgs = {}
        if hasattr(cfgs, 'env_cfgs') and cfgs.env_cfgs is not None:
            env_cfgs = cfgs.env_cfgs.todict() if hasattr(cfgs.env_cfgs, 'todict') else dict(cfgs.env_cfgs)
        self._env = env if env is not None else envs_mod.make(env_id, num_envs=num_envs,
                                                               device=self._device, **env_cfgs)
        # online_adapter.py:120-132: TimeLimit, then AutoReset, for envs that ask for them (single env, as in the
        # reference: wrapper.py:52,147); vector envs implement the gymnasium auto-reset convention themselves
        if getattr(self._env, 'need_time_limit_wrapper', False):
            assert self._env.max_episode_steps, ('You must define max_episode_steps as an integer\n'
                                                 'or cancel the use of the time_limit wrapper.')
            self._env = TimeLimit(self._env, time_limit=int(self._env.max_episode_steps), device=self._device)
        if getattr(self._env, 'need_auto_reset_wrapper', False):
            self._env = AutoReset(self._env, device=self._device)
        a = cfgs.algo_cfgs
        self._num_envs = int(self._env.num_envs)
        self._raw_obs_dim = int(self._env.observation_space.shape[0])
        self._obs_dim = self._raw_obs_dim + self._extra_obs_dims()  # what the agent / buffer see
        self._act_dim = int(self._env.action_space.shape[0])
        N, dev = self._num_envs, self._device
        f32 = dict(dtype=torch.float32, device=dev)
        self._obs_normalizer = (Normalizer((self._raw_obs_dim,), clip=5, device=dev)
                                if a.obs_normalize else None)
        # RewardNormalize / CostNormalize wrappers (envs/wrapper.py:280-423): scalar running statistics
        # over the batch of N rewards (costs) of each step, clip 5; episode metrics keep the ORIGINAL
        # reward / cost (info['original_reward'], onpolicy_adapter.py:155-156)
        self._reward_normalizer = (Normalizer((1,), clip=5, device=dev)
                                   if getattr(a, 'reward_normalize', False) else None)
        self._cost_normalizer = (Normalizer((1,), clip=5, device=dev)
                                 if getattr(a, 'cost_normalize', False) else None)
        # ActionScale(low=-1, high=1): agent acts in [-1, 1], env receives [space.low, space.high]
        self._old_min = torch.as_tensor(self._env.action_space.low, **f32).reshape(-1).contiguous()
        self._old_max = torch.as_tensor(self._env.action_space.high, **f32).reshape(-1).contiguous()
        self._act_env = torch.empty(N, self._act_dim, **f32)
        self._last_obs = torch.empty(N, self._obs_dim, **f32)
        self._final_norm = torch.empty(N, self._obs_dim, **f32)
        self._ep_ret = torch.zeros(N, **f32)
        self._ep_cost = torch.zeros(N, **f32)
        self._ep_len = torch.zeros(N, **f32)
        self._ep_rows: dict[str, torch.Tensor] = {}
        self._env.set_seed(seed)
        self._seed = seed

    # ------------------------------------------------------------------ reference surface
    def _extra_obs_dims(self) -> int:
        """Columns appended to the (normalised) env observation by the adapter (Saute/Simmer: 1)."""
        return 0

    def _after_reset(self, obs_rows: torch.Tensor) -> None:
        """Hook: the first observation rows of the epoch have been written (columns :raw_obs_dim)."""

    def _after_env_step(self, t: int, reward: torch.Tensor, cost: torch.Tensor, term: torch.Tensor,
                        trunc: torch.Tensor, next_rows: torch.Tensor, final_rows: torch.Tensor | None,
                        reward_row: torch.Tensor) -> None:
        """Hook between the wrapper chain and the bootstrap-value evaluation: may rewrite the reward row of
        the buffer and fill the adapter's extra observation columns of the next / final rows."""

    def _flush_extra(self, logger, idx: torch.Tensor) -> None:
        """Hook: extra per-episode metrics for the finished episodes `idx` (flat (t, n) indices)."""

    @property
    def observation_space(self):
        if self._obs_dim == self._raw_obs_dim:
            return self._env.observation_space
        from .spaces import Box

        return Box(-float('inf'), float('inf'), (self._obs_dim,))

    @property
    def action_space(self):
        from .spaces import Box

        return Box(-1.0, 1.0, (self._act_dim,))  # ActionScale's rescaled space

    @property
    def num_envs(self) -> int:
        return self._num_envs

    @property
    def env_spec_keys(self) -> list[str]:
        return list(getattr(self._env, 'env_spec_log', {}) or [])

    def save(self) -> dict:
        """online_adapter.py:232-246: objects to checkpoint."""
        saved = {}
        if self._obs_normalizer is not None:
            saved['obs_normalizer'] = self._obs_normalizer
        if self._reward_normalizer is not None:
            saved['reward_normalizer'] = self._reward_normalizer
        if self._cost_normalizer is not None:
            saved['cost_normalizer'] = self._cost_normalizer
        return saved

    def close(self) -> None:
        self._env.close()

    def reset(self, seed: int | None = None):
        obs, info = self._env.reset(seed=seed)
        return self._normalize(obs), info

    def _normalize(self, raw: torch.Tensor, mask: torch.Tensor | None = None,
                   out: torch.Tensor | None = None) -> torch.Tensor:
        raw = raw.reshape(self._num_envs, self._raw_obs_dim)
        if out is not None and out.shape[-1] != self._raw_obs_dim:
            out = out[:, :self._raw_obs_dim]  # the adapter's extra columns are filled by its hooks
        if self._obs_normalizer is None:
            if out is None:
                return raw
            out.copy_(raw)
            return out
        return self._obs_normalizer.normalize(raw, mask=mask, out=out)

    def _reset_log(self) -> None:
        self._ep_ret.zero_()
        self._ep_cost.zero_()
        self._ep_len.zero_()

    def _ensure_episode_rows(self, T: int) -> None:
        if not self._ep_rows or self._ep_rows['done'].shape[0] != T:
            N, dev = self._num_envs, self._device
            self._ep_rows = {'done': torch.zeros(T, N, dtype=torch.uint8, device=dev)}
            for k in ('ret', 'cost', 'len'):
                self._ep_rows[k] = torch.zeros(T, N, dtype=torch.float32, device=dev)

    # ------------------------------------------------------------------ rollout
    # The device part of an epoch's rollout is a FIXED sequence of launches on fixed device buffers when the
    # env is device-resident (`env.graph_safe`): which kernels run at step t depends only on t, T and the
    # horizon.  From the second epoch on it is therefore captured once as a hipGraph and replayed with ONE
    # host call per epoch instead of ~9 launches x T steps (the large-batch configuration of the benchmark is
    # otherwise bound by the host's launch rate, not by the GPU).  What changes between epochs lives in
    # device memory: the Philox stream positions of the env and of the policy noise (`commit()` /
    # `commit_rng()` advance their device-resident parts inside the graph), the normaliser state, the
    # network parameters.  OSA_ROLLOUT_GRAPH=0 disables it.
    _graph_safe_hooks = True

    def rollout(self, steps_per_epoch: int, agent: ConstraintActorCritic, buffer: VectorOnPolicyBuffer,
                logger) -> None:
        """onpolicy_adapter.py:58-136."""
        import os

        T = int(steps_per_epoch)
        self.last_rollout_path = 'launches'
        use_graph = (os.environ.get('OSA_ROLLOUT_GRAPH', '1') != '0' and self._graph_safe_hooks
                     and getattr(self._env, 'graph_safe', False) and hasattr(agent, 'commit_rng'))
        if not use_graph:
            self._rollout_device(T, agent, buffer)
            buffer.prefetch()
            self._flush_logs(logger, buffer)
            return
        st = self.__dict__.setdefault('_rollout_graph', {})
        # every by-value launch argument that can change between epochs is part of the key (a changed seed
        # re-captures instead of silently replaying the old stream)
        # (general networks: the layer-wise path's scratch block, whose address the captured launches bake in --
        # gmlp_ws() grows by reallocation, so a larger request after capture must re-capture, never replay)
        gws = getattr(agent, '_gws', None)
        key = (T, buffer.data['obs'].data_ptr(), agent.params.data_ptr(), self._num_envs,
               getattr(self._env, '_seed', None), getattr(agent, 'seed', None),
               int(gws.data_ptr()) if gws is not None else 0)
        if st.get('key') != key:  # first epoch (also sets kernel attributes, which must not happen under capture)
            st.clear()
            st.update(key=key, graph=None, failed=False)
            self._rollout_device(T, agent, buffer)
        elif st['graph'] is None and not st['failed']:
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):  # records the launches without executing them
                    self._rollout_device(T, agent, buffer)
                st['graph'] = g
                buffer.ptr = 0
                g.replay()
                buffer.ptr = T
            except Exception as exc:  # pragma: no cover - capture refused: stay on eager launches
                import warnings

                warnings.warn(f'omnisafe_amd: hipGraph capture of the rollout was refused ({exc!r}); the rollout '
                              'stays on eager launches (set OSA_ROLLOUT_GRAPH=0 to silence)', RuntimeWarning)
                st['failed'], st['graph'] = True, None
                buffer.ptr = 0
                self._rollout_device(T, agent, buffer)
        elif st['graph'] is not None:
            assert buffer.ptr == 0
            st['graph'].replay()
            buffer.ptr = T
        else:
            self._rollout_device(T, agent, buffer)
        self.last_rollout_graphed = st.get('graph') is not None
        # get()'s device work goes out before the host synchronises on the episode metrics (buffer.py:prefetch)
        buffer.prefetch()
        self._flush_logs(logger, buffer)

    def _rollout_device(self, T: int, agent: ConstraintActorCritic, buffer: VectorOnPolicyBuffer) -> None:
        """Everything of the rollout that runs on the device (no host synchronisation inside)."""
        lib, N = self._lib, self._num_envs
        assert buffer.size == T and buffer.num_buffers == N and buffer.ptr == 0
        self._reset_log()
        self._ensure_episode_rows(T)
        ep = self._ep_rows
        ep['done'].zero_()
        b = buffer.data
        obs_raw, _ = self._env.reset()  # the reference resets every epoch (:80)
        self._normalize(obs_raw, out=b['obs'][0])
        self._after_reset(b['obs'][0])
        import os

        # (agents whose step() does not know the fused epilogue -- test doubles that wrap step -- keep the launch)
        fuse_scale = (os.environ.get('OSA_FUSE_ACTION_SCALE', '1') != '0'
                      and getattr(agent.step, '__func__', None) is getattr(type(agent), 'step', None)
                      and hasattr(agent, '_rng_base'))
        for t in range(T):
            st = _lib.stream_ptr()
            obs = b['obs'][t]
            out = {'act': b['act'][t], 'value_r': b['value_r'][t], 'value_c': b['value_c'][t], 'logp': b['logp'][t]}
            if fuse_scale:  # ActionScale in the policy step's launch (one launch per vector step less)
                out['scale'] = (self._act_env, self._old_min, self._old_max, -1.0, 1.0)
                agent.step(obs, out=out)
            else:
                agent.step(obs, out=out)
                _lib.check(lib.osa_action_scale(_lib.ptr(b['act'][t]), self._act_dim, _lib.ptr(self._act_env),
                                                self._act_dim, N, self._act_dim, _lib.ptr(self._old_min),
                                                _lib.ptr(self._old_max), -1.0, 1.0, st), 'osa_action_scale')
            next_raw, reward, cost, terminated, truncated, info = self._env.step(self._act_env)
            reward, cost = reward.reshape(N), cost.reshape(N)
            # plain rows (no normaliser, no adapter hook that rewrites the reward row): written by the post-step
            # kernel below instead of two copy launches per vector step
            fuse_rows = type(self)._after_env_step is OnPolicyAdapter._after_env_step
            reward_row = cost_row = None
            if self._reward_normalizer is not None:
                self._reward_normalizer.normalize(reward.reshape(N, 1), out=b['reward'][t].view(N, 1))
            elif fuse_rows:
                reward_row = b['reward'][t]
            else:
                b['reward'][t].copy_(reward)
            if self._cost_normalizer is not None:
                self._cost_normalizer.normalize(cost.reshape(N, 1), out=b['cost'][t].view(N, 1))
            elif fuse_rows:
                cost_row = b['cost'][t]
            else:
                b['cost'][t].copy_(cost)
            reward = reward.to(torch.float32).contiguous()  # original values for the episode metrics
            cost = cost.to(torch.float32).contiguous()
            term = terminated.reshape(N).to(torch.uint8)
            trunc = truncated.reshape(N).to(torch.uint8)
            have_final = 'final_observation' in info
            if have_final:
                fmask = info.get('_final_observation', None)
                fmask = (term | trunc) if fmask is None else fmask.reshape(N).to(torch.uint8)
                self._normalize(info['final_observation'], mask=fmask, out=self._final_norm)
            epoch_end = t >= T - 1
            nxt = self._last_obs if epoch_end else b['obs'][t + 1]
            self._normalize(next_raw, out=nxt)
            self._after_env_step(t, reward, cost, term, trunc, nxt, self._final_norm if have_final else None,
                                 b['reward'][t])
            # Host env (HostEnvBridge): the bootstrap values and the episode accounting of this step are not needed
            # for the next action -- they are handed to the bridge, which enqueues them behind the next action's
            # device-to-host copy: the device does them while the host steps the env (round 4; device envs: in line)
            # (`defer` is a METHOD of the bridge: it resolves through the `__getattr__` forwarding of env wrappers such
            # as _EarlyTerminatedEnv, where an attribute ASSIGNMENT would land on the wrapper and never be seen)
            defer = getattr(self._env, 'defer', None) if getattr(self._env, 'host_resident', False) else None
            if callable(defer) and not epoch_end and os.environ.get('OSA_HOST_DEFER', '1') != '0':
                # (self._final_norm is rewritten by the NEXT step's normalisation, which is enqueued behind this work)
                final_src = self._final_norm if have_final else None

                def post_step(t=t, reward=reward, cost=cost, term=term, trunc=trunc, final_src=final_src,
                              reward_row=reward_row, cost_row=cost_row):
                    vfin = agent.values(final_src) if final_src is not None else (None, None)
                    _lib.check(lib.osa_rollout_post_step(
                        N, 0, _lib.ptr(reward), _lib.ptr(cost), _lib.ptr(term), _lib.ptr(trunc), None, None,
                        _lib.ptr(vfin[0]), _lib.ptr(vfin[1]), _lib.ptr(self._ep_ret), _lib.ptr(self._ep_cost),
                        _lib.ptr(self._ep_len), _lib.ptr(b['path_end'][t]), _lib.ptr(b['boot_r'][t]),
                        _lib.ptr(b['boot_c'][t]), _lib.ptr(ep['done'][t]), _lib.ptr(ep['ret'][t]),
                        _lib.ptr(ep['cost'][t]), _lib.ptr(ep['len'][t]), _lib.ptr(reward_row), _lib.ptr(cost_row),
                        _lib.stream_ptr()), 'osa_rollout_post_step')

                defer(post_step)
                buffer.advance()
                continue
            vfinal = agent.values(self._final_norm) if have_final else (None, None)
            vnext = agent.values(nxt) if epoch_end else (None, None)
            _lib.check(lib.osa_rollout_post_step(
                N, int(epoch_end), _lib.ptr(reward), _lib.ptr(cost), _lib.ptr(term),
                _lib.ptr(trunc), _lib.ptr(vnext[0]), _lib.ptr(vnext[1]), _lib.ptr(vfinal[0]),
                _lib.ptr(vfinal[1]), _lib.ptr(self._ep_ret), _lib.ptr(self._ep_cost),
                _lib.ptr(self._ep_len), _lib.ptr(b['path_end'][t]), _lib.ptr(b['boot_r'][t]),
                _lib.ptr(b['boot_c'][t]), _lib.ptr(ep['done'][t]), _lib.ptr(ep['ret'][t]),
                _lib.ptr(ep['cost'][t]), _lib.ptr(ep['len'][t]), _lib.ptr(reward_row), _lib.ptr(cost_row), st),
                'osa_rollout_post_step')
            buffer.advance()
        if hasattr(self._env, 'commit'):  # fold the epoch's stream positions into their device-resident parts
            self._env.commit()
        if hasattr(agent, 'commit_rng'):
            agent.commit_rng()

    def _flush_logs(self, logger, buffer: VectorOnPolicyBuffer) -> None:
        """One device->host transfer per epoch: finished episodes in (step, env) order
        (_log_metrics, :159-174) and the mean critic outputs (logger.store Value/*, :88-92)."""
        ep = self._ep_rows
        # the three windowed keys only ever keep the last `window` episodes: copy no more than those to the host;
        # un-windowed extras (Metrics/EpBudget of Saute / Simmer) average EVERY episode of the epoch
        window = logger.window_length('Metrics/EpRet') if hasattr(logger, 'window_length') else None
        import os

        if (type(self)._flush_extra is OnPolicyAdapter._flush_extra and os.environ.get('OSA_FLUSH_KERNEL', '1') != '0'
                and buffer.data['value_r'].numel() == ep['done'].numel()):
            # osa_episode_flush: the finished episodes compacted in (step, env) order + the two value means in two
            # launches and ONE synchronisation (the torch form below: nonzero, three gathers, two means, a stack)
            M = ep['done'].numel()
            fs = self.__dict__.setdefault('_flush_state', {})
            if fs.get('M') != M:
                dev = ep['done'].device
                fs.update(M=M, hdr=torch.zeros(4, dtype=torch.int32, device=dev),
                          idx=torch.empty(M, dtype=torch.int32, device=dev),
                          vals=torch.empty(4 * M, dtype=torch.float32, device=dev),
                          ws=torch.zeros(self._lib.osa_episode_flush_ws_doubles(M), dtype=torch.float64, device=dev))
            hdr = fs['hdr']
            _lib.check(self._lib.osa_episode_flush(
                _lib.ptr(ep['done']), _lib.ptr(ep['ret']), _lib.ptr(ep['cost']), _lib.ptr(ep['len']), None, M,
                _lib.ptr(buffer.data['value_r']), _lib.ptr(buffer.data['value_c']), _lib.ptr(hdr[0:1]),
                _lib.ptr(fs['idx']), _lib.ptr(fs['vals']), _lib.ptr(hdr[1:3]), _lib.ptr(fs['ws']),
                _lib.stream_ptr()), 'osa_episode_flush')
            if hasattr(logger, 'flush'):  # the previous epoch's csv row: host work while the device runs this rollout
                logger.flush()
            h = hdr.cpu()  # host sync (once per epoch)
            cnt = int(h[0])
            vmean = h[1:3].view(torch.float32).tolist()
            if cnt > 0:
                k0 = 0 if window is None else max(0, cnt - window)
                vals = fs['vals'].view(4, M)[:3, k0:cnt].cpu()
                logger.extend('Metrics/EpRet', vals[0].tolist())
                logger.extend('Metrics/EpCost', vals[1].tolist())
                logger.extend('Metrics/EpLen', vals[2].tolist())
            logger.store({'Value/reward': vmean[0]})
            if self._cfgs.algo_cfgs.use_cost:
                logger.store({'Value/cost': vmean[1]})
            return
        # everything the host will read is enqueued BEFORE the first synchronisation (each later one then finds its
        # result finished instead of leaving the device idle while the host enqueues the next reduction)
        vmean = torch.stack([buffer.data['value_r'].mean(), buffer.data['value_c'].mean()])
        if hasattr(logger, 'flush'):
            logger.flush()
        idx = ep['done'].reshape(-1).nonzero().reshape(-1)  # host sync (once per epoch)
        if idx.numel() > 0:
            widx = idx if window is None else idx[-window:]
            vals = torch.stack([ep[k].reshape(-1)[widx] for k in ('ret', 'cost', 'len')]).cpu()
            logger.extend('Metrics/EpRet', vals[0].tolist())
            logger.extend('Metrics/EpCost', vals[1].tolist())
            logger.extend('Metrics/EpLen', vals[2].tolist())
            self._flush_extra(logger, idx)
        vmean = vmean.tolist()
        logger.store({'Value/reward': vmean[0]})
        if self._cfgs.algo_cfgs.use_cost:
            logger.store({'Value/cost': vmean[1]})


class SauteAdapter(OnPolicyAdapter):
    """omnisafe/adapter/saute_adapter.py:31-259: the observation is augmented with the remaining safety
    budget z (1 at episode start, decremented by cost / budget and divided by saute_gamma every step), the
    reward is replaced by `unsafe_reward` once z <= 0.  One extra kernel per step (osa_saute_step)."""

    def __init__(self, env_id: str, num_envs: int, seed: int, cfgs, env=None) -> None:
        super().__init__(env_id, num_envs, seed, cfgs, env=env)
        a = cfgs.algo_cfgs
        assert not getattr(a, 'reward_normalize', False), 'Reward normalization is not supported'
        assert not getattr(a, 'cost_normalize', False), 'Cost normalization is not supported'
        N, dev = self._num_envs, self._device
        f32 = dict(dtype=torch.float32, device=dev)
        self._saute_gamma = float(a.saute_gamma)
        self._unsafe_reward = float(a.unsafe_reward)
        self._budget_scale = (1 - a.saute_gamma ** a.max_ep_len) / (1 - a.saute_gamma) / a.max_ep_len
        self._safety_budget = torch.full((N,), float(a.safety_budget * self._budget_scale), **f32)
        self._reset_value = torch.ones(N, **f32)
        self._safety_obs = torch.ones(N, **f32)
        self._ep_budget = torch.zeros(N, **f32)

    def _extra_obs_dims(self) -> int:
        return 1

    def _reset_log(self) -> None:
        super()._reset_log()
        self._ep_budget.zero_()

    def _ensure_episode_rows(self, T: int) -> None:
        super()._ensure_episode_rows(T)
        if 'budget' not in self._ep_rows or self._ep_rows['budget'].shape[0] != T:
            self._ep_rows['budget'] = torch.zeros(T, self._num_envs, dtype=torch.float32, device=self._device)

    def _epoch_start_value(self) -> torch.Tensor:
        return self._reset_value  # saute_adapter.py:118-121: ones

    def _after_reset(self, obs_rows: torch.Tensor) -> None:
        self._safety_obs.copy_(self._epoch_start_value())
        obs_rows[:, self._raw_obs_dim].copy_(self._safety_obs)

    def _after_env_step(self, t, reward, cost, term, trunc, next_rows, final_rows, reward_row) -> None:
        _lib.check(self._lib.osa_saute_step(
            self._num_envs, _lib.ptr(cost), _lib.ptr(reward), _lib.ptr(term), _lib.ptr(trunc),
            _lib.ptr(self._safety_obs), _lib.ptr(self._safety_budget), self._saute_gamma, self._unsafe_reward,
            _lib.ptr(self._reset_value), _lib.ptr(reward_row), _lib.ptr(next_rows), next_rows.stride(0),
            _lib.ptr(final_rows), 0 if final_rows is None else final_rows.stride(0), self._raw_obs_dim,
            _lib.ptr(self._ep_budget), _lib.ptr(self._ep_rows['budget'][t]), _lib.stream_ptr()), 'osa_saute_step')

    def _flush_extra(self, logger, idx: torch.Tensor) -> None:
        logger.extend('Metrics/EpBudget', self._ep_rows['budget'].reshape(-1)[idx].cpu().tolist())


class SimmerPIDController:
    """omnisafe/common/simmer_agent.py:93-189 (SimmerPIDAgent): PID on the blurred budget error, run once
    per epoch on host tensors with the reference's own sequence of torch CPU operations."""

    def __init__(self, cfgs, budget_bound: torch.Tensor, action_space=(-1, 1)) -> None:
        from collections import deque

        self._cfgs, self._budget_bound, self._action_space = cfgs, budget_bound, action_space
        self._sum_history = torch.zeros(1)
        self._prev_action = torch.zeros(1)
        self._prev_error = torch.zeros(1)
        self._prev_raw_action = torch.zeros(1)
        self._integral_history = deque([], maxlen=10)

    def act(self, safety_budget: torch.Tensor, observation: torch.Tensor) -> torch.Tensor:
        c = self._cfgs
        current_error = safety_budget - observation
        blured_error = c.polyak * self._prev_error + (1 - c.polyak) * current_error
        self._integral_history.append(blured_error)
        self._sum_history = torch.as_tensor(sum(self._integral_history))
        p_part = c.kp * blured_error
        i_part = c.ki * self._sum_history
        d_part = c.kd * (self._prev_action - self._prev_raw_action)
        raw_action = p_part + i_part + d_part
        action = torch.clamp(raw_action, min=self._action_space[0], max=self._action_space[1])
        next_safety_budget = torch.clamp(safety_budget + action, 1e-6 * torch.ones_like(safety_budget),
                                         self._budget_bound)
        action = next_safety_budget - safety_budget
        self._prev_action, self._prev_raw_action, self._prev_error = action, raw_action, blured_error
        return next_safety_budget


class SimmerAdapter(SauteAdapter):
    """omnisafe/adapter/simmer_adapter.py:31-131: Saute whose budget is steered by a PID controller; an
    epoch starts from the relative budget, episodes ending inside the epoch restart from 1 (the inherited
    SauteAdapter.step, saute_adapter.py:150-151)."""

    def __init__(self, env_id: str, num_envs: int, seed: int, cfgs, env=None) -> None:
        super().__init__(env_id, num_envs, seed, cfgs, env=env)
        a, N = cfgs.algo_cfgs, self._num_envs
        self._safety_budget_host = a.safety_budget * self._budget_scale * torch.ones(N, 1)
        self._upper_budget_host = a.upper_budget * self._budget_scale * torch.ones(N, 1)
        self._rel_budget = torch.ones(N, dtype=torch.float32, device=self._device)
        self._controller = SimmerPIDController(cfgs.control_cfgs, budget_bound=self._upper_budget_host)
        self._sync_budget()

    def _sync_budget(self) -> None:
        self._safety_budget.copy_(self._safety_budget_host.reshape(-1))
        self._rel_budget.copy_((self._safety_budget_host / self._upper_budget_host).reshape(-1))

    def _epoch_start_value(self) -> torch.Tensor:
        return self._rel_budget  # simmer_adapter.py:92-95

    def control_budget(self, ep_costs) -> None:
        """simmer_adapter.py:97-131."""
        ep_costs = torch.as_tensor(ep_costs, dtype=torch.float32).cpu() * self._budget_scale
        self._safety_budget_host = self._controller.act(safety_budget=self._safety_budget_host,
                                                        observation=ep_costs)
        self._sync_budget()


class _EarlyTerminatedEnv:
    """The step() override of the reference's EarlyTerminatedAdapter (early_terminated_adapter.py:50-88) as an
    env wrapper under the rollout loop: accumulate the cost, and once it exceeds the limit hand back a zero
    reward, terminated = 1 and the observation of a fresh episode.  The accumulated cost is only cleared
    there (not on a time-limit reset), as in the reference."""

    need_auto_reset_wrapper = False
    need_time_limit_wrapper = False
    graph_safe = False  # one host read of the accumulated cost per step decides what is launched next

    def __init__(self, env, adapter: 'EarlyTerminatedAdapter', cost_limit: float) -> None:
        self._env, self._adapter, self._cost_limit = env, adapter, float(cost_limit)
        self._cost_logger = torch.zeros(1, dtype=torch.float32, device=adapter._device)  # noqa: SLF001

    def __getattr__(self, name):  # spaces, num_envs, set_seed, close, ...
        return getattr(self._env, name)

    def reset(self, seed=None, options=None):
        return self._env.reset(seed=seed, options=options)

    def step(self, action: torch.Tensor):
        next_raw, reward, cost, terminated, truncated, info = self._env.step(action)
        self._cost_logger += cost.reshape(1).to(torch.float32)
        if float(self._cost_logger) > self._cost_limit:  # one env, one host read per step (as the reference)
            # the observation that is thrown away has already gone through ObsNormalize in the reference
            # (its running statistics saw it, early_terminated_adapter.py:78 -> wrapper.py:231-241)
            self._adapter._normalize(next_raw, out=self._adapter._scratch_obs)  # noqa: SLF001
            reward = torch.zeros_like(reward, dtype=torch.float32)
            terminated = torch.ones(1, dtype=torch.uint8, device=self._cost_logger.device)
            next_raw, _ = self._env.reset()
            self._cost_logger.zero_()
        return next_raw, reward, cost, terminated, truncated, info


class EarlyTerminatedAdapter(OnPolicyAdapter):
    """early_terminated_adapter.py:27-88: episodes end as soon as their accumulated cost exceeds
    ``algo_cfgs.cost_limit``.  Single environment only, like the reference (:42)."""

    _graph_safe_hooks = False

    def __init__(self, env_id: str, num_envs: int, seed: int, cfgs, env=None) -> None:
        assert num_envs == 1, 'EarlyTerminatedAdapter only supports num_envs=1.'
        super().__init__(env_id, num_envs, seed, cfgs, env=env)
        self._scratch_obs = torch.empty(1, self._obs_dim, dtype=torch.float32, device=self._device)
        self._cost_limit = float(cfgs.algo_cfgs.cost_limit)
        self._env = _EarlyTerminatedEnv(self._env, self, self._cost_limit)

