"""HBM-resident VectorOnPolicyBuffer.

Mirror of the reference interface (omnisafe/common/buffer/vector_onpolicy_buffer.py:54-138 and
onpolicy_buffer.py:84-238): same constructor arguments, ``store(**data)``, ``finish_path(last_value_r,
last_value_c, idx)`` and ``get() -> dict`` with the same 8 keys in the same env-major order, same
exception types.  The reference keeps N independent Python buffers and runs three fp64 Python loops
per path; here the rollout lives in one time-major (T, N, .) block of HBM and every path of every env
is finished by ONE launch of the HIP backward-scan kernel (osa_gae_scan), followed by a two-phase
statistics reduction and a fused standardise+transpose (osa_buffer_get).
"""
from __future__ import annotations

import torch

from . import _lib
from .spaces import is_box

_EST = {'gae': 0, 'gae-rtg': 1, 'plain': 2, 'vtrace': 3}


class VectorOnPolicyBuffer:  # pylint: disable=too-many-instance-attributes
    """Time-major device buffer for ``num_envs`` vectorised environments x ``size`` steps."""

    def __init__(  # pylint: disable=too-many-arguments
        self,
        obs_space,
        act_space,
        size: int,
        gamma: float,
        lam: float,
        lam_c: float,
        advantage_estimator: str,
        penalty_coefficient: float,
        standardized_adv_r: bool,
        standardized_adv_c: bool,
        num_envs: int = 1,
        device: torch.device | str = 'cuda:0',
        gae_variant: str | None = None,
    ) -> None:
        if num_envs < 1:
            raise ValueError('num_envs must be greater than 0.')  # vector_onpolicy_buffer.py:74-75
        if not is_box(obs_space) or not is_box(act_space):
            raise NotImplementedError  # buffer/base.py:73-80 (Box only)
        if advantage_estimator not in _EST:
            raise NotImplementedError  # onpolicy_buffer.py:333-334
        self._lib = _lib.load(require_gpu=True)
        self._device = torch.device(device)
        self._num_buffers = int(num_envs)
        self._size = int(size)
        self._gamma, self._lam, self._lam_c = float(gamma), float(lam), float(lam_c)
        self._estimator = _EST[advantage_estimator]
        # which backward-scan kernel finishes the paths: 'sequential' (one lane per env, bit-exact to the
        # reference), 'tiled' (time-parallel wavefront scan with LDS staging, tolerance-exact) or 'auto'
        # (by (T, N): see gae_variant_for).  Extension to the reference signature; env OSA_GAE_VARIANT.
        import os

        self._gae_variant = gae_variant or os.environ.get('OSA_GAE_VARIANT', 'auto')
        if self._gae_variant not in ('auto', 'sequential', 'tiled', 'chained'):
            raise ValueError(f'gae_variant must be auto, sequential, tiled or chained, not {self._gae_variant!r}')
        self._gae_ws: torch.Tensor | None = None  # carry workspace of the chained (time-split) scan
        self._penalty_coefficient = float(penalty_coefficient)
        self._standardized_adv_r = bool(standardized_adv_r)
        self._standardized_adv_c = bool(standardized_adv_c)
        self._obs_dim = int(obs_space.shape[0])
        self._act_dim = int(act_space.shape[0])
        T, N, dev = self._size, self._num_buffers, self._device
        f32 = dict(dtype=torch.float32, device=dev)
        # time-major rollout storage (a4: BaseBuffer/OnPolicyBuffer.__init__ allocations)
        self.data: dict[str, torch.Tensor] = {
            'obs': torch.zeros(T, N, self._obs_dim, **f32),
            'act': torch.zeros(T, N, self._act_dim, **f32),
        }
        for k in ('reward', 'cost', 'value_r', 'value_c', 'logp', 'adv_r', 'adv_c', 'target_value_r',
                  'target_value_c', 'discounted_ret', 'boot_r', 'boot_c'):
            self.data[k] = torch.zeros(T, N, **f32)
        self.data['path_end'] = torch.zeros(T, N, dtype=torch.uint8, device=dev)
        # env-major staging returned by get()
        M = T * N
        self._out = {
            'obs': torch.empty(M, self._obs_dim, **f32),
            'act': torch.empty(M, self._act_dim, **f32),
        }
        for k in ('logp', 'target_value_r', 'target_value_c', 'adv_r', 'adv_c', 'discounted_ret'):
            self._out[k] = torch.empty(M, **f32)
        self._stats = torch.zeros(8, dtype=torch.float64, device=dev)
        self._ws = torch.empty(self._lib.osa_reduce_ws_bytes() // 8, dtype=torch.float64, device=dev)
        self._ptr = 0
        self._prefetched = False

    # ------------------------------------------------------------------ properties
    @property
    def num_buffers(self) -> int:
        return self._num_buffers

    @property
    def size(self) -> int:
        return self._size

    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def standardized_adv_r(self) -> bool:
        return self._standardized_adv_r

    @property
    def standardized_adv_c(self) -> bool:
        return self._standardized_adv_c

    @property
    def buffers(self) -> list['_EnvBufferView']:
        """Per-env views with the attributes the reference's ``OnPolicyBuffer`` objects expose
        (vector_onpolicy_buffer.py:77-89: ``vector_buffer.buffers[idx].data[key]``, ``.ptr``,
        ``.path_start_idx``).  There is one storage block, so these are strided views of it."""
        return [_EnvBufferView(self, idx) for idx in range(self._num_buffers)]

    @property
    def stats(self) -> torch.Tensor:
        """Device statistics [sum_r, sum_c, n, sumsq_r, mean_r, mean_c, std_r, -] of the last get()."""
        return self._stats

    # ------------------------------------------------------------------ store / finish_path
    def row(self, key: str, t: int | None = None) -> torch.Tensor:
        """Row ``t`` (default: the current write row) of a time-major array -- producers may write into
        it directly instead of calling :meth:`store`."""
        return self.data[key][self.ptr if t is None else t]

    def advance(self) -> None:
        """Advance the write pointer after a producer filled ``row(...)`` in place."""
        assert self.ptr < self._size, 'No more space in the buffer!'
        self.ptr += 1
        self._prefetched = False

    def store(self, **data: torch.Tensor) -> None:
        """Store one vector step (reference keys: obs, act, reward, cost, value_r, value_c, logp)."""
        assert self.ptr < self._size, 'No more space in the buffer!'  # onpolicy_buffer.py:143
        self._prefetched = False
        d = {k: v.to(self._device, torch.float32).contiguous() for k, v in data.items()}
        N = self._num_buffers
        obs = d['obs'].reshape(N, self._obs_dim)
        act = d['act'].reshape(N, self._act_dim)
        b = self.data
        _lib.check(self._lib.osa_buffer_store_step(
            self.ptr, N, self._obs_dim, self._act_dim, _lib.ptr(obs), self._obs_dim, _lib.ptr(act),
            self._act_dim, _lib.ptr(d['reward'].reshape(N)), _lib.ptr(d['cost'].reshape(N)),
            _lib.ptr(d['value_r'].reshape(N)), _lib.ptr(d['value_c'].reshape(N)),
            _lib.ptr(d['logp'].reshape(N)), _lib.ptr(b['obs']), self._obs_dim, _lib.ptr(b['act']),
            self._act_dim, _lib.ptr(b['reward']), _lib.ptr(b['cost']), _lib.ptr(b['value_r']),
            _lib.ptr(b['value_c']), _lib.ptr(b['logp']), _lib.stream_ptr()), 'osa_buffer_store_step')
        self.ptr += 1

    def finish_path(self, last_value_r: torch.Tensor | None = None,
                    last_value_c: torch.Tensor | None = None, idx: int = 0) -> None:
        """Reference-compatible per-env call: ends env ``idx``'s current path after the last stored
        step with the given bootstrap values (defaults 0).  No arithmetic happens here -- the whole
        buffer is scanned once in :meth:`get`.  Vector callers should use :meth:`finish_paths`."""
        t = self.ptr - 1
        assert t >= 0, 'finish_path() before any store()'
        self._prefetched = False
        self.data['path_end'][t, idx] = 1
        for key, val in (('boot_r', last_value_r), ('boot_c', last_value_c)):
            if val is None:
                self.data[key][t, idx] = 0.0
            else:
                self.data[key][t, idx] = torch.as_tensor(val, dtype=torch.float32).reshape(-1)[0].to(
                    self._device, non_blocking=True)

    def finish_paths(self, mask: torch.Tensor, last_value_r: torch.Tensor,
                     last_value_c: torch.Tensor) -> None:
        """Vector form: ends the path of every env with ``mask[n]`` after the last stored step."""
        t = self.ptr - 1
        assert t >= 0, 'finish_paths() before any store()'
        self._prefetched = False
        m = mask.to(self._device).bool()
        self.data['path_end'][t] = m.to(torch.uint8)
        self.data['boot_r'][t] = torch.where(m, last_value_r.to(self._device, torch.float32), 0.0)
        self.data['boot_c'][t] = torch.where(m, last_value_c.to(self._device, torch.float32), 0.0)

    # ------------------------------------------------------------------ get
    @staticmethod
    def gae_variant_for(T: int, N: int, estimator: int) -> str:
        """The (T, N) rule of 'auto' (measured on MI355X, profiles/r3_gae_bandwidth.md):
          * `sequential` (lane per env, bit-exact): T < 64 -- the rollout-shaped buffers (BASELINE config 2:
            T = 16), where one lane walks a handful of steps and N alone fills the chip; v-trace always
            (a float32 chain);
          * `tiled` (time-parallel wavefront scan, 16 envs x 64 steps per tile): short horizons with few envs
            (64 <= T <= 256 and N <= 8192), where the whole buffer is a few hundred tiles;
          * `chained` (time split over workgroups, 64 envs x 16 steps per wave in registers, carries by decoupled
            look-back): everything else -- long horizons (BASELINE config 1: T = 5000, N = 4: 18 us against 108 us
            tiled) and large buffers (4096 x 4096: 3.7 TB/s against 2.5 TB/s tiled; 256 x 65 536: 3.4 against 2.3)."""
        if estimator == _EST['vtrace'] or T < 64:
            return 'sequential'
        if T <= 256 and N <= 8192:
            return 'tiled'
        return 'chained'

    def compute_advantages(self) -> None:
        """K5: one backward scan over the (T, N) buffer (osa_gae_scan / osa_gae_scan_tiled)."""
        b, T, N = self.data, self._size, self._num_buffers
        variant = self._gae_variant
        if variant == 'auto':
            variant = self.gae_variant_for(T, N, self._estimator)
        elif variant in ('tiled', 'chained') and self._estimator == _EST['vtrace']:
            variant = 'sequential'
        self.last_gae_variant = variant
        if variant == 'chained':
            need = self._lib.osa_gae_chained_ws_doubles(T, N)
            if self._gae_ws is None or self._gae_ws.numel() != need:
                # (zeros: the last double is the scan's sticky time-out word, never reset by a launch)
                self._gae_ws = torch.zeros(need, dtype=torch.float64, device=self._device)
                self._gae_ws_shape = (T, N)
            _lib.check(self._lib.osa_gae_scan_chained(
                _lib.ptr(b['reward']), _lib.ptr(b['cost']), _lib.ptr(b['value_r']), _lib.ptr(b['value_c']),
                _lib.ptr(b['path_end']), _lib.ptr(b['boot_r']), _lib.ptr(b['boot_c']), T, N, self._gamma,
                self._lam, self._lam_c, self._penalty_coefficient, self._estimator, _lib.ptr(b['adv_r']),
                _lib.ptr(b['adv_c']), _lib.ptr(b['target_value_r']), _lib.ptr(b['target_value_c']),
                _lib.ptr(b['discounted_ret']), _lib.ptr(self._gae_ws), _lib.stream_ptr()), 'osa_gae_scan_chained')
            return
        fn = self._lib.osa_gae_scan_tiled if variant == 'tiled' else self._lib.osa_gae_scan
        _lib.check(fn(
            _lib.ptr(b['reward']), _lib.ptr(b['cost']), _lib.ptr(b['value_r']), _lib.ptr(b['value_c']),
            _lib.ptr(b['path_end']), _lib.ptr(b['boot_r']), _lib.ptr(b['boot_c']), T, N, self._gamma,
            self._lam, self._lam_c, self._penalty_coefficient, self._estimator, _lib.ptr(b['adv_r']),
            _lib.ptr(b['adv_c']), _lib.ptr(b['target_value_r']), _lib.ptr(b['target_value_c']),
            _lib.ptr(b['discounted_ret']), _lib.stream_ptr()), 'osa_gae_scan')

    @property
    def ptr(self) -> int:
        return self._ptr

    @ptr.setter
    def ptr(self, value: int) -> None:
        """Every write of the pointer -- advance / store, get()'s reset, and the places that set it by hand (the
        adapter's hipGraph replay: `ptr = 0 ... replay ... ptr = T`; the timing tools) -- invalidates a prefetched
        batch: get() then re-assembles from the rows the buffer holds NOW, on the eager and the replayed path alike
        (round-3 advisor finding: a second replay without a get() in between returned the previous rollout's batch)."""
        self._ptr = int(value)
        self._prefetched = False

    def get(self) -> dict[str, torch.Tensor]:
        """Finish all paths, standardise, and return the env-major batch (reference key set).

        The buffer must be full (`ptr == size`): the reference's `get()` is only ever called then
        (policy_gradient.py:345-349) and a partially filled buffer would feed stale rows of the previous epoch
        into the advantages and their statistics.  The returned tensors alias the buffer's staging block: the
        next `get()` overwrites them (the reference returns fresh concatenations)."""
        assert self.ptr == self._size, f'get() on a partially filled buffer (ptr {self.ptr} of {self._size})'
        if not self._prefetched:  # (else: enqueued by prefetch() right behind the rollout)
            self._assemble()
        self.ptr = 0  # (also clears the prefetch mark)
        self.data['path_end'].zero_()
        return dict(self._out)

    def check_gae_sync(self) -> None:
        """Raise if a lane of the time-split GAE scan ever gave up waiting for a carry (its advantages and targets
        are NaN then).  A 4-byte synchronous copy: the algorithms call it once per update where the stream is drained
        anyway (behind the update's statistics)."""
        if self._gae_ws is None:
            return
        import ctypes as C

        flag = C.c_int(0)
        T, N = self._gae_ws_shape
        _lib.check(self._lib.osa_gae_chained_timed_out(_lib.ptr(self._gae_ws), T, N, C.byref(flag)),
                   'osa_gae_chained_timed_out')
        if flag.value:
            raise _lib.OsaError('osa_gae_scan_chained: a carry never arrived within the bounded spin (device shared '
                                'with another long-running kernel?) -- advantages of this epoch are invalid; set '
                                'OSA_GAE_VARIANT=sequential')

    def prefetch(self) -> None:
        """Enqueue get()'s device work now -- advantages, their statistics, the env-major batch -- and leave the
        buffer's visible state (ptr, path_end, the rows) as it is: the rollout adapter calls this right behind the last
        vector step, BEFORE it synchronises to read the episode metrics, so the device does not sit idle while the
        host walks from the logger through the Lagrange update to the first launch of get()."""
        if self.ptr == self._size and not self._prefetched:
            self._assemble()
            self._prefetched = True

    def _assemble(self) -> None:
        from . import distributed as dist

        b, T, N = self.data, self._size, self._num_buffers
        M = T * N
        st = _lib.stream_ptr()
        self.compute_advantages()
        _lib.check(self._lib.osa_adv_stats_phase1(_lib.ptr(b['adv_r']), _lib.ptr(b['adv_c']), M,
                                                  _lib.ptr(self._ws), _lib.ptr(self._stats), st),
                   'osa_adv_stats_phase1')
        if dist.collectives_active():
            dist.all_reduce_sum_(self._stats[0:3])
        _lib.check(self._lib.osa_adv_stats_phase2(_lib.ptr(b['adv_r']), M, _lib.ptr(self._ws),
                                                  _lib.ptr(self._stats), st), 'osa_adv_stats_phase2')
        if dist.collectives_active():
            dist.all_reduce_sum_(self._stats[3:4])
        o = self._out
        _lib.check(self._lib.osa_buffer_get(
            T, N, self._obs_dim, self._act_dim, _lib.ptr(b['obs']), self._obs_dim, _lib.ptr(b['act']),
            self._act_dim, _lib.ptr(b['logp']), _lib.ptr(b['target_value_r']),
            _lib.ptr(b['target_value_c']), _lib.ptr(b['adv_r']), _lib.ptr(b['adv_c']),
            _lib.ptr(b['discounted_ret']), _lib.ptr(self._stats), int(self._standardized_adv_r),
            int(self._standardized_adv_c), _lib.ptr(o['obs']), self._obs_dim, _lib.ptr(o['act']),
            self._act_dim, _lib.ptr(o['logp']), _lib.ptr(o['target_value_r']),
            _lib.ptr(o['target_value_c']), _lib.ptr(o['adv_r']), _lib.ptr(o['adv_c']),
            _lib.ptr(o['discounted_ret']), st), 'osa_buffer_get')


class _EnvBufferView:
    """What ``VectorOnPolicyBuffer.buffers[idx]`` looks like to code written against the reference:
    ``data[key]`` is env ``idx``'s (size, ...) column of the time-major block (a view: writes go through),
    ``ptr`` the shared write pointer, ``path_start_idx`` the step after the env's last finished path."""

    def __init__(self, owner: VectorOnPolicyBuffer, idx: int) -> None:
        self._owner, self._idx = owner, idx

    @property
    def data(self) -> dict[str, torch.Tensor]:
        return {k: v[:, self._idx] for k, v in self._owner.data.items()}

    @property
    def ptr(self) -> int:
        return self._owner.ptr

    @property
    def path_start_idx(self) -> int:
        ends = self._owner.data['path_end'][:self._owner.ptr, self._idx].nonzero()
        return int(ends[-1]) + 1 if ends.numel() else 0

