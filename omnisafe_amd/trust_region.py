"""Natural-gradient machinery on the device (K12-K16): full-batch policy gradient, Fisher-vector
product, conjugate gradients, batched line-search evaluation.

Restates the numerical core of omnisafe/algorithms/on_policy/base/natural_pg.py:74-182,
base/trpo.py:56-222 and second_order/cpo.py:57-462 on PADDED flat actor vectors (models.Layout):
  * g = -grad(-mean(ratio * adv))                 one launch of the matrix-core gradient kernel
  * F v                                           JVP -> backward, no autograd double backward
  * x = CG(F + damping I, g)                      device-resident scalars, no per-iteration host sync
                                                  (the reference syncs on `sqrt(r.r) < tol` every step)
  * line search                                   all candidates theta_old + decay^j * step evaluated
                                                  back to back, ONE host sync, then the reference's
                                                  sequential acceptance rule is applied to the results
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from . import distributed as dist
from .models import ConstraintActorCritic, HParams


class TrustRegionSolver:  # pylint: disable=too-many-instance-attributes
    def __init__(self, ac: ConstraintActorCritic, cg_iters: int, cg_damping: float, fvp_sample_freq: int = 1,
                 max_blocks: int = 256) -> None:
        self.ac, self.lib = ac, _lib.load(require_gpu=True)
        self.cg_iters, self.cg_damping, self.fvp_sample_freq = int(cg_iters), float(cg_damping), int(fvp_sample_freq)
        self.max_blocks = max_blocks
        dev, P = ac.device, ac.layout.P
        f32 = dict(dtype=torch.float32, device=dev)
        nws = self.lib.osa_minibatch_ws_floats(ac.obs_dim, ac.act_dim, ac.hidden, max_blocks)
        self._ws = torch.zeros(max(nws, 1), **f32)  # tail = arrival tickets (start at 0)
        self._stats = torch.zeros(16, **f32)
        self._eval_ws = torch.empty(4096, dtype=torch.float64, device=dev)
        self._scal = torch.zeros(4, **f32)
        self._hp = HParams(beta1=0.9, beta2=0.999, adam_eps=1e-8, use_cost=1)
        self._vecs = {k: torch.zeros(P, **f32) for k in ('x', 'r', 'p', 'z', 'raw')}
        self._old_mean: torch.Tensor | None = None
        self._old_log_std = torch.zeros(ac.layout.OUTP, **f32)
        self._fvp_obs: torch.Tensor | None = None
        # optional profiling: (start, end) HIP events around every Fisher-vector product's kernel (bench.py --algo CPO)
        self.profile_events: list | None = None
        self.fvp_calls = 0
        self.cg_solves = 0  # the reference's CG evaluates F(0) once more per solve (math.py:116-118: r = b - Ax(x))

    # ------------------------------------------------------------------ gradient
    def actor_loss_grad(self, data: dict, adv_key_r: str, adv_key_c: str, lagrange: torch.Tensor):
        """Full-batch gradient of L = -mean(ratio * (adv_r - l adv_c)/(1 + l)) (policy_gradient.py:574-578
        with the caller's surrogate).  Returns (loss device scalar, padded gradient [P], rank-averaged:
        distributed.avg_grads / dist_avg, trpo.py:181-185)."""
        ac, lib = self.ac, self.lib
        M = data['obs'].shape[0]
        if getattr(ac, 'general', False):
            ws, nws = ac.gmlp_ws(M)
            _lib.check(lib.osa_gmlp_minibatch(
                C.byref(ac.desc), _lib.ptr(ac.params), _lib.ptr(ac.adam_m), _lib.ptr(ac.adam_v), _lib.ptr(ac.adam_step),
                _lib.ptr(ac.grads), _lib.ptr(data['obs']), data['obs'].stride(0), _lib.ptr(data['act']),
                data['act'].stride(0), _lib.ptr(data['logp']), _lib.ptr(data['target_value_r']),
                _lib.ptr(data['target_value_c']), _lib.ptr(data[adv_key_r]), _lib.ptr(data[adv_key_c]), None, M,
                _lib.ptr(lagrange), C.byref(self._hp), 1, 2, 1, None, 0.0, _lib.ptr(ws), nws, _lib.ptr(self._stats),
                _lib.stream_ptr()), 'osa_gmlp_minibatch(full-batch actor gradient)')
            grad = ac.grads[0].clone()
            loss = self._stats[2:3].clone()
            dist.all_reduce_avg_(grad)
            dist.all_reduce_avg_(loss)
            return loss, grad
        _lib.check(lib.osa_ppo_minibatch(
            ac.obs_dim, ac.act_dim, ac.hidden, _lib.ptr(ac.params), _lib.ptr(ac.adam_m), _lib.ptr(ac.adam_v),
            _lib.ptr(ac.adam_step), _lib.ptr(ac.grads), _lib.ptr(data['obs']), data['obs'].stride(0),
            _lib.ptr(data['act']), data['act'].stride(0), _lib.ptr(data['logp']),
            _lib.ptr(data['target_value_r']), _lib.ptr(data['target_value_c']), _lib.ptr(data[adv_key_r]),
            _lib.ptr(data[adv_key_c]), None, M, _lib.ptr(lagrange), C.byref(self._hp), 1, 2, 1,
            self.max_blocks, _lib.ptr(self._ws), _lib.ptr(self._stats), _lib.stream_ptr()),
            'osa_ppo_minibatch(full-batch actor gradient)')
        grad = ac.grads[0].clone()
        loss = self._stats[2:3].clone()
        dist.all_reduce_avg_(grad)
        dist.all_reduce_avg_(loss)
        return loss, grad

    # ------------------------------------------------------------------ old distribution / fvp obs
    def begin(self, obs: torch.Tensor) -> None:
        """p_dist = actor(obs) and fvp_obs = obs[::fvp_sample_freq] (trpo.py:176-182)."""
        ac, M = self.ac, obs.shape[0]
        if self._old_mean is None or self._old_mean.shape[0] != M:
            self._old_mean = torch.empty(M, ac.act_dim, dtype=torch.float32, device=ac.device)
        if getattr(ac, 'general', False):
            ws, nws = ac.gmlp_ws(M)
            _lib.check(self.lib.osa_gmlp_actor_stats(
                C.byref(ac.desc), _lib.ptr(ac.params[0]), _lib.ptr(obs), obs.stride(0), M, None, 0, None, 0, 0, None, 0,
                None, None, None, None, _lib.ptr(self._old_mean), ac.act_dim, _lib.ptr(ws), nws, None,
                _lib.stream_ptr()), 'osa_gmlp_actor_stats(snapshot)')
        else:
            _lib.check(self.lib.osa_actor_kl(ac.obs_dim, ac.act_dim, ac.hidden, _lib.ptr(ac.params[0]),
                                             _lib.ptr(obs), obs.stride(0), M, None, 0, None, 0,
                                             _lib.ptr(self._old_mean), ac.act_dim, None, None,
                                             _lib.stream_ptr()), 'osa_actor_kl(snapshot)')
        lay = ac.layout
        self._old_log_std[:lay.act_dim].copy_(ac.params[0, lay.oLS:lay.oLS + lay.act_dim])
        self._fvp_obs = obs[::self.fvp_sample_freq]

    # ------------------------------------------------------------------ F v
    def fvp(self, v: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """NaturalPG._fvp (natural_pg.py:91-119) at the CURRENT actor parameters."""
        ac, lib, lay = self.ac, self.lib, self.ac.layout
        obs = self._fvp_obs
        M = obs.shape[0]
        raw = self._vecs['raw']
        ev = None
        if self.profile_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if getattr(ac, 'general', False):
            ws, nws = ac.gmlp_ws(M)
            _lib.check(lib.osa_gmlp_minibatch(
                C.byref(ac.desc), _lib.ptr(ac.params), _lib.ptr(ac.adam_m), _lib.ptr(ac.adam_v), _lib.ptr(ac.adam_step),
                _lib.ptr(ac.grads), _lib.ptr(obs), obs.stride(0), None, 0, None, None, None, None, None, None, M, None,
                C.byref(self._hp), 2, 2, 1, _lib.ptr(v), 1.0 / (M * ac.act_dim), _lib.ptr(ws), nws,
                _lib.ptr(self._stats), _lib.stream_ptr()), 'osa_gmlp_minibatch(fvp)')
        else:
            _lib.check(lib.osa_actor_fvp_raw(ac.obs_dim, ac.act_dim, ac.hidden, _lib.ptr(ac.params),
                                             _lib.ptr(ac.grads), _lib.ptr(obs), obs.stride(0), M, _lib.ptr(v),
                                             self.max_blocks, _lib.ptr(self._ws), _lib.ptr(self._stats),
                                             _lib.stream_ptr()), 'osa_actor_fvp_raw')
        if ev is not None:
            ev[1].record()
            import os

            if getattr(ac, 'general', False):
                name = 'gm_gemm_kernel (layer-wise Fisher-vector product)'
            elif ((ac.hidden & 0xFFFF) == 64 and ac.obs_dim <= 80 and ac.act_dim <= 16 and M > 64 and (ac.hidden >> 16) == 0
                  and os.environ.get('OSA_FVP_FAST', '1') != '0'):  # the shapes of csrc/fvp_kernel.hip
                name = 'osa_fvp_kernel + osa_fvp_reduce_kernel'
            else:
                name = 'osa_mb_grad_kernel<loss_kind 2: Fisher-vector product> + osa_slab_reduce_kernel'
            self.profile_events.append((name, M, ev))
        raw.copy_(ac.grads[0])
        dist.all_reduce_avg_(raw)  # C2: one flat message
        out = out if out is not None else torch.empty_like(v)
        _lib.check(lib.osa_fvp_finish(lay.P, _lib.ptr(raw), _lib.ptr(v), self.cg_damping, lay.oLS,
                                      lay.act_dim, 2.0 / lay.act_dim, _lib.ptr(out), _lib.stream_ptr()),
                   'osa_fvp_finish')
        self.fvp_calls += 1
        return out

    # ------------------------------------------------------------------ CG
    def conjugate_gradients(self, b: torch.Tensor, num_steps: int | None = None,
                            residual_tol: float = 1e-10, eps: float = 1e-6) -> torch.Tensor:
        """omnisafe/utils/math.py:86-132 with F = self.fvp; returns a new padded vector x."""
        lib, P, st = self.lib, self.ac.layout.P, _lib.stream_ptr()
        self.cg_solves += 1
        x, r, p, z = (self._vecs[k] for k in ('x', 'r', 'p', 'z'))
        _lib.check(lib.osa_cg_init(P, _lib.ptr(b), _lib.ptr(x), _lib.ptr(r), _lib.ptr(p),
                                   _lib.ptr(self._scal), st), 'osa_cg_init')
        for _ in range(self.cg_iters if num_steps is None else num_steps):
            self.fvp(p, out=z)
            _lib.check(lib.osa_cg_step(P, _lib.ptr(z), _lib.ptr(x), _lib.ptr(r), _lib.ptr(p),
                                       _lib.ptr(self._scal), residual_tol, eps, _lib.stream_ptr()),
                       'osa_cg_step')
        return x.clone()

    # ------------------------------------------------------------------ vector helpers
    def dot(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        out = torch.empty(1, dtype=torch.float32, device=self.ac.device)
        _lib.check(self.lib.osa_vec_dot(x.numel(), _lib.ptr(x), _lib.ptr(y), _lib.ptr(out),
                                        _lib.stream_ptr()), 'osa_vec_dot')
        return out

    def lincomb(self, a: float, x: torch.Tensor, b: float = 0.0, y: torch.Tensor | None = None,
                out: torch.Tensor | None = None) -> torch.Tensor:
        out = out if out is not None else torch.empty_like(x)
        _lib.check(self.lib.osa_vec_lincomb(x.numel(), float(a), _lib.ptr(x), float(b), _lib.ptr(y),
                                            _lib.ptr(out), _lib.stream_ptr()), 'osa_vec_lincomb')
        return out

    # ------------------------------------------------------------------ line search evaluation
    def evaluate_candidates(self, data: dict, theta_old: torch.Tensor, step: torch.Tensor, fracs: list[float],
                            adv_key_r: str, lagrange: torch.Tensor) -> torch.Tensor:
        """For every step fraction: [loss_pi, loss_cost, kl, mean ratio] of the actor at
        theta_old + frac * step, rank-averaged (dist_avg, trpo.py:114-118, cpo.py:140-143).  The actor
        parameters are left at theta_old.  One host sync for the whole search."""
        ac, lib, lay = self.ac, self.lib, self.ac.layout
        M = data['obs'].shape[0]
        res = torch.zeros(len(fracs), 4, dtype=torch.float32, device=ac.device)
        cand = torch.empty(3, lay.P, dtype=torch.float32, device=ac.device)  # eval reads block 0 only
        for k, frac in enumerate(fracs):
            self.lincomb(1.0, theta_old, frac, step, out=cand[0])
            if getattr(ac, 'general', False):
                ws, nws = ac.gmlp_ws(M)
                _lib.check(lib.osa_gmlp_actor_stats(
                    C.byref(ac.desc), _lib.ptr(cand), _lib.ptr(data['obs']), data['obs'].stride(0), M,
                    _lib.ptr(self._old_mean), ac.act_dim, _lib.ptr(self._old_log_std), 1, 1, _lib.ptr(data['act']),
                    data['act'].stride(0), _lib.ptr(data['logp']), _lib.ptr(data[adv_key_r]), _lib.ptr(data['adv_c']),
                    _lib.ptr(lagrange), None, 0, _lib.ptr(ws), nws, _lib.ptr(res[k]), _lib.stream_ptr()),
                    'osa_gmlp_actor_stats(eval)')
                continue
            _lib.check(lib.osa_actor_eval(
                ac.obs_dim, ac.act_dim, ac.hidden, _lib.ptr(cand), _lib.ptr(data['obs']),
                data['obs'].stride(0), M, _lib.ptr(data['act']), data['act'].stride(0), _lib.ptr(data['logp']),
                _lib.ptr(data[adv_key_r]), _lib.ptr(data['adv_c']), _lib.ptr(lagrange),
                _lib.ptr(self._old_mean), ac.act_dim, _lib.ptr(self._old_log_std), _lib.ptr(self._eval_ws),
                _lib.ptr(res[k]), _lib.stream_ptr()), 'osa_actor_eval')
        dist.all_reduce_avg_(res)
        return res.cpu()  # the one sync
