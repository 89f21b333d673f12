"""Minibatch update loop of PolicyGradient._update on the device.

Restates the control flow of omnisafe/algorithms/on_policy/base/policy_gradient.py:345-405 --
``update_iters`` passes over shuffled minibatches, reward-critic / cost-critic / actor step per
minibatch, full-batch KL(old || new) after each pass with optional early stop -- with every step of
arithmetic inside libomnisafe_amd kernels:

  * one launch per minibatch updates all three networks (osa_ppo_minibatch), instead of the
    reference's ~180 torch kernels + 3 ``.item()`` host syncs;
  * per-step statistics (losses, mean ratio, ...) stay on the device and are reduced once per epoch;
  * the only host synchronisation is the KL scalar once per pass when ``kl_early_stop`` is on.

Data parallelism (world_size > 1): local gradient-norm clip, then ONE flat all-reduce of the three
networks' gradients, then Adam (clip-then-average order of policy_gradient.py:437-442).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from . import distributed as dist
from .models import ConstraintActorCritic, HParams, SurrogateExt

NSTAT = 16


# What this process has learnt about the "workgroup b runs on XCC b mod 8" placement the one-XCC variants of the
# cooperative kernels rely on: None = not tried yet, True = verified by a kernel, False = a kernel found a
# network on two XCCs (every later launch takes the spread variant straight away).
_PLACEMENT = {'local_ok': None}


def _local_arg() -> int:
    """`local` argument of the cooperative passes: 1, or 3 under OSA_DEBUG_PLACEMENT=wrong (test hook: the one-XCC
    protocol on the spread grid, which the kernels' placement check must catch before anything is modified)."""
    return 3 if os.environ.get('OSA_DEBUG_PLACEMENT') == 'wrong' else 1


class PPOUpdater:  # pylint: disable=too-many-instance-attributes
    def __init__(self, ac: ConstraintActorCritic, *, batch_size: int, update_iters: int,
                 target_kl: float, kl_early_stop: bool, clip: float = 0.2, entropy_coef: float = 0.0,
                 use_critic_norm: bool = True, critic_norm_coef: float = 0.001,
                 use_max_grad_norm: bool = True, max_grad_norm: float = 40.0, use_cost: bool = True,
                 loss_kind: int = 0, max_blocks: int = 256, update_actor: bool = True,
                 persistent: bool = True, dp_mode: str = 'replicated', seed: int = 0,
                 update_critics: bool = True, ext: SurrogateExt | None = None) -> None:
        self.ac = ac
        # extended actor surrogate (FOCOPS / CUP / P3O): runs on the per-step kernels
        self.ext = ext
        # general networks (hidden_sizes outside the fused family: csrc/general_mlp.hip) run every optimiser step on
        # the layer-wise GEMM path -- no persistent pass, data parallelism by per-step all-reduce
        self.general = bool(getattr(ac, 'general', False))
        if self.general and ext is not None and int(batch_size) > 256 and (ext.kl_mask_eta >= 0 or ext.cost_kappa > 0):
            # (FOCOPS' trust-mask mean and P3O's penalty are minibatch-level quantities of one block of the loss kernel)
            raise NotImplementedError('FOCOPS / P3O on general networks: batch_size <= 256')
        self.update_critics = update_critics
        self.lib = _lib.load(require_gpu=True)
        self.batch_size, self.update_iters = int(batch_size), int(update_iters)
        self.target_kl, self.kl_early_stop = float(target_kl), bool(kl_early_stop)
        self.loss_kind = loss_kind
        self.update_actor = update_actor
        self.persistent = persistent  # persistent pass kernel (world_size 1) vs one launch per step
        self.persistent_max_batch = 512  # beyond this a step has enough rows to fill the chip per launch
        # world_size > 1: 'replicated' = all-gather the rollout once per epoch and compute the whole
        # global step on every rank (no per-step collective) -- as one cooperative persistent launch per
        # pass (osa_ppo_dp_pass) or, with 'replicated-steps', as two launches per step replayed from a
        # hipGraph (osa_ppo_dp_step); 'allreduce' = per-step flat RCCL all-reduce
        #   'p2p' (round 6) = every rank runs the single-GPU persistent pass on its own rows and the ranks exchange
        #   their clipped gradients by ONE-SHOT PEER WRITES into IPC-mapped exchange buffers (osa_ppo_p2p_pass): no
        #   collective and no W-fold recomputation on the step path; falls back to 'allreduce' where the pass kernel
        #   does not apply (general networks, wide observations, extended surrogates)
        self.dp_mode = dp_mode
        self._mode = 'allreduce' if dp_mode == 'p2p' else dp_mode  # what the steps run as (run() decides per update)
        self._p2p: dict = {}
        self.seed = int(seed)
        self._dp: dict = {}
        self._ar_k = 0  # steps of the current pass taken by the fast all-reduce mode (minibatch / _ar_end_pass)
        self.hp = HParams(clip=clip, entropy_coef=entropy_coef, critic_norm_coef=critic_norm_coef,
                          max_grad_norm=max_grad_norm, lr_actor=0.0, lr_critic=0.0, beta1=0.9,
                          beta2=0.999, adam_eps=1e-8, use_critic_norm=int(use_critic_norm),
                          use_max_grad_norm=int(use_max_grad_norm), use_cost=int(use_cost))
        self.max_blocks = int(max_blocks)
        dev = ac.device
        nws = self.lib.osa_minibatch_ws_floats(ac.obs_dim, ac.act_dim, ac.hidden, self.max_blocks)
        self._ws = torch.zeros(max(nws, 1), dtype=torch.float32, device=dev)  # tail = arrival tickets (start at 0)
        self._kl_ws = torch.empty(1024, dtype=torch.float64, device=dev)
        self._kl = torch.zeros(1, dtype=torch.float32, device=dev)
        self._old_mean: torch.Tensor | None = None
        self._old_log_std = torch.zeros(ac.layout.OUTP, dtype=torch.float32, device=dev)
        self._stats: torch.Tensor | None = None
        # optional profiling: list collecting (start, end) HIP events around every update launch on
        # torch's current stream (bench.py turns this on for the timed region)
        self.profile_events: list | None = None
        self.last_path: str | None = None
        self._graphed_pass = False
        self._use_wide = False
        self._split_xch: int | None = None  # uncached exchange buffer of the split wide pass (raw pointer)
        self._split_tried = False
        self._chunk: dict = {}  # exchange buffer / sync words of osa_ppo_chunked_pass
        self._ug: dict = {}  # captured hipGraph of one pass of per-step launches (large minibatches)
        self._split_local = False
        self._split_verified = False
        self._split_buf = None
        self._repl_wide = False

    def _chunk_ok(self, data: dict) -> bool:
        """osa_ppo_chunked_pass applies: plain surrogate, 64 < B <= 64 x (CUs / 8) rows, OSA_CHUNKED_PASS != 0.
        Allocates its exchange buffer (ordinary memory: the workgroups of a network share one XCC) on first use."""
        B = self.batch_size
        ck = self._chunk
        if ck.get('off') or self.ext is not None or B <= 64 or os.environ.get('OSA_CHUNKED_PASS', '1') == '0':
            return False
        W = (B + 63) // 64
        if ck.get('W') != W:
            ac = self.ac
            cus = torch.cuda.get_device_properties(ac.device).multi_processor_count
            if W > cus // 8:
                ck['off'] = True
                return False
            n = self.lib.osa_ppo_dp_pass_ws_floats(ac.obs_dim, ac.act_dim, ac.hidden, W)
            ck.update(W=W, xch=torch.zeros(max(n, 1), dtype=torch.float32, device=ac.device),
                      sync=torch.zeros(8, dtype=torch.int32, device=ac.device),
                      local=0 if _PLACEMENT['local_ok'] is False else 1, verified=False)
        return True

    def check_chunk_sync(self) -> None:
        """Sticky flag of the chunked pass (a workgroup never arrived: 1; a network's workgroups not on one XCC: 2)."""
        ck = self._chunk
        if 'sync' in ck and int(ck['sync'][3]) != 0:
            raise _lib.OsaError('osa_ppo_chunked_pass: ' + (
                'the workgroups of a network were not placed on one XCC' if int(ck['sync'][3]) == 2
                else 'a cooperating workgroup never arrived') + ' (results invalid); set OSA_CHUNKED_PASS=0')

    def _split_alloc(self) -> None:
        """Exchange buffer of osa_ppo_split_pass: ordinary memory with one XCC per network (default), uncached
        device memory with the workgroups spread over the XCCs (OSA_WIDE_SPLIT=spread, or after a tripped
        placement check); OSA_WIDE_SPLIT=0 keeps the one-CU kernel (A/B switch of tools/wide_pass_timing.py)."""
        if self._split_tried:
            return
        self._split_tried = True
        ac = self.ac
        if os.environ.get('OSA_WIDE_SPLIT', '1') == '0' or not bool(
                self.lib.osa_ppo_split_pass_supported(ac.obs_dim, ac.act_dim, ac.hidden)):
            return
        n = self.lib.osa_ppo_split_pass_xch_floats(ac.obs_dim, ac.act_dim, ac.hidden)
        # OSA_WIDE_SPLIT=local (default): one XCC per network, hand-offs through its L2 (ordinary memory);
        # OSA_WIDE_SPLIT=spread: workgroups over all XCCs, uncached exchange buffer
        self._split_local = os.environ.get('OSA_WIDE_SPLIT', 'local') != 'spread' and _PLACEMENT['local_ok'] is not False
        if self._split_local:
            self._split_buf = torch.zeros(max(n, 1), dtype=torch.float32, device=ac.device)
            self._split_xch = self._split_buf.data_ptr()
            return
        p = C.c_void_p()
        if self.lib.osa_dp_exchange_alloc(max(n, 1), C.byref(p)) == _lib.OSA_OK and p.value:
            self._split_xch = p.value

    def _split_free(self) -> None:
        if self._split_xch and not getattr(self, '_split_local', False):
            self.lib.osa_dp_exchange_free(C.c_void_p(self._split_xch))
        self._split_xch = None
        self._split_buf = None

    def check_split_sync(self) -> None:
        """Raises if a workgroup of a split wide pass ever gave up waiting for a peer (sticky device flag)."""
        if self._split_xch:
            flag = C.c_int(0)
            _lib.check(self.lib.osa_ppo_split_pass_timed_out(C.c_void_p(self._split_xch), C.byref(flag)),
                       'osa_ppo_split_pass_timed_out')
            if flag.value:
                raise _lib.OsaError('osa_ppo_split_pass: ' + (
                    'the workgroups of a network were not placed on one XCC; set OSA_WIDE_SPLIT=spread'
                    if flag.value == 2 else 'a cooperating workgroup never arrived') + ' (results invalid)')

    # ------------------------------------------------------------------
    def _nets_mask(self) -> int:
        m = (0b110 if self.hp.use_cost else 0b010) if self.update_critics else 0
        return m | (1 if self.update_actor else 0)

    def shuffles(self, rows: int, M: int, out: torch.Tensor | None = None,
                 generator: torch.Generator | None = None) -> torch.Tensor:
        """`rows` random permutations of 0 .. M-1 (one per pass: the reference's DataLoader(shuffle=True),
        policy_gradient.py:357-377) from ONE launch of osa_shuffle_rows -- a keyed bijection per row, seeded by one
        int64 per row from torch's (seeded) device generator.  OSA_SHUFFLE=sort: the batched argsort of random
        62-bit keys of rounds 1-3 (torch / rocPRIM merge sorts: ~45 launches, 0.3 ms for 8 x 65 536)."""
        dev = self.ac.device
        if os.environ.get('OSA_SHUFFLE', 'bijection') == 'sort':
            keys = torch.randint(0, 1 << 62, (rows, M), generator=generator, device=dev, dtype=torch.int64)
            p = keys.argsort(dim=1)
            if out is None:
                return p
            out.copy_(p.reshape(out.shape))
            return out
        seeds = torch.randint(0, 1 << 62, (rows,), generator=generator, device=dev, dtype=torch.int64)
        if out is None:
            out = torch.empty((rows, M), dtype=torch.int64, device=dev)
        assert out.is_contiguous() and out.numel() == rows * M
        _lib.check(self.lib.osa_shuffle_rows(_lib.ptr(seeds), rows, M, _lib.ptr(out), _lib.stream_ptr()),
                   'osa_shuffle_rows')
        return out

    def _graph_pass_ok(self, M: int) -> bool:
        """The per-step launches of a pass go through a captured hipGraph: large minibatches (the launches, not the
        rows, are what a step waits for), plain surrogate, single process.  OSA_UPDATE_GRAPH=0 keeps eager launches."""
        nmb = (M + self.batch_size - 1) // self.batch_size
        # (general networks: a step is ~10 launches of the layer-wise path -- captured whenever the pass stays below a
        # few thousand graph nodes, whatever the batch size)
        # `allreduce` data parallelism at ANY batch size over RCCL (round 5): per step gradient kernel -> ONE flat
        # all-reduce -> osa_adam_apply (policy_gradient.py:437-443's order with 1 message for 19), the pass's ~1000
        # steps incl. their collectives one captured hipGraph -- eager, a 64-row step costs 58 us of launches against
        # 9 us of kernel time at world 1 (profiles/r4_rccl_world1_timing.json)
        dp_graph = (dist.collectives_active() and self._mode == 'allreduce' and not self.general and nmb <= 1100
                    and dist.graph_capturable())
        return ((self.batch_size >= 2048 or (self.general and nmb <= 256) or dp_graph) and self.ext is None
                and (not dist.collectives_active() or dist.graph_capturable())
                and os.environ.get('OSA_UPDATE_GRAPH', '1') != '0' and not self._ug.get('failed', False))

    def _graph_pass(self, data: dict, perm: torch.Tensor | None, lagrange: torch.Tensor, stats_rows: torch.Tensor,
                    passes: int = 1) -> None:
        """One pass = ceil(M / B) optimiser steps of two launches each (partial gradients; slab reduce + clip + Adam),
        captured ONCE and replayed for every pass of every epoch: the permutation and the statistics rows live in
        fixed buffers, the learning rates in device memory (osa_ppo_hparams.lr_device) so that the LinearLR schedule
        can move between replays.  54 instead of 58 us per 16 384-row step (tools/large_batch_step_timing.py).

        Data parallelism (world_size > 1, round 4: `dp-large-batch-graph`): a step is partial gradients -> slab reduce +
        LOCAL clip (mode 1) -> ONE flat RCCL all-reduce of grads[3][P] (average) -> osa_adam_apply -- the reference's
        order (clip_grad_norm_, then avg_grads, then optimizer.step: policy_gradient.py:437-443; utils/distributed.py:
        167-198) with one message where it sends 19 -- and the whole pass, collectives included, is one captured
        hipGraph (RCCL enqueues kernels on its stream: capturable; over gloo the steps stay eager launches,
        `dp-large-batch`).

        `passes` > 1 (`perm` None): ALL passes of the update are one graph -- their permutations are drawn straight into
        the graph's index buffer by the one shuffle launch of the epoch, so that the per-pass copies (permutation in,
        learning rates in, statistics out: 32 `copyBuffer` operations per epoch of the large-batch benchmark, each a
        stream operation of its own between two graph launches) and 7 of the 8 graph launches disappear.  Used when no
        host decision sits between the passes (no KL early stop); same launches on the same data: same bits."""
        ac, B = self.ac, self.batch_size
        M = data['obs'].shape[0]
        nmb = (M + B - 1) // B
        hp = self.hp
        if dist.collectives_active() and self._allreduce_fast_ok(data, B):
            self._ar_slab()  # (allocated before the key is formed: its pointer is part of it)
        key = (M, B, passes, tuple(int(data[k].data_ptr()) for k in ('obs', 'act', 'logp', 'target_value_r',
                                                                     'target_value_c', 'adv_r', 'adv_c')),
               int(lagrange.data_ptr()),
               self._nets_mask(), self.loss_kind, self.max_blocks, dist.collectives_active(),
               # every pointer / stride the captured launches bake in (as the rollout graph's key does for params)
               tuple(int(t.data_ptr()) for t in (ac.params, ac.adam_m, ac.adam_v, ac.adam_step, ac.grads, self._ws)),
               int(ac.gmlp_ws(B)[0].data_ptr()) if self.general else 0,
               # what selects the fast all-reduce steps, and the slab their launches bake in
               (self._mode, os.environ.get('OSA_ALLREDUCE_FAST', '1'),
                int(self._ar_slab_t.data_ptr()) if getattr(self, '_ar_slab_t', None) is not None else 0),
               (data['obs'].stride(0), data['act'].stride(0)),
               (hp.clip, hp.entropy_coef, hp.critic_norm_coef, hp.max_grad_norm, hp.beta1, hp.beta2, hp.adam_eps,
                hp.use_critic_norm, hp.use_max_grad_norm, hp.use_cost))
        st = self._ug
        if st.get('key') != key:
            st.clear()
            st.update(key=key, perm=torch.empty(passes * M, dtype=torch.int64, device=ac.device),
                      stats=torch.zeros(passes * nmb, NSTAT, dtype=torch.float32, device=ac.device),
                      lr=torch.zeros(2, dtype=torch.float32, device=ac.device),
                      lr_host=torch.zeros(2, dtype=torch.float32).pin_memory(), graph=None, warm=False)
        if perm is None:  # one shuffle launch for all passes, written where the captured launches read
            self.shuffles(passes, M, out=st['perm'].view(passes, M))
        else:
            assert passes == 1
            st['perm'].copy_(perm)
        st['lr_host'][0], st['lr_host'][1] = float(hp.lr_actor), float(hp.lr_critic)
        st['lr'].copy_(st['lr_host'], non_blocking=True)

        def enqueue() -> None:
            self._ar_k = 0  # (a capture that was refused mid-pass, or an exception inside a pass, leaves no stale step index)
            for ip in range(passes):
                for k in range(nmb):
                    s0 = k * B
                    nb = min(B, M - s0)
                    self.minibatch(data, st['perm'][ip * M + s0:ip * M + s0 + nb], nb, lagrange, st['stats'][ip * nmb + k])
                self._ar_end_pass()

        hp.lr_device = st['lr'].data_ptr()
        pe, self.profile_events = self.profile_events, None  # (one event pair around the pass, none inside a capture)
        ev = None
        if pe is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        try:
            if st['graph'] is not None:
                st['graph'].replay()
            elif not st['warm']:  # first pass: eager (also sets the kernels' attributes, which must not happen under capture)
                enqueue()
                st['warm'] = True
            else:
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        enqueue()
                    st['graph'] = g
                    g.replay()
                except Exception as exc:  # pragma: no cover - capture refused: stay on eager launches
                    import warnings

                    warnings.warn(f'omnisafe_amd: hipGraph capture of the update pass was refused ({exc!r}); the '
                                  'steps stay on eager launches (OSA_UPDATE_GRAPH=0 silences this)', RuntimeWarning)
                    st['failed'], st['graph'] = True, None
                    enqueue()
        finally:
            hp.lr_device = None
            self.profile_events = pe
        if ev is not None:
            ev[1].record()
            pe.append(('gm_gemm_kernel' if self.general else 'osa_mb_grad_kernel', passes * M, ev))  # rows of the launch
        stats_rows.copy_(st['stats'])
        self._graphed_pass = st['graph'] is not None

    def _allreduce_fast_ok(self, data: dict, B: int) -> bool:
        """The `allreduce` mode's steps on osa_ppo_dp_step_phase: fused network family, plain surrogates, minibatches the
        persistent kernels take, 16-byte aligned rows (run() pads them once per update).  OSA_ALLREDUCE_FAST=0: the
        per-step kernels (osa_ppo_minibatch mode 1 -> all-reduce -> osa_adam_apply)."""
        obs = data['obs']
        return (self._mode == 'allreduce' and not self.general and self.ext is None and self.loss_kind in (0, 1)
                and self.batch_size <= self.persistent_max_batch and obs.stride(0) % 4 == 0 and obs.data_ptr() % 16 == 0
                and os.environ.get('OSA_ALLREDUCE_FAST', '1') != '0'
                and bool(self.lib.osa_ppo_pass_supported(self.ac.obs_dim, self.ac.act_dim, self.ac.hidden)))

    def _ar_slab(self) -> torch.Tensor:
        if getattr(self, '_ar_slab_t', None) is None:
            n = self.lib.osa_ppo_dp_ws_floats(self.ac.obs_dim, self.ac.act_dim, self.ac.hidden, 1)
            self._ar_slab_t = torch.zeros(n, dtype=torch.float32, device=self.ac.device)
        return self._ar_slab_t

    def _ar_end_pass(self) -> None:
        """After the last step of a pass of the fast all-reduce mode: the Adam step counters advance by its steps."""
        if self._ar_k:
            _lib.check(self.lib.osa_ppo_dp_end_pass(_lib.ptr(self.ac.adam_step),
                                                    self._nets_mask() & (7 if self.hp.use_cost else 3), self._ar_k,
                                                    _lib.stream_ptr()), 'osa_ppo_dp_end_pass')
            self._ar_k = 0

    def minibatch(self, data: dict, idx: torch.Tensor | None, B: int, lagrange: torch.Tensor,
                  stats_row: torch.Tensor) -> None:
        ac, lib, st = self.ac, self.lib, _lib.stream_ptr()
        dp = dist.collectives_active()
        mode = 1 if dp else 0  # 1: gradients only (clipped locally); all-reduce + Adam follow below
        ev = None
        if self.profile_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if self.general:
            ws, nws = ac.gmlp_ws(B)
            gext = None
            if self.ext is not None:  # FOCOPS / CUP / P3O on general networks (utils/model.py:73-111 builds any width)
                self.ext.old_mean = self._old_mean.data_ptr()
                self.ext.ld_old_mean = self._old_mean.stride(0)
                self.ext.old_log_std = self._old_log_std.data_ptr()
                gext = C.byref(self.ext)
            _lib.check(lib.osa_gmlp_minibatch_ext(
                C.byref(ac.desc), _lib.ptr(ac.params), _lib.ptr(ac.adam_m), _lib.ptr(ac.adam_v),
                _lib.ptr(ac.adam_step), _lib.ptr(ac.grads), _lib.ptr(data['obs']), data['obs'].stride(0),
                _lib.ptr(data['act']), data['act'].stride(0), _lib.ptr(data['logp']),
                _lib.ptr(data['target_value_r']), _lib.ptr(data['target_value_c']), _lib.ptr(data['adv_r']),
                _lib.ptr(data['adv_c']), _lib.ptr(idx), B, _lib.ptr(lagrange), C.byref(self.hp), self.loss_kind,
                mode, self._nets_mask(), None, 0.0, _lib.ptr(ws), nws, _lib.ptr(stats_row), gext, st),
                'osa_gmlp_minibatch_ext')
            if ev is not None:
                ev[1].record()
                self.profile_events.append(('gm_gemm_kernel', B, ev))
            if dp:
                dist.all_reduce_avg_(ac.grads)
                _lib.check(lib.osa_gmlp_adam_apply(C.byref(ac.desc), _lib.ptr(ac.params), _lib.ptr(ac.adam_m),
                                                   _lib.ptr(ac.adam_v), _lib.ptr(ac.adam_step), _lib.ptr(ac.grads),
                                                   C.byref(self.hp), self._nets_mask(), _lib.ptr(ac._gfin), st),
                           'osa_gmlp_adam_apply')
            return
        if dp and self._allreduce_fast_ok(data, B):
            # per-step all-reduce mode on the pass kernel's gradient-only form (round 5): clipped gradients of THIS rank's
            # minibatch -> ONE flat all-reduce (average) of the slab -> Adam (policy_gradient.py:437-443 order)
            slab = self._ar_slab()
            args = (ac.obs_dim, ac.act_dim, ac.hidden, _lib.ptr(ac.params), _lib.ptr(ac.adam_m), _lib.ptr(ac.adam_v),
                    _lib.ptr(ac.adam_step), _lib.ptr(data['obs']), data['obs'].stride(0), _lib.ptr(data['act']),
                    data['act'].stride(0), _lib.ptr(data['logp']), _lib.ptr(data['target_value_r']),
                    _lib.ptr(data['target_value_c']), _lib.ptr(data['adv_r']), _lib.ptr(data['adv_c']), _lib.ptr(idx),
                    B, self.batch_size, 1)
            tail = (_lib.ptr(lagrange), C.byref(self.hp), self.hp.lr_device, self.loss_kind, self._nets_mask(),
                    _lib.ptr(slab), _lib.ptr(stats_row))
            _lib.check(lib.osa_ppo_dp_step_phase(*args, 0, *tail, 1, st), 'osa_ppo_dp_step_phase(grad)')
            if ev is not None:
                ev[1].record()
                self.profile_events.append(('osa_ppo_pass_kernel', B, ev))
            dist.all_reduce_avg_(slab)
            _lib.check(lib.osa_ppo_dp_step_phase(*args, self._ar_k, *tail, 2, st), 'osa_ppo_dp_step_phase(apply)')
            self._ar_k += 1
            return
        ext = None
        if self.ext is not None:
            self.ext.old_mean = self._old_mean.data_ptr()
            self.ext.ld_old_mean = self._old_mean.stride(0)
            self.ext.old_log_std = self._old_log_std.data_ptr()
            ext = C.byref(self.ext)
        _lib.check(lib.osa_ppo_minibatch_ext(
            ac.obs_dim, ac.act_dim, ac.hidden, _lib.ptr(ac.params), _lib.ptr(ac.adam_m),
            _lib.ptr(ac.adam_v), _lib.ptr(ac.adam_step), _lib.ptr(ac.grads), _lib.ptr(data['obs']),
            data['obs'].stride(0), _lib.ptr(data['act']), data['act'].stride(0), _lib.ptr(data['logp']),
            _lib.ptr(data['target_value_r']), _lib.ptr(data['target_value_c']), _lib.ptr(data['adv_r']),
            _lib.ptr(data['adv_c']), _lib.ptr(idx), B, _lib.ptr(lagrange), C.byref(self.hp),
            self.loss_kind, mode, self._nets_mask(), self.max_blocks, _lib.ptr(self._ws),
            _lib.ptr(stats_row), ext, st), 'osa_ppo_minibatch_ext')
        if ev is not None:
            ev[1].record()
            self.profile_events.append(('osa_mb_grad_kernel', B, ev))
        if dp:
            dist.all_reduce_avg_(ac.grads)  # C1: one flat message for pi, V_r, V_c
            _lib.check(lib.osa_adam_apply(ac.obs_dim, ac.act_dim, ac.hidden, _lib.ptr(ac.params),
                                          _lib.ptr(ac.adam_m), _lib.ptr(ac.adam_v),
                                          _lib.ptr(ac.adam_step), _lib.ptr(ac.grads), C.byref(self.hp),
                                          self._nets_mask(), st), 'osa_adam_apply')

    def run_pass(self, data: dict, perm: torch.Tensor, lagrange: torch.Tensor,
                 stats_rows: torch.Tensor) -> None:
        """osa_ppo_pass: all ceil(M/B) minibatch steps of one pass in a single persistent launch."""
        ac = self.ac
        M = data['obs'].shape[0]
        ev = None
        if self.profile_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        name = 'osa_ppo_pass_kernel'
        if self._use_wide and self._split_xch:  # wide observations, first layer split over CUs (wide_split_kernel.hip)
            rc = self.lib.osa_ppo_split_pass(
                ac.obs_dim, ac.act_dim, ac.hidden, _lib.ptr(ac.params), _lib.ptr(ac.adam_m),
                _lib.ptr(ac.adam_v), _lib.ptr(ac.adam_step), _lib.ptr(data['obs']), data['obs'].stride(0),
                _lib.ptr(data['act']), data['act'].stride(0), _lib.ptr(data['logp']),
                _lib.ptr(data['target_value_r']), _lib.ptr(data['target_value_c']), _lib.ptr(data['adv_r']),
                _lib.ptr(data['adv_c']), _lib.ptr(perm), M, self.batch_size, _lib.ptr(lagrange),
                C.byref(self.hp), self.loss_kind, self._nets_mask(), C.c_void_p(self._split_xch),
                (1 if self._split_verified else _local_arg()) if self._split_local else 0, _lib.ptr(stats_rows),
                _lib.stream_ptr())
            if rc == _lib.OSA_EUNSUPPORTED:  # the device cannot hold the workgroups together: one CU per network
                self._split_free()
            else:
                _lib.check(rc, 'osa_ppo_split_pass')
                if self._split_local and not self._split_verified:
                    # first pass with one XCC per network: the kernel checks its placement before it modifies
                    # anything and returns untouched if it does not hold -> repeat spread over the XCCs
                    torch.cuda.synchronize()
                    flag = int(self._split_buf.view(torch.int32)[96])
                    if flag == 1:  # a workgroup never arrived at the placement check: not a placement question
                        raise _lib.OsaError('osa_ppo_split_pass: a cooperating workgroup never arrived at the '
                                            'placement check (device shared with another long-running kernel?)')
                    if flag != 0:
                        _PLACEMENT['local_ok'] = False
                        self._split_free()
                        self._split_tried = False
                        self._split_alloc()
                        return self.run_pass(data, perm, lagrange, stats_rows)
                    self._split_verified = _PLACEMENT['local_ok'] = True
                self.last_path = 'persistent-wide-split'
                if ev is not None:
                    ev[1].record()
                    self.profile_events.append(('osa_wide_split_kernel', M, ev))
                return
        if self._use_wide:  # wide observations: W1 and its Adam moments streamed from L2 (wide_pass_kernel.hip)
            _lib.check(self.lib.osa_ppo_wide_pass(
                ac.obs_dim, ac.act_dim, ac.hidden, _lib.ptr(ac.params), _lib.ptr(ac.adam_m),
                _lib.ptr(ac.adam_v), _lib.ptr(ac.adam_step), _lib.ptr(data['obs']), data['obs'].stride(0),
                _lib.ptr(data['act']), data['act'].stride(0), _lib.ptr(data['logp']),
                _lib.ptr(data['target_value_r']), _lib.ptr(data['target_value_c']), _lib.ptr(data['adv_r']),
                _lib.ptr(data['adv_c']), _lib.ptr(perm), M, self.batch_size, _lib.ptr(lagrange),
                C.byref(self.hp), self.loss_kind, self._nets_mask(), _lib.ptr(self._wide_ws), _lib.ptr(stats_rows),
                _lib.stream_ptr()), 'osa_ppo_wide_pass')
            if ev is not None:
                ev[1].record()
                self.profile_events.append(('osa_wide_pass_kernel', M, ev))
            return
        if self._chunk_ok(data):  # 64 < B: the minibatch's 64-row chunks on cooperating workgroups (one XCC per network)
            ck = self._chunk
            rc = self.lib.osa_ppo_chunked_pass(
                ac.obs_dim, ac.act_dim, ac.hidden, _lib.ptr(ac.params), _lib.ptr(ac.adam_m),
                _lib.ptr(ac.adam_v), _lib.ptr(ac.adam_step), _lib.ptr(data['obs']), data['obs'].stride(0),
                _lib.ptr(data['act']), data['act'].stride(0), _lib.ptr(data['logp']),
                _lib.ptr(data['target_value_r']), _lib.ptr(data['target_value_c']), _lib.ptr(data['adv_r']),
                _lib.ptr(data['adv_c']), _lib.ptr(perm), M, self.batch_size, _lib.ptr(lagrange),
                C.byref(self.hp), self.loss_kind, self._nets_mask(), _lib.ptr(ck['xch']), _lib.ptr(ck['sync']),
                (1 if ck.get('verified') else _local_arg()) if ck.get('local', 1) else 0, _lib.ptr(stats_rows),
                _lib.stream_ptr())
            if rc == _lib.OSA_EUNSUPPORTED:  # not co-resident: one workgroup walks through the chunks
                ck['off'] = True
            else:
                _lib.check(rc, 'osa_ppo_chunked_pass')
                if ck.get('local', 1) and not ck.get('verified'):
                    # (as above: an unverified placement leaves everything untouched; repeat with the workgroups
                    # spread over the XCCs and agent-scope release / acquire fences around the hand-offs)
                    torch.cuda.synchronize()
                    flag = int(ck['sync'][3])
                    if flag == 1:  # time-out, not placement: never re-run on possibly modified parameters
                        raise _lib.OsaError('osa_ppo_chunked_pass: a cooperating workgroup never arrived at the '
                                            'placement check (device shared with another long-running kernel?)')
                    if flag != 0:
                        _PLACEMENT['local_ok'] = False
                        ck['sync'].zero_()
                        ck['local'] = 0
                        return self.run_pass(data, perm, lagrange, stats_rows)
                    ck['verified'] = _PLACEMENT['local_ok'] = True
                self.last_path = 'persistent-chunked'
                if ev is not None:
                    ev[1].record()
                    self.profile_events.append(('osa_ppo_pass_kernel', M, ev))
                return
        ext = None
        if self.ext is not None:  # extended actor surrogate (FOCOPS / CUP / P3O) inside the persistent pass
            self.ext.old_mean = self._old_mean.data_ptr()
            self.ext.ld_old_mean = self._old_mean.stride(0)
            self.ext.old_log_std = self._old_log_std.data_ptr()
            ext = C.byref(self.ext)
        _lib.check(self.lib.osa_ppo_pass_ext(
            ac.obs_dim, ac.act_dim, ac.hidden, _lib.ptr(ac.params), _lib.ptr(ac.adam_m),
            _lib.ptr(ac.adam_v), _lib.ptr(ac.adam_step), _lib.ptr(data['obs']), data['obs'].stride(0),
            _lib.ptr(data['act']), data['act'].stride(0), _lib.ptr(data['logp']),
            _lib.ptr(data['target_value_r']), _lib.ptr(data['target_value_c']), _lib.ptr(data['adv_r']),
            _lib.ptr(data['adv_c']), _lib.ptr(perm), M, self.batch_size, _lib.ptr(lagrange),
            C.byref(self.hp), self.loss_kind, self._nets_mask(), _lib.ptr(stats_rows), ext,
            _lib.stream_ptr()), 'osa_ppo_pass_ext')
        if ev is not None:
            ev[1].record()
            self.profile_events.append((name, M, ev))

    # ------------------------------------------------------------------ one-shot peer exchange (dp_mode 'p2p')
    def _p2p_ok(self, data: dict) -> bool:
        """osa_ppo_p2p_pass applies: fused network family with narrow observations, plain surrogates, minibatches the
        persistent kernels take, 16-byte aligned rows (run() pads them once per update), at most 16 ranks."""
        obs = data['obs']
        return (self.dp_mode == 'p2p' and not self.general and self.ext is None and self.loss_kind in (0, 1)
                and self.persistent and self.batch_size <= self.persistent_max_batch and dist.world_size() <= 16
                and obs.stride(0) % 4 == 0 and obs.data_ptr() % 16 == 0 and not self._p2p.get('off')
                and bool(self.lib.osa_ppo_pass_supported(self.ac.obs_dim, self.ac.act_dim, self.ac.hidden)))

    def _p2p_setup(self) -> bool:
        """Allocate this rank's exchange buffer (uncached device memory), hand its IPC handle to every rank and map
        theirs (a set-up step: one all-gather of 64 bytes + a barrier; the step path has no collective).  Returns False
        -- on EVERY rank, the decision is all-reduced -- when any rank's runtime refuses the allocation or a mapping."""
        import torch.distributed as tdist

        ac, lib, st = self.ac, self.lib, self._p2p
        W, rank = dist.world_size(), dist.rank()
        if st.get('W') == W and st.get('peers') is not None:
            return True
        self._p2p_free()
        n = lib.osa_p2p_exchange_floats(ac.obs_dim, ac.act_dim, ac.hidden, W)
        own, handle = C.c_void_p(), (C.c_ubyte * 64)()
        ok = n > 0 and lib.osa_p2p_exchange_alloc(n, C.byref(own), handle) == _lib.OSA_OK and bool(own.value)
        handles: list = [None] * W
        if W > 1:
            tdist.all_gather_object(handles, bytes(handle) if ok else None)
        else:
            handles[0] = bytes(handle) if ok else None
        ok = all(h is not None for h in handles)
        ptrs = [None] * W
        opened = []
        if ok:
            for q in range(W):
                if q == rank:
                    ptrs[q] = own.value
                    continue
                pq = C.c_void_p()
                buf = (C.c_ubyte * 64).from_buffer_copy(handles[q])
                if lib.osa_p2p_exchange_open(buf, C.byref(pq)) != _lib.OSA_OK or not pq.value:
                    ok = False
                    break
                ptrs[q] = pq.value
                opened.append(pq.value)
        # everybody proceeds, or nobody does (and nobody writes into a buffer that somebody is about to free)
        flag = torch.tensor([1.0 if ok else 0.0], device=ac.device if dist.graph_capturable() else 'cpu')
        if W > 1:
            tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)
        if float(flag) < 1.0:
            for pq in opened:
                lib.osa_p2p_exchange_release(C.c_void_p(pq))
            if own.value:
                lib.osa_p2p_exchange_release(own)
            st['off'] = True
            import warnings

            warnings.warn('omnisafe_amd: the peer-exchange buffers could not be allocated / mapped on every rank '
                          '(hipIpc refused); dp_mode falls back to the per-step all-reduce', RuntimeWarning)
            return False
        st.update(W=W, own=own.value, opened=opened, peers=(C.c_void_p * W)(*ptrs), seq=0,
                  timeout=float(os.environ.get('OSA_P2P_TIMEOUT_S', '20')))
        return True

    def _p2p_free(self) -> None:
        st = self._p2p
        for pq in st.get('opened', []):
            self.lib.osa_p2p_exchange_release(C.c_void_p(pq))
        if st.get('own'):
            self.lib.osa_p2p_exchange_release(C.c_void_p(st['own']))
        for k in ('own', 'opened', 'peers', 'W'):
            st.pop(k, None)

    def check_p2p_sync(self) -> None:
        """Sticky time-out word of the peer exchange: a rank that never arrived leaves every other rank with Adam steps
        on incompletely averaged gradients -- an error, never a silent fallback."""
        st = self._p2p
        if st.get('own'):
            flag = C.c_int(0)
            _lib.check(self.lib.osa_p2p_exchange_timed_out(C.c_void_p(st['own']), C.byref(flag)),
                       'osa_p2p_exchange_timed_out')
            if flag.value:
                raise _lib.OsaError('osa_ppo_p2p_pass: a peer rank never arrived at an optimiser step within '
                                    f"{st['timeout']} s (results invalid); set OSA_DP_MODE=allreduce")

    def run_pass_p2p(self, data: dict, perm: torch.Tensor, lagrange: torch.Tensor, stats_rows: torch.Tensor) -> None:
        """osa_ppo_p2p_pass: one pass of THIS rank's minibatches in one persistent launch, gradients exchanged by peer
        writes (policy_gradient.py:366-382 + 437-443 under torch.distributed)."""
        ac, st = self.ac, self._p2p
        M = data['obs'].shape[0]
        nmb = (M + self.batch_size - 1) // self.batch_size
        ev = None
        if self.profile_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        _lib.check(self.lib.osa_ppo_p2p_pass(
            ac.obs_dim, ac.act_dim, ac.hidden, _lib.ptr(ac.params), _lib.ptr(ac.adam_m), _lib.ptr(ac.adam_v),
            _lib.ptr(ac.adam_step), _lib.ptr(data['obs']), data['obs'].stride(0), _lib.ptr(data['act']),
            data['act'].stride(0), _lib.ptr(data['logp']), _lib.ptr(data['target_value_r']),
            _lib.ptr(data['target_value_c']), _lib.ptr(data['adv_r']), _lib.ptr(data['adv_c']), _lib.ptr(perm), M,
            self.batch_size, st['W'], dist.rank(), st['peers'], st['seq'] & 0xFFFFFFFF, st['timeout'],
            _lib.ptr(lagrange), C.byref(self.hp), self.loss_kind, self._nets_mask(), _lib.ptr(stats_rows),
            _lib.stream_ptr()), 'osa_ppo_p2p_pass')
        st['seq'] += nmb
        if ev is not None:
            ev[1].record()
            self.profile_events.append(('osa_ppo_p2p_pass_kernel', M, ev))

    # ------------------------------------------------------------------ replicated-data DP path
    _DP_KEYS = ('obs', 'act', 'logp', 'target_value_r', 'target_value_c', 'adv_r', 'adv_c')

    def gather_for_replicated(self, data: dict, W: int) -> dict:
        """All-gather this epoch's env-major batch of every rank into persistent [W*M, ...] buffers
        (one collective per tensor per epoch instead of one all-reduce per optimiser step)."""
        M = data['obs'].shape[0]
        st = self._dp
        if st.get('M') != M or st.get('W') != W:
            self._dp_free()
            st.clear()
            st.update(M=M, W=W, graph=None)
            st['data'] = {k: torch.empty((W * M,) + tuple(data[k].shape[1:]), dtype=torch.float32,
                                         device=self.ac.device) for k in self._DP_KEYS}
        for k in self._DP_KEYS:
            dist.all_gather_rows(data[k], out=st['data'][k])
        return st['data']

    def _dp_state(self, M: int, W: int, nmb: int) -> dict:
        ac, st = self.ac, self._dp
        if st.get('M') != M or st.get('W') != W:
            self._dp_free()
            st.clear()
            st.update(M=M, W=W, graph=None)
        if 'perm' not in st:
            dev = ac.device
            st['perm'] = torch.zeros(W, M, dtype=torch.int64, device=dev)
            st['pass_stats'] = torch.zeros(nmb, NSTAT, dtype=torch.float32, device=dev)
            st['slabs'] = torch.empty(self.lib.osa_ppo_dp_ws_floats(ac.obs_dim, ac.act_dim, ac.hidden, W),
                                      dtype=torch.float32, device=dev)
            st['lr'] = torch.zeros(2, dtype=torch.float32, device=dev)
            # one generator, seeded from the (rank-independent) config seed: every rank draws the SAME W
            # permutations without communicating
            st['gen'] = torch.Generator(device=dev)
            st['gen'].manual_seed(self.seed + 7919)
            st['graph'] = None
        return st

    def _dp_enqueue_pass(self, data_all: dict, M: int, W: int, lagrange: torch.Tensor, st: dict, nmb: int) -> None:
        ac, lib = self.ac, self.lib
        for k in range(nmb):
            _lib.check(lib.osa_ppo_dp_step(
                ac.obs_dim, ac.act_dim, ac.hidden, _lib.ptr(ac.params), _lib.ptr(ac.adam_m),
                _lib.ptr(ac.adam_v), _lib.ptr(ac.adam_step), _lib.ptr(data_all['obs']),
                data_all['obs'].stride(0), _lib.ptr(data_all['act']), data_all['act'].stride(0),
                _lib.ptr(data_all['logp']), _lib.ptr(data_all['target_value_r']),
                _lib.ptr(data_all['target_value_c']), _lib.ptr(data_all['adv_r']), _lib.ptr(data_all['adv_c']),
                _lib.ptr(st['perm']), M, self.batch_size, W, k, _lib.ptr(lagrange), C.byref(self.hp),
                _lib.ptr(st['lr']), self.loss_kind, self._nets_mask(), _lib.ptr(st['slabs']),
                _lib.ptr(st['pass_stats'][k]), _lib.stream_ptr()), 'osa_ppo_dp_step')
        _lib.check(lib.osa_ppo_dp_end_pass(_lib.ptr(ac.adam_step), self._nets_mask() & (7 if self.hp.use_cost else 3),
                                           nmb, _lib.stream_ptr()), 'osa_ppo_dp_end_pass')

    def _dp_free(self) -> None:
        self._split_free()
        st = self._dp
        if st.get('xch') is None and st.get('xch_ptr'):
            self.lib.osa_dp_exchange_free(st['xch_ptr'])
        st.pop('xch_ptr', None)
        if st.get('wide_xch'):
            self.lib.osa_dp_exchange_free(C.c_void_p(st['wide_xch']))
        st.pop('wide_xch', None)

    def _wide_dp_fits(self, W: int) -> bool:
        """The data-parallel split pass (osa_ppo_split_dp_pass) applies: wide observations, B <= 64, plain surrogate,
        and the device holds W x 3 x (1 + ceil(KB / 6)) workgroups (one per compute unit) together."""
        ac = self.ac
        if (self.ext is not None or self.batch_size > 64 or self.loss_kind not in (0, 1) or not self.persistent
                or os.environ.get('OSA_WIDE_SPLIT', '1') == '0'
                or not bool(self.lib.osa_ppo_split_pass_supported(ac.obs_dim, ac.act_dim, ac.hidden))):
            return False
        helpers = ((ac.obs_dim + 15) // 16 + 5) // 6
        cus = torch.cuda.get_device_properties(ac.device).multi_processor_count
        return W * 3 * (helpers + 1) <= cus

    def _wide_dp_pass(self, data_all: dict, M: int, W: int, lagrange: torch.Tensor, st: dict) -> None:
        """osa_ppo_split_dp_pass: one cooperative launch = one pass of the global update for wide observations
        (BASELINE config 4 under world_size > 1): W virtual ranks x 3 networks x (leader + helpers)."""
        ac, lib = self.ac, self.lib
        if not st.get('wide_xch'):
            n = lib.osa_ppo_split_dp_xch_floats(ac.obs_dim, ac.act_dim, ac.hidden, W)
            p = C.c_void_p()
            _lib.check(lib.osa_dp_exchange_alloc(max(n, 1), C.byref(p)), 'osa_dp_exchange_alloc')
            st['wide_xch'] = p.value
            # OSA_WIDE_DP=place (default): the W owners of the same parameters on ONE XCC, their exchange slabs in
            # ordinary memory served by that XCC's L2 (every replica otherwise reads the W slabs of its owners from
            # the device-coherent level: W^2 x 25 KB per owner group and step); OSA_WIDE_DP=spread: everything uncached
            helpers = ((ac.obs_dim + 15) // 16 + 5) // 6
            cus = torch.cuda.get_device_properties(ac.device).multi_processor_count
            st['wide_place'] = (os.environ.get('OSA_WIDE_DP', 'place') == 'place' and _PLACEMENT['local_ok'] is not False
                                and ((3 * (helpers + 1) + 7) // 8) * W <= cus // 8)
            st['wide_verified'] = False
            if st['wide_place']:
                st['wide_dpx'] = torch.zeros(lib.osa_ppo_split_dp_dpx_floats(ac.obs_dim, ac.act_dim, ac.hidden, W),
                                             dtype=torch.float32, device=ac.device)
        place = st['wide_place']
        _lib.check(lib.osa_ppo_split_dp_pass(
            ac.obs_dim, ac.act_dim, ac.hidden, _lib.ptr(ac.params), _lib.ptr(ac.adam_m), _lib.ptr(ac.adam_v),
            _lib.ptr(ac.adam_step), _lib.ptr(data_all['obs']), data_all['obs'].stride(0), _lib.ptr(data_all['act']),
            data_all['act'].stride(0), _lib.ptr(data_all['logp']), _lib.ptr(data_all['target_value_r']),
            _lib.ptr(data_all['target_value_c']), _lib.ptr(data_all['adv_r']), _lib.ptr(data_all['adv_c']),
            _lib.ptr(st['perm']), M, self.batch_size, W, _lib.ptr(lagrange), C.byref(self.hp), self.loss_kind,
            self._nets_mask(), C.c_void_p(st['wide_xch']), _lib.ptr(st['wide_dpx']) if place else None,
            ((1 if st['wide_verified'] else _local_arg()) if place else 0), _lib.ptr(st['pass_stats']),
            _lib.stream_ptr()), 'osa_ppo_split_dp_pass')
        if place and not st['wide_verified']:
            # first pass with the owner groups on one XCC each: an unverified placement returns with everything
            # untouched -> repeat rank-major with the uncached exchange
            torch.cuda.synchronize()
            flag = C.c_int(0)
            _lib.check(lib.osa_ppo_split_pass_timed_out(C.c_void_p(st['wide_xch']), C.byref(flag)),
                       'osa_ppo_split_pass_timed_out')
            if flag.value == 1:
                raise _lib.OsaError('osa_ppo_split_dp_pass: a cooperating workgroup never arrived at the placement '
                                    'check (device shared with another long-running kernel?)')
            if flag.value != 0:
                _PLACEMENT['local_ok'] = False
                st['wide_place'] = False
                _lib.check(lib.osa_ppo_split_pass_clear_flag(C.c_void_p(st['wide_xch'])),
                           'osa_ppo_split_pass_clear_flag')
                return self._wide_dp_pass(data_all, M, W, lagrange, st)
            st['wide_verified'] = True

    def __del__(self):
        try:
            self._p2p_free()
            self._dp_free()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def _dp_coop_pass(self, data_all: dict, M: int, W: int, lagrange: torch.Tensor, st: dict) -> bool:
        """osa_ppo_dp_pass: the whole pass as ONE cooperative launch of 3 x W persistent workgroups.
        Returns False when the device cannot hold them (caller falls back to the stepwise path)."""
        ac, lib = self.ac, self.lib
        # batch > 64: the minibatch's 64-row chunks of every rank on their own workgroups (osa_ppo_dp_chunked_pass:
        # W x ceil(B / 64) peers per network, two hand-offs per step) instead of one workgroup per rank walking
        # through the chunks; OSA_CHUNKED_PASS=0 or a device too small for the peers keeps the latter
        cus = torch.cuda.get_device_properties(ac.device).multi_processor_count
        nch = (self.batch_size + 63) // 64
        chunked = (nch > 1 and os.environ.get('OSA_CHUNKED_PASS', '1') != '0' and 3 * W * nch <= cus
                   and W <= 16 and not st.get('chunk_off'))
        peers = W * nch if chunked else W
        if st.get('xch_peers') != (peers, chunked):
            if st.get('xch') is None and st.get('xch_ptr'):
                lib.osa_dp_exchange_free(st['xch_ptr'])
            for k in ('xch', 'xch_ptr', 'sync', 'verified'):
                st.pop(k, None)
            st['xch_peers'] = (peers, chunked)
        st['chunked'] = chunked
        if 'xch' not in st:
            n = (lib.osa_ppo_dp_chunked_pass_ws_floats(ac.obs_dim, ac.act_dim, ac.hidden, self.batch_size, W)
                 if chunked else lib.osa_ppo_dp_pass_ws_floats(ac.obs_dim, ac.act_dim, ac.hidden, W))
            # uncached device memory for the hand-off (no L2 write-back / invalidate per step); an ordinary
            # tensor if the runtime refuses (OSA_DP_XCH=cached forces that)
            # OSA_DP_XCH=local (default while world <= CUs / 8): one XCC per network, ordinary memory served by
            # that XCC's L2 (osa_ppo_dp_pass_placed(local = 1), placement verified on the device)
            p = C.c_void_p()
            mode = os.environ.get('OSA_DP_XCH', 'local')
            st['local'] = mode == 'local' and peers <= cus // 8 and _PLACEMENT['local_ok'] is not False
            if not st['local'] and mode in ('local', 'uncached') and lib.osa_dp_exchange_alloc(
                    max(n, 1), C.byref(p)) == _lib.OSA_OK and p.value:
                st['xch_ptr'], st['xch'] = p.value, None
            else:
                st['xch'] = torch.zeros(max(n, 1), dtype=torch.float32, device=ac.device)
                st['xch_ptr'] = st['xch'].data_ptr()
            st['sync'] = torch.zeros(64, dtype=torch.int32, device=ac.device)
        # (the reduction sliced over the ranks + Adam on the slice + parameters exchanged -- round 3's
        # osa_ppo_dp_slice_pass -- measured 17.9 us per step at 8 virtual ranks against 15.7 for the direct sum, 15.9 v
        # 13.2 at 4: the second hand-off costs more than the W - 2 slab reads and the 7/8 of Adam it saves; removed in
        # round 4, profiles/HISTORY.md §5.2, profiles/r3_dp_shapes_timing.md)
        st['sliced'] = False
        fn = lib.osa_ppo_dp_chunked_pass if chunked else lib.osa_ppo_dp_pass_placed
        rc = fn(
            ac.obs_dim, ac.act_dim, ac.hidden, _lib.ptr(ac.params), _lib.ptr(ac.adam_m),
            _lib.ptr(ac.adam_v), _lib.ptr(ac.adam_step), _lib.ptr(data_all['obs']),
            data_all['obs'].stride(0), _lib.ptr(data_all['act']), data_all['act'].stride(0),
            _lib.ptr(data_all['logp']), _lib.ptr(data_all['target_value_r']),
            _lib.ptr(data_all['target_value_c']), _lib.ptr(data_all['adv_r']), _lib.ptr(data_all['adv_c']),
            _lib.ptr(st['perm']), M, self.batch_size, W, _lib.ptr(lagrange), C.byref(self.hp),
            self.loss_kind, self._nets_mask(), st['xch_ptr'], _lib.ptr(st['sync']),
            (1 if st.get('verified') else _local_arg()) if st['local'] else 0,
            _lib.ptr(st['pass_stats']), _lib.stream_ptr())
        if rc == _lib.OSA_EUNSUPPORTED and chunked:  # not co-resident: one workgroup per rank walks through the chunks
            st['chunk_off'] = True
            return self._dp_coop_pass(data_all, M, W, lagrange, st)
        if rc == _lib.OSA_EUNSUPPORTED:
            return False
        _lib.check(rc, 'osa_ppo_dp_chunked_pass' if chunked else 'osa_ppo_dp_pass_placed')
        if st['local'] and not st.get('verified'):
            # first pass with one XCC per network: an unverified placement returns with everything untouched
            # -> repeat spread over the XCCs (same buffer, agent-scope release / acquire fences)
            torch.cuda.synchronize()
            flag = int(st['sync'][3])
            if flag == 1:  # time-out, not placement: never re-run on possibly modified parameters
                raise _lib.OsaError('osa_ppo_dp_pass_placed: a peer workgroup never arrived at the placement check '
                                    '(workgroups not co-resident?); set OSA_DP_MODE=replicated-steps')
            if flag != 0:
                _PLACEMENT['local_ok'] = False
                st['sync'].zero_()
                st['local'] = False
                return self._dp_coop_pass(data_all, M, W, lagrange, st)
            st['verified'] = _PLACEMENT['local_ok'] = True
        st['coop_passes'] = st.get('coop_passes', 0) + 1
        return True

    def check_wide_dp_sync(self) -> None:
        """Sticky time-out word of the data-parallel split pass (a cooperating workgroup never arrived)."""
        st = self._dp
        if st.get('wide_xch'):
            flag = C.c_int(0)
            _lib.check(self.lib.osa_ppo_split_pass_timed_out(C.c_void_p(st['wide_xch']), C.byref(flag)),
                       'osa_ppo_split_pass_timed_out')
            if flag.value:
                raise _lib.OsaError('osa_ppo_split_dp_pass: a cooperating workgroup never arrived (results invalid); '
                                    'set OSA_DP_MODE=allreduce')

    def check_dp_sync(self) -> None:
        """Raise if ANY cooperative pass since allocation flagged a peer workgroup that never arrived (host
        sync).  sync[3] is sticky: osa_ppo_dp_pass resets only the arrival counters sync[0..2], so a time-out
        in an early pass of an update is still visible here.  After a time-out this rank has applied Adam
        steps to incompletely averaged gradients: its replica is no longer trustworthy, hence an error and
        not a silent fallback."""
        st = self._dp
        if 'sync' in st and int(st['sync'][3]) != 0:
            if int(st['sync'][3]) == 2:
                raise RuntimeError('osa_ppo_dp_pass_placed: the workgroups of a network were not placed on one XCC '
                                   '(results invalid); set OSA_DP_XCH=uncached')
            raise RuntimeError('osa_ppo_dp_pass: a peer workgroup timed out (workgroups not co-resident?); '
                               'set OSA_DP_MODE=replicated-steps')

    def check_reduce_sync(self) -> None:
        """The fused slab-reduce + clip/Adam launch of the large-batch step meets its workgroups at a grid
        barrier with a bounded spin; a time-out is flagged in the workspace tail (sticky) instead of hanging."""
        if int(self._ws.view(torch.int32)[-5]) != 0:
            raise RuntimeError('osa_ppo_minibatch: the workgroups of the slab reduction did not all arrive '
                               '(device shared with another long-running kernel?)')

    def run_pass_replicated(self, data_all: dict, M: int, W: int, lagrange: torch.Tensor,
                            stats_rows: torch.Tensor, perms_all: torch.Tensor | None = None,
                            use_graph: bool = True, coop: bool | None = None) -> None:
        """One pass of the global update on the all-gathered data.  Default: one cooperative persistent
        launch (osa_ppo_dp_pass).  Fallback (`coop=False`, dp_mode 'replicated-steps', or more than
        CUs/3 ranks): ceil(M/B) steps x (W x 3 gradient workgroups + reduce/Adam), captured once as a
        hipGraph and replayed per pass (2 launches per step would otherwise be host-launch-bound)."""
        ac = self.ac
        nmb = (M + self.batch_size - 1) // self.batch_size
        st = self._dp_state(M, W, nmb)
        if perms_all is not None:
            st['perm'].copy_(perms_all.reshape(W, M))
        else:  # W shuffles, one launch (every rank draws the same seeds from the generator they all seeded alike)
            self.shuffles(W, M, out=st['perm'], generator=st['gen'])
        if self._repl_wide:  # wide observations: the data-parallel split pass (no stepwise variant)
            ev = None
            if self.profile_events is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            self._wide_dp_pass(data_all, M, W, lagrange, st)
            if ev is not None:
                ev[1].record()
                self.profile_events.append(('osa_wide_split_kernel', W * M, ev))
            stats_rows.copy_(st['pass_stats'][:stats_rows.shape[0]])
            return
        if coop is None:
            coop = self.dp_mode != 'replicated-steps'
        if coop:
            ev = None
            if self.profile_events is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            if self._dp_coop_pass(data_all, M, W, lagrange, st):
                if ev is not None:
                    ev[1].record()
                    self.profile_events.append(('osa_ppo_dp_pass', W * M, ev))
                stats_rows.copy_(st['pass_stats'][:stats_rows.shape[0]])
                return
        st['lr'][0] = float(self.hp.lr_actor)
        st['lr'][1] = float(self.hp.lr_critic)
        key = (M, W, nmb, self._nets_mask(), int(lagrange.data_ptr()), data_all['obs'].data_ptr())
        ev = None
        if self.profile_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        graph_ok = use_graph and not st.get('graph_failed', False) and st.get('warm', False)
        if graph_ok and (st.get('graph') is None or st.get('graph_key') != key):
            try:  # capture records the launches without executing them
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._dp_enqueue_pass(data_all, M, W, lagrange, st, nmb)
                st['graph'], st['graph_key'] = g, key
            except Exception:  # pragma: no cover - capture unsupported: stay on eager launches
                st['graph_failed'], st['graph'] = True, None
                graph_ok = False
        if graph_ok and st.get('graph') is not None:
            st['graph'].replay()
        else:
            # first pass (also sets the kernels' LDS attributes, which must not happen under capture)
            self._dp_enqueue_pass(data_all, M, W, lagrange, st, nmb)
            st['warm'] = True
        if ev is not None:
            ev[1].record()
            self.profile_events.append(('osa_ppo_dp_step', W * M, ev))
        stats_rows.copy_(st['pass_stats'][:stats_rows.shape[0]])

    def _aligned_rows(self, data: dict) -> dict:
        """The persistent kernels read observation rows with 16-byte loads: rows must be 16-byte aligned
        with a leading dimension that is a multiple of 4 floats.  buffer.get() of the BASELINE shapes
        complies (D_o = 60, 72, 376); anything else (e.g. D_o = 27) is padded once per update."""
        obs = data['obs']
        if obs.stride(0) % 4 == 0 and obs.data_ptr() % 16 == 0 and obs.stride(1) == 1:
            return data
        M, D = obs.shape
        pad = torch.zeros(M, (D + 3) // 4 * 4, dtype=torch.float32, device=obs.device)
        pad[:, :D].copy_(obs)
        out = dict(data)
        out['obs'] = pad[:, :D]  # a view: shape (M, D), row stride = padded width
        return out

    def snapshot_old_distribution(self, obs: torch.Tensor) -> None:
        """old_distribution = actor(obs) (policy_gradient.py:357)."""
        ac, M = self.ac, obs.shape[0]
        if self._old_mean is None or self._old_mean.shape[0] != M:
            self._old_mean = torch.empty(M, ac.act_dim, dtype=torch.float32, device=ac.device)
        if self.general:
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

    def kl(self, obs: torch.Tensor, reduce_mode: int = 0) -> torch.Tensor:
        """Device scalar KL(old || new), rank-averaged (dist_avg, policy_gradient.py:390)."""
        ac, M = self.ac, obs.shape[0]
        if self.general:
            ws, nws = ac.gmlp_ws(M)
            _lib.check(self.lib.osa_gmlp_actor_stats(
                C.byref(ac.desc), _lib.ptr(ac.params[0]), _lib.ptr(obs), obs.stride(0), M, _lib.ptr(self._old_mean),
                ac.act_dim, _lib.ptr(self._old_log_std), 0, reduce_mode, None, 0, None, None, None, None, None, 0,
                _lib.ptr(ws), nws, _lib.ptr(self._kl), _lib.stream_ptr()), 'osa_gmlp_actor_stats(kl)')
            dist.all_reduce_avg_(self._kl)
            return self._kl
        _lib.check(self.lib.osa_actor_kl(ac.obs_dim, ac.act_dim, ac.hidden, _lib.ptr(ac.params[0]),
                                         _lib.ptr(obs), obs.stride(0), M, _lib.ptr(self._old_mean),
                                         ac.act_dim, _lib.ptr(self._old_log_std), reduce_mode, None, 0,
                                         _lib.ptr(self._kl_ws), _lib.ptr(self._kl), _lib.stream_ptr()),
                   'osa_actor_kl')
        dist.all_reduce_avg_(self._kl)
        return self._kl

    # ------------------------------------------------------------------
    def run(self, data: dict, lagrange: torch.Tensor, perms=None, actor_lr: float | None = None,
            critic_lr: float | None = None) -> dict:
        """One call of PolicyGradient._update after ``data = buf.get()``.  ``perms`` (optional, list of
        update_iters index tensors) injects the minibatch order for parity tests; by default a fresh
        device permutation is drawn per pass (DataLoader(shuffle=True), policy_gradient.py:359-363)."""
        ac = self.ac
        self.hp.lr_actor = float(actor_lr if actor_lr is not None else (ac.actor_lr or 0.0))
        self.hp.lr_critic = float(critic_lr if critic_lr is not None else ac.critic_lr)
        obs = data['obs']
        M, B = obs.shape[0], self.batch_size
        nmb = (M + B - 1) // B
        total = self.update_iters * nmb
        if self._stats is None or self._stats.shape[0] < total:
            self._stats = torch.zeros(total, NSTAT, dtype=torch.float32, device=ac.device)
        stats = self._stats
        if self.update_actor:
            self.snapshot_old_distribution(obs)
        update_counts, final_kl, step = 0, 0.0, 0
        kl_dev = None
        self._pass_fn = None
        self._graphed_pass = False
        # (with the minibatch's chunks on cooperating workgroups the persistent pass wins up to 1024 rows: 24.6 us per
        # step against 50.9 for the per-step launches; one workgroup walking through 16 chunks: 127)
        pmb = self.persistent_max_batch if self._chunk.get('off') or self.ext is not None or os.environ.get(
            'OSA_CHUNKED_PASS', '1') == '0' else max(self.persistent_max_batch, 1024)
        if (self.persistent and not self.general and (self.ext is None or B <= 64) and not dist.collectives_active()
                and B <= pmb and bool(
                self.lib.osa_ppo_pass_supported(ac.obs_dim, ac.act_dim, ac.hidden))):
            self._pass_fn = ('osa_ppo_pass_kernel', self.lib.osa_ppo_pass)
        self._use_wide = False
        if (self._pass_fn is None and self.persistent and not self.general and self.ext is None
                and not dist.collectives_active() and B <= 64 and self.loss_kind in (0, 1)
                and bool(self.lib.osa_ppo_wide_pass_supported(ac.obs_dim, ac.act_dim, ac.hidden))):
            self._pass_fn = ('osa_wide_pass_kernel', self.lib.osa_ppo_wide_pass)
            self._use_wide = True
            if getattr(self, '_wide_ws', None) is None:
                n = self.lib.osa_ppo_wide_pass_ws_floats(ac.obs_dim, ac.act_dim, ac.hidden)
                self._wide_ws = torch.empty(n, dtype=torch.float32, device=ac.device)
            self._split_alloc()
        use_pass = self._pass_fn is not None
        W = dist.world_size()
        if not self.general:  # (the layer-wise path gathers its rows into aligned scratch itself)
            data = self._aligned_rows(data)
        # one-shot peer exchange: the single-GPU pass on this rank's rows + peer writes of the clipped gradients
        use_p2p = dist.collectives_active() and self._p2p_ok(data) and self._p2p_setup()
        self._mode = 'allreduce' if self.dp_mode == 'p2p' else self.dp_mode  # (p2p not applicable: per-step all-reduce)
        use_repl = (dist.collectives_active() and not self.general and self.ext is None and self.update_critics
                    and self.dp_mode in ('replicated', 'replicated-steps') and B <= self.persistent_max_batch and bool(
            self.lib.osa_ppo_pass_supported(ac.obs_dim, ac.act_dim, ac.hidden)))
        # wide observations (BASELINE config 4): the data-parallel form of the split pass
        self._repl_wide = (not use_repl and not self.general and dist.collectives_active() and self.update_critics
                           and self.dp_mode in ('replicated', 'replicated-steps') and self._wide_dp_fits(W))
        use_repl = use_repl or self._repl_wide
        if use_repl:
            gathered = self._aligned_rows(self.gather_for_replicated(data, W))
        # which machinery ran (tests assert the timed path, not a fallback)
        self.last_path = ('replicated-wide-split' if self._repl_wide else 'replicated') if use_repl else (
            ('persistent-wide' if self._use_wide else 'persistent') if use_pass else ('p2p' if use_p2p else 'per-step'))
        # all passes' permutations in one launch (one shuffle per row, DataLoader(shuffle=True) semantics) instead of
        # update_iters separate randperm launches
        all_perms = None
        # all passes as ONE captured graph when nothing on the host sits between them (see _graph_pass)
        whole = (perms is None and not use_repl and not use_pass and not use_p2p and self._graph_pass_ok(M)
                 and not (self.update_actor and self.kl_early_stop)
                 and self.update_iters * nmb * (16 if self.general else 1) <= 1024
                 and os.environ.get('OSA_UPDATE_GRAPH_WHOLE', '1') != '0')
        if whole:
            self._graph_pass(data, None, lagrange, stats[:total], passes=self.update_iters)
            step, update_counts = total, self.update_iters
            last_perm = self._ug['perm'][(self.update_iters - 1) * M:]
            if self.update_actor:
                kl_dev = self.kl(obs)
        elif perms is None and not use_repl:
            all_perms = self.shuffles(self.update_iters, M)
        for i in range(0 if whole else self.update_iters):
            if use_repl:
                perm = None
            elif perms is not None:
                perm = torch.as_tensor(perms[i]).to(ac.device, torch.int64)
            else:
                perm = all_perms[i]
            if use_repl:
                pa = None if perms is None else torch.as_tensor(perms[i]).to(ac.device, torch.int64)
                self.run_pass_replicated(gathered, M, W, lagrange, stats[step:step + nmb], perms_all=pa)
                step += nmb
            elif use_pass:  # one persistent launch for the whole pass
                self.run_pass(data, perm, lagrange, stats[step:step + nmb])
                step += nmb
            elif use_p2p:  # the same launch per rank, gradients exchanged by peer writes
                self.run_pass_p2p(data, perm, lagrange, stats[step:step + nmb])
                step += nmb
            elif self._graph_pass_ok(M):  # large minibatches: the pass's steps (two launches each) as one hipGraph
                self._graph_pass(data, perm, lagrange, stats[step:step + nmb])
                step += nmb
            else:
                for s in range(0, M, B):
                    nb = min(B, M - s)
                    self.minibatch(data, perm[s:s + nb], nb, lagrange, stats[step])
                    step += 1
                self._ar_end_pass()
            update_counts += 1
            last_perm = self._dp['perm'][dist.rank()] if use_repl else perm
            # the KL pass feeds the early-stop test; without early stop only the last pass's value is
            # logged (policy_gradient.py:383-404), so the earlier full-batch passes are not computed
            if self.update_actor and (self.kl_early_stop or i == self.update_iters - 1):
                kl_dev = self.kl(obs)
                if self.kl_early_stop:
                    final_kl = float(kl_dev)  # the one host sync per pass
                    if use_repl:  # the stream is drained anyway: see a lost peer before the KL decision
                        self.check_dp_sync()
                        self.check_wide_dp_sync()
                    if use_p2p:
                        self.check_p2p_sync()
                    if final_kl > self.target_kl:
                        break
        if use_repl:
            self.check_dp_sync()  # (run() ends in a host read of the statistics anyway)
            self.check_wide_dp_sync()
        if use_p2p:
            self.check_p2p_sync()  # (a host read; run() ends in one of the statistics anyway)
        if not use_repl and not use_pass and not use_p2p and B > 64 and not self.general:
            self.check_reduce_sync()
        if use_p2p:
            pass
        elif not use_repl and not use_pass and B >= 2048 and dist.collectives_active() and self.ext is None:
            # (what ran, for the tests and the bench line: the captured pass incl. its RCCL all-reduces, or eager steps)
            self.last_path = 'dp-large-batch-graph' if getattr(self, '_graphed_pass', False) else 'dp-large-batch'
        elif (not use_repl and not use_pass and dist.collectives_active() and self._mode == 'allreduce'
              and getattr(self, '_graphed_pass', False)):
            self.last_path = 'allreduce-graph'  # (eager steps -- gloo, or before the capture -- stay 'per-step')
        if self.general:
            self.last_path = 'general-' + self.last_path  # (layer-wise GEMM path, csrc/general_mlp.hip)
        if self._use_wide:
            self.check_split_sync()
        if self.last_path == 'persistent-chunked':
            self.check_chunk_sync()
        used = stats[:step]
        # rows of the LAST minibatch of the last executed pass (the reference logs Value/Adv from the loop variable
        # that shadows the full batch: policy_gradient.py:369-377, 402)
        out = {'stop_iter': update_counts, 'steps': step, 'stats': used,
               'last_minibatch': last_perm[(nmb - 1) * B:M]}
        if self.update_actor:
            out['kl'] = float(kl_dev) if kl_dev is not None else final_kl
        return out

    @staticmethod
    def summarize(run_out: dict, critic_norm_coef: float, use_critic_norm: bool) -> dict:
        """Per-key means over the optimiser steps of the epoch (what Logger.get_stats averages,
        omnisafe/common/logger.py:359-374), from ONE device->host copy."""
        s = run_out['stats'].cpu().numpy().astype(np.float64)  # numpy: no OpenMP region per reduction
        coef = critic_norm_coef if use_critic_norm else 0.0
        loss_r = s[:, 0] + coef * s[:, 5]
        loss_c = s[:, 1] + coef * s[:, 6]
        return {
            'Loss/Loss_reward_critic': float(loss_r.mean()), 'Loss/Loss_cost_critic': float(loss_c.mean()),
            'Loss/Loss_pi': float(s[:, 2].mean()), 'Train/PolicyRatio': float(s[:, 3].mean()),
            'Train/PolicyRatio/Std': float(s[:, 3].std()) if len(s) > 1 else 0.0,
            'Train/Entropy': float(s[:, 4].mean()),
            'per_step': {'loss_r': loss_r, 'loss_c': loss_c, 'loss_pi': s[:, 2], 'ratio_mean': s[:, 3],
                         'entropy': s[:, 4]},
        }
