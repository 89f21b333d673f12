"""Data-parallel communication for the hot path: one process per GPU, torch.distributed with the
``nccl`` backend (= RCCL over xGMI on ROCm) on device tensors, ``gloo`` on CPU tensors (tests).

Replaces the collective call sites of omnisafe/utils/distributed.py:142-393 (C1-C5 in SURVEY.md 2.2).
The reference issues one blocking all-reduce per parameter tensor (avg_grads, :193-198: 19 messages per
minibatch for three nets); here every exchange is ONE flat buffer:
  * gradients of pi, V_r, V_c of one optimiser step   -> all_reduce_avg_(flat)      (C1)
  * Fisher-vector product                              -> all_reduce_avg_(flat)      (C2)
  * scalar KL / loss statistics packed in one vector   -> all_reduce_avg_(vec)       (C3)
  * advantage / episode statistics [sum, n, sumsq]     -> all_reduce_sum_(vec)       (C4)
  * initial parameters                                 -> broadcast_(flat, src=0)    (C5)
All payloads here are <= a few hundred KB, i.e. latency-bound on xGMI (7 point-to-point links per
GPU): fewer, fused messages matter, ring bandwidth does not.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def is_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size() -> int:
    return dist.get_world_size() if is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if is_initialized() else 0


def collectives_active() -> bool:
    """True when the exchange steps of the path must be issued: world_size > 1, or a process group exists
    and OSA_DIST_FORCE_COLLECTIVES=1 -- a world of ONE rank then still runs every RCCL call of the
    multi-GPU path (init, all-reduce, all-gather, broadcast, the DP update modes), which is how the
    single-GPU test box exercises the `nccl` backend (tests/test_rccl_gpu.py)."""
    if not is_initialized():
        return False
    return dist.get_world_size() > 1 or os.environ.get('OSA_DIST_FORCE_COLLECTIVES', '0') == '1'


def init_from_env(device: torch.device | str | None = None) -> bool:
    """Join the process group described by torchrun's environment (RANK/WORLD_SIZE/MASTER_*), as the
    reference does in setup_distributed (omnisafe/utils/distributed.py:59-104).  Returns True when
    running with world_size > 1."""
    if is_initialized():
        return world_size() > 1
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if ws <= 1 and os.environ.get('OSA_DIST_FORCE_COLLECTIVES', '0') != '1':
        return False
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    use_gpu = device is not None and torch.device(device).type == 'cuda'
    if use_gpu:
        torch.cuda.set_device(torch.device(device))
    # OSA_DIST_BACKEND=gloo forces gloo on device tensors (staged through the host): used by the tests
    # that run several ranks on ONE GPU, where RCCL refuses duplicate devices.  Production: nccl = RCCL.
    backend = os.environ.get('OSA_DIST_BACKEND', 'nccl' if use_gpu else 'gloo')
    kwargs = {}
    if use_gpu and backend == 'nccl':
        kwargs['device_id'] = torch.device(device)
    dist.init_process_group(backend=backend, **kwargs)
    return ws > 1


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    if collectives_active():
        if not t.is_contiguous():
            tmp = t.contiguous()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM)
            t.copy_(tmp)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def graph_capturable() -> bool:
    """Collectives of this process group can be recorded into a hipGraph: the `nccl` (= RCCL) backend enqueues
    kernels on a stream; gloo stages through the host and synchronises."""
    return is_initialized() and dist.get_backend() == 'nccl'


def all_reduce_avg_(t: torch.Tensor) -> torch.Tensor:
    """SUM then divide by world size (avg_grads / avg_tensor semantics, distributed.py:160-198).  Over RCCL with a
    power-of-two world the division rides in the collective (ReduceOp.AVG: every operand pre-scaled by 1 / world --
    exact for powers of two, so the bits equal sum / world) instead of a separate elementwise launch behind it."""
    if collectives_active():
        ws = world_size()
        # (float device tensors only; RCCL's AVG pre-scales every operand by 1 / world: exact for a power-of-two world
        # except where a gradient element is so small that the pre-scaled value is subnormal -- |g| < 2^-126 x world,
        # far below Adam's eps -- where sum / world would round once instead of per operand)
        if (ws > 1 and (ws & (ws - 1)) == 0 and t.is_contiguous() and t.is_cuda and t.is_floating_point()
                and dist.get_backend() == 'nccl'):
            dist.all_reduce(t, op=dist.ReduceOp.AVG)
            return t
        all_reduce_sum_(t)
        if ws > 1:
            t.div_(ws)
    return t


def broadcast_(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    if collectives_active():
        dist.broadcast(t, src=src)
    return t


def barrier() -> None:
    if collectives_active():
        dist.barrier()


def all_gather_rows(t: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """Concatenate every rank's tensor along dim 0 (rank r occupies rows r*n .. r*n+n-1).  Used once per
    epoch by the replicated-data update path to hand every rank the whole rollout."""
    ws = world_size()
    if not collectives_active():
        if out is None:
            return t
        out.copy_(t)
        return out
    t = t.contiguous()
    if out is None:
        out = torch.empty((ws * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    if dist.get_backend() == 'nccl':
        dist.all_gather_into_tensor(out, t)
    else:
        parts = list(out.chunk(ws, dim=0))
        dist.all_gather(parts, t)
    return out
