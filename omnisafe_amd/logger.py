"""Epoch logger -- mirror of the part of omnisafe/common/logger.py:59-389 the on-policy path uses:
registered keys with window deques, ``/Min /Max /Std /Delta`` variants, cross-rank statistics,
``progress.csv`` with the reference's column names, ``torch_save/epoch-N.pt`` checkpoints with keys
``pi`` and ``obs_normalizer``.  Values arrive as Python floats once per epoch (the kernels accumulate
on the device), so the per-step ``.item()`` host syncs of the reference (logger.py:277-278) are gone.
"""
from __future__ import annotations

import csv
import json
import os
import time
from collections import deque
from typing import Any

import numpy as np
import torch

from . import distributed as dist


class Logger:  # pylint: disable=too-many-instance-attributes
    def __init__(self, output_dir: str, exp_name: str, seed: int = 0, use_tensorboard: bool = False,
                 use_wandb: bool = False, config=None, verbose: bool = True) -> None:
        hms = time.strftime('%Y-%m-%d-%H-%M-%S')
        rel = f'seed-{str(seed).zfill(3)}-{hms}'
        self._log_dir = os.path.join(output_dir, exp_name, rel)
        self._maste_proc = dist.rank() == 0
        self._verbose = verbose
        self._epoch = 0
        self._first_row = True
        self._what_to_save: dict[str, Any] | None = None
        self._data: dict[str, deque | list] = {}
        self._headers_windows: dict[str, int | None] = {}
        self._headers_minmax: dict[str, bool] = {}
        self._headers_delta: dict[str, bool] = {}
        self._current_row: dict[str, float] = {}
        self._csv = None
        self._pending = None  # (snapshot of the epoch's values, epoch) of a deferred dump_tabular
        # a deferred row must reach the csv whatever ends the process (sys.exit, an uncaught exception elsewhere):
        # flushed at interpreter exit through a weak reference (learn() also flushes in its `finally`)
        import atexit
        import weakref

        ref = weakref.ref(self)

        def _flush_at_exit() -> None:
            lg = ref()
            try:
                if lg is not None and lg._pending is not None and not lg._output_file.closed:  # noqa: SLF001
                    lg.flush()
            except Exception:  # noqa: BLE001 - interpreter shutdown
                pass

        atexit.register(_flush_at_exit)
        self._atexit_hook = _flush_at_exit  # (unregistered by close(): one closure per Logger otherwise piles up)
        # Optional sinks of the reference (logger.py:130-150, 312-318): the epoch row goes to TensorBoard
        # (`<log_dir>/tb`) and / or Weights & Biases when their packages are importable; when one is asked for and
        # missing, the run continues on csv alone and SAYS so (a drop-in must not silently ignore a config key).
        self._tb_writer = None
        self._wandb = None
        if self._maste_proc and use_tensorboard:
            try:
                from torch.utils.tensorboard.writer import SummaryWriter

                self._tb_writer = SummaryWriter(log_dir=os.path.join(self._log_dir, 'tb'))
            except Exception as exc:  # noqa: BLE001 - the tensorboard package is optional
                import warnings

                warnings.warn(f'omnisafe_amd.Logger: use_tensorboard=True but TensorBoard is unavailable ({exc!r}); '
                              'logging to progress.csv only', RuntimeWarning)
        if self._maste_proc and use_wandb:
            try:
                import wandb

                cfgd = config.todict() if hasattr(config, 'todict') else (dict(config) if config is not None else {})
                lc = cfgd.get('logger_cfgs', {}) if isinstance(cfgd, dict) else {}
                wandb.init(project=lc.get('wandb_project', 'omnisafe'), name=f'{exp_name}-{rel}',
                           dir=self._log_dir, config=cfgd)
                self._wandb = wandb
            except Exception as exc:  # noqa: BLE001 - the wandb package is optional
                import warnings

                warnings.warn(f'omnisafe_amd.Logger: use_wandb=True but wandb is unavailable ({exc!r}); '
                              'logging to progress.csv only', RuntimeWarning)
        if self._maste_proc:
            os.makedirs(self._log_dir, exist_ok=True)
            self._output_file = open(os.path.join(self._log_dir, 'progress.csv'), 'w', encoding='utf-8',
                                     newline='')
            self._csv_writer = csv.writer(self._output_file)
            if config is not None:
                with open(os.path.join(self._log_dir, 'config.json'), 'w', encoding='utf-8') as f:
                    json.dump(config.todict() if hasattr(config, 'todict') else dict(config), f,
                              indent=4, default=str)

    # ------------------------------------------------------------------
    @property
    def current_epoch(self) -> int:
        return self._epoch

    @property
    def log_dir(self) -> str:
        return self._log_dir

    def log(self, msg: str, color: str = 'green', bold: bool = False) -> None:
        if self._maste_proc and self._verbose:
            print(msg, flush=True)

    def register_key(self, key: str, window_length: int | None = None, min_and_max: bool = False,
                     delta: bool = False) -> None:
        """logger.py:196-251."""
        assert key not in self._current_row, f'Key {key} has been registered'
        self._current_row[key] = 0
        if min_and_max:
            for suf in ('/Min', '/Max', '/Std'):
                self._current_row[key + suf] = 0
        if delta:
            self._current_row[key + '/Delta'] = 0
        self._headers_minmax[key], self._headers_delta[key] = min_and_max, delta
        self._headers_windows[key] = window_length
        self._data[key] = deque(maxlen=window_length) if window_length is not None else []

    def window_length(self, key: str) -> int | None:
        """The window a key was registered with (None = all values of the epoch are kept)."""
        return self._headers_windows.get(key)

    def store(self, data: dict[str, Any] | None = None, /, **kwargs: Any) -> None:
        """logger.py:253-282: scalars appended; tensors/arrays contribute their mean."""
        if data is not None:
            kwargs.update(data)
        for key, val in kwargs.items():
            assert key in self._current_row, f'Key {key} has not been registered'
            if isinstance(val, (int, float)):
                self._data[key].append(val)
            elif isinstance(val, torch.Tensor):
                self._data[key].append(val.float().mean().item())
            elif isinstance(val, np.ndarray):
                self._data[key].append(float(val.mean()))
            else:
                raise ValueError(f'Unsupported type {type(val)}')

    def extend(self, key: str, values) -> None:
        """Append many scalars at once (per-episode metrics extracted from the device once per epoch)."""
        assert key in self._current_row, f'Key {key} has not been registered'
        self._data[key].extend(values if isinstance(values, list) else [float(v) for v in values])

    def get_stats(self, key: str, min_and_max: bool = False) -> tuple[float, ...]:
        """logger.py:344-374 via dist_statistics_scalar (distributed.py:361-393): global mean (and
        population std / min / max) over all ranks' values.  `/Min` and `/Max` reproduce what the reference
        WRITES, not what the names say: dist_min / dist_max reduce the vector of stored values element-wise
        across ranks (distributed.py:388-390) and get_stats takes the MEAN of the result (logger.py:366) -- with
        one rank both columns equal the mean.  (Round 1 logged the true extrema; the csv values of a drop-in
        have to be the reference's.)"""
        if not dist.collectives_active():
            return _local_stats(np.asarray(self._data[key], dtype=np.float32), min_and_max)
        vals = torch.tensor(list(self._data[key]), dtype=torch.float32)
        return _dist_stats(vals, min_and_max)

    def dump_tabular(self) -> None:
        """logger.py:284-319: compute this epoch's row, write csv, reset non-window keys.

        The row's statistics, the csv line and the optional sinks are host work the device would sit idle through
        (the epoch's kernels are finished by now, the next rollout is not enqueued yet: 0.15 ms of a 2.9 ms epoch at
        the large-batch setting).  Without cross-rank statistics the epoch's values are therefore only SNAPSHOT here
        and the row is written by `flush()` -- which the adapter calls right after it has enqueued the next epoch's
        rollout, `close()` / `torch_save()` / the next `dump_tabular()` at the latest: same rows, same order, one
        epoch later on disk at most; an exception out of `learn()` and interpreter exit flush it too.  OSA_LOG_DEFER=0
        writes in place, and so does a logger with `verbose` output or an external sink (TensorBoard / wandb): whoever
        watches those sees every epoch when the reference would show it."""
        self.flush()
        live_sinks = self._verbose or self._tb_writer is not None or self._wandb is not None
        if not dist.collectives_active() and not live_sinks and os.environ.get('OSA_LOG_DEFER', '1') != '0':
            snap = {key: np.asarray(self._data[key], dtype=np.float32) for key in self._data}
            for key in self._data:
                if self._headers_windows[key] is None:
                    self._data[key] = []
            self._pending = (snap, self._epoch)
            self._epoch += 1
            return
        self._update_current_row()
        self._write_row(self._epoch)
        self._epoch += 1

    def flush(self) -> None:
        """Write the row a deferred `dump_tabular()` left pending (no-op otherwise)."""
        pending, self._pending = getattr(self, '_pending', None), None
        if pending is None:
            return
        snap, epoch = pending
        for key, vals in snap.items():
            old = self._current_row[key]
            st = _local_stats(vals, self._headers_minmax[key])
            self._current_row[key] = st[0]
            if self._headers_minmax[key]:
                self._current_row[key + '/Min'], self._current_row[key + '/Max'] = st[1], st[2]
                self._current_row[key + '/Std'] = st[3]
            if self._headers_delta[key]:
                self._current_row[key + '/Delta'] = st[0] - old
        self._write_row(epoch)

    def _write_row(self, epoch: int) -> None:
        if self._maste_proc:
            if self._first_row:
                self._csv_writer.writerow(self._current_row.keys())
                self._first_row = False
            self._csv_writer.writerow(self._current_row.values())
            self._output_file.flush()
            if self._tb_writer is not None:
                for key, val in self._current_row.items():
                    self._tb_writer.add_scalar(key, val, global_step=epoch)
                self._tb_writer.flush()
            if self._wandb is not None:
                self._wandb.log(self._current_row, step=epoch)
            if self._verbose:
                width = max(len(k) for k in self._current_row)
                print('\n'.join(f'  {k:<{width}}  {v}' for k, v in self._current_row.items()), flush=True)

    def _update_current_row(self) -> None:
        for key in self._data:
            old = self._current_row[key]
            if self._headers_minmax[key]:
                mean, mn, mx, std = self.get_stats(key, True)
                self._current_row[key] = mean
                self._current_row[key + '/Min'], self._current_row[key + '/Max'] = mn, mx
                self._current_row[key + '/Std'] = std
            else:
                mean = self.get_stats(key, False)[0]
                self._current_row[key] = mean
            if self._headers_delta[key]:
                self._current_row[key + '/Delta'] = mean - old
            if self._headers_windows[key] is None:
                self._data[key] = []

    def setup_torch_saver(self, what_to_save: dict[str, Any]) -> None:
        self._what_to_save = what_to_save

    def torch_save(self) -> None:
        """logger.py:183-194: {'pi': state_dict, 'obs_normalizer': state_dict} (CPU tensors)."""
        self.flush()
        if not self._maste_proc:
            return
        assert self._what_to_save is not None, 'Please setup torch saver first'
        path = os.path.join(self._log_dir, 'torch_save', f'epoch-{self._epoch}.pt')
        os.makedirs(os.path.dirname(path), exist_ok=True)
        params = {}
        for k, v in self._what_to_save.items():
            sd = v.state_dict() if hasattr(v, 'state_dict') else v
            params[k] = {n: t.detach().cpu() for n, t in sd.items()} if isinstance(sd, dict) else sd
        torch.save(params, path)

    def __del__(self) -> None:
        try:  # a logger dropped without close(): the pending row still reaches the csv
            if getattr(self, '_pending', None) is not None and not self._output_file.closed:
                self.flush()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def close(self) -> None:
        self.flush()
        hook = self.__dict__.pop('_atexit_hook', None)
        if hook is not None:
            import atexit

            atexit.unregister(hook)
        if self._maste_proc:
            self._output_file.close()
            if self._tb_writer is not None:
                self._tb_writer.close()
            if self._wandb is not None:
                self._wandb.finish()


def _local_stats(vals: np.ndarray, min_and_max: bool) -> tuple[float, ...]:
    """get_stats without other ranks.  float32 numpy reductions: torch CPU ops would open an OpenMP region per call,
    which costs milliseconds on many-core hosts and sits on the epoch's critical path (the GPU idles meanwhile)."""
    n = vals.size
    if n == 0:
        nan = float('nan')
        return (nan, nan, nan, nan) if min_and_max else (nan,)
    mean = vals.sum(dtype=np.float32) / np.float32(n)
    if not min_and_max:
        return (float(mean),)
    std = np.sqrt(((vals - mean) ** 2).sum(dtype=np.float32) / np.float32(n))
    elem_mean = float(vals.mean(dtype=np.float32))  # min_val.mean() / max_val.mean() of the reference
    return float(mean), elem_mean, elem_mean, float(std)


def _dist_stats(vals: torch.Tensor, min_and_max: bool):
    """Cross-rank statistics through gloo/RCCL: [sum, n] -> mean; [sumsq] -> std; min / max as the reference
    computes them: element-wise MIN / MAX of the ranks' vectors, then the mean (distributed.py:388-390,
    logger.py:366).  The reference requires equally long vectors on all ranks (its all-reduce would fail
    otherwise); with unequal lengths the common prefix is reduced."""
    dev = torch.device('cuda', torch.cuda.current_device()) if (
        torch.cuda.is_available() and torch.distributed.get_backend() == 'nccl') else torch.device('cpu')
    n = len(vals)
    s = torch.tensor([float(vals.sum()) if n else 0.0, float(n)], dtype=torch.float64, device=dev)
    dist.all_reduce_sum_(s)
    if s[1].item() == 0:
        nan = float('nan')
        return (nan, nan, nan, nan) if min_and_max else (nan,)
    mean = (s[0] / s[1]).item()
    if not min_and_max:
        return (mean,)
    sq = torch.tensor([float(((vals.double() - mean) ** 2).sum()) if n else 0.0], dtype=torch.float64,
                      device=dev)
    dist.all_reduce_sum_(sq)
    std = float(torch.sqrt(sq[0] / s[1]))
    nmin = torch.tensor([float(n)], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(nmin, op=torch.distributed.ReduceOp.MIN)
    k = int(nmin.item())
    if k == 0:
        return mean, mean, mean, std
    lo = vals[:k].to(device=dev, dtype=torch.float32).clone()
    hi = lo.clone()
    torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
    torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    return mean, lo.mean().item(), hi.mean().item(), std
