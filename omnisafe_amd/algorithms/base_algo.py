"""BaseAlgo (mirror of omnisafe/algorithms/base_algo.py:27-83): construction order
_init_env -> _init_model -> _init -> _init_log; per-rank seed = cfg.seed + 1000 * rank (:40)."""
from __future__ import annotations

import random
from abc import ABC, abstractmethod

import numpy as np
import torch

from .. import distributed as dist


def seed_all(seed: int) -> None:
    """omnisafe/utils/tools.py:132-154."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def get_device(device) -> torch.device:
    """omnisafe/utils/tools.py:338-358 -- except that a missing GPU is an error here, not a silent
    fall back to the CPU: this package has no CPU path."""
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise RuntimeError(f"omnisafe_amd runs on an AMD GPU only (train_cfgs.device='{device}'); use the "
                           'reference implementation for CPU runs')
    if not torch.cuda.is_available():
        raise RuntimeError('no GPU visible to torch (ROCm) -- omnisafe_amd has no CPU fallback')
    return dev


class BaseAlgo(ABC):
    def __init__(self, env_id: str, cfgs) -> None:
        self._env_id = env_id
        self._cfgs = cfgs
        dist.init_from_env(cfgs.train_cfgs.device)
        self._seed = int(cfgs.seed) + dist.rank() * 1000
        seed_all(self._seed)
        self._device = get_device(cfgs.train_cfgs.device)
        if self._device.index is not None:
            torch.cuda.set_device(self._device)
        self._init_env()
        self._init_model()
        self._init()
        self._init_log()

    @property
    def logger(self):
        return self._logger  # pylint: disable=no-member

    @property
    def cost_limit(self):
        return getattr(self._cfgs.algo_cfgs, '_cost_limit', None)

    @abstractmethod
    def _init(self) -> None: ...

    @abstractmethod
    def _init_env(self) -> None: ...

    @abstractmethod
    def _init_model(self) -> None: ...

    @abstractmethod
    def _init_log(self) -> None: ...

    @abstractmethod
    def learn(self) -> tuple[float, float, float]: ...
