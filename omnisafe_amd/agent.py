"""``omnisafe_amd.Agent`` -- mirror of omnisafe.Agent = AlgoWrapper
(omnisafe/algorithms/algo_wrapper.py:36-184): default configs <- custom_cfgs <- train_terminal_cfgs,
epochs = total_steps // steps_per_epoch, registry lookup, ``learn()``.

Process model: the reference re-executes itself under torchrun when ``parallel > 1``
(omnisafe/utils/distributed.py:83-139).  Here ``omnisafe_amd.distributed.init_from_env`` joins the RCCL process
group from torchrun's environment (one process per GPU); the ranks are started either by the caller (``python -m
torch.distributed.run --nproc-per-node N ...``) or by the script itself -- ``python bench.py --gpus N`` re-executes
itself under ``torch.distributed.run`` exactly as the reference's ``fork`` does (distributed.py:121-137).
"""
from __future__ import annotations

import os

from . import distributed as dist
from .algorithms import registry
from .config import Config, check_all_configs, get_default_kwargs, recursive_check_config


class Agent:  # the reference exposes AlgoWrapper under this name (omnisafe/__init__.py:25)
    def __init__(self, algo: str, env_id: str, train_terminal_cfgs: dict | None = None,
                 custom_cfgs: dict | None = None) -> None:
        self.algo, self.env_id = algo, env_id
        self.train_terminal_cfgs, self.custom_cfgs = train_terminal_cfgs, custom_cfgs
        self.cfgs = self._init_config()
        self._init_algo()

    def _init_config(self) -> Config:
        """algo_wrapper.py:75-138."""
        if self.algo not in registry.REGISTRY._module_dict:  # noqa: SLF001
            raise AssertionError(f"{self.algo} doesn't exist in omnisafe_amd "
                                 f'({sorted(registry.REGISTRY._module_dict)})')  # noqa: SLF001
        default = get_default_kwargs(self.algo)
        cfgs = Config.dict2config(default)
        cfgs.recurisve_update({'exp_name': f'{self.algo}-{{{self.env_id}}}', 'env_id': self.env_id,
                               'algo': self.algo})
        exclude = ('exp_name', 'env_id', 'algo', 'exp_increment_cfgs')
        if self.custom_cfgs:
            recursive_check_config(self.custom_cfgs, default, exclude_keys=exclude)
            cfgs.recurisve_update(self.custom_cfgs)
        if self.train_terminal_cfgs:
            recursive_check_config(self.train_terminal_cfgs, default['train_cfgs'])
            cfgs.recurisve_update({'train_cfgs': self.train_terminal_cfgs})
        total, spe = cfgs.train_cfgs.total_steps, cfgs.algo_cfgs.steps_per_epoch
        cfgs.train_cfgs.recurisve_update({'epochs': total // spe})  # algo_wrapper.py:133-136
        return cfgs

    def _init_algo(self) -> None:
        """algo_wrapper.py:149-170."""
        check_all_configs(self.cfgs)
        dev = str(self.cfgs.train_cfgs.device)
        if dist.world_size() > 1 or int(os.environ.get('WORLD_SIZE', '1')) > 1:
            local = int(os.environ.get('LOCAL_RANK', '0'))
            if os.environ.get('OSA_SINGLE_DEVICE_RANKS'):  # test hook: all ranks share cuda:0
                local = 0
            dev = f'cuda:{local}'
            self.cfgs.train_cfgs.recurisve_update({'device': dev})
        os.environ['OMNISAFE_DEVICE'] = dev
        # algo_wrapper.py:160-165: on a GPU device the reference runs the host side on ONE torch thread (the
        # remaining host work is scalar: Lagrange step, logger); without the cap every tiny CPU reduction
        # spins up an OpenMP team as large as the host
        import torch

        torch.set_num_threads(1)
        self.agent = registry.get(self.algo)(env_id=self.env_id, cfgs=self.cfgs)

    def learn(self) -> tuple[float, float, float]:
        """algo_wrapper.py:172-184."""
        return self.agent.learn()
