"""Configuration objects consumed by the algorithms -- mirror of omnisafe/utils/config.py:27-262.

``Config`` is an attribute dictionary with the reference's recursive-update semantics.  The default
hyper-parameters below are the values of the reference's YAML files
(omnisafe/configs/on-policy/{PPOLag,TRPOLag,CPO}.yaml ``defaults`` blocks); when the algorithms are
used behind the reference's own ``omnisafe.Agent`` the reference passes its own Config built from those
YAML files and this table is not consulted.
"""
from __future__ import annotations

import copy
from typing import Any


class Config:
    """Attribute-style nested config (config.py:27-140)."""

    def __init__(self, **kwargs: Any) -> None:
        for key, value in kwargs.items():
            self[key] = Config.dict2config(value) if isinstance(value, dict) else value

    def __getitem__(self, item: str) -> Any:
        return self.__dict__[item]

    def __setitem__(self, key: str, value: Any) -> None:
        self.__dict__[key] = value

    def __contains__(self, key: str) -> bool:
        return key in self.__dict__

    def __repr__(self) -> str:
        return f'Config({self.todict()})'

    @staticmethod
    def dict2config(d: dict) -> 'Config':
        cfg = Config()
        for k, v in d.items():
            cfg[k] = Config.dict2config(v) if isinstance(v, dict) else v
        return cfg

    def todict(self) -> dict:
        return {k: (v.todict() if isinstance(v, Config) else v) for k, v in self.__dict__.items()}

    def recurisve_update(self, update_args: dict) -> None:  # (sic) reference spelling, config.py:185
        for key, value in update_args.items():
            if key in self.__dict__ and isinstance(self[key], Config) and isinstance(value, dict):
                self[key].recurisve_update(value)
            else:
                self[key] = Config.dict2config(value) if isinstance(value, dict) else value


_COMMON_LOGGER = {'use_wandb': False, 'wandb_project': 'omnisafe', 'use_tensorboard': False,
                  'save_model_freq': 100, 'log_dir': './runs', 'window_lens': 100,
                  'verbose': True}  # 'verbose' is an omnisafe_amd extension (quiet benchmark runs)
_COMMON_MODEL = {
    'weight_initialization_mode': 'kaiming_uniform', 'actor_type': 'gaussian_learning',
    'linear_lr_decay': True, 'exploration_noise_anneal': False, 'std_range': [0.5, 0.1],
    'actor': {'hidden_sizes': [64, 64], 'activation': 'tanh', 'lr': 0.0003},
    'critic': {'hidden_sizes': [64, 64], 'activation': 'tanh', 'lr': 0.0003},
}

# PPOLag.yaml:16-132
_PPOLAG = {
    'seed': 0,
    'train_cfgs': {'device': 'cuda:0', 'torch_threads': 16, 'vector_env_nums': 1, 'parallel': 1,
                   'total_steps': 10000000},
    'algo_cfgs': {
        'steps_per_epoch': 20000, 'update_iters': 40, 'batch_size': 64, 'target_kl': 0.02,
        'entropy_coef': 0.0, 'reward_normalize': False, 'cost_normalize': False, 'obs_normalize': True,
        'kl_early_stop': True, 'use_max_grad_norm': True, 'max_grad_norm': 40.0,
        'use_critic_norm': True, 'critic_norm_coef': 0.001, 'gamma': 0.99, 'cost_gamma': 0.99,
        'lam': 0.95, 'lam_c': 0.95, 'clip': 0.2, 'adv_estimation_method': 'gae',
        'standardized_rew_adv': True, 'standardized_cost_adv': True, 'penalty_coef': 0.0,
        'use_cost': True},
    'logger_cfgs': _COMMON_LOGGER,
    'model_cfgs': _COMMON_MODEL,
    'lagrange_cfgs': {'cost_limit': 25.0, 'lagrangian_multiplier_init': 0.001, 'lambda_lr': 0.035,
                      'lambda_optimizer': 'Adam'},
    'env_cfgs': {},
}

# TRPOLag.yaml:16-136 (differences from PPOLag: no clip, cg_*, critic lr 1e-3, actor lr None, B 128,
# 10 critic iterations, target_kl 0.01, no early stop)
_TRPOLAG = copy.deepcopy(_PPOLAG)
_TRPOLAG['algo_cfgs'].update({'update_iters': 10, 'batch_size': 128, 'target_kl': 0.01,
                              'kl_early_stop': False, 'cg_damping': 0.1, 'cg_iters': 15,
                              'fvp_obs': 'None', 'fvp_sample_freq': 1})
_TRPOLAG['algo_cfgs'].pop('clip')
_TRPOLAG['model_cfgs'] = copy.deepcopy(_COMMON_MODEL)
_TRPOLAG['model_cfgs']['actor']['lr'] = None
_TRPOLAG['model_cfgs']['critic']['lr'] = 0.001
_TRPOLAG['model_cfgs']['linear_lr_decay'] = False  # TRPOLag.yaml (CPO.yaml keeps True)

# CPO.yaml (as TRPOLag, cost_limit in algo_cfgs, no lagrange_cfgs)
_CPO = copy.deepcopy(_TRPOLAG)
_CPO['algo_cfgs']['cost_limit'] = 25
_CPO['model_cfgs']['linear_lr_decay'] = True
_CPO.pop('lagrange_cfgs')

# PPO / TRPO bases (same blocks without the Lagrange section) -- used by the class hierarchy
_PPO = copy.deepcopy(_PPOLAG)
_PPO.pop('lagrange_cfgs')
_PPO['algo_cfgs']['use_cost'] = False          # PPO.yaml:80
_TRPO = copy.deepcopy(_TRPOLAG)
_TRPO.pop('lagrange_cfgs')
_TRPO['algo_cfgs']['use_cost'] = False         # TRPO.yaml

_PG = copy.deepcopy(_PPO)                      # PolicyGradient.yaml: 10 passes, no clip, window 50
_PG['algo_cfgs']['update_iters'] = 10
_PG['algo_cfgs'].pop('clip')
_PG['logger_cfgs'] = dict(_COMMON_LOGGER, window_lens=50)
_NPG = copy.deepcopy(_TRPO)                    # NaturalPG.yaml keeps an (unused) clip entry
_NPG['algo_cfgs']['clip'] = 0.2

# ---- sibling algorithms (configs/on-policy/{PDO,RCPO,IPO,OnCRPO,CPPOPID,TRPOPID,PCPO}.yaml): differences
# from the blocks above only
_PID_LAGRANGE = {'cost_limit': 25.0, 'lagrangian_multiplier_init': 0.001, 'pid_kp': 0.1, 'pid_ki': 0.01,
                 'pid_kd': 0.01, 'pid_d_delay': 10, 'pid_delta_p_ema_alpha': 0.95,
                 'pid_delta_d_ema_alpha': 0.95, 'sum_norm': True, 'diff_norm': False, 'penalty_max': 100.0}


def _derive(base: dict, algo: dict | None = None, drop_algo=(), model: dict | None = None,
            lagrange: dict | None | bool = False) -> dict:
    d = copy.deepcopy(base)
    d['algo_cfgs'].update(algo or {})
    for k in drop_algo:
        d['algo_cfgs'].pop(k, None)
    for k, v in (model or {}).items():
        d['model_cfgs'][k] = v
    if lagrange is None:
        d.pop('lagrange_cfgs', None)
    elif lagrange is not False:
        d['lagrange_cfgs'] = copy.deepcopy(lagrange)
    return d


_PDO = _derive(_PPOLAG, {'reward_normalize': True, 'cost_normalize': True})
_RCPO = _derive(_TRPOLAG)                                   # linear_lr_decay False, as TRPOLag.yaml
_IPO = _derive(_PPOLAG, {'update_iters': 10, 'reward_normalize': True, 'cost_normalize': True, 'kappa': 0.01,
                         'penalty_max': 1.0, 'cost_limit': 25.0})   # IPO.yaml keeps an (unused) lagrange_cfgs
_ONCRPO = _derive(_TRPO, {'cost_limit': 25.0, 'distance': 2.0, 'use_cost': True})
_CPPOPID = _derive(_PPOLAG, lagrange=_PID_LAGRANGE)
_TRPOPID = _derive(_TRPOLAG, {'clip': 0.2}, lagrange=_PID_LAGRANGE)
_PCPO = _derive(_CPO)
_LAG_BOUNDED = dict(_PPOLAG['lagrange_cfgs'], lagrangian_upper_bound=2.0)
_FOCOPS = _derive(_PPOLAG, {'focops_eta': 0.02, 'focops_lam': 1.5}, lagrange=_LAG_BOUNDED)
_CUP = _derive(_PPOLAG, {'target_kl': 0.01}, lagrange=_LAG_BOUNDED)
_P3O = _derive(_PPOLAG, {'update_iters': 10, 'kappa': 20.0, 'cost_limit': 25.0}, lagrange=None)
_SAUTE = {'safety_budget': 25.0, 'saute_gamma': 0.999, 'max_ep_len': 1000, 'unsafe_reward': -1.0}
_PPOSAUTE = _derive(_PPO, _SAUTE)
_TRPOSAUTE = _derive(_TRPO, _SAUTE)
_PPOSIMMER = _derive(_PPO, dict(_SAUTE, upper_budget=25.0))
_TRPOSIMMER = _derive(_TRPO, dict(_SAUTE, upper_budget=25.0))
for _d in (_PPOSIMMER, _TRPOSIMMER):
    _d['control_cfgs'] = {'kp': 0.0005, 'ki': 1e-05, 'kd': 0.0, 'polyak': 0.995}

_PPOET = _derive(_PPO, {'cost_limit': 25.0})     # PPOEarlyTerminated.yaml = PPO.yaml + cost_limit
_TRPOET = _derive(_TRPO, {'cost_limit': 25.0})

DEFAULTS = {'PPOLag': _PPOLAG, 'TRPOLag': _TRPOLAG, 'CPO': _CPO, 'PPO': _PPO, 'TRPO': _TRPO,
            'PolicyGradient': _PG, 'NaturalPG': _NPG, 'PDO': _PDO, 'RCPO': _RCPO, 'IPO': _IPO,
            'OnCRPO': _ONCRPO, 'CPPOPID': _CPPOPID, 'TRPOPID': _TRPOPID, 'PCPO': _PCPO,
            'FOCOPS': _FOCOPS, 'CUP': _CUP, 'P3O': _P3O, 'PPOSaute': _PPOSAUTE, 'TRPOSaute': _TRPOSAUTE,
            'PPOSimmerPID': _PPOSIMMER, 'TRPOSimmerPID': _TRPOSIMMER,
            'PPOEarlyTerminated': _PPOET, 'TRPOEarlyTerminated': _TRPOET}


def get_default_kwargs(algo: str) -> dict:
    """get_default_kwargs_yaml (config.py:235-262) for the accelerated algorithms."""
    if algo not in DEFAULTS:
        raise KeyError(f'{algo} is not an omnisafe_amd algorithm (have {sorted(DEFAULTS)})')
    return copy.deepcopy(DEFAULTS[algo])


def recursive_check_config(config: dict, default: dict, exclude_keys=()) -> None:
    """omnisafe/utils/tools.py:246-270: custom keys must exist in the defaults."""
    assert isinstance(config, dict), 'custom_cfgs must be a dict!'
    for key in config:
        if key not in default and key not in exclude_keys:
            raise KeyError(f'Invalid key: {key}')
        if isinstance(config[key], dict) and key not in ('env_cfgs',):
            recursive_check_config(config[key], default[key])


def check_all_configs(cfgs: Config) -> None:
    """Subset of config.py:265-409 that guards this path."""
    a = cfgs.algo_cfgs
    assert isinstance(a.update_iters, int) and a.update_iters > 0, 'update_iters must be positive int'
    assert isinstance(a.steps_per_epoch, int) and a.steps_per_epoch > 0
    assert isinstance(a.batch_size, int) and a.batch_size > 0
    assert 0.0 <= a.gamma <= 1.0 and 0.0 <= a.lam <= 1.0 and 0.0 <= a.lam_c <= 1.0
    assert a.adv_estimation_method in ('gae', 'gae-rtg', 'vtrace', 'plain')
    assert cfgs.train_cfgs.vector_env_nums >= 1 and cfgs.train_cfgs.parallel >= 1
