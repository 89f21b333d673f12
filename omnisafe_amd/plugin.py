"""Drop-in installation behind the reference's own ``omnisafe.Agent``.

``omnisafe_amd.install()`` (call it after ``import omnisafe``) swaps the registry entries of the
accelerated algorithms for the classes of this package -- ``Registry._register_module`` raises KeyError
on duplicates (omnisafe/algorithms/registry.py:55-58), so the entries are replaced in ``_module_dict``
directly -- and registers the synthetic device environments with the reference's env registry.  After
that ``omnisafe.Agent('PPOLag', env_id, custom_cfgs={'train_cfgs': {'device': 'cuda:0'}}).learn()``
constructs ``omnisafe_amd.algorithms.PPOLag`` with the reference's own Config (YAML defaults) and
runs the HIP path; nothing else in the reference changes.  See INTEGRATION.md.
"""
from __future__ import annotations

ACCELERATED = ('PolicyGradient', 'PPO', 'PPOLag', 'NaturalPG', 'TRPO', 'TRPOLag', 'CPO', 'PDO', 'RCPO', 'IPO',
               'OnCRPO', 'CPPOPID', 'TRPOPID', 'PCPO', 'FOCOPS', 'CUP', 'P3O', 'PPOSaute', 'TRPOSaute',
               'PPOSimmerPID', 'TRPOSimmerPID', 'PPOEarlyTerminated', 'TRPOEarlyTerminated')


_saved_entries: dict[str, type] = {}  # the reference's own classes, kept for uninstall()
_installed: dict[str, type] = {}      # what install() registered: class NAME(omnisafe_amd NAME, reference NAME)


def _as_subclass_of_reference(name: str, ours: type, ref_cls: type) -> type:
    """The class that goes into the reference's registry: named like the reference's, a subclass of BOTH this
    package's implementation (first in the MRO: every hook of the path -- `_init_env`, `_init_model`, `_init`,
    `_init_log`, `learn`, `_update`, ... -- resolves to the HIP implementation) and of the reference's class, so
    that `isinstance(agent.agent, omnisafe.algorithms.PPOLag)` and every `issubclass` test written against the
    reference keep holding after the swap.  The reference's `__init__` is never reached: ours
    does not chain to it."""
    if name in _installed and _installed[name].__mro__[1] is ours and ref_cls in _installed[name].__mro__:
        return _installed[name]
    if issubclass(ours, ref_cls):
        return ours
    import types

    def __new__(cls, *args, **kwargs):  # noqa: N807
        """Dispatch on `cfgs.train_cfgs.device` (the reference's YAML default is `cpu`, PPOLag.yaml:22; BASELINE
        config 1): a CPU run is the reference's business, so the registry call `registry.get(algo)(env_id=...,
        cfgs=...)` (algo_wrapper.py:167-170) then returns an instance of the SAVED reference class -- fully
        constructed by the reference's own `__init__`, incl. its `distributed.fork` / `setup_distributed`
        handling of `train_cfgs.parallel > 1` upstream of this call -- and Python does not run our `__init__` on
        it (it is not an instance of `cls`).  This is the user's explicit device choice, not a fallback: a
        `cuda:N` request without a usable GPU still raises (base_algo.get_device)."""
        cfgs = kwargs.get('cfgs', args[1] if len(args) > 1 else None)
        device = getattr(getattr(cfgs, 'train_cfgs', None), 'device', None)
        if device is not None and _is_cpu_device(device):
            return ref_cls(*args, **kwargs)
        return object.__new__(cls)

    cls = types.new_class(name, (ours, ref_cls), {}, lambda ns: ns.update(
        {'__module__': ours.__module__, '__doc__': ours.__doc__, '__qualname__': name, '__new__': __new__}))
    _installed[name] = cls
    return cls


def _is_cpu_device(device) -> bool:
    import torch

    try:
        return torch.device(device).type == 'cpu'
    except (RuntimeError, TypeError):
        return False


def install(algorithms: tuple[str, ...] | None = None) -> list[str]:
    """Replace the reference's registry entries; returns the names that were swapped."""
    import omnisafe  # the reference; must be importable by the caller's environment
    from omnisafe.algorithms import registry as ref_registry
    from omnisafe.envs import core as ref_env_core

    from . import envs as amd_envs
    from .algorithms import registry as amd_registry

    swapped = []
    for name in (algorithms or ACCELERATED):
        if name in amd_registry.REGISTRY._module_dict and name in ref_registry.REGISTRY._module_dict:  # noqa: SLF001
            ours = amd_registry.get(name)
            current = ref_registry.REGISTRY._module_dict[name]  # noqa: SLF001
            if not issubclass(current, ours):  # (a second install() finds its own class there)
                _saved_entries.setdefault(name, current)
            ref_cls = _saved_entries.get(name, current)
            ref_registry.REGISTRY._module_dict[name] = _as_subclass_of_reference(name, ours, ref_cls)  # noqa: SLF001
            swapped.append(name)
    # make the synthetic ids valid env ids for the reference's config checks (envs/core.py:362-386)
    reg = ref_env_core.ENV_REGISTRY
    known = set(reg.support_envs())
    new_ids = [e for e in amd_envs.SYNTH_DIMS if e not in known]
    if new_ids:
        reg._class['OmnisafeAmdSynthVectorEnv'] = amd_envs.SynthVectorEnv  # noqa: SLF001
        reg._support_envs['OmnisafeAmdSynthVectorEnv'] = new_ids  # noqa: SLF001
    reach_ids = [e for e in amd_envs.ReachVectorEnv._support_envs if e not in known]  # noqa: SLF001
    if reach_ids:
        reg._class['OmnisafeAmdReachVectorEnv'] = amd_envs.ReachVectorEnv  # noqa: SLF001
        reg._support_envs['OmnisafeAmdReachVectorEnv'] = reach_ids  # noqa: SLF001
    del omnisafe
    return swapped


def uninstall() -> list[str]:
    """Put the reference's own classes back (A/B runs of the same script: reference path vs HIP path)."""
    from omnisafe.algorithms import registry as ref_registry

    restored = []
    for name, cls in list(_saved_entries.items()):
        ref_registry.REGISTRY._module_dict[name] = cls  # noqa: SLF001
        restored.append(name)
    _saved_entries.clear()
    return restored
