"""Accelerated on-policy algorithms; same names as the reference's registry entries."""
from . import registry
from .policy_gradient import PPO, PolicyGradient, PPOLag
from .trust_region_algos import CPO, TRPO, NaturalPG, TRPOLag
from .siblings import (CPPOPID, CUP, FOCOPS, IPO, P3O, PCPO, PDO, RCPO, TRPOPID, OnCRPO, PPOSaute,
                       PPOSimmerPID, TRPOSaute, TRPOSimmerPID)

ALGORITHMS = {'on-policy': tuple(sorted(registry.REGISTRY._module_dict))}  # noqa: SLF001
__all__ = ['PolicyGradient', 'PPO', 'PPOLag', 'NaturalPG', 'TRPO', 'TRPOLag', 'CPO', 'PDO', 'RCPO', 'IPO',
           'OnCRPO', 'CPPOPID', 'TRPOPID', 'PCPO', 'FOCOPS', 'CUP', 'P3O', 'PPOSaute', 'TRPOSaute',
           'PPOSimmerPID', 'TRPOSimmerPID', 'registry', 'ALGORITHMS']
