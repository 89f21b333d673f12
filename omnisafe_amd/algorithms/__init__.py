"""Accelerated on-policy algorithms; same names as the reference's registry entries."""
from . import registry
from .policy_gradient import PPO, PolicyGradient, PPOLag

ALGORITHMS = {'on-policy': tuple(sorted(registry.REGISTRY._module_dict))}  # noqa: SLF001
__all__ = ['PolicyGradient', 'PPO', 'PPOLag', 'registry', 'ALGORITHMS']
