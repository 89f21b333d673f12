"""omnisafe_amd -- MI355X-native on-policy SafeRL hot path (rollout -> GAE buffer -> PPOLag / TRPOLag /
CPO update) behind the interfaces of PKU-Alignment/omnisafe.  See DESIGN.md and INTEGRATION.md."""
from .agent import Agent  # noqa: F401
from .plugin import install, uninstall  # noqa: F401

__version__ = '0.1.0'
