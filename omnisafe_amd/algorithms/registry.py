"""Name -> class registry (mirror of omnisafe/algorithms/registry.py:23-70)."""
from __future__ import annotations

import inspect
from typing import Any


class Registry:
    def __init__(self, name: str) -> None:
        self._name = name
        self._module_dict: dict[str, type] = {}

    @property
    def name(self) -> str:
        return self._name

    def _register_module(self, module_class: type) -> None:
        if not inspect.isclass(module_class):
            raise TypeError(f'module must be a class, but got {type(module_class)}')
        module_name = module_class.__name__
        if module_name in self._module_dict:
            raise KeyError(f'{module_name} is already registered in {self.name}')  # registry.py:55-58
        self._module_dict[module_name] = module_class

    def register(self, cls: type) -> type:
        self._register_module(cls)
        return cls

    def get(self, name: str) -> Any:
        if name in self._module_dict:
            return self._module_dict[name]
        raise KeyError(f'{name} is not in {self.name} registry')


REGISTRY = Registry('omnisafe_amd')
register = REGISTRY.register
get = REGISTRY.get
