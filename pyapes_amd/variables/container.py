"""Named bags of first / second derivatives (the reference's ``pyapes/variables/container.py``):
``Jac(x=..., y=...)`` and ``Hess(xx=..., xy=...)``; ``len`` counts the components that were given,
iteration yields them in declaration order, ``obj["zx"]`` sorts the key (``xz``) and raises
``KeyError`` for a component that was not given."""
from __future__ import annotations

import torch
from torch import Tensor


class _Derivatives:
    _names: tuple[str, ...] = ()

    def __init__(self, **components: Tensor):
        unknown = set(components) - set(self._names)
        if unknown:
            raise TypeError(f"{type(self).__name__}: unknown component(s) {sorted(unknown)}")
        empty = torch.tensor([])
        for n in self._names:
            setattr(self, n, components.get(n, empty))
        self.keys = [n for n in self._names if getattr(self, n).shape[0] != 0]
        self.max = len(self.keys)

    def __getitem__(self, key: str) -> Tensor:
        name = "".join(sorted(key.lower()))
        item = getattr(self, name, None)
        if item is None or item.shape[0] == 0:
            raise KeyError(f"Derivative: key {key} not found.")
        return item

    def __len__(self) -> int:
        return self.max

    def __iter__(self):
        return iter([getattr(self, k) for k in self.keys])


class Jac(_Derivatives):
    _names = ("x", "y", "z", "r")


class Hess(_Derivatives):
    _names = ("xx", "xy", "xz", "yy", "yz", "zz", "rr", "rz")
