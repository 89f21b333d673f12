"""Minimal Box space used when gymnasium is not installed (the reference accepts only Box on this
path: omnisafe/common/buffer/base.py:73-80, omnisafe/models/base.py:66-74)."""
from __future__ import annotations

import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.asarray(low).shape
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

    def __repr__(self):
        return f'Box(shape={self.shape})'


def is_box(space) -> bool:
    """Duck-typed Box check: gymnasium.spaces.Box, this Box, or anything with a 1-D shape + bounds."""
    return (hasattr(space, 'shape') and hasattr(space, 'low') and hasattr(space, 'high')
            and len(tuple(space.shape)) == 1)
