"""Device-resident running normaliser -- mirror of omnisafe/common/normalizer.py:27-158.

Same state names (``_mean, _sumsq, _var, _std, _count, _clip``) so that ``state_dict()`` written into
``torch_save/epoch-N.pt`` under key ``obs_normalizer`` is loadable by the reference Evaluator."""
from __future__ import annotations

from collections import OrderedDict

import torch

from . import _lib


class Normalizer:
    def __init__(self, shape: tuple[int, ...], clip: float = 1e6, device='cuda:0') -> None:
        self._lib = _lib.load(require_gpu=True)
        self._shape = tuple(shape)
        self.device = torch.device(device)
        D = 1
        for n in self._shape:  # any shape, as the reference: statistics are kept flat, exposed reshaped
            D *= int(n)
        self._D = D
        f32 = dict(dtype=torch.float32, device=self.device)
        self._mean = torch.zeros(D, **f32)
        self._sumsq = torch.zeros(D, **f32)
        self._var = torch.zeros(D, **f32)
        self._std = torch.zeros(D, **f32)
        self._count = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._clip = clip * torch.ones(D, **f32)
        self._clip_value = float(clip)
        self._ws: torch.Tensor | None = None

    @property
    def shape(self):
        return self._shape

    @property
    def mean(self) -> torch.Tensor:
        return self._mean.view(self._shape)

    @property
    def std(self) -> torch.Tensor:
        return self._std.view(self._shape)

    def __call__(self, data: torch.Tensor) -> torch.Tensor:
        """nn.Module.forward of the reference (normalizer.py:84-86)."""
        return self.normalize(data)

    def _workspace(self, N: int) -> torch.Tensor:
        need = self._lib.osa_normalizer_ws_doubles(N, self._D)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.zeros(need, dtype=torch.float64, device=self.device)  # ticket word starts at 0
        return self._ws

    def push(self, data: torch.Tensor, mask: torch.Tensor | None = None) -> None:
        """_push (normalizer.py:109-139) with the rows selected by ``mask`` (uint8/bool (N,))."""
        x = data.reshape(-1, self._D)
        N, D = x.shape
        m = None if mask is None else mask.to(torch.uint8)
        _lib.check(self._lib.osa_normalizer_push(
            _lib.ptr(x), x.stride(0), N, D, _lib.ptr(m), _lib.ptr(self._mean), _lib.ptr(self._sumsq),
            _lib.ptr(self._var), _lib.ptr(self._std), _lib.ptr(self._count),
            _lib.ptr(self._workspace(N)), _lib.stream_ptr()), 'osa_normalizer_push')

    def normalize(self, data: torch.Tensor, mask: torch.Tensor | None = None,
                  out: torch.Tensor | None = None) -> torch.Tensor:
        """normalize (normalizer.py:88-107): push, then clamp((x - mean)/std, -clip, clip)."""
        x = data.reshape(-1, self._D).to(self.device, torch.float32)
        if x.stride(-1) != 1:
            x = x.contiguous()
        self.push(x, mask)
        N, D = x.shape
        y = out if out is not None else torch.empty(N, D, dtype=torch.float32, device=self.device)
        m = None if mask is None else mask.to(torch.uint8)
        _lib.check(self._lib.osa_normalizer_apply(
            _lib.ptr(x), x.stride(0), _lib.ptr(y), y.stride(0), N, D, _lib.ptr(m), _lib.ptr(self._mean),
            _lib.ptr(self._std), _lib.ptr(self._count), self._clip_value, _lib.stream_ptr()),
            'osa_normalizer_apply')
        return y.reshape(data.shape) if out is None else y

    def state_dict(self) -> 'OrderedDict[str, torch.Tensor]':
        sh = self._shape
        return OrderedDict([('_mean', self._mean.clone().view(sh)), ('_sumsq', self._sumsq.clone().view(sh)),
                            ('_var', self._var.clone().view(sh)), ('_std', self._std.clone().view(sh)),
                            ('_count', self._count[0].clone()), ('_clip', self._clip.clone().view(sh))])

    def load_state_dict(self, sd) -> None:
        for k in ('_mean', '_sumsq', '_var', '_std', '_clip'):
            getattr(self, k).copy_(torch.as_tensor(sd[k]).to(self.device).reshape(-1))
        self._count[0] = int(sd['_count'])
