"""paddle.nn.initializer — the initialisers the rank nets name; [EXT] defaults."""
import math as _m
import torch as _t


class Constant:
    def __init__(self, value=0.0):
        self.value = value

    def __call__(self, p):
        p.fill_(self.value)


class Normal:
    def __init__(self, mean=0.0, std=1.0):
        self.mean, self.std = mean, std

    def __call__(self, p):
        p.normal_(self.mean, self.std)


class TruncatedNormal(Normal):
    def __call__(self, p):
        _t.nn.init.trunc_normal_(p, self.mean, self.std, self.mean - 2 * self.std,
                                 self.mean + 2 * self.std)


def _fans(p):
    if p.dim() < 2:
        return p.numel(), p.numel()
    rec = 1
    for s in p.shape[2:]:
        rec *= s
    return p.shape[0] * rec, p.shape[1] * rec


class XavierUniform:
    def __call__(self, p):
        fi, fo = _fans(p)
        lim = _m.sqrt(6.0 / (fi + fo))
        p.uniform_(-lim, lim)


class XavierNormal:
    def __call__(self, p):
        fi, fo = _fans(p)
        p.normal_(0.0, _m.sqrt(2.0 / (fi + fo)))


class Uniform:
    """paddle.nn.initializer.Uniform(low=-1.0, high=1.0) [EXT defaults]."""

    def __init__(self, low=-1.0, high=1.0):
        self.low, self.high = low, high

    def __call__(self, p):
        p.uniform_(self.low, self.high)
