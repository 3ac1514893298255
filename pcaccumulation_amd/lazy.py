"""Scalar results that the reference reads with `.item()` (models/egomotion.py:456, models/alignnet.py:280-281,
libs/loss.py:30-35,226), kept asynchronous: every `.item()` drains the launch queue, and a step has two places where a batch of
them is produced (end of MotionNet.forward, end of FuseLoss.forward) right before thousands of small backward launches.  Here the
values are copied to pinned host memory without waiting and turn into the reference's Python numbers the first time somebody
reads them from the result dictionary -- same keys, same types, one wait at most, and none at all before the backward pass
has been queued."""
import torch


class HostCopy(object):
    """One device -> host copy of a flat tensor, asynchronous for GPU tensors; numpy() waits for it (once)."""

    def __init__(self, flat):
        flat = flat.detach()
        if flat.is_cuda:
            self.host = torch.empty(flat.shape, dtype=flat.dtype, pin_memory=True)
            self.host.copy_(flat, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
        else:
            self.host, self.event = flat, None
        self._np = None

    def numpy(self):
        if self._np is None:
            if self.event is not None:
                self.event.synchronize()
            self._np = self.host.numpy()
        return self._np


class LazyValue(object):
    """Elements [lo, hi) of a HostCopy, converted by `convert` when read."""

    def __init__(self, copy, lo, hi, convert):
        self.copy, self.lo, self.hi, self.convert = copy, lo, hi, convert

    def get(self):
        return self.convert(self.copy.numpy()[self.lo:self.hi])


def lazy_scalars(tensors, convert=float):
    """0-d device tensors -> LazyValues sharing one transfer."""
    copy = HostCopy(torch.stack([t.detach().double().reshape(()) for t in tensors]))
    return [LazyValue(copy, i, i + 1, lambda v, c=convert: c(v[0])) for i in range(len(tensors))]


class LazyDict(dict):
    """dict whose LazyValue entries become their value on first read (`[]`, get, items, values, pop); raw() hands an entry over
    untouched so that another LazyDict can carry it along."""

    def _settle(self, key, value):
        if isinstance(value, LazyValue):
            value = value.get()
            dict.__setitem__(self, key, value)
        return value

    def __getitem__(self, key):
        return self._settle(key, dict.__getitem__(self, key))

    def get(self, key, default=None):
        return self[key] if key in self else default

    def pop(self, key, *default):
        if key in self:
            self[key]
        return dict.pop(self, key, *default)

    def raw(self, key):
        return dict.__getitem__(self, key)

    def resolve(self):
        for key in list(self.keys()):
            self[key]
        return self

    def items(self):
        self.resolve()
        return dict.items(self)

    def values(self):
        self.resolve()
        return dict.values(self)


def raw(mapping, key):
    return mapping.raw(key) if isinstance(mapping, LazyDict) else mapping[key]
