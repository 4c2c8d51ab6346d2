"""TEST INFRASTRUCTURE ONLY (tests/conftest.py installs it when X265HIP_EMU=1): with the emulated library of tests/emu ("device" memory is the host heap) the tests' device
tensors are ordinary CPU tensors.  This maps the handful of torch.cuda entry points the tests and the Python plumbing use onto the CPU, so that the `-m gpu` tests can drive the
emulated library unchanged.  Never active on a GPU box, never imported by the package."""
import types

import torch


def _host(kw):
    d = kw.get("device")
    if d is not None and str(d).startswith("cuda"):
        kw["device"] = "cpu"
    return kw


def _wrap(f):
    def g(*a, **k):
        return f(*a, **_host(k))
    g.__name__ = getattr(f, "__name__", "wrapped")
    return g


class _Event:
    def __init__(self, *a, **k):
        pass

    def record(self, *a, **k):
        pass

    def synchronize(self):
        pass

    def elapsed_time(self, other):
        return 1.0          # (a millisecond: the tests print rates from it)


class _Stream:
    cuda_stream = 0

    def __init__(self, *a, **k):
        pass

    def synchronize(self):
        pass

    def wait_stream(self, s):
        pass

    def wait_event(self, e):
        pass

    def record_event(self, e=None):
        return e or _Event()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def install():
    for name in ("zeros", "empty", "ones", "full", "tensor", "arange", "zeros_like", "empty_like", "as_tensor", "randint", "rand", "randn"):
        setattr(torch, name, _wrap(getattr(torch, name)))
    torch.Tensor.cuda = lambda self, *a, **k: self.clone()          # a copy, as a transfer to the device is (the source array must not alias the "device" tensor)
    real_cpu = torch.Tensor.cpu
    torch.Tensor.cpu = lambda self, *a, **k: real_cpu(self, *a, **k).clone()
    real_to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, (str, torch.device)) and str(x).startswith("cuda")) else x for x in a)
        return real_to(self, *a, **_host(k))
    torch.Tensor.to = to
    c = torch.cuda
    c.is_available = lambda: True
    c.synchronize = lambda *a, **k: None
    c.device_count = lambda: 1
    c.set_device = lambda *a, **k: None
    c.current_device = lambda: 0
    c.get_device_name = lambda *a, **k: "host emulation of tests/emu"
    c.current_stream = lambda *a, **k: _Stream()
    c.Stream = _Stream
    c.Event = _Event
    c.stream = lambda s: s
    c.empty_cache = lambda: None
    c.mem_get_info = lambda *a, **k: (8 << 30, 8 << 30)
