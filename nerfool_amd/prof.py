"""Per-kernel timing with HIP events on the launch stream (used by bench.py for the roofline figures).

When a `KernelTimer` is active every C-ABI launch made through nerfool_amd.ops is bracketed by a pair of
`torch.cuda.Event`s recorded on torch's current stream -- the stream the kernels are enqueued on -- and the elapsed
times are read after one synchronisation at the end, so the timed region itself is not perturbed by host waits."""
import contextlib
from collections import defaultdict

import torch

_active = None


class KernelTimer:
    def __init__(self, only=None):
        """only: iterable of entry-point names to time (None = every launch).  Two event records cost ~6 us of host time,
        which matters once a step is ~250 launches and close to launch-bound: bench.py times only the kernels its roofline
        table prices inside the timed region."""
        self.pairs = defaultdict(list)
        self.meta = defaultdict(list)
        self.only = None if only is None else frozenset(only)

    def summary(self):
        """name -> (launches, mean ms, total ms, list of per-launch metadata) -- synchronises once."""
        torch.cuda.synchronize()
        out = {}
        for name, lst in self.pairs.items():
            ms = [a.elapsed_time(b) for a, b in lst]
            out[name] = {'launches': len(ms), 'mean_ms': sum(ms) / max(1, len(ms)), 'total_ms': sum(ms), 'ms': ms,
                         'meta': self.meta[name]}
        return out


@contextlib.contextmanager
def timing(timer):
    global _active
    prev, _active = _active, timer
    try:
        yield timer
    finally:
        _active = prev


@contextlib.contextmanager
def launch(name, tensor, **meta):
    """Bracket one C-ABI call; a no-op unless a timer is active and `tensor` lives on a GPU."""
    if _active is None or not tensor.is_cuda or (_active.only is not None and name not in _active.only):
        yield
        return
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    stream = torch.cuda.current_stream(tensor.device)      # the tensor's device, which need not be the current one
    a.record(stream)
    yield
    b.record(stream)
    _active.pairs[name].append((a, b))
    _active.meta[name].append(meta)
