"""Import the (read-only, Python) NeRFool reference in THIS container only.

Used exclusively by tests/golden/make_golden.py and tests/golden/check_oracle_vs_reference.py
to produce / re-verify the committed golden vectors.  /root/reference does not exist on the GPU
box, so nothing imported by the pytest suite, bench.py or __graft_entry__ may import this file.

Stubbing recipe: modules the image lacks (cv2, imageio, tensorflow, ...) are replaced by
permissive ModuleType subclasses that carry a real __spec__ (torch._dynamo's find_spec walks
sys.modules when torch.optim.Adam is constructed) and `.cuda()` is neutralised.
"""
import importlib.machinery
import os
import sys
import types

REF_ROOT = '/root/reference'


class _Anything:
    """Callable / attribute sink used for every symbol of a stubbed module."""

    def __init__(self, name='stub'):
        self._name = name

    def __call__(self, *a, **k):
        return _Anything(self._name + '()')

    def __getattr__(self, item):
        if item.startswith('__') and item.endswith('__'):
            raise AttributeError(item)
        return _Anything(self._name + '.' + item)

    def __iter__(self):
        return iter(())

    def __getitem__(self, item):
        return _Anything(self._name + '[]')

    def __contains__(self, item):
        return False


class _StubModule(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith('__') and item.endswith('__'):
            raise AttributeError(item)
        return _Anything(self.__name__ + '.' + item)


_STUBS = ['cv2', 'imageio', 'tensorflow', 'configargparse', 'tensorboardX', 'lpips_tensorflow',
          'lpips', 'torchvision', 'torchvision.transforms', 'matplotlib', 'matplotlib.pyplot',
          'matplotlib.cm', 'matplotlib.backends', 'matplotlib.backends.backend_agg',
          'matplotlib.figure', 'skimage', 'skimage.metrics', 'skimage.transform']


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'ibrnet'))


def install(flavour='ibrnet'):
    """Make `import ibrnet...` / `import eval_adv` resolve to the reference. Returns nothing."""
    if not available():
        raise RuntimeError('reference tree not present; golden vectors can only be made in the build container')
    os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
    sys.dont_write_bytecode = True
    import torch
    for name in _STUBS:
        if name in sys.modules:
            continue
        try:
            importlib.import_module(name)
            continue
        except Exception:
            pass
        mod = _StubModule(name)
        mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
        mod.__path__ = []
        sys.modules[name] = mod
    torch.Tensor.cuda = lambda self, *a, **k: self
    paths = [REF_ROOT, os.path.join(REF_ROOT, 'eval', flavour)]
    for p in reversed(paths):
        if p not in sys.path:
            sys.path.insert(0, p)
