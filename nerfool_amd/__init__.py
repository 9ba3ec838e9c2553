"""nerfool_amd: MI355X-native adversarial inner loop of NeRFool.

Build: gfx950 only (`hipcc --offload-arch=gfx950`; the kernels use CDNA4-only instructions such as `v_permlane32_swap` without a
fallback).  The package reads no environment variable; the `NERFOOL_*` kernel-selection switches of earlier revisions are gone
and a process that still sets one is told so once at import."""
import os as _os
import warnings as _warnings

_REMOVED_ENV = ('NERFOOL_CNN', 'NERFOOL_CONV3X3', 'NERFOOL_CONV_S2', 'NERFOOL_GATHER_BWD', 'NERFOOL_GATHER_FUSION',
                'NERFOOL_GNT_KERNELS', 'NERFOOL_IBRNET_KERNELS', 'NERFOOL_IBRNET_PRECISION')
_set = [k for k in _REMOVED_ENV if k in _os.environ]
if _set:
    _warnings.warn('nerfool_amd ignores %s: the environment switches were removed -- the default kernels run; precision is chosen '
                   'by args.ibrnet_precision, test / diagnostic paths by module attributes (DESIGN.md section 1)' % ', '.join(_set),
                   RuntimeWarning, stacklevel=2)
del _set
