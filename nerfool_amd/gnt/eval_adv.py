"""Attack half of eval/gnt/eval_adv.py (optimize_adv_perturb :282-339 with the `criterion=` keyword, the view-specific loop of
:967-1054 and the universal loop of :739-878) for the GNT flavour.  The loss evaluation and the fused PGD update are shared
with the IBRNet flavour (nerfool_amd/eval_adv.py detects the GNT network and switches renderer and criterion); purification
(optimize_purif) and the random-noise defence are out of scope."""
from ..eval_adv import PGDAttack, RayShard, clamp, init_adv_perturb, optimize_adv_perturb  # noqa: F401
from .criterion import Criterion  # noqa: F401
