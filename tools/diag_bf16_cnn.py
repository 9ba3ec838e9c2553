"""What do plain bf16 operands in the 3x3 convolutions (feature_network.WINO_OPERANDS = 'bf16') cost in accuracy?  Runs the bf16 attack
checks of tests/parity_cases.py with the feature CNN's Winograd products on plain bf16 operands and prints the achieved numbers
(assertion failures are reported, not raised).  usage: python tools/diag_bf16_cnn.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import parity_cases as pc                                         # noqa: E402
from nerfool_amd.ibrnet import feature_network                     # noqa: E402

for operands in ('bf16x3', 'bf16'):
    feature_network.WINO_OPERANDS = operands
    print('==== WINO_OPERANDS =', operands, flush=True)
    for fn in (pc.check_bf16_attack, pc.check_fused_resunet):
        try:
            fn('cuda')
        except AssertionError as e:
            print('ASSERT', fn.__name__, str(e)[:300])
