"""CPU oracle for the NeRFool adversarial inner loop (IBRNet flavour, GNT flavour in gnt_ref.py).

TEST INFRASTRUCTURE ONLY.  This package is a plain PyTorch-CPU fp32 restatement of the reference
algorithm (GATECH-EIC/NeRFool, files cited per function).  It exists to check the HIP path and to be
timed as the `cpu_baseline` ("port") leg of bench.py.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; nothing under nerfool_amd/ does, and the product path
raises if its HIP library is missing instead of falling back to this code.

Parity pin: the reference ships no tests or golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, produced in the build container by
tests/golden/make_golden.py (imports /root/reference with stubbed cv2/tensorflow/...) and committed as
tests/golden/*.npz.  tests/test_oracle_golden.py replays every fixture through this package.
"""
