"""CPU oracle for the BayesLMs hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Nothing under bayeslms_amd/ imports it; the product path raises
when the HIP library is missing instead of falling back to this code.

Parity status: PINNED.  Every function here is checked against outputs of the
reference itself (imported read-only in the build container by
tests/golden/make_golden.py) through the fixtures committed in tests/golden/.
"""
