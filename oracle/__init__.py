"""ORACLE — CPU restatement of the reference's hot-path arithmetic.  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by cvpr2021_vspw_implement_amd/.
Parity status: PINNED against tests/golden/*.npz, generated from the reference itself by tests/golden/make_golden.py."""
