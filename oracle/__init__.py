"""ORACLE: CPU restatements of the reference forward pass.  Test infrastructure only.

PARITY UNPINNED (no TensorFlow in the build container, no reference golden vectors).
Import allowed only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
