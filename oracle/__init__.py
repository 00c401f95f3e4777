"""oracle/ -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY: importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the product package.
PARITY UNPINNED by the reference (no tests/fixtures there; TensorFlow 1.x cannot run here)."""
