"""CPU restatement of the reference's Chebyshev-conv path.  Test infrastructure only:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
