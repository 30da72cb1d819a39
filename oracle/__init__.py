"""CPU oracle: TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  Imported by tests/,
__graft_entry__.smoke() and the cpu_baseline leg of bench.py, never by the product."""
