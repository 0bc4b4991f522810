"""ORACLE -- CPU restatement of the reference hot path (test infrastructure only).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import anything from this package; the product (`mmt_amd`) never does.
"""
