"""CPU oracle for the DeepCV hot path (TEST INFRASTRUCTURE, not product code).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
package.  See `oracle/deepcv_oracle.py` for the restatement and its pinning status.
"""
