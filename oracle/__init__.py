"""CPU ORACLE package -- test infrastructure, NOT the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this.  See oracle/rbl_oracle.h for the parity pin.
"""
from .oracle import Oracle, RefPair, oracle_lib_path, ref_lib_path  # noqa: F401
