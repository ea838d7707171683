"""CPU oracle for the TRPL hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
package.  `oracle.lib` is a ctypes binding of oracle/liboracle.so (built from
trpl_oracle.c by oracle/Makefile, or by __graft_entry__.build()).
"""
from .binding import (OracleLib, load, pvsim, fastlog, prob, pcreduce, norm2, scales,  # noqa: F401
                      simulate_loglik, fma_variant)
