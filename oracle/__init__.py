"""CPU oracle for the FA2 hot path -- TEST INFRASTRUCTURE, never product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker (see oracle/naive_attention.c header).
"""
from .cpu_oracle import *  # noqa: F401,F403
