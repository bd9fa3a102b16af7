"""CPU oracle for the KSPSolve hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  See oracle/sp_oracle.c for the parity status ("parity unpinned"
for the solver; inputs pinned on SURVEY.md Appendix B).
"""
from .oracle import *  # noqa: F401,F403
