"""CPU parity oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (visionworkbench_amd) never does.
"""
from .binding import *  # noqa: F401,F403
