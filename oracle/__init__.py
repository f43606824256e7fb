"""CPU oracle for the AnoDDPM hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under anoddpm_amd/ imports this package.  Allowed importers: tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
