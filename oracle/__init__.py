"""TEST INFRASTRUCTURE ONLY.  CPU oracle for the die_amd parity tests: importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never from die_amd/."""
