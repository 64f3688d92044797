"""oracle — TEST INFRASTRUCTURE ONLY (CPU checker for the HIP BuildGraph path).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Nothing under disco_amd/ does.
"""
