"""CPU oracle package -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Nothing under dilithium_amd/ does: the product path is HIP only.
"""
