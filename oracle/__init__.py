"""CPU oracle for the gateway compress+hash stage -- TEST INFRASTRUCTURE ONLY.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import
this package.  The product (skyplane_amd/, libskyhip.so) never does.
"""
