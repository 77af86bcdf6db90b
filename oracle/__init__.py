"""
TEST INFRASTRUCTURE -- CPU oracle of the Rouse Kalman-filter log-likelihood.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may import
this package.  The product package ``bild_amd`` never does and has no CPU fallback.
"""
