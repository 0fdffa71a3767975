"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/cg_oracle.c header).

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product (``conjugategradient_amd``)
never does.
"""
