"""CPU oracle for the VN-PCN hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and there only as the checker / the timed CPU baseline.  The product path
(``vn_pointcloudcompletion_amd``) never imports this package and fails loudly when
the HIP library is missing.
"""
