"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the GeoMop/MLMC
moment-estimation / max-entropy hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker.  The product (``mlmc_amd``) never imports it and
fails loudly when its HIP library is missing.
"""
