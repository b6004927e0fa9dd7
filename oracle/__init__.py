"""CPU oracle for the ALS / IMC / DirtyIMC hot path.  TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package; the product package
(``recommendation-models_amd/alsmi``) never does.
"""
