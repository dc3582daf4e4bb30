"""CPU oracle for the onssen STFT-domain separation forward pass.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker / the timed CPU
baseline.  The product path (``onssen_amd``) never imports this package and
fails loudly when its HIP library is missing.
"""
