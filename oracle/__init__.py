"""CPU oracle for the EGTR hot path -- TEST INFRASTRUCTURE ONLY.

This package is a from-scratch CPU restatement (plain PyTorch-CPU / numpy, functional style, fp32 or fp64)
of the algorithms on the hot path of naver-ai/egtr, each function citing the reference file:line it follows.
It exists to CHECK the hand-written HIP path in ``egtr_amd`` and to serve as the timed CPU baseline in
``bench.py``.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; nothing under ``egtr_amd/`` does, and the product path raises if its HIP library is missing rather than
falling back to anything in here.

Parity pinning: the reference ships no tests or golden vectors, so the oracle is pinned
against outputs of the reference itself, produced in the build container by importing ``/root/reference``
under import shims (``tests/golden/make_golden.py``) and committed as ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` checks every oracle function against those fixtures.
"""
