"""CPU oracle for the PaddleRec models/rank hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``paddlerec_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the timed CPU baseline.

Parity status: **unpinned at the Paddle-kernel boundary** — PaddlePaddle (the
un-vendored dependency that executes every op of the reference's ``net.py``;
``README_EN.md:46,55`` pins it only as ``>=2.0``) is not installable here.  The
restatement is pinned one level above that: ``oracle/make_golden.py`` imports
the reference's *unmodified* ``net.py`` files over ``oracle/paddle_shim`` (a
torch-CPU stand-in for the ~40 paddle symbols they touch) and stores their
outputs under ``tests/golden/``; the NumPy / C restatements here are checked
against those fixtures, against torch-CPU autograd, and against public
known-answer vectors (xxh32).
"""
