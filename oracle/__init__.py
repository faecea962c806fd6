"""CPU oracle for the BiFold VLM training-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``bifold_amd/`` may import this
package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the timed CPU baseline.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the reference's own
modules from ``/root/reference`` (this container only), runs them on seeded
inputs and freezes inputs + outputs under ``tests/golden/``; the oracle is
checked against those fixtures in ``tests/test_oracle_golden.py``.
"""
