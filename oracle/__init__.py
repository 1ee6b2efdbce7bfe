"""CPU oracle for the batched box-QP ADMM path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``lqp_py_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker / the timed CPU baseline.
"""
