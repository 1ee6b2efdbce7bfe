"""Import-path alias: ``lqp_py.<module>`` -> ``lqp_py_amd.<module>``.

The reference's callers import ``lqp_py.solve_box_qp_admm_torch``, ``lqp_py.control`` ... (experiments/
experiment_1.py:6-9, demo/demo_solve_box_qp_torch.py:3-4).  With this directory on ``sys.path`` in place of the
reference package, those imports resolve to the MI355X layer: no code lives here, every submodule IS the
``lqp_py_amd`` module of the same name (``lqp_py.solve_box_qp_admm_torch.SolveBoxQP is lqp_py_amd.SolveBoxQP``).
Modules of the reference outside the box-QP path (``scs_qp``) have no counterpart and raise ImportError as any
missing module would.
"""
import importlib
import sys

_ALIASED = ("solve_box_qp_admm_torch", "lu_layer", "solve_qp_eqcon_torch", "solve_qp_uncon_torch", "control", "utils",
            "solve_box_qp_admm", "solve_qp_uncon", "optnet")

for _name in _ALIASED:
    _mod = importlib.import_module("lqp_py_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    globals()[_name] = _mod
del _name, _mod
