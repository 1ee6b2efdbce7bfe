"""Control-dict factory for the box-QP layer.

Mirrors ``box_qp_control`` of the reference (lqp_py/control.py:1-24) key for
key, including the two keys the solver never reads back
(``check_terimnation`` and ``adaptive_rho_max_iter``; the solver reads
``check_solved`` / ``adaptive_max_iter``, solve_box_qp_admm_torch.py:139,148),
so a dict built here behaves exactly like one built by the reference.
"""


def box_qp_control(max_iters=10_000, eps_abs=1e-3, eps_rel=1e-3, check_solved=None,
                   rho=None, rho_min=1e-6, rho_max=1e6, adaptive_rho=True, adaptive_rho_tol=10,
                   adaptive_rho_iter=100, adaptive_rho_max_iter=1000, adaptive_rho_threshold=1e-5,
                   verbose=False, scale=True, beta=None, unroll=False, backward='fixed_point', **kwargs):
    control = dict(
        max_iters=max_iters, eps_abs=eps_abs, eps_rel=eps_rel,
        check_terimnation=check_solved,          # sic: key name of the reference
        rho=rho, rho_min=rho_min, rho_max=rho_max,
        adaptive_rho=adaptive_rho, adaptive_rho_tol=adaptive_rho_tol,
        adaptive_rho_iter=adaptive_rho_iter, adaptive_rho_max_iter=adaptive_rho_max_iter,
        adaptive_rho_threshold=adaptive_rho_threshold,
        verbose=verbose, scale=scale, unroll=unroll, beta=beta, backward=backward,
    )
    control.update(**kwargs)                     # unknown keys (e.g. reduce='max') are carried, ignored
    return control


def optnet_control(max_iters=10, tol=1e-3, check_solved=1, verbose=False, reduce='max', int_reg=1e-6, **kwargs):
    """Control dict of ``OptNet`` (lqp_py/control.py:27-36, same keys, typo included)."""
    control = dict(max_iters=max_iters, tol=tol, check_terimnation=check_solved, verbose=verbose, reduce=reduce,
                   int_reg=int_reg)
    control.update(**kwargs)
    return control
