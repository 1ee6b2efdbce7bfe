import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.synthetic import create_qp_data
from lqp_py_amd.control import box_qp_control
dev = torch.device("cuda:0")
for B in (4, 128):
    inp = [t.to(dev) for t in create_qp_data(500, B, seed=0)]
    ctl = L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5, max_iters=61, check_solved=1000)   # fixed 61 iterations, one check
    for _ in range(2): L.torch_solve_box_qp(*inp, dict(ctl))
    torch.cuda.synchronize(); _lib.profile(enable=True, reset=True)
    for _ in range(5): L.torch_solve_box_qp(*inp, dict(ctl))
    torch.cuda.synchronize(); pr = _lib.profile(); _lib.profile(enable=False)
    print(f"B={B}: loop {pr['admm_loop'][0]/5*1e3:.1f} us for 61 iterations")
