import os, sys, ctypes, torch
os.environ.setdefault("LQP_ENV_NOCACHE", "1")      # (this tool flips library knobs between solves)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LQP_RESIDENT"] = "0"
import lqp_py_amd as L
from lqp_py_amd import _lib
from lqp_py_amd.synthetic import create_qp_data
from lqp_py_amd.control import box_qp_control
lib = _lib.load()
fn = ctypes.CDLL(_lib.LIB_PATH).lqp_debug_read_cycles
dev = torch.device("cuda:0")
for B in (4, 128):
    inp = [t.to(dev) for t in create_qp_data(500, B, seed=0)]
    ctl = L.box_qp_control(eps_abs=1e-5, eps_rel=1e-5)
    L.torch_solve_box_qp(*inp, dict(ctl)); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)(); fn(buf)
    sol = L.torch_solve_box_qp(*inp, dict(ctl)); torch.cuda.synchronize()
    fn(buf)
    n = buf[3]
    print(f"B={B}: solves {n}  per solve: offdiag {buf[0]/n:.0f}  diag {buf[1]/n:.0f}  vmwait {buf[2]/n:.0f}  | loop total/iter {buf[4]/n:.0f} cycles")
