"""Measured parity errors of the GPU tests, written to gpurun_out/parity_report.json at the end of the session
(committed as profiles/rNN_parity_report.json): per golden case, per output, per x-update path -- the largest
absolute error against the reference-made golden vector, the scale it is judged on, and (where the test computed
one) the error of the reference's own fp32 result and of the HIP result against an fp64 solve of the same inputs."""
import json
import os

_records = []


def record(case, output, err, scale=1.0, **extra):
    rec = {"case": case, "output": output, "max_abs_err": float(err), "scale": float(scale),
           "rel_to_scale": float(err) / (float(scale) + 1e-300)}
    rec.update({k: (float(v) if isinstance(v, (int, float)) and not isinstance(v, bool) else v) for k, v in extra.items()})
    _records.append(rec)
    return rec


def dump():
    if not _records:
        return None
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, "parity_report.json")
        with open(path, "w") as f:
            json.dump({"tolerances": {"north_star": "primal/dual residuals to 1e-5, gradients rtol 1e-4",
                                      "fp64_criterion": "|HIP - fp64| <= |reference fp32 golden - fp64| + eps"},
                       "records": _records}, f, indent=1)
        return path
    except OSError:
        return None
