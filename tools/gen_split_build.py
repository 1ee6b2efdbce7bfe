#!/usr/bin/env python3
"""Split build of the HIP library: one translation unit per group of kernels, compiled in parallel.

The library is ONE source (csrc/lqp_amd.hip) whose ~80 kernel instances take ~3.5 minutes of single-threaded device
code generation.  Every kernel is a template, so the host code can be compiled with `extern template` declarations
(no device code at all) while each group's translation unit instantiates its share explicitly -- same headers, same
flags, the same code per kernel as the single-source build.

    python tools/gen_split_build.py            # (re)writes csrc/split/*.inc / *.hip from the BUILT library's kernel list

Run it after a single-source build (LQP_UNITY_BUILD=1) whenever kernel instances were added or removed; an instance
that is missing from the lists is simply compiled in the host translation unit (slower, still correct).
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "lqp_py_amd", "csrc")
OUT = os.path.join(CSRC, "split")

# group -> predicate on the demangled name; first match wins.  Balanced by measured compile time (seconds at -O3).
GROUPS = [
    ("bwd_f", lambda n: re.search(r"k_bwd_chol_solve<\d, true>", n)),
    ("resident_f", lambda n: re.search(r"k_spd_resident<8, 2, true>", n)),
    ("resident_g", lambda n: re.search(r"k_spd_resident<(5|6|7), 2, true>", n)),
    ("resident_h", lambda n: re.search(r"k_spd_resident<\d, \d, true>", n)),
    ("resident_c", lambda n: re.search(r"k_spd_resident<(3|4), 2(, false)?>", n)),
    ("resident_a", lambda n: re.search(r"k_spd_resident<(5|6|7), 2(, false)?>", n)),
    ("resident_b", lambda n: re.search(r"k_spd_resident<8, 2(, false)?>|k_spd_resident<\d, 4(, false)?>", n)),
    ("split_c", lambda n: re.search(r"k_admm_loop_split<(3|4), 512, false, 2>", n)),
    ("split_a", lambda n: re.search(r"k_admm_loop_split<(5|6|7), 512, false, 2>", n)),
    ("split_b", lambda n: "k_admm_loop_split<" in n),
    ("loop_tail", lambda n: re.search(r"k_admm_loop<\w+, \w+, true,", n)),
    ("loop_hot", lambda n: "k_admm_loop<" in n),
    ("dense", lambda n: "k_lu_inverse<" in n or "k_admm_loop_dense" in n),
    ("lu2", lambda n: "k_lu_factor2<" in n),
    ("lu_a", lambda n: re.search(r"k_lu_factor<float, (32|16), true|k_lu_factor_big|k_lu_factor_wide", n)),
    ("lu_b", lambda n: "k_lu_factor<" in n),
    ("spd", lambda n: re.search(r"k_spd_|k_bwd_chol_solve|k_bwd_build_chol", n)),
    ("unroll", lambda n: "k_unroll_" in n or "k_admm_loop_small" in n),
    ("misc", lambda n: True),
]


# instances that are not in the built library yet (added since the last single-source build)
EXTRA = [
    "void lqp::k_bwd_gather_rhs<float>(lqp::BwdParams<float>)",
    "void lqp::k_bwd_gather_rhs<double>(lqp::BwdParams<double>)",
    "void lqp::k_report_info<0>(int const*, int*, int)",
    "void lqp::k_admm_loop_dense_w<float>(lqp::FwdParams<float>, int, int, int, int)",
    "void lqp::k_admm_loop_dense_w<double>(lqp::FwdParams<double>, int, int, int, int)",
    "void lqp::k_admm_loop_dense<float>(lqp::FwdParams<float>, int, int, int)",
    "void lqp::k_admm_loop_dense<double>(lqp::FwdParams<double>, int, int, int)",
    "void lqp::k_lu_inverse<float, false>(float const*, unsigned long, int, int, int const*, int, float*, unsigned long, int, int const*)",
    "void lqp::k_lu_inverse<float, true>(float const*, unsigned long, int, int, int const*, int, float*, unsigned long, int, int const*)",
    "void lqp::k_lu_inverse<double, false>(double const*, unsigned long, int, int, int const*, int, double*, unsigned long, int, int const*)",
    "void lqp::k_lu_factor2<float, 32>(float*, int, int, unsigned long, int*, int, int*, int const*, int const*, unsigned long long*, unsigned long, unsigned int, unsigned long long*, int, int)",
    "void lqp::k_lu_factor2<double, 16>(double*, int, int, unsigned long, int*, int, int*, int const*, int const*, unsigned long long*, unsigned long, unsigned int, unsigned long long*, int, int)",
    "void lqp::k_lu_factor_wide<float>(float*, int, int, unsigned long, int*, int, int*, int const*, int const*, int*, unsigned long, unsigned int, int, unsigned long long*)",
    "void lqp::k_lu_factor_wide<double>(double*, int, int, unsigned long, int*, int, int*, int const*, int const*, int*, unsigned long, unsigned int, int, unsigned long long*)",
    "void lqp::k_lu_factor_big<float>(float*, int, int, unsigned long, int*, int, int*, int const*, int const*)",
    "void lqp::k_lu_factor_big<double>(double*, int, int, unsigned long, int*, int, int*, int const*, int const*)",
    "void lqp::k_unroll_sweep<1>(lqp::FwdParams<float>, lqp::UnrollParams)",
    "void lqp::k_unroll_sweep<16>(lqp::FwdParams<float>, lqp::UnrollParams)",
    "void lqp::k_unroll_outer<0>(float const*, float const*, float*, int, int)",
    "void lqp::k_unroll_sweep_split<8, 1>(lqp::FwdParams<float>, lqp::UnrollParams, unsigned int)",
    "void lqp::k_unroll_sweep_split<8, 16>(lqp::FwdParams<float>, lqp::UnrollParams, unsigned int)",
    "void lqp::k_unroll_sweep_split<7, 16>(lqp::FwdParams<float>, lqp::UnrollParams, unsigned int)",
    "void lqp::k_unroll_sweep_split<6, 16>(lqp::FwdParams<float>, lqp::UnrollParams, unsigned int)",
    "void lqp::k_unroll_sweep_split<5, 16>(lqp::FwdParams<float>, lqp::UnrollParams, unsigned int)",
    "void lqp::k_unroll_scale_colmax<0>(float const*, int, float*, int*, int*)",
    "void lqp::k_unroll_scale_grad<0>(float const*, float const*, float const*, float*, int, float*)",
    "void lqp::k_unroll_scale_vectors<0>(lqp::ScaleVecParams)",
    "void lqp::k_unroll_scale_scatter<0>(float const*, float const*, int const*, int const*, float const*, float*, int)",
    "void lqp::k_admm_loop_small<0>(lqp::FwdParams<float>, int, int, int)",
    "void lqp::k_admm_loop_split<3, 512, false, 2>(lqp::FwdParams<float>, int, int, int)",
    "void lqp::k_admm_loop_split<4, 512, false, 2>(lqp::FwdParams<float>, int, int, int)",
    "void lqp::k_spd_inverse<1>(lqp::FwdParams<float>, int const*)",
    "void lqp::k_spd_inverse<2>(lqp::FwdParams<float>, int const*)",
    "void lqp::k_spd_inverse_dense<1>(float const*, float*, float*, int*, int, int, float*)",
    "void lqp::k_spd_inverse_dense<2>(float const*, float*, float*, int*, int, int, float*)",
]


# instances a built library may still hold that no longer exist in the sources
DROP = [
    "void lqp::k_bwd_chol_solve<0>(lqp::BwdParams<float>)",
    "void lqp::k_bwd_chol_solve<4>(lqp::BwdParams<float>)",
    "void lqp::k_spd_resident<3, 2>(lqp::FwdParams<float>, int const*)",
    "void lqp::k_spd_resident<4, 2>(lqp::FwdParams<float>, int const*)",
    "void lqp::k_spd_resident<5, 2>(lqp::FwdParams<float>, int const*)",
    "void lqp::k_spd_resident<6, 2>(lqp::FwdParams<float>, int const*)",
    "void lqp::k_spd_resident<7, 2>(lqp::FwdParams<float>, int const*)",
    "void lqp::k_spd_resident<8, 2>(lqp::FwdParams<float>, int const*)",
    "void lqp::k_spd_resident<7, 4>(lqp::FwdParams<float>, int const*)",
    "void lqp::k_spd_resident<8, 4>(lqp::FwdParams<float>, int const*)",
    "void lqp::k_lu_inverse<float>(float const*, unsigned long, int, int, int const*, int, float*, unsigned long, int, int const*)",
    "void lqp::k_lu_inverse<double>(double const*, unsigned long, int, int, int const*, int, double*, unsigned long, int, int const*)",
    "void lqp::k_lu_factor_wide<0>(float*, int, int, unsigned long, int*, int, int*, int const*, int const*, int*, unsigned long, unsigned int, int, unsigned long long*)",
    "void lqp::k_lu_factor_wide<0>(float*, int, int, unsigned long, int*, int, int*, int const*, int const*, int*, unsigned long, unsigned int, int)",
    "void lqp::k_unroll_scale_fro<0>(float const*, float const*, int, float*)",
    "void lqp::k_admm_loop_lu2<float>(lqp::FwdParams<float>, int, int, int)",
    "void lqp::k_admm_loop_lu2<double>(lqp::FwdParams<double>, int, int, int)",
    "void lqp::k_lu_factor<float, 32, true, 1024>(float*, int, int, unsigned long, int*, int, int*, int const*, unsigned long long*, int const*)",
    "void lqp::k_lu_factor_la<16, 1024>(float*, int, int, unsigned long, int*, int, int*, int const*, int const*, unsigned long long*)",
    "void lqp::k_lu_factor_la<32, 768>(float*, int, int, unsigned long, int*, int, int*, int const*, int const*, unsigned long long*)",
    "void lqp::k_lu_factor2<float, 32>(float*, int, int, unsigned long, int*, int, int*, int const*, int const*, unsigned long long*, unsigned long, unsigned int, unsigned long long*, int)",
    "void lqp::k_lu_factor2<double, 16>(double*, int, int, unsigned long, int*, int, int*, int const*, int const*, unsigned long long*, unsigned long, unsigned int, unsigned long long*, int)",
    "void lqp::k_unroll_sweep<0>(lqp::FwdParams<float>, lqp::UnrollParams)",
    "void lqp::k_lu_factor<float, 32, false, 1024>(float*, int, int, unsigned long, int*, int, int*, int const*, unsigned long long*, int const*)",
    "void lqp::k_spd_inverse<0>(lqp::FwdParams<float>, int const*)",
    "void lqp::k_spd_inverse_dense<0>(float const*, float*, float*, int*, int, int, float*)",
]


def kernel_names(lib):
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        subprocess.run(["cp", lib, so], check=True)
        subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", "lib.so"], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        names = []
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            nm = subprocess.run(["nm", "--defined-only", os.path.join(tmp, f)], capture_output=True, text=True, check=True).stdout
            syms = [ln.split()[2] for ln in nm.splitlines() if len(ln.split()) == 3 and ln.split()[1] == "T"]
            dem = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True, check=True).stdout
            names += [d.strip() for d in dem.splitlines() if d.strip()]
    out = set()
    for n in names:
        if n.startswith("lqp::k_"):          # (a library built before these kernels became templates)
            n = "void " + n.replace("(", "<0>(", 1)
        if n.startswith("void lqp::"):
            out.add(n)
    return sorted((out | set(EXTRA)) - set(DROP))


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(CSRC, "liblqp_amd.so")
    names = kernel_names(lib)
    os.makedirs(OUT, exist_ok=True)
    by_group = {g: [] for g, _ in GROUPS}
    for n in names:
        for g, pred in GROUPS:
            if pred(n):
                by_group[g].append(n)
                break
    head = "// generated by tools/gen_split_build.py from the kernel list of a single-source build -- do not edit\n"
    with open(os.path.join(OUT, "lqp_extern.inc"), "w") as f:
        f.write(head + "// host translation unit: no kernel of these lists is instantiated here\n")
        for g, _ in GROUPS:
            f.write(f"// ---- {g}\n")
            for n in by_group[g]:
                f.write("extern template __global__ " + n + ";\n")
    for g, _ in GROUPS:
        with open(os.path.join(OUT, f"lqp_tu_{g}.hip"), "w") as f:
            f.write(head + '#include "../lqp_unroll.hpp"\n')
            for n in by_group[g]:
                f.write("template __global__ " + n + ";\n")
    print({g: len(v) for g, v in by_group.items()}, "kernels:", len(names))


if __name__ == "__main__":
    main()
