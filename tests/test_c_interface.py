"""The reference's C interface (interfaces/c/include/piqp.h) served by libpiqp_amd.so: include/piqp_c_compat.h.

CPU part: the header is valid C, its struct layouts are the reference's (sizes / offsets that its C clients were compiled
against), every entry point is exported.  GPU part: a plain C client (tests/c/c_interface_kat.c) built with gcc runs the
reference's own C-interface known answers (interfaces/c/tests/src/c_interface_test.cpp) on the device."""
import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "piqp_amd", "lib")
ENTRY_POINTS = ["piqp_csc_matrix", "piqp_set_default_settings_dense", "piqp_set_default_settings_sparse", "piqp_setup_dense", "piqp_setup_sparse",
                "piqp_update_settings", "piqp_update_dense", "piqp_update_sparse", "piqp_solve", "piqp_cleanup"]


def test_header_is_c_and_layouts_match_the_reference(tmp_path):
    """sizes / offsets computed from the reference's declarations (piqp_typedef.h:27-190) on LP64: piqp_float = double, piqp_int = int"""
    src = tmp_path / "layout.c"
    src.write_text(r'''
#include <stddef.h>
#include <stdio.h>
#include "piqp_c_compat.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(piqp_csc), sizeof(piqp_data_dense), sizeof(piqp_data_sparse), sizeof(piqp_settings), sizeof(piqp_info),
           sizeof(piqp_result), sizeof(piqp_solver_info), sizeof(piqp_workspace));
    printf("%zu %zu %zu %zu %zu %zu\n", offsetof(piqp_settings, check_duality_gap), offsetof(piqp_settings, tau), offsetof(piqp_settings, kkt_solver),
           offsetof(piqp_settings, verbose), offsetof(piqp_info, primal_res), offsetof(piqp_info, run_time));
    printf("%d %d %d %d\n", (int)PIQP_SPARSE_MULTISTAGE, (int)PIQP_INVALID_SETTINGS, (int)PIQP_NUMERICS, (int)PIQP_SOLVED);
    return 0;
}''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", f"-I{INC}", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    sizes = [int(v) for v in out[:8]]
    # piqp_csc: 3 int + pad, 3 pointers; data_*: 3 int + pad, 9 pointers; settings: see offsets; info: 2 x 4 + 22 doubles + ...
    assert sizes[0] == 40 and sizes[1] == 88 and sizes[2] == 88
    assert sizes[6] == 16 and sizes[7] == 8 + 16 + 8
    assert sizes[5] == 10 * 8 + sizes[4]
    offs = [int(v) for v in out[8:14]]
    assert offs[0] == 32                     # 4 doubles before check_duality_gap
    assert offs[1] == 32 + 8 + 5 * 8 + 7 * 4 + 4  # tau: after 5 doubles and 7 ints (+ 4 padding)
    assert offs[2] == offs[1] + 8 and offs[3] == sizes[3] - 8
    assert offs[4] == 8 + 6 * 8 and offs[5] == sizes[4] - 8
    assert [int(v) for v in out[14:]] == [5, -10, -8, 1]


def test_entry_points_exported():
    sys.path.insert(0, ROOT)
    from piqp_amd import _lib  # loads torch's HIP runtime before the library (one HIP runtime per process)
    lib = _lib.load()
    for name in ENTRY_POINTS:
        assert hasattr(lib, name), name
    text = open(os.path.join(INC, "piqp_c_compat.h")).read()
    for name in ENTRY_POINTS:
        assert name + "(" in text


def test_default_settings_without_device():
    """piqp_set_default_settings_* need no GPU: the values of settings.hpp:45-82"""
    sys.path.insert(0, ROOT)
    import piqp_amd  # noqa: F401  (loads torch's HIP runtime first, then the library)
    from piqp_amd import _lib
    L = _lib.load()

    class S(C.Structure):
        _fields_ = [("rho_init", C.c_double), ("delta_init", C.c_double), ("eps_abs", C.c_double), ("eps_rel", C.c_double), ("check_duality_gap", C.c_int),
                    ("eps_duality_gap_abs", C.c_double), ("eps_duality_gap_rel", C.c_double), ("infeasibility_threshold", C.c_double), ("reg_lower_limit", C.c_double),
                    ("reg_finetune_lower_limit", C.c_double), ("a", C.c_int), ("b", C.c_int), ("max_iter", C.c_int), ("max_factor_retires", C.c_int), ("c", C.c_int),
                    ("d", C.c_int), ("preconditioner_iter", C.c_int), ("tau", C.c_double), ("kkt_solver", C.c_int),
                    ("rest", C.c_byte * 128)]  # the remaining fields (the callee writes the whole piqp_settings)
    s = S()
    L.piqp_set_default_settings_sparse(C.byref(s))
    assert (s.rho_init, s.delta_init, s.eps_abs, s.eps_rel, s.max_iter, s.preconditioner_iter, s.tau, s.kkt_solver) == (1e-6, 1e-4, 1e-8, 1e-9, 250, 10, 0.99, 1)
    L.piqp_set_default_settings_dense(C.byref(s))
    assert s.kkt_solver == 0 and s.reg_finetune_lower_limit == 1e-13


@pytest.mark.gpu
def test_c_client_known_answers(tmp_path):
    exe = tmp_path / "c_interface_kat"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", f"-I{INC}", os.path.join(ROOT, "tests", "c", "c_interface_kat.c"), "-o", str(exe), f"-L{LIBDIR}", "-lpiqp_amd",
                    f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-lm"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "all checks passed" in r.stdout
