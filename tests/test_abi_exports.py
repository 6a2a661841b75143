"""libfasp_hip.so loads on a machine without a GPU and exports every symbol that
include/fasp_hip.h (the drop-in boundary) and include/fasp_hip_dev.h (measurement / test entries) declare; host-only entry points work; compute entry points refuse
to run (there is no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

from _libs import ROOT, T, poisson7pt


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "fasp_hip.h")).read() + open(os.path.join(ROOT, "include", "fasp_hip_dev.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(fasp_[a-z0-9_A-Z]+)\s*\(", src)
    return sorted(set(names))


def test_every_declared_symbol_is_exported(fa):
    L = fa.lib()
    declared = _declared_functions()
    assert len(declared) >= 30
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(set(fa.EXPORTS)) == declared


def test_library_has_no_dependency_on_the_oracle():
    out = subprocess.run(["ldd", os.path.join(ROOT, "faspsolver_amd", "libfasp_hip.so")],
                         capture_output=True, text=True).stdout
    assert "liboracle" not in out and "libfasp_ref" not in out
    for fn in os.listdir(os.path.join(ROOT, "faspsolver_amd", "csrc")):
        if fn.endswith((".cpp", ".hip", ".h")):
            text = open(os.path.join(ROOT, "faspsolver_amd", "csrc", fn)).read()
            assert "oracle/" not in text and "fasp_oracle" not in text, fn


def test_header_compiles_as_c_and_cxx(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "fasp_hip.h"\nint main(void){AMG_param p; ITS_param q; '
                   '_Static_assert(sizeof(dCSRmat)==40,"dCSRmat"); _Static_assert(sizeof(AMG_param)==224,"AMG_param");'
                   '_Static_assert(sizeof(ITS_param)==40,"ITS_param"); _Static_assert(sizeof(dvector)==16,"dvector");'
                   'fasp_param_amg_init(&p); fasp_param_solver_init(&q); return p.max_levels==20?0:1;}\n')
    exe = tmp_path / "t"
    lib = os.path.join(ROOT, "faspsolver_amd")
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", lib,
                    "-lfasp_hip", f"-Wl,-rpath,{lib}"], check=True)
    assert subprocess.run([str(exe)]).returncode == 0
    srcpp = tmp_path / "t.cpp"
    srcpp.write_text('#include "fasp_hip.h"\nint main(){ITS_param q; fasp_param_solver_init(&q); return q.maxit==500?0:1;}\n')
    subprocess.run(["g++", "-I", os.path.join(ROOT, "include"), str(srcpp), "-o", str(exe), "-L", lib,
                    "-lfasp_hip", f"-Wl,-rpath,{lib}"], check=True)
    assert subprocess.run([str(exe)]).returncode == 0


def test_unsupported_parameters_are_refused_before_any_work(fa):
    ia, ja, a, f, ue = poisson7pt(5)
    x = np.zeros(len(f))
    cases = []
    it, am = fa.param_solver_init(), fa.param_amg_init(); am.smoother = 21   # SMOOTHER_BLKOIL: an application-specific smoother
    cases.append((it, am, T.ERROR_AMG_SMOOTH_TYPE))
    it, am = fa.param_solver_init(), fa.param_amg_init(); am.smoother = T.SMOOTHER_JACOBI; am.AMG_type = T.UA_AMG; am.aggregation_type = 3  # NPAIR: unfinished in the reference
    cases.append((it, am, T.ERROR_INPUT_PAR))
    it, am = fa.param_solver_init(), fa.param_amg_init(); am.smoother = T.SMOOTHER_JACOBI; am.cycle_type = 7   # not a cycle type of the reference
    cases.append((it, am, T.ERROR_INPUT_PAR))
    it, am = fa.param_solver_init(), fa.param_amg_init(); am.smoother = T.SMOOTHER_JACOBI; am.ILU_levels = 1
    cases.append((it, am, T.ERROR_INPUT_PAR))
    it, am = fa.param_solver_init(), fa.param_amg_init(); am.smoother = T.SMOOTHER_JACOBI; it.itsolver_type = 13  # SOLVER_SMinRes
    cases.append((it, am, T.ERROR_SOLVER_TYPE))
    it, am = fa.param_solver_init(), fa.param_amg_init(); am.smoother = T.SMOOTHER_JACOBI; am.interpolation_type = 3   # INTERP_ENG
    cases.append((it, am, T.ERROR_AMG_INTERP_TYPE))
    it, am = fa.param_solver_init(), fa.param_amg_init(); am.smoother = T.SMOOTHER_JACOBI
    am.interpolation_type = T.INTERP_RDC; am.coarsening_type = T.COARSE_AC   # undefined in the reference
    cases.append((it, am, T.ERROR_AMG_INTERP_TYPE))
    it, am = fa.param_solver_init(), fa.param_amg_init(); am.smoother = T.SMOOTHER_JACOBI; am.coarsening_type = T.COARSE_CR
    cases.append((it, am, T.ERROR_AMG_COARSE_TYPE))
    for it, am, code in cases:
        assert fa.solver_dcsr_krylov_amg(ia, ja, a, f, x, it, am) == code
        assert np.all(x == 0.0)


def test_no_cpu_fallback_without_gpu(fa):
    if fa.available():
        import pytest
        pytest.skip("a GPU is present")
    ia, ja, a, f, ue = poisson7pt(5)
    x = np.zeros(len(f))
    it, am = fa.param_solver_init(), fa.param_amg_init(); am.smoother = T.SMOOTHER_JACOBI
    assert fa.solver_dcsr_krylov_amg(ia, ja, a, f, x, it, am) == T.ERROR_MISC
    assert np.all(x == 0.0)
