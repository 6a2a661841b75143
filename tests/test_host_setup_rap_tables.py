"""The Galerkin product of the host setup has two forms (csrc/host_setup.cpp, galerkin_rap): per-thread marker arrays, and -- for
levels of 12 M rows and more, where filling the markers cost more than the product -- small per-row hash tables.  Same discovery
order, same accumulation order: the hierarchies must agree bit for bit.  The table form is forced in a child process
(FASP_HIP_RAP_TABLE_MIN=0 is read once per process)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import faspsolver_amd as fa
ia, ja, a, f, ue = fa.poisson7pt(%d)
H = fa.AMG(ia, ja, a, fa.param_amg_init(), host_only=True)
out = {}
for l in range(H.num_levels):
    r, c, lia, lja, lv = H.matrix(l, 0)
    out[f"ia{l}"] = lia; out[f"ja{l}"] = lja; out[f"v{l}"] = lv
np.savez(sys.argv[1], nl=H.num_levels, **out)
"""


def _hierarchy(tmp_path, tag, n, env_extra):
    path = str(tmp_path / f"h_{tag}.npz")
    env = dict(os.environ); env.update(env_extra)
    subprocess.run([sys.executable, "-c", CHILD % (ROOT, n), path], check=True, env=env, stdout=subprocess.DEVNULL)
    return np.load(path)


def test_table_form_of_the_galerkin_product_is_bit_identical(tmp_path):
    n = 28
    a = _hierarchy(tmp_path, "markers", n, {"FASP_HIP_RAP_TABLE_MIN": "2000000000"})
    b = _hierarchy(tmp_path, "tables", n, {"FASP_HIP_RAP_TABLE_MIN": "0"})
    assert int(a["nl"]) == int(b["nl"]) >= 3
    for l in range(int(a["nl"])):
        assert np.array_equal(a[f"ia{l}"], b[f"ia{l}"]) and np.array_equal(a[f"ja{l}"], b[f"ja{l}"])
        assert np.array_equal(a[f"v{l}"].view(np.uint64), b[f"v{l}"].view(np.uint64)), l
