"""Hop latency of the dataflow triangular solve (seq_split.hip.h): a banded matrix whose row p couples to the `bw` rows before it --
one row per dependency class, a chain of n dependent rows -- swept by the Gauss-Seidel kernel of a one-level hierarchy.
microseconds per sweep / n = what ONE link of the chain costs.   python tools/perf_chain.py n bw [tune=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import faspsolver_amd as fa
n, bw = int(sys.argv[1]), int(sys.argv[2])
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    fa.lib().fasp_hip_tune(k.encode(), int(v))
offs = list(range(-bw, 0)) + list(range(1, bw + 1))
A = sp.diags([-1.0 / (2 * bw)] * len(offs), offs, shape=(n, n), format="csr") + 2.0 * sp.identity(n, format="csr")
A = A.tocsr(); A.sort_indices()
p = fa.param_amg_init(); p.max_levels = 1
H = fa.AMG(A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy(), p)
H.set_rhs(np.ones(n))
us = H.time_kernel(10, 0, 5) * 1e3
print(f"n {n} bandwidth {bw}: {us:.1f} us per ascending sweep = {us / n * 1e3:.0f} ns per row", flush=True)
