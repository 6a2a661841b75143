"""k_csr_estream's decomposition tables (csrc/kernels3.hip.h, device_csr.hip.h: build_estream_host) checked on the CPU: the host
walks them exactly as the kernel does (fasp_hip_estream_selftest) -- every entry in exactly one chunk of at most 512 entries starting at
a multiple of 8, every row finished exactly once (inside one wave, or by as many parts as its first wave's table says, the last part in
the last wave), for row shapes that stress the cuts.  The arithmetic itself is tests/test_gpu_estream.py (-m gpu)."""
import ctypes as C

import numpy as np
import pytest

import faspsolver_amd as fa


def _walk(lens, per_wave, wmax):
    ia = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    info = (C.c_int * 3)()
    L = fa.lib()
    L.fasp_hip_estream_selftest.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    st = L.fasp_hip_estream_selftest(ia.ctypes.data_as(C.POINTER(C.c_int)), len(lens), int(ia[-1]), per_wave, wmax, info)
    return st, list(info)


@pytest.mark.parametrize("per_wave,wmax", [(3072, 6144), (512, 6144), (64, 32), (100000, 64), (8, 4096)])
def test_tables_cover_every_row_once(per_wave, wmax):
    rng = np.random.default_rng(per_wave + wmax)
    shapes = {
        "long": rng.integers(50, 1200, 700),
        "one-row": np.array([200000]),
        "across-waves": np.concatenate([[0, 0, 0], rng.integers(60, 400, 400), [40000, 0, 0, 25000], rng.integers(60, 400, 300), [52000, 0, 0]]),
        "short-runs": np.concatenate([rng.integers(100, 900, 200), rng.integers(0, 3, 900), rng.integers(100, 900, 200)]),
        "aligned": np.full(640, 512),
        "all-tiny": rng.integers(0, 2, 70000),
        "eights": np.full(9000, 8),
    }
    for name, lens in shapes.items():
        if lens.sum() < 1:
            continue
        st, info = _walk(lens, per_wave, wmax)
        assert st == 0, (name, st, info)
        assert info[0] % 32 == 0 and info[0] <= max(32, wmax) and info[1] >= 1


def test_cut_rows_are_counted():
    """A 200 000-entry row between short ones at 32 wave ranges: it is cut (its parts counted), the short rows are not."""
    lens = np.concatenate([np.full(50, 100), [200000], np.full(50, 100)])
    st, info = _walk(lens, 100000, 32)
    assert st == 0 and info[0] == 32 and info[2] >= 1


# ---- k_csr_pstream (csrc/kernels4.hip.h): panel-major copy + tables, walked on the host -----------------------------------------------
def _rand_rows(lens, m, seed, band=None):
    rng = np.random.default_rng(seed)
    n = len(lens)
    ia = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cols = []
    for i, l in enumerate(lens):
        if l == 0:
            continue
        if band is None:
            c = rng.choice(m, size=l, replace=False)
        else:
            lo = int(min(max(0, i * (m / n) - band / 2), m - band))
            c = lo + rng.choice(band, size=l, replace=False)
        cols.append(c)
    ja = np.concatenate(cols).astype(np.int32)
    return ia, ja, rng.standard_normal(len(ja))


@pytest.mark.parametrize("case", ["one-panel", "many-panels", "band", "ragged", "tiny-rows", "odd-columns"])
def test_panel_tables_reproduce_the_product(case):
    from faspsolver_amd import _types as T
    rng = np.random.default_rng(len(case))
    if case == "one-panel":
        lens, m, band = rng.integers(50, 600, 300), 8000, None
    elif case == "many-panels":
        lens, m, band = rng.integers(60, 900, 400), 100000, None          # rows scattered over 13 panels
    elif case == "band":
        lens, m, band = rng.integers(50, 400, 3000), 300000, 20000         # what a coarse level looks like: 2-4 panels per row
    elif case == "ragged":
        lens = rng.integers(60, 400, 900); lens[[0, 1, 450, 898, 899]] = 0; lens[[100, 452]] = [40000, 25000]
        m, band = 60000, None
    elif case == "tiny-rows":
        lens, m, band = np.concatenate([rng.integers(100, 900, 100), rng.integers(0, 3, 2000), rng.integers(100, 900, 100)]), 30000, None
    else:
        lens, m, band = rng.integers(50, 600, 300), 8192 * 3 + 5, None     # the last panel holds five columns
    ia, ja, a = _rand_rows(lens, m, 17, band)
    n = len(lens)
    x = rng.standard_normal(m)
    A, keep = T.as_csr(ia, ja, a, ncol=m)
    y = np.zeros(n); info = (C.c_int * 4)()
    L = fa.lib()
    L.fasp_hip_pstream_selftest.argtypes = [C.POINTER(T.dCSRmat), T.c_double_p, T.c_double_p, C.POINTER(C.c_int)]
    st = L.fasp_hip_pstream_selftest(C.byref(A), T.dp(x), T.dp(y), info)
    assert st == 0, (case, st, list(info))
    import scipy.sparse as sp
    M = sp.csr_matrix((a, ja, ia), shape=(n, m))
    yref = M @ x
    rowabs = abs(M) @ np.abs(x)
    assert np.all(np.abs(y - yref) <= 1e-13 * np.maximum(rowabs, 1e-300) + 1e-300), case
    assert np.all(y[lens == 0] == 0.0)
    assert info[0] >= np.count_nonzero(lens)        # at least one sub-row per non-empty row
