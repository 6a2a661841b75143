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

