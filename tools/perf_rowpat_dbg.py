"""Timing experiments on the row-pattern kernel (dev tool): gathers per row, tile schedule."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import faspsolver_amd as fa
from faspsolver_amd import _types as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = fa.lib()
ia, ja, a, f, ue = fa.poisson7pt(n)
amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667; amgp.max_levels = 2
H = fa.AMG(ia, ja, a, amgp)
print("levels", H.num_levels, flush=True)
for xp in (16, -1, 1, 4, 64, 256):
    L.fasp_hip_tune(b"xcd_pat", xp)
    for rpl in (1, 2):
        L.fasp_hip_tune(b"rpl", rpl)
        line = f"xcd_pat {xp:4d} rpl {rpl}:"
        for dbg in (0, 1, 2, 3, 5, 7):
            L.fasp_hip_tune(b"dbg", dbg)
            ms = H.time_kernel(0, 0, 20)
            line += f" dbg{dbg} {ms*1e3:7.1f}us |"
        print(line, flush=True)
L.fasp_hip_tune(b"dbg", 0)
for mg in (256, 512, 1024, 1792, 2048):
    L.fasp_hip_tune(b"maxgrid", mg)
    L.fasp_hip_tune(b"xcd_pat", -1); L.fasp_hip_tune(b"rpl", 1)
    print(f"maxgrid {mg}: mxv {H.time_kernel(0,0,20)*1e3:.1f} us jacobi {H.time_kernel(2,0,20)*1e3:.1f} us", flush=True)
H.close()
