"""Sweep kernel-selection / launch-geometry knobs for the CSR kernels (dev tool).
usage: sweep_spmv.py n levels(comma) [ops(comma of time_kernel kinds)]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faspsolver_amd as fa
from faspsolver_amd import _types as T

GRIDS = (-1,)
NAMES = {0: "mxv", 1: "aAxpy-1", 2: "jacobi", 5: "mxv+dot", 6: "R mxv", 7: "P aAxpy"}

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    levels = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1]
    ops = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 2]
    xcd = int(os.environ.get("SWEEP_XCD", "0"))
    ia, ja, a, f, ue = fa.poisson7pt(n)
    amgp = fa.param_amg_init(); amgp.smoother = T.SMOOTHER_JACOBI; amgp.relaxation = 0.6667
    H = fa.AMG(ia, ja, a, amgp)
    L = fa.lib()
    def tune(**kw):
        for k in ("maxgrid", "xcd", "nt", "kind", "lanes", "wrows", "wcap"):
            L.fasp_hip_tune(k.encode(), kw.get(k, {"xcd": 1, "nt": 1}.get(k, -1)))
    for l in levels:
        if l >= H.num_levels: continue
        r, c, _, _, v = H.matrix(l, 0)
        avg = len(v) / r
        print(f"=== level {l}: rows {r} nnz {len(v)} avg {avg:.1f}   (us per launch; columns = maxgrid {GRIDS}, xcd={xcd})", flush=True)
        fams = [("vec L=4", dict(kind=0, lanes=4)), ("vec L=8", dict(kind=0, lanes=8)), ("vec L=16", dict(kind=0, lanes=16)),
                ("vec L=32", dict(kind=0, lanes=32)), ("vec L=64", dict(kind=0, lanes=64))]
        if avg <= 80:
            fams += [("ws 64/512", dict(kind=2, wrows=64, wcap=512)), ("ws 64/1024", dict(kind=2, wrows=64, wcap=1024)),
                     ("ws 32/512", dict(kind=2, wrows=32, wcap=512)), ("ws 32/1024", dict(kind=2, wrows=32, wcap=1024)),
                     ("ws 16/1024", dict(kind=2, wrows=16, wcap=1024))]
        if avg > 300:
            fams += [("blockrow", dict(kind=3))]
        if avg <= 80:
            fams = [(f"ws 64/512 G={g}", dict(kind=2, wrows=64, wcap=512, xcdg=g)) for g in (0, 2, 4, 8, 16, 32, 64, 128)] + [("vec L=4", dict(kind=0, lanes=4)), ("vec L=8", dict(kind=0, lanes=8))]
        for op in ops:
            if op in (6, 7) and l == H.num_levels - 1: continue
            for name, cfg in fams:
                if op in (6, 7) and cfg.get("lanes", 0) > 16: continue
                if op not in (6, 7):
                    if (cfg.get("lanes") or 0) > 16 * max(avg, 1): continue
                row = []
                for g in GRIDS:
                    cfg2 = dict(cfg); xg = cfg2.pop("xcdg", xcd)
                    tune(maxgrid=g, xcd=xg, nt=int(os.environ.get("SWEEP_NT", "1")), **cfg2)
                    row.append(H.time_kernel(op, l, 5) * 1e3)
                print(f"  {NAMES[op]:8s} {name:11s} " + " ".join(f"{x:8.1f}" for x in row), flush=True)
    tune()
    H.close()

if __name__ == "__main__":
    main()
