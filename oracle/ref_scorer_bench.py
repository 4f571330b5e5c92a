"""TEST INFRASTRUCTURE (bench.py's cpu_baseline leg): times the reference's own Poisson scorer
(mutationRulesPoissonQualityScore, VC:3834-3884, as compiled into oracle/_ref/libvc_scorer_ref.so by oracle/Makefile) on a
stream of (k, RD, err) triples -- one core, then one process per core.  Runs as its own process so that the worker pool
never forks from a process that holds the GPU.

usage: python ref_scorer_bench.py <stream.npz> [processes]   ->   one JSON line
"""
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_ref", "libvc_scorer_ref.so")


def score(args):
    k, rd, e = args
    L = C.CDLL(LIB)
    q = np.empty(k.size, np.float64)
    L.ref_score_batch(k.ctypes.data_as(C.c_void_p), rd.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p), C.c_long(k.size),
                      q.ctypes.data_as(C.c_void_p))
    return float(np.nansum(q))


def main():
    z = np.load(sys.argv[1])
    k, rd, e = (np.ascontiguousarray(z[n]) for n in ("k", "rd", "err"))
    n = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, min(os.cpu_count() or 1, 16))
    score((k[:1000], rd[:1000], e[:1000]))
    t0 = time.perf_counter()
    score((k, rd, e))
    t_one = time.perf_counter() - t0
    parts = [(k[i::n].copy(), rd[i::n].copy(), e[i::n].copy()) for i in range(n)]
    with mp.get_context("fork").Pool(n) as pool:
        pool.map(score, [(k[:1000], rd[:1000], e[:1000])] * n)  # start the workers
        t0 = time.perf_counter()
        pool.map(score, parts, chunksize=1)
        t_all = time.perf_counter() - t0
    print(json.dumps(dict(evaluations=int(k.size), seconds_1_core=t_one, processes=n, seconds_n_processes=t_all)))


if __name__ == "__main__":
    main()
