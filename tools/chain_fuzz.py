"""Fuzz of the chained-workgroup ksw kernel (ksw_chain_kernel.h) on the wavefront emulator against the oracle:
   python tools/chain_fuzz.py [seed] [n_cases] [max_len] [bp indices, e.g. 0,1] [start skew in us]
Prints the runs, the mismatches (asserts on the first) and the event counters."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kswcases
import test_kernels_emu as T

if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    max_len = int(sys.argv[3]) if len(sys.argv) > 3 else 700
    geos = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2]
    E = T._load_chain()
    if len(sys.argv) > 5:
        E.emu_chain_start_skew(int(sys.argv[5]))
    t0 = time.time()
    cases = kswcases.stripe_edge_cases(seed, n, max_len) + kswcases.stripe_cases(seed + 1, n // 4, max_len)
    n_run = T._chain_run(E, cases, [400 + g * 10 + v for g in geos for v in (0, 2, 3)])
    print("chain fuzz: seed %d, %d runs, 0 mismatches, %.0f s, events %s" % (seed, sum(n_run.values()), time.time() - t0, T._chain_events(E)), flush=True)
