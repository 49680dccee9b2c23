"""A larger look for forked histories than the parity suite takes: the engine's per-history final records against the
oracle's on many histories of the models whose histories are dense in interface solves and bends (GPU box; the oracle
runs on the host cores, one id range per thread).  python tools/fork_hunt.py [scale]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radiative3d_amd import Engine, Model
from radiative3d_amd.configs import CONFIGS
from oracle import oracle_ffi as O

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
threads = min(16, len(os.sched_getaffinity(0)))
t0 = time.time()
for name, n in (("lopnor_moho", 60000), ("lopnor", 200000), ("sphere", 30000), ("sphere_deep", 30000), ("crustpinch", 200000),
                ("upthrust", 100000), ("scat_params_study", 200000), ("halfspace", 400000)):
    n = int(n * scale)
    m = Model(CONFIGS[name](4))
    res, mine = Engine(m).run(n, 0, 0x5EED, trace=True)
    per = -(-n // threads)

    def work(i):
        lo, hi = i * per, min(n, (i + 1) * per)
        return O.run(m, hi - lo, lo, 0x5EED, trace=True)[1] if hi > lo else []

    with ThreadPoolExecutor(threads) as pool:
        parts = list(pool.map(work, range(threads)))
    want = [f for part in parts for f in part]
    forks = [i for i, (a, b) in enumerate(zip(mine, want)) if (a.fate, a.moves, a.type, a.n_catch) != (b.fate, b.moves, b.type, b.n_catch)]
    dev = max((abs(a.time - b.time) / max(1.0, abs(b.time)) for k, (a, b) in enumerate(zip(mine, want)) if k not in set(forks)), default=0.0)
    print(f"{name:18s} {n:7d} histories, {res.events['rtsolve']:9d} interface solves, {res.events['transfer']:9d} crossings: "
          f"{len(forks)} forked {forks[:5]}; times to {dev:.1e}; {time.time() - t0:.0f} s", flush=True)
