"""Quick GPU-vs-oracle check + timing, run on the GPU box:  python tools/gpu_check.py [model] [n]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radiative3d_amd import Model, Engine
from oracle import oracle_ffi as O
from radiative3d_amd.configs import CONFIGS

def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "halfspace"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    deg = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    big = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    t = time.time(); m = Model(CONFIGS[which](deg)); print("model build %.2fs" % (time.time() - t), m.n_cells, m.n_scatterers, m.n_seismometers, flush=True)
    t = time.time(); eng = Engine(m); print("engine create %.2fs" % (time.time() - t), flush=True)
    t = time.time(); rg, fg = eng.run(n, trace=True); tg = time.time() - t
    print("gpu traced run %.3fs kernel %.3f ms" % (tg, eng.last_kernel_ms()), flush=True)
    t = time.time(); ro, fo = O.run(m, n, trace=True); to = time.time() - t
    print("oracle %.2fs (%.0f hist/s)" % (to, n / to), flush=True)
    print("O", ro.n_lost, ro.n_timeout, ro.n_invalid, {k: round(v / n, 4) for k, v in ro.events.items()})
    print("G", rg.n_lost, rg.n_timeout, rg.n_invalid, {k: round(v / n, 4) for k, v in rg.events.items()})
    bad = 0
    for i in range(n):
        a, b = fo[i], fg[i]
        ok = (a.fate == b.fate and a.moves == b.moves and a.type == b.type and a.n_catch == b.n_catch
              and abs(a.time - b.time) <= 1e-9 * max(1, abs(a.time)) and abs(a.amp - b.amp) <= 1e-9)
        if not ok:
            bad += 1
            if bad <= 5:
                print("MISMATCH", i, (a.fate, a.moves, a.type, a.n_catch, a.time, a.amp), (b.fate, b.moves, b.type, b.n_catch, b.time, b.amp))
    print("mismatching histories:", bad, "of", n)
    de = np.abs(ro.energy - rg.energy)
    print("max |dE|", de.max(), "sumE", ro.energy.sum(), rg.energy.sum(), "counts equal:", bool((ro.counts == rg.counts).all()))
    if big:
        for rep in range(3):
            t = time.time(); r = eng.run(big, first_id=1 << 32); dt = time.time() - t
            ms = eng.last_kernel_ms()
            print("big run n=%d wall %.3fs kernel %.2f ms -> %.3e hist/s" % (big, dt, ms, big / (ms * 1e-3)), {k: round(v / big, 3) for k, v in r.events.items()}, flush=True)

main()
