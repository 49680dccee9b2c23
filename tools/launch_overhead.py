"""Kernel time against histories per launch: T(n) = a + b n exposes the per-launch fixed cost
(table staging, drain of the last histories, end-of-kernel flushes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radiative3d_amd import Model, Engine
from radiative3d_amd.configs import CONFIGS
name, deg = sys.argv[1], int(sys.argv[2])
m = Model(CONFIGS[name](deg) + ["--device-tables"]); e = Engine(m, lib=os.environ.get("R3D_HIP_LIB"))
e.run(1_000_000)
rows = []
for n in (100_000, 1_000_000, 2_000_000, 5_000_000, 10_000_000, 20_000_000, 50_000_000, 100_000_000):
    best = 1e9
    for rep in range(2):
        e.run(n, first_id=(rep + 1) << 36); best = min(best, e.last_kernel_ms())
    rows.append((n, best))
    print(f"{name} n {n:>10}: {best:8.2f} ms   {best / n * 1e7:7.2f} ms per 1e7", flush=True)
(n0, t0), (n1, t1) = rows[-3], rows[-1]
b = (t1 - t0) / (n1 - n0); a = t1 - b * n1
print(f"fit on the two largest: fixed {a:.2f} ms + {b * 1e7:.2f} ms per 1e7")
