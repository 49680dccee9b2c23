"""Long-run stability check on the GPU: python tools/long_run.py <config> <deg> <n_total> [chunk]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radiative3d_amd import Model, Engine
from radiative3d_amd.configs import CONFIGS
name, deg, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 10_000_000
m = Model(CONFIGS[name](deg)); e = Engine(m, lib=os.environ.get("R3D_HIP_LIB")); res = m.new_result()
t0 = time.time(); done = 0; kms = 0.0
while done < n:
    c = min(chunk, n - done)
    e.run(c, first_id=done, result=res); kms += e.last_kernel_ms(); done += c
    print(f"  {done} histories, kernel {kms/1e3:.2f} s so far", flush=True)
dt = time.time() - t0
ev = {k: round(v / n, 4) for k, v in res.events.items()}
print(f"{name} deg {deg}: {n} histories, wall {dt:.2f} s, kernel {kms/1e3:.2f} s -> {n/(kms*1e-3):.3e} hist/s (kernel)")
print("  lost/timeout/invalid", res.n_lost, res.n_timeout, res.n_invalid, "diag", hex(res.diag_invalid))
print("  events/history", ev)
print("  counts total", int(res.counts.sum()), "energy total %.6e" % res.energy[:, :, 3:].sum())
