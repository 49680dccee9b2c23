"""What the host-buffer form of the seam costs beside the resident one: r3d_run (kernel + the 8 MB result block read back
over PCIe and added on the host) against the same launch's kernel time.  python tools/host_result_rate.py [config] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radiative3d_amd import Model, Engine
from radiative3d_amd.configs import CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "crustpinch"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
m = Model(CONFIGS[name](9) + ["--device-tables"]); e = Engine(m)
res = m.new_result()
e.run(n // 10, result=res)
walls, kernels = [], []
for rep in range(5):
    t = time.perf_counter()
    e.run(n, first_id=(rep + 1) << 36, result=res)
    walls.append(1e3 * (time.perf_counter() - t)); kernels.append(e.last_kernel_ms())
walls.sort(); kernels.sort()
w, k = walls[2], kernels[2]
print(f"{name} n {n}: r3d_run wall {w:.2f} ms (kernel {k:.2f} ms + {w - k:.2f} ms result block over PCIe and host add) -> "
      f"{n / w * 1e3:.3e} histories/s PCIe-inclusive, {n / k * 1e3:.3e} resident", flush=True)
