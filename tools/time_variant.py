"""Time one build of the engine on a config:  R3D_HIP_LIB=... python tools/time_variant.py crustpinch 9 10000000"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radiative3d_amd import Model, Engine
from radiative3d_amd.configs import CONFIGS
name, deg, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
m = Model(CONFIGS[name](deg)); e = Engine(m, lib=os.environ.get("R3D_HIP_LIB"))
e.run(n // 10)
best = 1e9
for rep in range(3):
    r = e.run(n, first_id=(rep + 1) << 33); best = min(best, e.last_kernel_ms())
print("%-40s %s deg %d n %d: kernel %.2f ms -> %.3e hist/s  iters/hist %.2f" % (os.path.basename(os.environ.get("R3D_HIP_LIB", "libr3d_hip.so")), name, deg, n, best, n / best * 1e3, r.events["iterations"] / n), flush=True)
