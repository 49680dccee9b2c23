"""Per-phase wave-cycle shares from a -DR3D_PHASE_TIMING build (diagnostic only)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["R3D_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "radiative3d_amd", "lib", os.environ.get("R3D_PHASE_LIB", "variant_PHASE.so"))
from radiative3d_amd import Model, Engine, _ffi
from radiative3d_amd.configs import CONFIGS
name, deg, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
m = Model(CONFIGS[name](deg)); e = Engine(m)
L = _ffi.hip_lib(); out = (C.c_ulonglong * 8)()
e.run(n // 10); L.r3d_debug_phase_cycles(out)
e.run(n); ms = e.last_kernel_ms(); L.r3d_debug_phase_cycles(out)
tot = sum(out)
names = ["refill", "move", "collect", "light events", "parked R/T", "tallies+deaths", "cell fetch wait", "-"]
print(f"{name} deg {deg} n {n}: kernel {ms:.2f} ms")
for k, v in zip(names, out):
    if v: print(f"  {k:16s} {100.0 * v / tot:5.1f} %")
