"""Kernel time of BASELINE config 5 (NSCP + dense scatter-event grid) next to the plain run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radiative3d_amd import Model, Engine
from radiative3d_amd.configs import crustpinch
m = Model(crustpinch(9) + ["--device-tables"]); e = Engine(m, lib=os.environ.get("R3D_HIP_LIB"))
n = 10_000_000
e.run(n // 10)
e.run(n); t0 = e.last_kernel_ms()
e.set_volume(origin=(-1000.0, -1000.0, -250.0), cell_size=(2000.0 / 256, 2000.0 / 256, 250.0 / 64), dims=(256, 256, 64),
             n_frames=300, frame_dt=2.0)
e.run(n, first_id=n); t1 = e.last_kernel_ms()
vol = e.read_volume()
print(f"NSCP deg 9, {n} histories: kernel {t0:.2f} ms plain, {t1:.2f} ms with a 2x300x64x256x256 u32 grid "
      f"({vol.nbytes / 1e9:.1f} GB); {int(vol.sum())} events binned ({vol.sum() / n:.2f} per history)")
