"""Per-queue batch statistics of the pool kernel from a -DR3D_PHASE_TIMING build (diagnostic only):
batches served, mean lanes per batch, share of the wave cycles, cycles per batch.
  R3D_PHASE_LIB=variant_PHASE.so python tools/pool_stats.py crustpinch 9 10000000"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["R3D_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "radiative3d_amd", "lib", os.environ.get("R3D_PHASE_LIB", "variant_PHASE.so"))
import torch
from radiative3d_amd import Model, Engine, _ffi
from radiative3d_amd.parallel import DeviceResult
from radiative3d_amd.configs import CONFIGS
name, deg, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
m = Model(CONFIGS[name](deg) + ["--device-tables"]); e = Engine(m, lib=os.environ.get("R3D_HIP_LIB"))
L = _ffi.hip_lib(path=os.environ.get("R3D_HIP_LIB")); out = (C.c_ulonglong * 40)()
buf = DeviceResult(m, "cuda:0")
e.run_device(n, 0, 0x5EED, *buf.pointers(), carry="carry"); torch.cuda.synchronize(); L.r3d_debug_pool_stats(out)
e.run_device(n, n, 0x5EED, *buf.pointers(), carry="carry"); torch.cuda.synchronize(); ms = e.last_kernel_ms(); L.r3d_debug_pool_stats(out)
e.run_device(0, 0, 0x5EED, *buf.pointers(), carry="final"); torch.cuda.synchronize()
names = ["MOVE", "COLLECT", "RT", "SCATTER", "FREE/refill"]
tot = sum(out[16:21])
print(f"{name} deg {deg} n {n}: chained launch {ms:.2f} ms (instrumented build); idle polls {out[6]}")
for q, nm in enumerate(names):
    b, l, c = out[q], out[8 + q], out[16 + q]
    if q == 0 and out[7]:
        print(f"  move sub-iterations {out[7]} ({out[7] / max(1, b):.2f} per batch), lanes still live after one {out[15] / out[7]:.1f}")
    if b:
        print(f"  {nm:12s} batches {b:9d}  lanes/batch {l / b:5.1f}  cycles/batch {c / b:8.0f} (take {out[24 + q] / b:6.0f}, hand-off {out[32 + q] / b:6.0f})  share {100.0 * c / tot:5.1f} %")
