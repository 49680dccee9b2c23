"""Per-queue statistics of a chain's FLUSH launch alone (-DR3D_PHASE_TIMING build):
    make variant NAME=PHASE DEFS="-DR3D_PHASE_TIMING";  python tools/flush_stats.py lopnor 9 10000000"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from radiative3d_amd import Model, Engine, _ffi
from radiative3d_amd.parallel import DeviceResult
from radiative3d_amd.configs import CONFIGS
name, deg, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lib = os.path.join(REPO, "radiative3d_amd", "lib", os.environ.get("R3D_PHASE_LIB", "variant_PHASE.so"))
m = Model(CONFIGS[name](deg) + ["--device-tables"]); e = Engine(m, lib=lib)
L = _ffi.hip_lib(path=lib); out = (C.c_ulonglong * 40)()
buf = DeviceResult(m, "cuda:0")
for i in range(3):
    e.run_device(n, i * n, 0x5EED, *buf.pointers(), carry="carry")
torch.cuda.synchronize(); L.r3d_debug_pool_stats(out)
e.run_device(0, 0, 0x5EED, *buf.pointers(), carry="final"); torch.cuda.synchronize(); ms = e.last_kernel_ms(); L.r3d_debug_pool_stats(out)
names = ["MOVE", "COLLECT", "RT", "SCATTER", "FREE/refill"]
print(f"{name} deg {deg}: flush launch after 3 x {n}: {ms:.2f} ms = {ms * 2.4e6:.0f} cycles at 2.4 GHz; idle polls {out[6]}")
for q, nm in enumerate(names):
    b, l, c = out[q], out[8 + q], out[16 + q]
    if b:
        print(f"  {nm:12s} batches {b:9d}  lanes/batch {l / b:5.1f}  cycles/batch {c / b:8.0f} (take {out[24 + q] / b:6.0f}, hand-off {out[32 + q] / b:6.0f})  "
              f"wave-cycles {c:.3e} = {c / 256 / (ms * 2.4e6):.2f} waves busy per workgroup on average")
print(f"  move sub-iterations {out[7]} ({out[7] / max(1, out[0]):.2f} per MOVE batch), lanes live in them {out[15] / max(1, out[7]):.1f}")
