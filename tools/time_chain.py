"""Per-launch kernel time of a carry chain (r3d_run_device_carry), i.e. without the drain phase
of a self-contained launch:  [R3D_HIP_LIB=...] python tools/time_chain.py crustpinch 9 10000000 [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiative3d_amd import Model, Engine
from radiative3d_amd.parallel import DeviceResult
from radiative3d_amd.configs import CONFIGS
name, deg, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
k = int(sys.argv[4]) if len(sys.argv) > 4 else 6
m = Model(CONFIGS[name](deg) + ["--device-tables"]); e = Engine(m, lib=os.environ.get("R3D_HIP_LIB"))
buf = DeviceResult(m, "cuda:0")
ms = []
for i in range(k + 1):
    e.run_device(n, i * n, 0x5EED, *buf.pointers(), carry="carry")
    torch.cuda.synchronize()
    if i:
        ms.append(e.last_kernel_ms())
e.run_device(0, 0, 0x5EED, *buf.pointers(), carry="final"); torch.cuda.synchronize()
flush = e.last_kernel_ms()
lone = []
for i in range(3):   # self-contained launches of the same size (each drains its own stragglers)
    e.run_device(n, (1 << 40) + i * n, 0x5EED, *buf.pointers()); torch.cuda.synchronize()
    lone.append(e.last_kernel_ms())
lone.sort()
ms.sort()
print("%-28s %s deg %d n %d: chained launch %.2f ms median (min %.2f), flush %.2f ms, self-contained launch %.2f ms -> %.3e hist/s" % (
    os.path.basename(os.environ.get("R3D_HIP_LIB", "libr3d_hip.so")), name, deg, n, ms[len(ms) // 2], ms[0], flush, lone[1],
    n / ms[len(ms) // 2] * 1e3), flush=True)
