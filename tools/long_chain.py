"""Soak of the carry chain: K launches of n histories + flush; the totals must account for every
history (generated == lost + timeout + invalid == K n).  python tools/long_chain.py crustpinch 9 10000000 100"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiative3d_amd import Model, Engine
from radiative3d_amd.parallel import DeviceResult
from radiative3d_amd.configs import CONFIGS
name, deg, n, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
m = Model(CONFIGS[name](deg) + ["--device-tables"]); e = Engine(m, lib=os.environ.get("R3D_HIP_LIB"))
total, buf = DeviceResult(m, "cuda:0"), DeviceResult(m, "cuda:0")
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(k):
    buf.zero_()
    e.run_device(n, i * n, 0x5EED, *buf.pointers(), carry="carry")
    total.add_(buf)
buf.zero_()
e.run_device(0, 0, 0x5EED, *buf.pointers(), carry="final")
total.add_(buf)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
r = total.to_result()
done = r.n_lost + r.n_timeout + r.n_invalid
print(f"{name} deg {deg}: {k} chained launches of {n}: {dt:.2f} s wall -> {k * n / dt:.3e} hist/s; generated {r.events['generated']}, "
      f"ended {done} (lost {r.n_lost} timeout {r.n_timeout} invalid {r.n_invalid}), catches {r.events['catch']} == counts {int(r.counts.sum())}")
assert r.events["generated"] == done == k * n and int(r.counts.sum()) == r.events["catch"]
