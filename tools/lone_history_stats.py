"""Where the time of ONE history goes when it is alone on the chip (what a drain's last stretch is):
per-queue batches, take / hand-off / phase cycles from a -DR3D_PHASE_TIMING build.
    make variant NAME=PHASE DEFS="-DR3D_PHASE_TIMING"
    python tools/lone_history_stats.py crustpinch 9 2937872"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from radiative3d_amd import Model, Engine, _ffi
from radiative3d_amd.configs import CONFIGS
name, deg, hid = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lib = os.path.join(REPO, "radiative3d_amd", "lib", os.environ.get("R3D_PHASE_LIB", "variant_PHASE.so"))
m = Model(CONFIGS[name](deg) + ["--device-tables"]); e = Engine(m, lib=lib)
L = _ffi.hip_lib(path=lib); out = (C.c_ulonglong * 40)()
e.run(1, first_id=hid); L.r3d_debug_pool_stats(out)
r = e.run(1, first_id=hid); ms = e.last_kernel_ms(); L.r3d_debug_pool_stats(out)
names = ["MOVE", "COLLECT", "RT", "SCATTER", "FREE/refill"]
moves = r.events["iterations"]
print(f"{name} deg {deg} history {hid}: {moves} moves, {r.events['rtsolve']} R/T solves, {r.events['collect']} collections, "
      f"{r.events['scatter']} scatterings in {ms:.3f} ms = {1e3 * ms / moves:.2f} us = {2400 * ms / moves * 1e3 / 1e3:.0f} cycles (at 2.4 GHz) per move; idle polls {out[6]}")
tot = 0
for q, nm in enumerate(names):
    b, l, c = out[q], out[8 + q], out[16 + q]
    if b:
        tot += c
        print(f"  {nm:12s} batches {b:7d}  lanes/batch {l / b:5.1f}  cycles/batch {c / b:8.0f} (take {out[24 + q] / b:6.0f}, hand-off {out[32 + q] / b:6.0f})  cycles/move {c / moves:7.0f}")
print(f"  sum of the serving waves' batch cycles per move: {tot / moves:.0f}; move sub-iterations {out[7]} ({out[7] / max(1, out[0]):.2f} per MOVE batch)")
