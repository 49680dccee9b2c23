"""One history alone on the chip, timed on several builds of the engine in one call:
    python tools/lone_history_time.py lopnor 9 3142726 variant_X.so libr3d_hip.so"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from radiative3d_amd import Model, Engine
from radiative3d_amd.configs import CONFIGS
name, deg, hid = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
m = Model(CONFIGS[name](deg) + ["--device-tables"])
for lib in sys.argv[4:]:
    e = Engine(m, lib=os.path.join(REPO, 'radiative3d_amd', 'lib', lib))
    e.run(1, first_id=hid)
    ts = []
    for _ in range(5):
        r = e.run(1, first_id=hid); ts.append(e.last_kernel_ms())
    ts.sort()
    print(f"{lib}: {name} history {hid}: {r.events['iterations']} moves, {r.events['rtsolve']} solves in {ts[2]:.3f} ms = {1e3*ts[2]/r.events['iterations']:.2f} us per move", flush=True)
    e.close()
