"""How far inside the 1e-9 tolerance of the parity tests the engine sits: largest deviation of the
engine's final records and energies from the oracle's, per model (a diagnostic, run by hand on the
GPU box:  python -m tests.margins).  Lives under tests/ because it runs the oracle."""
import numpy as np

from radiative3d_amd import Engine, Model
from radiative3d_amd.configs import CONFIGS
from oracle import oracle_ffi as O


def main():
    for name, n in (("halfspace", 50000), ("crustpinch", 20000), ("lopnor", 20000), ("sphere_deep", 3000), ("sphere", 3000)):
        m = Model(CONFIGS[name](4))
        rg, fg = Engine(m).run(n, 0, 0x5EED, trace=True)
        ro, fo = O.run(m, n, 0, 0x5EED, trace=True)
        dt = max(abs(a.time - b.time) / max(1.0, abs(b.time)) for a, b in zip(fg, fo))
        dp = max(abs(a.path - b.path) / max(1.0, abs(b.path)) for a, b in zip(fg, fo))
        da = max(abs(a.amp - b.amp) for a, b in zip(fg, fo))
        forks = sum((a.fate, a.moves, a.type, a.n_catch) != (b.fate, b.moves, b.type, b.n_catch) for a, b in zip(fg, fo))
        eg, eo = rg.energy[:, :, 3:], ro.energy[:, :, 3:]   # by wave type (an axis normal to the motion holds rounding noise)
        mask = eo > 0
        de = float(np.max(np.abs(eg[mask] - eo[mask]) / eo[mask])) if mask.any() else 0.0
        moves = max(a.moves for a in fg)
        print(f"{name:12s} n {n}: forks {forks}; max rel dev time {dt:.1e} path {dp:.1e}, abs dev amplitude {da:.1e}, "
              f"rel dev bin energy {de:.1e}; longest history {moves} moves", flush=True)


if __name__ == "__main__":
    main()
