"""Regenerate tests/golden/oracle_vectors.npz -- REGRESSION vectors of this repository's
own CPU oracle (oracle/r3d_oracle.cpp), not reference output: the reference is not built
here, so these pin the oracle and the engine against drift and let
the GPU tests check the engine without running the oracle.

    python tests/golden/make_golden.py

Contents, per configuration of tests/configs.py at TOA degree 3, seed 0x5EED, ids 0..N-1:
  <name>_finals   structured array: fate, type, moves, n_catch, time, path, amp, loc[3], dir[3]
  <name>_scalars  lost, timeout, invalid, 7 invalid reasons, 8 event counters
  <name>_counts   per-seismometer catch totals by type [n_seis, 2]
  <name>_energy   per-seismometer energy totals X, Y, Z, P, S [n_seis, 5]
and the reflection / transmission probabilities of the reference's --rtcoef-test interface
(rtcoef.cpp:687-742: rho, alpha, beta = 10, 8, 4 over 8, 4, 2; 100 sines x P, SV, SH):
  rt_table        [3, 100, 7]: sine, R_P, R_SV, R_SH, T_P, T_SV, T_SH
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

N = {"halfspace": 400, "crustpinch": 300, "lopnor": 300, "sphere": 60, "toysphere_vids": 40, "lopnor_vids": 40}
FINAL = np.dtype([("fate", "u1"), ("type", "u1"), ("moves", "<u4"), ("n_catch", "<u2"), ("time", "<f8"),
                  ("path", "<f8"), ("amp", "<f8"), ("loc", "<f8", 3), ("dir", "<f8", 3)])


def finals_array(finals):
    out = np.zeros(len(finals), dtype=FINAL)
    for i, f in enumerate(finals):
        out[i] = (f.fate, f.type, f.moves, f.n_catch, f.time, f.path, f.amp, tuple(f.loc), tuple(f.dir))
    return out


def main():
    from oracle import oracle_ffi as O
    from radiative3d_amd import Model
    from tests.configs import CONFIGS
    data = {}
    for name, n in N.items():
        m = Model(CONFIGS[name](3))
        res, fin = O.run(m, n, trace=True)
        data[name + "_finals"] = finals_array(fin)
        data[name + "_scalars"] = res.scalars()
        data[name + "_counts"] = res.counts.sum(axis=1)
        data[name + "_energy"] = res.energy.sum(axis=1)
    table = np.zeros((3, 100, 7))
    for t in range(3):          # 0 P, 1 SH, 2 SV incidence -- stored in the order P, SV, SH below
        for i in range(100):
            s = i / 100.0
            table[(0, 2, 1)[t], i] = [s] + list(O.rt_probs(10, 8, 4, 8, 4, 2, s, t))
    data["rt_table"] = table
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **data)
    print("wrote", os.path.join(HERE, "oracle_vectors.npz"))


if __name__ == "__main__":
    main()
