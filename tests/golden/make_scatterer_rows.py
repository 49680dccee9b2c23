"""Writes tests/golden/oracle_scatterer_rows_toa9.json: for every scatterer of the four benchmark
configurations, the row of the reference's scatterer dump (scatterers.cpp:455-476) as
oracle/r3d_tables_oracle.cpp computes it at TOA degree 9 -- medium parameters, mean free paths,
dipole moments, and the four numbers as the reference's stream formatting prints them
(setprecision(6) / setprecision(4)).  The medium parameters (nu, eps, a, kappa, el, gam0) come from
the host builder's grid; everything else is the oracle's.  ~3 minutes (44 scatterers x 5.2 M angles).

    python tests/golden/make_scatterer_rows.py
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import tables_ffi as T          # noqa: E402
from radiative3d_amd import Model          # noqa: E402
from radiative3d_amd.configs import CONFIGS  # noqa: E402

toa = T.toa(9)
out = {"_provenance": "oracle/r3d_tables_oracle.cpp (CPU restatement of scatterers.cpp:97-259 + scatparams.cpp:75-194) "
                      "at TOA degree 9; NOT reference output.  The halfspace row equals the row the survey recorded on "
                      "the unmodified reference (reference_recorded.json) in every printed digit."}
for name in ("halfspace", "crustpinch", "lopnor", "sphere"):
    m = Model(CONFIGS[name](2))
    rows = []
    for s in range(m.n_scatterers):
        het = list(m.desc.scatterers[s].het)
        r = T.scatterer(het, toa)
        rows.append({"het": het, "mfp": list(r["mfp"]), "dipole": list(r["dipole"]),
                     "printed": [f"{r['mfp'][0]:.6g}", f"{r['mfp'][1]:.6g}", f"{r['dipole'][0]:.4g}", f"{r['dipole'][1]:.4g}"]})
        print(name, s, rows[-1]["printed"], flush=True)
    out[name] = rows
json.dump(out, open(os.path.join(REPO, "tests", "golden", "oracle_scatterer_rows_toa9.json"), "w"), indent=1)
