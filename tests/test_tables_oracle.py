"""The TABLES the hot path reads, against oracle/r3d_tables_oracle.cpp -- the reference's table
builders restated in its own formulation: take-off set (geom_s2.cpp:60-292), GSATO / XSATO / PSATO
-> cumulative tables, mean free paths, dipole moments, conversion weights (scatparams.cpp:75-194,
scatterers.cpp:97-259), moment tensor -> P / SH / SV patterns in the local NED frame
(tensors.hpp:147-280, ecs.cpp:704-722, events.cpp:42-107), seismometer axes and areas
(dataout.cpp:42-71, ecs.cpp:147-306).

CPU: the oracle against what the survey recorded on the reference and against closed forms, then
the product's HOST builder against the oracle on the four benchmark models and the override path.
The HIP builder (--device-tables) meets the same oracle in tests/test_device_tables.py (-m gpu)."""
import json
import math
import os

import numpy as np
import pytest

from oracle import tables_ffi as T
from radiative3d_amd import Model
from tests.configs import CONFIGS, halfspace

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# what each configuration's command line says about the event (main.cpp:485-560)
SOURCES = {"halfspace": ("SDR", [0, 90, 0, 0.0, 1.0]), "crustpinch": ("SDR", [22.5, 90, 0, 0.0, 1.0]),
           "crustpinch_vids": ("SDR", [22.5, 90, 0, 0.0, 1.0]), "lopnor": ("USGS", [1, 1, 1, 0, 0, 0]),
           "sphere": ("SDR", [22.5, 90, 0, 0.0, 1.0]), "sphere_deep": ("SDR", [22.5, 90, 0, 0.0, 1.0]),
           "lopnor_vids": ("SDR", [125, 40, 90, 0.0, 1.0])}


# ------------------------------------------------------------ the oracle itself, pinned ----
def test_oracle_prints_the_reference_scatterer_row_at_toa_degree_9():
    """SURVEY.md 8(c) item 3, measured on the unmodified reference: the Halfspace scatterer at TOA
    degree 9 prints MFP 1564.43 / 594.537 (setprecision(6)) and dipoles 0.6881 / 0.8776
    (setprecision(4)), scatterers.cpp:455-476.  The oracle's numbers print to the same digits."""
    ref = json.load(open(os.path.join(GOLDEN, "reference_recorded.json")))["halfspace_scatterer_dump_toa9"]
    m = Model(halfspace(2))                      # (only for the medium's parameters: el = 2 pi f / Vs, gam0 = Vp / Vs)
    het = list(m.desc.scatterers[0].het)
    assert het[:4] == [0.8, 0.01, 1.0, 0.5]
    assert het[4] == pytest.approx(2 * math.pi * 2.0 / 3.63, rel=1e-15) and het[5] == pytest.approx(6.40 / 3.63, rel=1e-15)
    r = T.scatterer(het, T.toa(9))
    assert f"{r['mfp'][0]:.6g}" == f"{ref['mfp_p']:.6g}" == "1564.43"
    assert f"{r['mfp'][1]:.6g}" == f"{ref['mfp_s']:.6g}" == "594.537"
    assert f"{r['dipole'][0]:.4g}" == f"{ref['dipole_p']:.4g}" == "0.6881"
    assert f"{r['dipole'][1]:.4g}" == f"{ref['dipole_s']:.4g}" == "0.8776"


def test_oracle_take_off_set_is_the_quadrisected_icosahedron():
    for deg in (0, 1, 3):
        toa = T.toa(deg)
        assert toa.shape == (20 * 4 ** deg, 2)
        xyz = np.stack([np.sin(toa[:, 0]) * np.cos(toa[:, 1]), np.sin(toa[:, 0]) * np.sin(toa[:, 1]),
                        np.cos(toa[:, 0])], 1)
        assert np.abs(xyz.mean(0)).max() < 1e-12                     # a symmetric point set
        assert len({tuple(np.round(v, 9)) for v in xyz}) == len(xyz)
    # degree 0: the face centres of the icosahedron, 20 directions at one angular distance from their nearest
    toa = T.toa(0)
    xyz = np.stack([np.sin(toa[:, 0]) * np.cos(toa[:, 1]), np.sin(toa[:, 0]) * np.sin(toa[:, 1]), np.cos(toa[:, 0])], 1)
    g = xyz @ xyz.T - 2 * np.eye(20)
    assert np.allclose(g.max(1), math.sqrt(5) / 3, atol=1e-12)       # cos of the angle between adjacent face normals


def test_oracle_sato_fehler_closed_forms():
    """GSATO at psi = 0: no P->S or S->P conversion in the forward direction, and the S->S polarisation
    is atan2(2 sin zeta, -2 cos zeta) (xss_psi = -2 cos zeta, xss_zeta = 2 sin zeta there, whatever nu); the
    "< 1e-30 -> 0" clamps (scatparams.cpp:98-113)."""
    het = [0.8, 0.01, 1.0, 0.5, 3.4618, 1.7631]
    for zeta in (0.0, 0.7, 2.0, -1.3):
        gpp, gps, gsp, gss, spol = T.gsato(het, 0.0, zeta)
        assert gps == 0.0 and gsp == 0.0 and gpp > 0 and gss > 0
        assert spol == pytest.approx(math.atan2(math.sin(zeta) * 2.0, -math.cos(zeta) * 2.0), abs=1e-15)
    # exponential medium (kappa = 1/2): PSATO = 8 pi eps^2 a^3 / (1 + a^2 m^2)^2, hence at psi = pi
    # gpp = el^4 / (4 pi) * xpp^2 * PSATO(2 el / gam0), xpp = (nu (-2) - 2) / gam0^2
    nu, eps, a, kappa, el, gam0 = het
    m = 2 * el / gam0
    psdf = 8 * math.pi * eps ** 2 * a ** 3 / (1 + a * a * m * m) ** 2
    want = el ** 4 / (4 * math.pi) * ((-2 * nu - 2) / gam0 ** 2) ** 2 * psdf
    assert T.gsato(het, math.pi, 0.3)[0] == pytest.approx(want, rel=1e-12)
    tiny = [0.8, 1e-16, 1.0, 0.5, 3.4618, 1.7631]      # eps^2 = 1e-32: every weight falls below the clamp
    assert list(T.gsato(tiny, 1.0, 0.5)[:4]) == [0.0, 0.0, 0.0, 0.0]


def test_oracle_moment_tensors_and_radiation_patterns():
    toa = T.toa(5)
    # a double couple: unit Frobenius norm, trace 0; whole-space energies P : (SH + SV) = 2 : 3 (no
    # velocity weighting in the reference's patterns), to the accuracy of the equal-area sum
    mt, asym = T.moment_tensor("SDR", [22.5, 90, 0, 0.0, 1.0], 0, 6371.0, [0, 0, -10])
    assert asym < 1e-15 and mt[0] + mt[1] + mt[2] == pytest.approx(0, abs=1e-15)
    assert sum(mt[:3] ** 2) + 2 * sum(mt[3:] ** 2) == pytest.approx(1.0, rel=1e-14)
    cdf, whole = T.source(mt, toa)
    assert all(np.all(np.diff(c) >= 0) for c in cdf)
    assert whole[0] / whole[2] == pytest.approx(2 / 5, rel=2e-4)
    # strike 0, dip 90, rake 0 (Aki & Richards 4.89: M_NE = sin(dip) cos(rake) cos(2 strike) M0): the only
    # non-zero element is M_NE = M_EN = +1/sqrt(2) at unit Frobenius norm; model x = east, y = north here
    mt0, _ = T.moment_tensor("SDR", [0, 90, 0, 0.0, 1.0], 0, 6371.0, [0, 0, -5])
    assert np.allclose(mt0, [0, 0, 0, 1 / math.sqrt(2), 0, 0], atol=1e-15)
    # an explosion radiates P only, uniformly
    mte, _ = T.moment_tensor("USGS", [1, 1, 1, 0, 0, 0], 0, 6371.0, [425.54, -169.53, -1.02])
    assert np.allclose(mte, [1, 1, 1, 0, 0, 0], atol=1e-15)
    cdf, whole = T.source(mte, toa)
    assert whole[1] == whole[0] and whole[2] == whole[0]
    assert np.allclose(np.diff(cdf[0]), 1.0, rtol=1e-13)
    # in a spherical Earth (model space: x through the eastern pode, y through the north pole, z through null
    # island, ecs.cpp:100-128) the local frame below null island is the flat one's: N = +y, E = +x, D = -z ...
    mts, _ = T.moment_tensor("SDR", [22.5, 90, 0, 0.0, 1.0], 3, 6371.0, [0, 0, 6371.0 - 600])
    assert np.allclose(mts, mt, atol=1e-15)
    # ... and on the equator a quarter turn east it is N = +y, E = -z, D = -x: M_NN -> yy, M_EE -> zz, M_NE -> -yz
    mtq, _ = T.moment_tensor("SDR", [22.5, 90, 0, 0.0, 1.0], 3, 6371.0, [6371.0 - 600, 0, 0])
    assert np.allclose(mtq, [0, mt[1], mt[0], 0, 0, -mt[3]], atol=1e-15)


def test_oracle_seismometer_axes():
    r_in, r_out = [0.0, 0.0], [3.0, 2.0]
    ax, area = T.seismometer(0, 6371.0, [0, 0, -5], [100, 100, 0], 1, r_in, r_out)
    s = 1 / math.sqrt(2)
    assert np.allclose(ax, [[s, s, 0], [-s, s, 0], [0, 0, 1]], atol=1e-15)    # radial, X2 = up x radial (dataout.cpp:56), up
    assert np.allclose(area, [9 * math.pi, 4 * math.pi])
    ax, _ = T.seismometer(0, 6371.0, [0, 0, -5], [100, 100, 0], 0, r_in, r_out)
    assert np.allclose(ax, [[1, 0, 0], [0, 1, 0], [0, 0, 1]], atol=1e-15)     # east, north, up
    # straight above the event ECS.GetTransverse falls back on south (ecs.cpp:262-306): radial = up x south = east
    ax, _ = T.seismometer(0, 6371.0, [0, 0, -5], [0, 0, 0], 1, r_in, r_out)
    assert np.allclose(ax, [[1, 0, 0], [0, 1, 0], [0, 0, 1]], atol=1e-15)
    # spherical Earth: up is radial, the triple is orthonormal and right-handed like the flat one
    ax, _ = T.seismometer(3, 6371.0, [0, 0, 5771.0], [1000.0, 2000.0, 5965.0], 1, [0, 0], [400, 400])
    assert np.allclose(ax @ ax.T, np.eye(3), atol=1e-14)
    assert np.allclose(ax[2], np.array([1000.0, 2000.0, 5965.0]) / np.linalg.norm([1000.0, 2000.0, 5965.0]))
    assert np.linalg.det(ax) == pytest.approx(1.0, abs=1e-14)


# ------------------------------------------------- the product's HOST builder vs the oracle ----
def _host_tables(m):
    d = m.desc
    n = d.n_toa
    toa = np.ctypeslib.as_array(d.toa, shape=(n, 2))
    scat = []
    for s in range(d.n_scatterers):
        S = d.scatterers[s]
        scat.append(dict(het=list(S.het), mfp=list(S.mfp), whole=np.array([list(S.whole_cdf[0]), list(S.whole_cdf[1])]),
                         cdf=np.stack([np.ctypeslib.as_array(S.cdf[k], shape=(n,)) for k in range(4)]),
                         spol=np.ctypeslib.as_array(S.spol, shape=(n,)), fixed=bool(S.mfp_fixed)))
    src = dict(moment=np.array(list(d.source.moment)), loc=list(d.source.loc), whole=np.array(list(d.source.whole_cdf)),
               cdf=np.stack([np.ctypeslib.as_array(d.source.cdf[k], shape=(n,)) for k in range(3)]))
    return toa, scat, src


@pytest.mark.parametrize("name,deg", [("halfspace", 5), ("crustpinch", 4), ("lopnor", 4), ("sphere_deep", 4),
                                      ("crustpinch_vids", 4), ("lopnor_vids", 4)])
def test_host_builder_matches_the_tables_oracle(name, deg):
    """Every table of the four benchmark models (and of the --overridemfp / --nodeflect video runs)
    as radiative3d_amd/host builds it == the oracle's: the two restate the same formulas in the
    same operation order, so the agreement asked for is 1e-13 relative (observed: exact)."""
    m = Model(CONFIGS[name](deg))
    toa, scat, src = _host_tables(m)
    want_toa = T.toa(deg)
    assert toa.shape == want_toa.shape and np.max(np.abs(toa - want_toa)) <= 1e-15
    override = "--overridemfp" in " ".join(CONFIGS[name](deg))
    nodeflect = "--nodeflect" in CONFIGS[name](deg)
    for s, h in enumerate(scat):
        given = [float(x) for x in [a for a in CONFIGS[name](deg) if a.startswith("--overridemfp")][0].split("=")[1].split(",")] \
            if override else None
        o = T.scatterer(h["het"], want_toa, mfp_override=given, no_deflect=nodeflect)
        assert h["fixed"] == override
        for k in range(4):
            assert np.max(np.abs(h["cdf"][k] - o["cdf"][k])) <= 1e-13 * max(o["cdf"][k][-1], 1e-300), (name, s, k)
        assert np.max(np.abs(h["spol"] - o["spol"])) <= 1e-14
        assert np.allclose(h["whole"], o["whole"], rtol=1e-13, atol=0)
        assert np.allclose(h["mfp"], o["mfp"], rtol=1e-13)
        info = m.scatterer_info(s)
        assert [info["dipole_p"], info["dipole_s"]] == pytest.approx(list(o["dipole"]), rel=1e-12, abs=1e-14)
    # event: moment tensor in the local NED frame, radiation patterns
    code, rad_e, _ = m.coordinates
    kind, params = SOURCES[name]
    want_mt, asym = T.moment_tensor(kind, params, code, rad_e, src["loc"])
    assert asym < 1e-14 and np.allclose(src["moment"], want_mt, rtol=0, atol=1e-15)
    cdf, whole = T.source(src["moment"], want_toa)
    for k in range(3):
        assert np.max(np.abs(src["cdf"][k] - cdf[k])) <= 1e-13 * max(whole[2], 1e-300)
    assert np.allclose(src["whole"], whole, rtol=1e-13, atol=0)
    # receivers: axes and gather areas from the placed location and the radii
    d = m.desc
    for i in range(d.n_seismometers):
        S = d.seismometers[i]
        ax, area = T.seismometer(code, rad_e, src["loc"], list(S.loc), m.seismometer_axes(i), list(S.r_in), list(S.r_out))
        got = np.array([list(S.axes[0]), list(S.axes[1]), list(S.axes[2])])
        assert np.allclose(got, ax, rtol=0, atol=1e-14), (name, i)
        assert np.allclose(list(S.area), area, rtol=1e-14)


def test_committed_scatterer_rows_at_toa_degree_9():
    """tests/golden/oracle_scatterer_rows_toa9.json: the scatterer dump rows (scatterers.cpp:455-476:
    el, gam0, MFP_P, MFP_S, DM_P, DM_S) the ORACLE gives for every scatterer of the four benchmark
    configurations at TOA degree 9 (made by tests/golden/make_scatterer_rows.py), beside the one row
    the survey recorded on the reference.  The host builder's medium parameters at degree 2 (they do
    not depend on the take-off set) must be the rows' own, and its degree-9 halfspace tables their MFPs."""
    rows = json.load(open(os.path.join(GOLDEN, "oracle_scatterer_rows_toa9.json")))
    assert rows["halfspace"][0]["printed"] == ["1564.43", "594.537", "0.6881", "0.8776"]    # = the reference's row
    for name, n_scat in (("halfspace", 1), ("crustpinch", 7), ("lopnor", 21), ("sphere", 15)):
        m = Model(CONFIGS[name](2))
        assert len(rows[name]) == m.n_scatterers == n_scat
        for s, row in enumerate(rows[name]):
            assert list(m.desc.scatterers[s].het) == pytest.approx(row["het"], rel=1e-15)
    m9 = Model(CONFIGS["halfspace"](9))
    info = m9.scatterer_info(0)
    row = rows["halfspace"][0]
    assert [info["mfp_p"], info["mfp_s"]] == pytest.approx(row["mfp"], rel=1e-12)
    assert [info["dipole_p"], info["dipole_s"]] == pytest.approx(row["dipole"], rel=1e-11)


# ------------------------------------------------------------ cell arrays ----
def _cell_fields(c, kind):
    """What the traversal reads of an r3d_cell, as one flat list (faces: normal, point, radius, neighbour, flags)."""
    out = [list(c.vel_c), list(c.vel_a), [x for g in c.vel_grad for x in g], [c.rho_c, c.rho_a], list(c.rho_grad),
           list(c.q), list(c.zero_rad2)]
    faces = []
    for f in range(c.n_faces):
        F = c.faces[f]
        faces.append((list(F.normal), list(F.point), F.radius, F.neighbor, F.flags))
    return out, faces, c.scatterer, c.n_faces


@pytest.mark.parametrize("name", ["halfspace", "lopnor", "lopnor_vids", "crustpinch", "upthrust", "sphere", "toysphere_vids",
                                  "crustpinch_vids"])
def test_host_cell_builders_match_the_oracle(name):
    """The three cell-array builders (layered cylinder, warped-Cartesian-grid tetra, spherical shells) of
    radiative3d_amd/host against the oracle's, from the same grid nodes: every cell's velocity / density
    fit, attenuation Q, faces (normal, point, radius), links, flags (collect, reflect, adjoin, discontinuity)
    and scatterer index, and the scatterers' medium parameters in creation order -- for the four benchmark
    grids and the remaining user models (model.cpp:647-934, :1017-1228; media.cpp:130-156, :353-395, :578-626)."""
    args = CONFIGS[name](2)
    m = Model(args)
    dims, nodes = m.grid_nodes()
    d = m.desc
    rng = [float(a.split("=")[1]) for a in args if a.startswith("--range=")]
    freq = [float(a.split("=")[1]) for a in args if a.startswith("--frequency=")][0]
    one_dummy = any(a.startswith("--overridemfp") for a in args) and "--nodeflect" in args
    cells, n, het = T.build_cells(d.cell_kind, dims, nodes, freq, rng[0] if rng else 600.0, one_dummy)
    assert n == d.n_cells and len(het) == d.n_scatterers
    for s in range(d.n_scatterers):
        assert list(d.scatterers[s].het) == pytest.approx(list(het[s]), rel=1e-15, abs=0), (name, s)
    worst = 0.0
    for i in range(n):
        got_num, got_faces, got_scat, got_nf = _cell_fields(d.cells[i], d.cell_kind)
        want_num, want_faces, want_scat, want_nf = _cell_fields(cells[i], d.cell_kind)
        assert (got_scat, got_nf) == (want_scat, want_nf), (name, i)
        for g, w in zip(got_num, want_num):
            # the two solve the same 4 x 4 systems with different pivoting: agreement to rounding of the solve
            assert np.allclose(g, w, rtol=1e-9, atol=1e-9 * max(1e-3, float(np.max(np.abs(w))) if len(w) else 0)), (name, i, g, w)
            fin = np.isfinite(w)          # (a uniform shell has an infinite zero-velocity radius, on both sides)
            if fin.any():
                ga, wa = np.array(g)[fin], np.array(w)[fin]
                worst = max(worst, float(np.max(np.abs(ga - wa) / np.maximum(np.abs(wa), 1e-3))))
        for (gn, gp, gr, gnb, gfl), (wn, wp, wr, wnb, wfl) in zip(got_faces, want_faces):
            assert (gnb, gfl) == (wnb, wfl), (name, i)
            assert gr == wr and np.allclose(gn, wn, rtol=0, atol=1e-14) and np.allclose(gp, wp, rtol=0, atol=1e-12 * (1 + abs(wr)))
    print(f"{name}: {n} cells, {len(het)} scatterers; largest relative difference of a fitted coefficient {worst:.1e}")


# ------------------------------------------- the CLI's --event-test mission ----
@pytest.mark.parametrize("source,kind,params", [
    ("EXPL", "USGS", [1, 1, 1, 0, 0, 0]), ("EQ", "USGS", [0, -1, 1, 0, 0, 0]),
    ("SDR,22.5,90,0", "SDR", [22.5, 90, 0, 0.0, 1.0]), ("SDR,125,40,90", "SDR", [125, 40, 90, 0.0, 1.0])])
def test_event_test_mission_prints_the_oracles_radiation_patterns(tmp_path, source, kind, params):
    """`./main --event-test --source=...` (main.cpp:86-104): 3 x 1280 rows `lon lat size symbol` on the degree-3
    take-off set, size = sqrt(whole-space share x differential probability x nTOA) x 0.2
    (PhononSource::output_differential_probabilities, sources.cpp:71-87) -- against the same rows formed from
    the oracle's tessellation, moment tensor and P / SH / SV tables."""
    import subprocess
    main = os.path.join(os.path.dirname(GOLDEN.rstrip("/")), "..", "main")
    main = os.path.abspath(main)
    if not os.path.exists(main):
        subprocess.check_call(["make", "-C", os.path.dirname(main), "cli"])
    out = subprocess.run([main, "--event-test", f"--source={source}", "--source-loc=0,0,-10"], capture_output=True,
                         text=True, cwd=tmp_path, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    rows = [ln.split() for ln in out.stdout.splitlines() if len(ln.split()) == 4 and ln.split()[3] in ("c", "-", "y")]
    assert len(rows) == 3 * 1280
    toa = T.toa(3)
    mt, _ = T.moment_tensor(kind, params, 0, 6371.0, [0, 0, -10])
    cdf, whole = T.source(mt, toa)
    n = len(toa)
    at = 0
    for t, sym in enumerate(("c", "-", "y")):
        share = (whole[t] - (whole[t - 1] if t else 0.0)) / whole[2]
        mag = cdf[t][-1]
        diff = np.diff(np.concatenate([[0.0], cdf[t]])) / mag if mag else np.zeros(n)
        want = np.sqrt(share * diff * n) * 0.2
        for k in range(n):
            lon, lat, size, s_ = rows[at]
            at += 1
            assert s_ == sym
            # (printed with the stream's default 6 significant digits)
            assert float(lon) == pytest.approx(np.degrees(toa[k, 1]), rel=6e-6, abs=1e-6)
            assert float(lat) == pytest.approx(90.0 - np.degrees(toa[k, 0]), rel=6e-6, abs=1e-6)
            assert float(size) == pytest.approx(want[k], rel=6e-6, abs=1e-9), (source, sym, k)


@pytest.mark.parametrize("name", ["halfspace", "lopnor", "lopnor_vids", "crustpinch", "upthrust", "sphere", "toysphere_vids"])
def test_coordinate_conversion_of_the_grid_matches_the_oracle(name):
    """The step between the model definition's grid (anchored to the reference's own user.cpp by byte-identical
    grid dumps, tests/test_user_models_compile.py) and the nodes the cell builders read: ECS.Convert of every
    location (ortho, range-azimuth, curved, spherical; Earth-flattened depths), the flattened velocities, and
    Qp / Qs solved from the two Q values given (ecs.cpp:319-372, :540-577; grid.cpp:95-124; elastic.cpp:10-52)."""
    m = Model(CONFIGS[name](2))
    code, rad_e, flat = m.coordinates
    dims, nodes = m.grid_nodes()
    raw = m.grid_nodes_raw()
    want = T.convert_nodes(code, rad_e, flat, raw)
    assert len(want) == len(nodes) == dims[0] * dims[1] * dims[2]
    if name == "lopnor":
        assert flat and any(abs(raw[i].x[2] - nodes[i].loc[2]) > 1e-3 for i in range(len(raw)))   # depths really move
    for i in range(len(nodes)):
        g, w = nodes[i], want[i]
        assert g.n_sets == w.n_sets
        assert np.allclose(list(g.loc), list(w.loc), rtol=1e-14, atol=1e-11), (name, i)
        assert g.radius == pytest.approx(w.radius, rel=1e-15, abs=0)
        for side in range(2):
            assert np.allclose(list(g.side[side]), list(w.side[side]), rtol=1e-14, atol=0, equal_nan=True), (name, i, side)
