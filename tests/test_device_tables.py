"""Tables built in HBM (--device-tables; include/r3d.h build-on-device forms, csrc/r3d_tables_build.hip)
against oracle/r3d_tables_oracle.cpp -- the reference's table builders restated (take-off set
geom_s2.cpp:60-292; GSATO -> cumulative tables, mean free paths, dipoles scatparams.cpp:75-194,
scatterers.cpp:97-259; source patterns events.cpp:42-107) -- and history parity of a run on them
against the path oracle fed the very same (downloaded) tables.  (The product's HOST builder meets
the same oracle on the CPU: tests/test_tables_oracle.py.)"""
import ctypes as C

import numpy as np
import pytest

from radiative3d_amd import Model, _ffi
from tests.configs import CONFIGS
from tests.conftest import finals_differ


def test_device_table_model_carries_parameters_not_tables():
    host = Model(CONFIGS["crustpinch"](3))
    dev = Model(CONFIGS["crustpinch"](3) + ["--device-tables"])
    assert dev.device_tables and not host.device_tables
    assert dev.n_scatterers == host.n_scatterers == 7
    for s in range(dev.n_scatterers):
        d, h = dev.desc.scatterers[s], host.desc.scatterers[s]
        assert not d.cdf[0] and not d.spol and h.cdf[0] and h.spol
        info = host.scatterer_info(s)
        assert list(d.het) == [info[k] for k in ("nu", "eps", "a", "kappa", "el", "gam0")] == list(h.het)
        assert d.psdf_numer == h.psdf_numer > 0 and d.mfp_fixed == 0
        assert np.isnan(dev.scatterer_info(s)["mfp_p"])          # unknown until an engine exists
    # the take-off set and the source's radiation patterns are left to the engine too: only the
    # tessellation degree and the moment tensor (local north-east-down frame) travel
    assert not dev.desc.toa and dev.desc.toa_degree == 3 and dev.desc.n_toa == host.desc.n_toa == 20 * 4 ** 3
    assert host.desc.toa and host.desc.toa_degree == 3
    assert not dev.desc.source.cdf[0] and host.desc.source.cdf[0]
    assert list(dev.desc.source.moment) == list(host.desc.source.moment) and any(dev.desc.source.moment)
    assert dev.desc.source.cell == host.desc.source.cell and list(dev.desc.source.loc) == list(host.desc.source.loc)


class _Patched:
    """A model description whose scatterers point at tables downloaded from the engine."""

    def __init__(self, model, engine):
        self._model = model
        n = model.n_scatterers
        self._keep = []
        self._scat = (_ffi.Scatterer * n)()
        for s in range(n):
            C.memmove(C.byref(self._scat[s]), C.byref(model.desc.scatterers[s]), C.sizeof(_ffi.Scatterer))
            cdf, spol = engine.download_scatterer(s)
            st = engine.scatterer_stats(s)
            self._keep += [cdf, spol]
            for k in range(4):
                self._scat[s].cdf[k] = cdf[k].ctypes.data_as(_ffi._dp)
            self._scat[s].spol = spol.ctypes.data_as(_ffi._dp)
            self._scat[s].mfp[0], self._scat[s].mfp[1] = st[0], st[1]
            tot = st[4:8]
            for t, w in enumerate(([tot[0], tot[1], 0, 0], [0, 0, tot[2], tot[3]])):
                acc = 0.0
                for k in range(4):
                    acc += w[k]
                    self._scat[s].whole_cdf[t][k] = acc
        self._desc = _ffi.ModelDesc()
        C.memmove(C.byref(self._desc), C.byref(model.desc), C.sizeof(_ffi.ModelDesc))
        self._desc.scatterers = C.cast(self._scat, C.POINTER(_ffi.Scatterer))
        if not model.desc.toa:                       # the take-off set the engine generated
            toa = engine.download_toa()
            self._keep.append(toa)
            self._desc.toa = toa.ctypes.data_as(_ffi._dp)
        if not model.desc.source.cdf[0]:             # ... and its source tables
            cdf, whole = engine.download_source()
            self._keep.append(cdf)
            for k in range(3):
                self._desc.source.cdf[k] = cdf[k].ctypes.data_as(_ffi._dp)
                self._desc.source.whole_cdf[k] = whole[k]
        self.desc_p = C.pointer(self._desc)

    def new_result(self):
        return self._model.new_result()


@pytest.mark.gpu
@pytest.mark.parametrize("name,deg", [("crustpinch", 5), ("lopnor", 4), ("sphere", 4), ("halfspace", 6),
                                      ("sphere_deep", 4), ("crustpinch_vids", 4)])
def test_device_tables_match_the_tables_oracle_and_the_path_oracle(name, deg):
    from oracle import oracle_ffi
    from oracle import tables_ffi as T
    from radiative3d_amd import Engine
    args = CONFIGS[name](deg)
    host = Model(args)                      # (for the statistical comparison at the end only)
    dev = Model(args + ["--device-tables"])
    e = Engine(dev)
    n_toa = dev.n_toa
    override = [a for a in args if a.startswith("--overridemfp")]
    given = [float(x) for x in override[0].split("=")[1].split(",")] if override else None
    # the take-off set generated in HBM: the oracle's recursion, up to the last place of acos / atan2
    want_toa = T.toa(deg)
    toa = e.download_toa()
    assert np.max(np.abs(toa[:, 0] - want_toa[:, 0])) < 1e-14
    assert np.max(np.abs(np.angle(np.exp(1j * (toa[:, 1] - want_toa[:, 1]))))) < 1e-14
    for s in range(dev.n_scatterers):
        o = T.scatterer(list(dev.desc.scatterers[s].het), want_toa, mfp_override=given, no_deflect="--nodeflect" in args)
        st = e.scatterer_stats(s)
        assert st[0] == pytest.approx(o["mfp"][0], rel=1e-11) and st[1] == pytest.approx(o["mfp"][1], rel=1e-11)
        if "--nodeflect" not in args:       # (under --nodeflect the reference reports 1, 1 without computing)
            assert st[2] == pytest.approx(o["dipole"][0], abs=1e-10) and st[3] == pytest.approx(o["dipole"][1], abs=1e-10)
        assert dev.scatterer_info(s)["mfp_p"] == st[0]            # the model now shows the engine's numbers
        cdf, spol = e.download_scatterer(s)
        for k in range(4):
            assert np.all(np.diff(cdf[k]) >= 0)
            assert np.max(np.abs(cdf[k] - o["cdf"][k])) <= 1e-12 * max(o["cdf"][k][-1], 1e-300)
            assert st[4 + k] == pytest.approx(o["cdf"][k][-1], rel=1e-12)
        assert np.max(np.abs(np.angle(np.exp(1j * (spol - o["spol"]))))) < 1e-9
    # the source's cumulative radiation patterns, from the moment tensor the model carries
    want_cdf, want_whole = T.source(list(dev.desc.source.moment), want_toa)
    cdf, whole = e.download_source()
    for k in range(3):
        assert np.all(np.diff(cdf[k]) >= 0)
        assert np.max(np.abs(cdf[k] - want_cdf[k])) <= 1e-12 * max(want_cdf[k][-1], want_whole[2])
        assert whole[k] == pytest.approx(want_whole[k], rel=1e-12)
    # a run on the device-built tables == the oracle on those same tables, history for history
    n = 20000 if not name.startswith("sphere") else 4000
    res_g, fin_g = e.run(n, trace=True)
    res_o, fin_o = oracle_ffi.run(_Patched(dev, e), n, trace=True)
    from oracle.check import assert_aggregates_equal_without, forked_ids
    forked = forked_ids(fin_g, fin_o, 0)
    assert len(forked) <= n * 0.0005, forked[:20]
    patched = _Patched(dev, e)
    assert_aggregates_equal_without(res_g, res_o, forked, lambda k, i: e.run(k, i),
                                    lambda k, i: oracle_ffi.run(patched, k, i), "run on device-built tables")
    assert res_g.events["scatter"] == pytest.approx(res_o.events["scatter"], rel=2e-3)
    # and statistically the same physics as the host-built model (different rounding in the
    # tables can move single draws, not the distribution)
    res_h = Engine(host).run(n)
    assert res_h.events["scatter"] == pytest.approx(res_g.events["scatter"], rel=0.05, abs=30)
    assert res_h.n_lost == pytest.approx(res_g.n_lost, rel=0.02, abs=30)


@pytest.mark.gpu
def test_device_tables_build_time_at_degree_9():
    """The point of the exercise: NSCP TOA degree 9 (5.2 M directions x 7 scatterers)."""
    import time
    from radiative3d_amd import Engine
    t0 = time.perf_counter()
    dev = Model(CONFIGS["crustpinch"](9) + ["--device-tables"])
    t1 = time.perf_counter()
    e = Engine(dev)
    t2 = time.perf_counter()
    print(f"\nhost part {t1 - t0:.2f} s, engine create incl. table build {t2 - t1:.2f} s")
    st = e.scatterer_stats(0)
    assert 4000 < st[0] < 20000 and (t2 - t0) < 1.0          # SURVEY: NSCP MFP range 4 600 - 18 800 km
    res = e.run(1_000_000)
    assert res.events["iterations"] / 1e6 == pytest.approx(27.8, rel=0.02)
