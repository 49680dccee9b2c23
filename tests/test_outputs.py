"""Output writers and the ./main command-line shell:
the file formats the reference's do-*.sh + Octave pipeline consumes
(dataout.cpp:222-406, :623-694; model.cpp:88-197; scatterers.cpp:420-478)."""
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import oracle_ffi as O
from radiative3d_amd import Model, _ffi
from tests.configs import halfspace, lopnor

MAIN = os.path.join(_ffi.REPO, "main")


def parse_octave_struct(text):
    """Minimal reader of the GNU-Octave text struct the reference writes."""
    out, lines, i = {}, text.splitlines(), 0
    while i < len(lines):
        m = re.match(r"# name: (\w+)", lines[i])
        if not m or m.group(1) == "SEIS":
            i += 1
            continue
        name, kind = m.group(1), lines[i + 1].split(":")[1].strip()
        i += 2
        if kind == "scalar":
            out[name] = float(lines[i].split()[0])
        elif kind == "string":
            out[name] = lines[i + 2]
        elif kind == "matrix":
            rows = int(lines[i].split(":")[1])
            cols = int(lines[i + 1].split(":")[1])
            vals = [[float(x) for x in lines[i + 2 + r].split()] for r in range(rows)]
            assert all(len(v) == cols for v in vals), name
            out[name] = np.array(vals)
        i += 1
    return out


def test_seismometer_files_carry_the_result(tmp_path):
    m = Model(halfspace(4))
    res = O.run(m, 30000)
    summary = m.write_outputs(res, str(tmp_path), mparams_path=str(tmp_path / "out_mparams.octv"))
    assert "->  Phonons lost due to:" in summary
    assert f"Loss surfaces:  {res.n_lost}" in summary and f"Timeout:        {res.n_timeout}" in summary
    assert "(Diag: 0x0)" in summary
    files = sorted(f for f in os.listdir(tmp_path) if f.startswith("seis_") and f.endswith(".octv"))
    assert files == [f"seis_{i:03d}.octv" for i in range(144)]
    s = parse_octave_struct(open(tmp_path / "seis_060.octv").read())
    assert s["NumBins"] == 400 and s["Frequency"] == 2 and s["AxesDesc"] == "RTZ"
    assert s["TimeWindow"].tolist() == [[0, 200]]
    assert s["EventLoc"].tolist() == [[0, 0, -5]]
    assert np.allclose(s["Location"], [m.desc.seismometers[60].loc[k] for k in range(3)], rtol=1e-5)
    assert s["TraceXYZ"].shape == (400, 3) and s["TracePS"].shape == (400, 2) and s["CountPS"].shape == (400, 2)
    assert np.allclose(s["TraceXYZ"], res.energy[60, :, :3], rtol=1e-5, atol=0)   # 6 significant digits
    assert np.allclose(s["TracePS"], res.energy[60, :, 3:], rtol=1e-5, atol=0)
    assert (s["CountPS"] == res.counts[60]).all()
    assert s["GatherRadius"][0, 1] == pytest.approx(m.desc.seismometers[60].r_out[0], rel=1e-5)
    # ASCII trace file: one block per seismometer, n_bins rows of 7 columns
    asc = open(tmp_path / "seis_traces_asc.dat").read()
    assert asc.count("#### BEGIN TRACE ####") == 144
    assert "SEIS: ------------|   Seismometer Number: 143   |------------" in asc
    block = asc.split("#### BEGIN TRACE ####\n")[61].split("SEIS: #### END TRACE")[0].splitlines()
    assert len(block) == 400 and all(len(r.split()) == 7 for r in block)
    assert sum(int(r.split()[5]) for r in block) == int(res.counts[60, :, 0].sum())
    # parameter file
    p = parse_octave_struct(open(tmp_path / "out_mparams.octv").read())
    assert p["TOA_Degree"] == 4 and p["PhononTTL"] == 200 and p["CylinderRange"] == 900
    assert p["CompiledArgs"].shape == (1, 18)


def test_scatterer_dump_and_params_echo():
    m = Model(lopnor(3))
    dump = m.scatterer_dump().splitlines()
    assert dump[0] == "#  BEGIN SCATTERER DUMP:" and dump[1] == "#  Overrides: None" and dump[-1] == "#  END SCATTERERS"
    rows = [l for l in dump if not l.startswith("#")]
    assert len(rows) == 21
    first = rows[0].split()
    assert [float(first[0]), float(first[1]), float(first[2]), float(first[3])] == [0.8, 0.01, 0.5, 0.2]
    assert float(first[7]) == pytest.approx(m.scatterer_info(0)["mfp_p"], rel=1e-5)
    echo = m.params_echo()
    assert "TOA_Degree: 3" in echo and "320 Seismometers Requested." in echo and "Event Moment Tensor:" in echo


@pytest.mark.skipif(not os.path.exists(MAIN), reason="./main not built")
def test_main_cli_missions_without_gpu(tmp_path):
    out = subprocess.run([MAIN, "--rtcoef-test"], capture_output=True, text=True, cwd=tmp_path)
    assert out.returncode == 0
    rows = [l.split() for l in out.stdout.splitlines() if re.match(r"^\s+[0-9]", l)]
    assert len(rows) == 300                      # 3 incident types x 100 sines (rtcoef.cpp:687-742)
    # first P row: normal incidence on (10,8,4)/(8,4,2): R = ((32-80)/(32+80))^2 of the flux rho1*alpha1
    r = ((8 * 4 - 10 * 8) / (8 * 4 + 10 * 8)) ** 2
    assert float(rows[0][0]) == 0 and float(rows[0][1]) == pytest.approx(80 * r, rel=1e-5)
    assert float(rows[0][2]) == pytest.approx(80 * (1 - r), rel=1e-5)
    bad = subprocess.run([MAIN, "--no-such-flag"], capture_output=True, text=True, cwd=tmp_path)
    assert bad.returncode == 1 and "Unrecognized option" in bad.stdout
    grid = subprocess.run([MAIN, "--grid-compiled=40", "--toa-degree=2", "--dump-grid", "--rtcoef-test"],
                          capture_output=True, text=True, cwd=tmp_path)
    assert "#  R3D_GRID:" in grid.stdout and "#  END R3D_GRID" in grid.stdout
    assert "#  BEGIN SCATTERER DUMP:" in grid.stdout


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(MAIN), reason="./main not built")
def test_main_cli_end_to_end_on_gpu(tmp_path):
    # (--host-tables: the comparison below is history by history against the oracle on host-built
    #  tables; by default a ./main run builds its tables in HBM, equal to 1e-12, not to the bit)
    args = [a for a in halfspace(4)] + ["--num-phonons=200K", "--seed=7", f"--output-dir={tmp_path}", "--host-tables",
                                        "--mparams-outfile=out_mparams.octv", "--reports=INV", "--dump-grid"]
    run = subprocess.run([MAIN] + args, capture_output=True, text=True, cwd=tmp_path)
    assert run.returncode == 0, run.stdout[-2000:]
    for marker in ("@@ __BEGIN_MODEL_INITIALIZATION__", "@@ __BEGINNING_SIMULATION__",
                   "@@ __SIMULATION_COMPLETE__", "Printing Post-Sim Summary:"):
        assert marker in run.stdout
    lost = int(re.search(r"Loss surfaces:\s+(\d+)", run.stdout).group(1))
    tmo = int(re.search(r"Timeout:\s+(\d+)", run.stdout).group(1))
    assert lost + tmo == 200000
    assert "|  Shards: 1 (summed by host: one shard" in run.stdout     # (a node of one shard has nothing to reduce)
    assert "RCCL version" not in run.stdout                   # libraries' banners do not land in the reference's format
    m = Model(halfspace(4))
    want = O.run(m, 200000, seed=7)
    assert (lost, tmo) == (want.n_lost, want.n_timeout)
    s = parse_octave_struct(open(tmp_path / "seis_100.octv").read())
    assert (s["CountPS"] == want.counts[100]).all()
    assert np.allclose(s["TracePS"], want.energy[100, :, 3:], rtol=1e-5)
    assert os.path.exists(tmp_path / "seis_traces_asc.dat") and os.path.exists(tmp_path / "out_mparams.octv")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(MAIN), reason="./main not built")
def test_main_cli_report_file_on_gpu(tmp_path):
    """--reports / --report-file (reference main.cpp:223-263): the file holds one line per
    requested event, in the reference's line format, grouped by history."""
    args = [a for a in halfspace(4)] + ["--num-phonons=500", "--seed=11", f"--output-dir={tmp_path}", "--host-tables",
                                        "--reports=GEN,LST,TMO", "--report-file=reports.dat"]
    run = subprocess.run([MAIN] + args, capture_output=True, text=True, cwd=tmp_path)
    assert run.returncode == 0, run.stdout[-2000:]
    lines = open(tmp_path / "reports.dat").read().splitlines()
    tags = [l[:3] for l in lines]
    assert tags.count("GEN") == 500 and tags.count("LST") + tags.count("TMO") == 500 and len(lines) == 1000
    assert tags[0::2] == ["GEN"] * 500                       # GEN then its history's end, id by id
    ids = [int(l[4:11]) for l in lines]
    assert ids == sorted(ids) and ids[0] == 0 and ids[-1] == 499
    m = Model(halfspace(4))
    _, ev, _ = O.run_with_events(m, 500, mask=1 | 32 | 64, seed=11)
    want = m.format_reports(ev).splitlines()
    assert len(want) == len(lines)
    num = re.compile(r"-?\d+\.?\d*(?:e[-+]?\d+)?")
    for a, b in zip(want, lines):               # same text up to rounding of the printed numbers
        assert num.sub("#", a).split() == num.sub("#", b).split(), (a, b)
        va, vb = [float(x) for x in num.findall(a)], [float(x) for x in num.findall(b)]
        assert np.allclose(va, vb, rtol=1e-5, atol=1e-9), (a, b)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(MAIN), reason="./main not built")
def test_main_cli_builds_its_tables_on_the_device_by_default(tmp_path):
    """A plain ./main run (no --host-tables): take-off set, source and scattering tables made in HBM;
    same physics as the host-table run (statistically: the tables agree to 1e-12, not to the bit)."""
    base = [a for a in halfspace(5)] + ["--num-phonons=400K", "--seed=5", f"--output-dir={tmp_path}"]
    runs = {}
    for tag, extra in (("device", []), ("host", ["--host-tables"])):
        run = subprocess.run([MAIN] + base + extra, capture_output=True, text=True, cwd=tmp_path)
        assert run.returncode == 0, run.stdout[-2000:]
        runs[tag] = (int(re.search(r"Loss surfaces:\s+(\d+)", run.stdout).group(1)),
                     int(re.search(r"Timeout:\s+(\d+)", run.stdout).group(1)), run.stdout)
    assert sum(runs["device"][:2]) == sum(runs["host"][:2]) == 400000
    assert runs["device"][0] == pytest.approx(runs["host"][0], rel=0.01)
    assert "SCATTERER" in runs["device"][2].upper()          # (the dump shows the engine's mean free paths)


@pytest.mark.gpu
def test_plain_c_example_runs_on_gpu(tmp_path):
    """examples/run_model.c: model from option tokens, r3d_run_model, the reference's files."""
    exe = str(tmp_path / "run_model")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(repo, "radiative3d_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(repo, "include"),
                           os.path.join(repo, "examples", "run_model.c"), "-L", lib, "-lr3d_host", "-lr3d_hip",
                           "-L/opt/rocm/lib", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe, "1", str(tmp_path)] + halfspace(4) + ["--num-phonons=50K", "--seed=3"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    m = Model(halfspace(4))
    want = O.run(m, 50000, seed=3)
    assert f"lost {want.n_lost} timeout {want.n_timeout} invalid {want.n_invalid}" in out.stdout
    assert os.path.exists(tmp_path / "seis_000.octv") and os.path.exists(tmp_path / "seis_traces_asc.dat")


@pytest.mark.skipif(not os.path.exists(MAIN), reason="./main not built")
def test_main_cli_scatter_grid_options_are_parsed(tmp_path):
    """--scatter-grid / --devices: malformed values are refused by the option parser (no GPU needed)."""
    for bad in ("--scatter-grid=8,8,8", "--scatter-grid=8,8,0,10,0,0,0,1,1,1", "--scatter-grid=8,8,8,10,0,0,0,1,1,0",
                "--scatter-grid=8,8,8,0,0,0,0,1,1,1", "--devices="):
        out = subprocess.run([MAIN, "--grid-compiled=40", "--toa-degree=2", bad], capture_output=True, text=True, cwd=tmp_path)
        assert out.returncode == 1 and "Error processing command-line options" in out.stdout, (bad, out.stdout[-300:])


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(MAIN), reason="./main not built")
def test_main_cli_writes_the_scatter_grid_of_a_video_run_on_gpu(tmp_path):
    """./main --scatter-grid on the crust-pinch video run (do-crustpinch-vids.sh arguments; the option would
    travel in the script's ADDITIONAL=): three shards on the box's one GPU (--devices=0,0,0), grids added by
    frame, raw file + Octave header -- equal to the oracle's histogram of the whole job, bit for bit.
    Reference semantics: vis/scattervid/preprocess.sh:17-29 + scattervid_above.m:111 (frame = floor(t / dt))."""
    from radiative3d_amd.configs import crustpinch_vids
    from radiative3d_amd.model import volume_desc
    n = 30000
    dims, frames, lo, hi = (64, 60, 14), 35, (-200.0, -600.0, -130.0), (1080.0, 600.0, 10.0)
    args = crustpinch_vids(4) + [f"--num-phonons={n}", "--seed=13", f"--output-dir={tmp_path}", "--host-tables", "--devices=0,0,0",
                                 "--scatter-grid=" + ",".join(str(v) for v in dims + (frames,) + lo + hi),
                                 "--scatter-grid-file=grid"]
    run = subprocess.run([MAIN] + args, capture_output=True, text=True, cwd=tmp_path)
    assert run.returncode == 0, run.stdout[-2000:]
    hdr = parse_octave_struct(open(tmp_path / "grid.octv").read())
    assert list(hdr["GridDims"].reshape(-1)) == list(dims) and int(hdr["GridFrames"]) == frames
    assert float(hdr["GridFrameSeconds"]) == pytest.approx(350.0 / frames)
    got = np.fromfile(tmp_path / "grid.u32", dtype="<u4").reshape(2, frames, dims[2], dims[1], dims[0])
    m = Model(crustpinch_vids(4))
    cell = tuple((h - l) / d for l, h, d in zip(lo, hi, dims))
    res, want = O.run_with_volume(m, n, volume_desc(lo, cell, dims, frames, 350.0 / frames), seed=13)
    assert (got == want).all() and int(hdr["GridEventsBinned"]) == int(want.sum()) > 100000
    assert int(hdr["GridSaturatedCells"]) == 0
    assert f"{int(want.sum())} events binned" in run.stdout
    assert "|  Shards: 3 (summed by host: shards share a device)" in run.stdout     # (three shards on ONE device: RCCL refuses that communicator)
    lost = int(re.search(r"Loss surfaces:\s+(\d+)", run.stdout).group(1))
    assert lost == res.n_lost
    assert not os.path.exists(tmp_path / "grid.u32.part")
    # a grid that the shards could not add up is refused BEFORE the run, and nothing is left behind
    big = subprocess.run([MAIN] + crustpinch_vids(4) + ["--num-phonons=1000", f"--output-dir={tmp_path}", "--host-tables", "--devices=0,0",
                          "--scatter-grid=1024,1024,64,64," + ",".join(str(v) for v in lo + hi), "--scatter-grid-file=big"],
                         capture_output=True, text=True, cwd=tmp_path)
    assert big.returncode == 1 and "do not fit the 32-bit cell" in big.stdout and "__BEGINNING_SIMULATION__" not in big.stdout
    assert not os.path.exists(tmp_path / "big.u32") and not os.path.exists(tmp_path / "big.u32.part")
