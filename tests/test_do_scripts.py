"""The reference's own run scripts against this repository's command line and run configurations
(build container only: skipped where /root/reference is absent; nothing of the reference is copied
into the repository and nothing of it travels to the GPU box).

north_star: "the existing do-*.sh drivers ... work unchanged".  Each do-script is run AS IT LIES in
the reference tree, from a scratch directory that holds symbolic links to the scripts, `scripts/`
and `vis/`, a Makefile for which `make -q` succeeds (scripts/do-fundamentals.sh:145-152) and a
stand-in `./main` that writes its argument list to a file.  The captured tokens -- what
RunSimulation assembles, scripts/do-fundamentals.sh:387-427 -- must be (a) exactly the run
configuration radiative3d_amd/configs.py transcribes and bench.py runs, at the script's TOA degree,
and (b) accepted by host/cmdline.cpp: the model builds from them."""
import os
import shutil
import subprocess

import pytest

from radiative3d_amd import Model, configs

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "scripts")), reason="reference tree not present")

# what a script adds for its own bookkeeping and every run shares (do-fundamentals.sh:396-419)
BOOKKEEPING = ("--reports=", "--output-dir=", "--report-file=", "--mparams-outfile=", "--num-phonons=", "--dump-grid")


def run_do_script(tmp_path, script, edit=None):
    """Run reference do-script `script` in a scratch directory; returns the tokens ./main was given."""
    work = tmp_path / script.replace(".sh", "")
    work.mkdir()
    for entry in os.listdir(REF):
        if entry.startswith("do-") and entry.endswith(".sh"):
            os.symlink(os.path.join(REF, entry), work / entry)
    os.symlink(os.path.join(REF, "scripts"), work / "scripts")
    os.symlink(os.path.join(REF, "vis"), work / "vis")
    if edit:   # a user's one-line choice inside the script (the scripts say "copy and edit this file")
        text = open(os.path.join(REF, script)).read()
        assert edit[0] in text
        os.unlink(work / script)
        (work / script).write_text(text.replace(edit[0], edit[1], 1))
    argv_file = work / "argv.txt"
    main = work / "main"
    main.write_text('#!/bin/bash\nprintf \'%s\\n\' "$@" > "$R3D_ARGV_OUT"\n'
                    'echo "#  R3D_GRID:"; echo "#  END R3D_GRID"\n')
    main.chmod(0o755)
    (work / "Makefile").write_text("main:\n")
    assert subprocess.run(["make", "-q"], cwd=work).returncode == 0
    env = dict(os.environ, R3D_ARGV_OUT=str(argv_file))
    # (figure generation follows the run and needs octave: absent here, its errors are not the test's)
    subprocess.run(["bash", "./" + script, "noseis"], cwd=work, env=env, capture_output=True, text=True, timeout=120)
    assert argv_file.exists(), "the script never reached ./main"
    tokens = argv_file.read_text().split("\n")[:-1]
    shutil.rmtree(work / "data", ignore_errors=True)
    return tokens


def split(tokens):
    run = [t for t in tokens if not t.startswith(BOOKKEEPING)]
    return run, [t for t in tokens if t.startswith(BOOKKEEPING)]


def toa_degree(tokens):
    return int(next(t for t in tokens if t.startswith("--toa-degree=")).split("=")[1])


CASES = [
    ("do-halfspace.sh", None, lambda d: configs.halfspace(d)),
    ("do-crustpinch.sh", None, lambda d: configs.crustpinch(d)),
    ("do-lopnor.sh", None, lambda d: configs.lopnor(d, event="eq")),
    ("do-lopnor.sh", ("event=eq ", "event=expl "), lambda d: configs.lopnor(d)),     # BASELINE config 3
    ("do-spherical.sh", None, lambda d: configs.sphere(d)),
    ("do-crustpinch-vids.sh", None, lambda d: configs.crustpinch_vids(d)),
    ("do-toysphere-vids.sh", None, lambda d: configs.toysphere_vids(d)),
    ("do-lopnor-vids.sh", None, lambda d: configs.lopnor_vids(d)),
]


@pytest.mark.parametrize("script,edit,config", CASES, ids=[c[0] + ("+expl" if c[1] else "") for c in CASES])
def test_do_script_tokens_are_the_run_configuration(tmp_path, script, edit, config):
    tokens = run_do_script(tmp_path, script, edit)
    run, book = split(tokens)
    deg = toa_degree(run)
    assert deg == (8 if "vids" in script else 9)                  # PopDefaults: waveform 9, video 8
    # (a) the transcription bench.py and the tests run: same tokens, whatever their order
    assert sorted(run) == sorted(config(deg)), (sorted(run), sorted(config(deg)))
    # every bookkeeping option the scripts pass is there once
    assert sorted(t.split("=")[0] for t in book) == sorted(b.rstrip("=") for b in BOOKKEEPING)
    # (b) host/cmdline.cpp takes the script's command line as it is (tables at a small TOA degree:
    # the option parser and the model builder are what is under test, not 5 M take-off angles)
    small = [("--toa-degree=3" if t.startswith("--toa-degree=") else t) for t in tokens
             if not t.startswith("--output-dir=")]
    m = Model(small)
    want = Model(config(3))
    assert (m.n_cells, m.n_scatterers, m.n_seismometers, m.n_bins) == (want.n_cells, want.n_scatterers,
                                                                      want.n_seismometers, want.n_bins)


def test_the_cli_accepts_a_do_script_command_line_end_to_end(tmp_path):
    """./main itself (host/main.cpp) on the tokens of do-halfspace.sh: option parsing, model build,
    grid and scatterer dumps on stdout -- up to the point where it asks for a GPU."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(repo, "main")
    if not os.path.exists(exe):
        pytest.skip("./main not built")
    tokens = run_do_script(tmp_path, "do-halfspace.sh")
    out = tmp_path / "out"
    out.mkdir()
    args = [("--toa-degree=3" if t.startswith("--toa-degree=") else f"--output-dir={out}" if t.startswith("--output-dir=")
             else "--num-phonons=1K" if t.startswith("--num-phonons=") else t) for t in tokens]
    r = subprocess.run([exe] + args, cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert "#  R3D_GRID:" in r.stdout and "#  END R3D_GRID" in r.stdout
    if r.returncode != 0:   # no GPU in the build container: refused with the engine's message, after the model was built
        assert "no HIP device" in r.stdout + r.stderr or "no CPU path" in r.stdout + r.stderr, r.stderr[-2000:]
