"""Evidence that checks itself (CPU): every traversal kernel compiled into libr3d_hip.so is named by a
GPU parity case that holds it against the oracle, and the committed hardware-counter files that
bench.py's `roofline` reads were recorded on the kernel sources as they stand."""
import glob
import json
import os
import re
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
sys.path.insert(0, REPO)


def _kernel_cases():
    """KERNEL_CASES of tests/test_gpu_parity.py, read from its source (importing the module needs torch + a GPU-less
    pass through its fixtures, which is fine, but the list is all that is wanted here)."""
    text = open(os.path.join(REPO, "tests", "test_gpu_parity.py")).read()
    block = re.search(r"KERNEL_CASES = \[(.*?)\]\n", text, re.S).group(1)
    return [(int(k), int(r), name) for k, r, name in re.findall(r"\((\d), (\d), \"(\w+)\", \d+\)", block)]


@pytest.mark.parametrize("lib", ["libr3d_hip.so", "libr3d_hip_repro.so"])
def test_every_compiled_traversal_kernel_has_a_parity_case(lib):
    import kernel_resources as K
    compiled = K.traversal_variants(os.path.join(REPO, "radiative3d_amd", "lib", lib))
    assert compiled, "no traversal kernels found in " + lib
    cases = {(k, r) for k, r, _ in _kernel_cases()}
    # each GPU case runs the diagnostic kernel (r3d_run_traced), the self-contained production kernel (r3d_run:
    # pool_job_kernel), the chain-step kernel (r3d_run_device_carry) and the drain kernel (the chain's flush) of its variant
    covered = {(k, r, role) for (k, r) in cases for role in ("trace", "job", "production", "drain")}
    assert compiled == covered, (sorted(compiled - covered), sorted(covered - compiled))


def test_kernel_cases_use_models_of_the_right_kind():
    from radiative3d_amd import Model
    from radiative3d_amd.configs import CONFIGS
    for kind, res, name in _kernel_cases():
        assert Model(CONFIGS[name](2)).desc.cell_kind == kind, name


def test_committed_counter_files_belong_to_these_kernel_sources():
    """A kernel edit without a re-collection would silently turn roofline.frac into null (= unmeasured)
    in the driver's bench line: fail here instead."""
    import bench
    files = sorted(glob.glob(os.path.join(REPO, "profiles", bench.PROFILE_ROUND, "pmc_*.json")))
    assert files, f"no counter files under profiles/{bench.PROFILE_ROUND}"
    names = {os.path.basename(f)[4:-5] for f in files}
    assert set(bench.workloads()) <= names, sorted(set(bench.workloads()) - names)
    want = bench.kernel_source_hash()
    for f in files:
        rec = json.load(open(f))
        assert rec["kernel_source_hash"] == want, (
            f"{os.path.relpath(f, REPO)} was recorded on other kernel sources ({rec['kernel_source_hash']} != {want}): "
            "re-run tools/collect_profiles.sh on the GPU box and commit the new summaries")
        assert rec["SQ_ACTIVE_INST_VALU"] and rec["histories_per_launch"] == bench.workloads()[rec["config"]]["histories"]
