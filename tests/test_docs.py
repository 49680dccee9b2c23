"""The documents and the committed evidence stay usable (CPU): DESIGN.md is a design document a maintainer can read in one
sitting (the experiment history lives in LOG.md), and every bench line committed under profiles/<round>/ is one JSON object."""
import glob
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_is_short_and_the_log_is_wrapped():
    design = open(os.path.join(REPO, "DESIGN.md"), encoding="utf-8").read().split("\n")
    assert len(design) <= 400, len(design)
    for name in ("DESIGN.md", "LOG.md"):
        for k, line in enumerate(open(os.path.join(REPO, name), encoding="utf-8").read().split("\n"), 1):
            assert len(line) <= 160, f"{name}:{k} has {len(line)} characters"
    text = "\n".join(design)
    for needle in ("parity unpinned", "model.cpp:602-633", "## 2. Data layout", "## 5. Multi-GPU", "## 7. How to verify",
                   "No N > 1 run exists"):
        assert needle.lower() in text.lower(), needle


def test_every_committed_bench_line_is_one_json_object():
    import bench
    files = sorted(glob.glob(os.path.join(REPO, "profiles", bench.PROFILE_ROUND, "bench_line_*.json")))
    assert len(files) >= 6, files
    for f in files:
        line = json.load(open(f))          # (the whole file: no banner before the line, nothing after it)
        assert line["metric"] == "phonon-histories/sec" and line["unit"] == "histories/s" and line["value"] > 0, f
        if line.get("cpu_baseline") is None:      # (a --timed-only line of a profiling pass: the timed region and nothing else)
            continue
        r = line["roofline"]
        assert r["frac"] == r["useful_frac"] and 0.05 < r["frac"] < 1.0, (f, r["frac"])
        assert r["valu_busy"] is None or r["frac"] < r["valu_busy"] < 1.0, f
        assert "CHAINED" in line["config"]["workload"] and "`job`" in line["config"]["workload"], f
        assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1, f
