"""Rewrite the two measured tables of DESIGN.md (between their BEGIN / END markers) from profiles/<round>/:
python tools/design_tables.py r06"""
import json
import os
import re
import sys

rnd = sys.argv[1]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(repo, "profiles", rnd)


def line(name):
    return json.load(open(os.path.join(prof, f"bench_line_{name}.json")))


def f(x, fmt):
    return "-" if x is None else format(x, fmt)


names = {"crustpinch": "NSCP (config 2)", "halfspace": "half-space, one receiver (1)", "lopnor": "LopNor, explosion (3)",
         "sphere": "SphereEarth, 600 km source (4)", "crustpinch_volume": "NSCP video run + 10 GB grid (5)"}
roof = ["| workload (BASELINE config) | step launch | useful frac | VALU busy | lanes active | waiting | TA busy | HBM per launch |",
        "|---|---:|---:|---:|---:|---:|---:|---:|"]
for c in names:
    b, pm = line(c), json.load(open(os.path.join(prof, f"pmc_{c}.json")))
    r, hb, ms = b["roofline"], pm["hbm_traffic_bytes_per_launch"], b["roofline"]["kernel_ms_step_avg"]
    roof.append(f"| {names[c]} | {ms:.2f} ms | {f(r.get('useful_frac'), '.3f')} | {f(r.get('valu_busy'), '.2f')} | {pm['lane_activity']:.2f} | "
                f"{pm['wave_cycles_waiting']:.2f} | {f(r.get('ta_busy'), '.2f')} | {hb / 1e9:.1f} GB ({100 * hb / (ms * 1e-3) / 8e12:.0f} % of 8 TB/s) |")


def row(b, label):
    r, sl, cpu, env = b["roofline"], b.get("single_launch") or {}, b.get("cpu_baseline") or {}, b.get("envelope") or {}
    job = b.get("job") or {}
    e = f"{env['rms_sigma']:.2f} / {env['rms_sigma_gpu_vs_gpu_same_batches']:.2f}" if env.get("rms_sigma") else "-"
    return (f"| {label} | **{b['value']:.3e}** | {r['kernel_ms_step_avg']:.2f} | {sum(r['kernel_ms_flush']):.2f} | {sl.get('kernel_ms', 0):.1f} | "
            f"{job.get('histories', 0):.0e}: {job.get('ms', 0):.1f} ms | {cpu.get('value', 0):.2e} | {e} |")


meas = ["| workload | histories/s | step ms | flush ms | lone 1e7 launch ms | the job as stated, one GPU | CPU port, 16 threads | envelope RMS: GPU-CPU / GPU-GPU |",
        "|---|---:|---:|---:|---:|---:|---:|---:|",
        row(line("crustpinch_steps20_warmup5"), "NSCP, the driver's flags (20 steps + 5)")]
for c in names:
    meas.append(row(line(c), names[c]))

path = os.path.join(repo, "DESIGN.md")
text = open(path).read()
for tag, rows in (("roofline-table", roof), ("measured-table", meas)):
    pat = re.compile(rf"(<!-- BEGIN {tag} -->\n).*?(<!-- END {tag} -->)", re.S)
    assert pat.search(text), tag
    text = pat.sub(lambda m: m.group(1) + "\n".join(rows) + "\n" + m.group(2), text)
open(path, "w").write(text)
print("\n".join(roof), "\n\n", "\n".join(meas))
