"""Rewrite the two measured tables of DESIGN.md (between their BEGIN / END markers) from profiles/<round>/:
python tools/design_tables.py r03"""
import json
import os
import re
import sys

rnd = sys.argv[1]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(repo, "profiles", rnd)


def line(name):
    return json.loads(open(os.path.join(prof, f"bench_line_{name}.json")).read().strip().splitlines()[-1])


names = {"crustpinch": "crustpinch (NSCP)", "halfspace": "halfspace", "lopnor": "lopnor", "sphere": "sphere, 600 km source",
         "crustpinch_volume": "crustpinch_volume (config 5)"}
roof = ["| config | ms / launch | VALU instr / launch | `roofline.frac` (busy) | lanes active | wave-cycles waiting | SALU / VALU | HBM bytes / launch (% of 8 TB/s) |",
        "|---|---:|---:|---:|---:|---:|---:|---:|"]
for c in names:
    b, pm = line(c), json.load(open(os.path.join(prof, f"pmc_{c}.json")))
    r, hb, ms = b["roofline"], pm["hbm_traffic_bytes_per_launch"], b["roofline"]["kernel_ms_step_avg"]
    roof.append(f"| {names[c]} | {ms:.2f} | {pm['SQ_INSTS_VALU']:.2e} | {r['frac']:.2f} | {pm['lane_activity']:.2f} | "
                f"{pm['wave_cycles_waiting']:.2f} | {pm['SQ_INSTS_SALU'] / pm['SQ_INSTS_VALU']:.2f} | {hb / 1e9:.1f} GB ({100 * hb / (ms * 1e-3) / 8e12:.0f} %) |")

labels = {"crustpinch": "crustpinch, 5 steps", "halfspace": "halfspace (one receiver)", "lopnor": "lopnor (explosion)",
          "sphere": "sphere (600 km source)", "crustpinch_volume": "crustpinch_volume (config 5: video run + 10 GB grid)"}


def row(b, label, bold=False):
    r, sl, cpu, env = b["roofline"], b.get("single_launch") or {}, b.get("cpu_baseline") or {}, b.get("envelope") or {}
    job = b.get("job") or {}
    v = f"{b['value']:.2e}"
    v = f"**{v}**" if bold else v
    e = (f"{env['rms_sigma']:.2f} / {env['rms_sigma_gpu_vs_gpu_same_batches']:.2f}, {env['bins']} bins"
         if env.get("rms_sigma") else "-")
    return (f"| {label} | {v} | {r['kernel_ms_step_avg']:.2f} ms | {sum(r['kernel_ms_flush']):.2f} ms | "
            f"{sl.get('kernel_ms', 0):.1f} ms = {sl.get('value', 0):.2e}/s | {job.get('histories', 0):.0e} in {job.get('ms', 0):.1f} ms = {job.get('value', 0):.2e}/s | "
            f"{cpu.get('value', 0):.2e}/s ({cpu.get('cores')} threads) | {e} |")


meas = ["| config (`bench.py --config`) | histories/s | step launch | flush | one self-contained launch of 1e7 | the BASELINE job as stated (one GPU) | CPU port (1e7-history sample where it fits 25 s) | envelope RMS (GPU vs CPU / GPU vs GPU) |",
        "|---|---:|---:|---:|---:|---:|---:|---:|",
        row(line("crustpinch_steps20_warmup5"), "crustpinch (headline), 20 steps + 5 warm-up (the driver's command)", True)]
for c in labels:
    meas.append(row(line(c), labels[c], c == "sphere"))

path = os.path.join(repo, "DESIGN.md")
text = open(path).read()
for tag, rows in (("roofline-table", roof), ("measured-table", meas)):
    pat = re.compile(rf"(<!-- BEGIN {tag} -->\n).*?(<!-- END {tag} -->)", re.S)
    assert pat.search(text), tag
    text = pat.sub(lambda m: m.group(1) + "\n".join(rows) + "\n" + m.group(2), text)
open(path, "w").write(text)
print("\n".join(roof), "\n\n", "\n".join(meas))
