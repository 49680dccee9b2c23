"""Vector instructions of each phase's common path (tools/microbench/phase_floor.hip), counted in the
compiler's gfx950 assembly:  python tools/phase_floor.py [out.json]
Writes {kernel: v_* instructions} + the kernel sources' hash; bench.py turns them into
roofline.floor_lane_insts_per_history with the run's own event counts."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def count(asm_text):
    out = {}
    for name, body in re.findall(r"^(floor_\w+):(.*?)s_endpgm", asm_text, re.S | re.M):
        c = collections.Counter()
        for ln in body.split("\n"):
            m = re.match(r"^\s+([a-z_0-9]+)\s", ln)
            if not m:
                continue
            op = m.group(1)
            c["valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "mem"] += 1
        out[name] = dict(c)
    return out


def main():
    import bench
    src = os.path.join(REPO, "tools", "microbench", "phase_floor.hip")
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "floor.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-mllvm", "-disable-machine-licm",
                               "-DR3D_FLOOR_TIERS", "-I" + os.path.join(REPO, "include"), "--cuda-device-only", "-S", "-o", asm, src])
        counts = count(open(asm).read())
    rec = {"kernel_source_hash": bench.kernel_source_hash(),
           "what": "v_* instructions per lane of each phase's common path (tools/microbench/phase_floor.hip, short tiers, no rare branch, "
                   "no queue / slot / tally code), static count in the gfx950 assembly",
           "valu": {k[6:]: v["valu"] for k, v in sorted(counts.items())},
           "salu": {k[6:]: v.get("salu", 0) for k, v in sorted(counts.items())}}
    text = json.dumps(rec, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
