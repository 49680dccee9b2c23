"""Registers, spills, scratch and LDS of every kernel in libr3d_hip.so, from the code object's
metadata notes:  python tools/kernel_resources.py [lib.so] [name filter]"""
import os, re, subprocess, sys, tempfile
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                          "radiative3d_amd", "lib", "libr3d_hip.so")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = tempfile.mkdtemp()
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + lib,
                       "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + tmp + "/dev.co"],
                      stderr=subprocess.DEVNULL) if False else None
# the fat binary sits in .hip_fatbin: pull it out, then unbundle
subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, tmp + "/fat.bin"])
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + tmp + "/fat.bin",
                       "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + tmp + "/dev.co"])
notes = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", tmp + "/dev.co"], text=True)
cur = {}
rows = []
for line in notes.split("\n"):
    m = re.match(r"\s+- \.(agpr_count|args):", line)
    if m and m.group(1) == "agpr_count" and cur:
        rows.append(cur); cur = {}
    m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)$", line)
    if m and m.group(1) in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                            "private_segment_fixed_size", "group_segment_fixed_size", "agpr_count"):
        if m.group(1) == "name" and "name" in cur and not m.group(2).startswith("_Z") and "kernel" not in m.group(2):
            continue
        cur[m.group(1)] = m.group(2).strip("'")
if cur:
    rows.append(cur)
seen = set()
for r in rows:
    n = r.get("name", "?")
    if n in seen or flt not in n:
        continue
    seen.add(n)
    dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().replace("r3d::", "").replace("(KArgs)", "")
    print(f"{dem:58s} vgpr {r.get('vgpr_count','?'):>4} sgpr {r.get('sgpr_count','?'):>4} vspill {r.get('vgpr_spill_count','?'):>4} "
          f"sspill {r.get('sgpr_spill_count','?'):>4} scratch {r.get('private_segment_fixed_size','?'):>5} lds {r.get('group_segment_fixed_size','?'):>6}")
