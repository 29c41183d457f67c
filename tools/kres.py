"""Compile one .hip file for gfx950 and print registers / scratch per kernel (no GPU needed):  python tools/kres.py raindrop_amd/csrc/x.hip [-DFLAG ..]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(root, "include"),
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"] + sys.argv[2:]
r = subprocess.run(cmd, capture_output=True, text=True)
cur = None
for line in r.stderr.splitlines():
    if "error" in line:
        print(line)
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split()[0]] = int(m.group(2))
        if m.group(1).startswith("LDS"):
            n = re.sub(r"^_ZN2rd12_GLOBAL__N_1\d+", "", cur["name"])[:70]
            print("%-72s VGPR %3d AGPR %3d scratch %3d occ %d" % (n, cur.get("VGPRs", 0), cur.get("AGPRs", 0), cur.get("ScratchSize", 0), cur.get("Occupancy", 0)))
sys.exit(r.returncode)
