#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch usage of libowwhip as the compiler reports it (no GPU needed).
usage: tools/kernel_resources.py [-DDEFINE ...] [--filter REGEX]"""
import os
import re
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
args = sys.argv[1:]
flt = None
if "--filter" in args:
    i = args.index("--filter")
    flt = re.compile(args[i + 1])
    del args[i:i + 2]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only",
       os.path.join(ROOT, "openwakeword_amd/csrc/owwhip.hip"), "-I" + os.path.join(ROOT, "include"),
       "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null"] + args
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
keys = {"VGPRs": "vgpr", "AGPRs": "agpr", "SGPRs": "sgpr", "ScratchSize [bytes/lane]": "scratch",
        "Occupancy [waves/SIMD]": "occ", "LDS Size [bytes/block]": "lds"}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    for k, short in keys.items():
        m = re.search(re.escape(k) + r": (\d+)", line)
        if m and cur:
            rows[cur][short] = int(m.group(1))
if not rows:
    sys.stderr.write(out[-3000:])
    sys.exit(1)
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.split("\n")
print(f"{'kernel':78s} VGPR AGPR SGPR scratch occ    LDS")
for (k, r), d in zip(rows.items(), names):
    d = re.sub(r"^void ", "", d).split("(")[0]
    if flt and not flt.search(d):
        continue
    print(f"{d[:78]:78s} {r.get('vgpr', 0):4d} {r.get('agpr', 0):4d} {r.get('sgpr', 0):4d} {r.get('scratch', 0):7d} {r.get('occ', 0):3d} {r.get('lds', 0):6d}")
