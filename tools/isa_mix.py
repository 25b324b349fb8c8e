#!/usr/bin/env python3
"""Static instruction mix of the kernels in an assembly listing of owwhip.hip (hipcc -S --cuda-device-only).
usage: tools/isa_mix.py /tmp/owwhip.s [REGEX]"""
import collections
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for m in re.finditer(r"^(_Z\w+):\s*; @.*?\n(.*?)\.Lfunc_end\d+:", txt, re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    if flt and not flt.search(name):
        continue
    c = collections.Counter()
    for line in m.group(2).splitlines():
        line = line.strip()
        mm = re.match(r"([a-z_0-9]+)", line)
        if not mm or line.startswith(";") or line.startswith("."):
            continue
        op = mm.group(1)
        if op.startswith("v_mfma"): c["mfma"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): c["vmem"] += 1
        elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
        elif op.startswith("s_nop"): c["nop"] += 1
        elif op.startswith("s_barrier"): c["barrier"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        if op.startswith("v_cvt"): c["cvt"] += 1
        if op.startswith("v_pk_"): c["pk"] += 1
        if "dpp" in line: c["dpp"] += 1
        if op.startswith(("v_mov", "v_accvgpr")): c["mov"] += 1
    print(name[:100])
    print("   ", dict(c), " valu/mfma = %.2f" % (c["valu"] / max(1, c["mfma"])))
