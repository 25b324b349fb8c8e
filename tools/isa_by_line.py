#!/usr/bin/env python3
"""Attribute the instructions of one kernel to source lines (needs an assembly listing built with -gline-tables-only):
hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -gline-tables-only openwakeword_amd/csrc/owwhip.hip -Iinclude -o /tmp/owwhip_g.s
usage: tools/isa_by_line.py /tmp/owwhip_g.s 'hmelA_kernel<false>' [bucket_lines=1]"""
import collections
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2]
bucket = int(sys.argv[3]) if len(sys.argv) > 3 else 1
files = {int(m.group(1)): m.group(2) for m in re.finditer(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', txt)}
for m in re.finditer(r"^(_Z\w+):\s*; @.*?\n(.*?)\.Lfunc_end\d+:", txt, re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    if want not in name:
        continue
    cur = ("?", 0)
    cnt = collections.defaultdict(collections.Counter)
    for line in m.group(2).splitlines():
        line = line.strip()
        lm = re.match(r"\.loc\s+(\d+)\s+(\d+)", line)
        if lm:
            cur = (files.get(int(lm.group(1)), "?").split("/")[-1], int(lm.group(2)) // bucket * bucket)
            continue
        mm = re.match(r"([a-z_0-9]+)", line)
        if not mm or line.startswith((";", ".")):
            continue
        op = mm.group(1)
        kind = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
                "vmem" if op.startswith(("global_", "buffer_", "scratch_")) else "wait" if op.startswith(("s_waitcnt", "s_nop")) else
                "salu" if op.startswith("s_") else None)
        if kind:
            cnt[cur][kind] += 1
    print(name)
    tot = collections.Counter()
    for k in sorted(cnt):
        c = cnt[k]
        tot.update(c)
        if c["valu"] + c["mfma"] + c["lds"] + c["vmem"] >= 8:
            print(f"  {k[0]:>18}:{k[1]:<5} " + " ".join(f"{n}={c[n]}" for n in ("mfma", "valu", "lds", "vmem", "salu", "wait") if c[n]))
    print("  total", dict(tot))
