#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel average duration of the FULL-GRID launches only
(oww_commit also launches each CNN kernel 12 times on a 32-stream grid to derive the reset state; those
tiny launches are excluded so the averages are comparable with bench.py's hipEvent numbers).
usage: summarize.py <kernel_trace.csv> [<counter_collection.csv> ...]"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("owk::", "")
    for tag, s in (("StageCfg<24, 48", "stageB"), ("StageCfg<48, 72", "stageC"), ("StageCfg<72, 96", "stageD"), ("StageCfg<96, 96", "stageE")):
        if tag in name:
            return s + ("_lds(valu)" if ", false, " in name.split(">")[1] else "_lds")
    for tag, s in (("RCfg<24, 48", "stageB"), ("RCfg<48, 72", "stageC"), ("RCfg<72, 96", "stageD"), ("RCfg<96, 96", "stageE"), ("stageA_kernel", "stageA")):
        if tag in name and ("owr::rstage" in ("owr::" + name) or name.startswith("rstage")):
            return s + "_rr"
        if tag in name and name.startswith(("owh::hstage", "hstage")):
            return s + "_hx"
    if "hmelA_kernel" in name:
        return "stageA_hx"                       # mel front end + stage A in one launch (owwhip_fused.h); same key as the separate stage A
    if "vad_front_kernel" in name:
        return "vad_front"
    if "vad_lstm_kernel" in name:
        return "vad_lstm"
    if "heads_hx_kernel" in name:
        return "heads_hx"
    return name.split("(")[0][:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    by = defaultdict(list)
    for r in rows:
        by[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]), r))
    print(f"{'kernel':32s} {'launches':>8s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'grid':>10s} {'wg':>5s} {'vgpr':>5s} {'lds_B':>7s}  (full-grid launches only)")
    tot = 0.0
    out = []
    for k, v in by.items():
        gmax = max(g for _, g, _ in v)
        full = [d for d, g, _ in v if g == gmax]
        r = [x for _, g, x in v if g == gmax][0]
        out.append((sum(full), k, len(full), sum(full) / len(full) / 1e3, min(full) / 1e3, max(full) / 1e3, gmax, r))
    for s, k, n, avg, mn, mx, g, r in sorted(out, reverse=True):
        if not k.startswith(("stage", "mel", "heads", "postproc", "advance", "vad_", "verifier")):
            continue
        tot += s
        print(f"{k:32s} {n:8d} {avg:10.1f} {mn:10.1f} {mx:10.1f} {g:10d} {r['Workgroup_Size_X']:>5s} {r['VGPR_Count']:>5s} {r['LDS_Block_Size']:>7s}")
    print(f"total hot-path kernel time {tot / 1e6:.2f} ms")
    for path in sys.argv[2:]:
        rows = list(csv.DictReader(open(path)))
        agg = defaultdict(lambda: defaultdict(list))
        gmax = defaultdict(int)
        for r in rows:
            k = short(r["Kernel_Name"])
            gmax[k] = max(gmax[k], int(r["Grid_Size"]))
        for r in rows:
            k = short(r["Kernel_Name"])
            if int(r["Grid_Size"]) == gmax[k]:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(f"\ncounters from {path} (mean per full-grid launch)")
        for k in sorted(agg):
            if k.startswith(("stage", "mel", "heads", "r", "vad_")):
                print(f"  {k:20s} " + "  ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(agg[k].items())))


def traffic_json(paths, out_path):
    """HBM bytes per full-grid launch from the FETCH_SIZE / WRITE_SIZE passes (KB counters; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950) -> JSON that bench.py reports as roofline.traffic."""
    import json
    agg = defaultdict(lambda: defaultdict(list))
    grid = {}
    for path in paths:
        rows = list(csv.DictReader(open(path)))
        gmax = defaultdict(int)
        for r in rows:
            gmax[short(r["Kernel_Name"])] = max(gmax[short(r["Kernel_Name"])], int(r["Grid_Size"]))
        for r in rows:
            k = short(r["Kernel_Name"])
            if int(r["Grid_Size"]) == gmax[k] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                grid[k] = gmax[k]
    out = {}
    for k, v in agg.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            f = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]) * 1024 * 2
            w = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"]) * 1024
            out[k] = {"fetch_bytes": round(f), "write_bytes": round(w), "hbm_bytes": round(f + w), "grid": grid[k]}
    out["_csrc_sha16"] = build_hash()
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)


def build_hash():
    """Source hash of the library the counters were taken on (bench.py prices its composite bound only from a pass of the build it runs)."""
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from openwakeword_amd import _build
    return os.environ.get("OWW_SRC_SHA16") or _build.source_hash()


def instr_json(paths, out_path):
    """Per-kernel instruction and cycle counters of the full-grid launches (mean per launch) from all --pmc passes -> JSON that
    bench.py turns into the composite issue bound of the longest launch (VALU / MFMA / LDS instructions per stream-step)."""
    import json
    agg = defaultdict(lambda: defaultdict(list))
    grid = {}
    for path in paths:
        rows = list(csv.DictReader(open(path)))
        gmax = defaultdict(int)
        for r in rows:
            gmax[short(r["Kernel_Name"])] = max(gmax[short(r["Kernel_Name"])], int(r["Grid_Size"]))
        for r in rows:
            k = short(r["Kernel_Name"])
            if int(r["Grid_Size"]) == gmax[k]:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                grid[k] = gmax[k]
    out = {}
    for k, v in agg.items():
        if k.startswith(("stage", "heads_hx", "mel", "vad_")) and "SQ_INSTS_VALU" in v:
            out[k] = {c: sum(x) / len(x) for c, x in v.items()}
            out[k]["grid"] = grid[k]
    out["_csrc_sha16"] = build_hash()
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "--traffic":
        traffic_json(sys.argv[3:], sys.argv[2])
    elif sys.argv[1] == "--instr":
        instr_json(sys.argv[3:], sys.argv[2])
    else:
        main()
