#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes of tools/gpu_profile.sh into profiles/pmc_traffic.json.

usage: python tools/pmc_summary.py gpurun_out/TAG profiles/NAME [workload]

HBM bytes per launch of the dominant kernel = FETCH_SIZE x 2 + WRITE_SIZE (both reported in KB;
gfx950 reports half of wide reads, see /opt/skills/guides/MI355X_MICROARCH.md), each counter from
its own pass; the minimum over the launches of a pass is taken (steady state: the first launch
also pays first-touch reads of the scene).  Copies the small CSVs next to the summary."""
import csv
import glob
import json
import os
import shutil
import sys


def per_kernel(tag_dir, counter):
    files = glob.glob(os.path.join(tag_dir, "pmc_" + counter, "**", "*counter_collection.csv"), recursive=True)
    files = sorted(files, key=os.path.getmtime)[-1:]  # gpurun merges into existing directories: newest run only
    out = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            out.setdefault(name, {}).setdefault(row["Dispatch_Id"], 0.0)
            out[name][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {k: sorted(v.values()) for k, v in out.items()}, files


def main():
    tag_dir, prof_dir = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else "final_scene 1920x1080x1000spp"
    os.makedirs(prof_dir, exist_ok=True)
    fetch, ff = per_kernel(tag_dir, "FETCH_SIZE")
    write, wf = per_kernel(tag_dir, "WRITE_SIZE")
    for src, dst in ((ff, "pmc_FETCH_SIZE.csv"), (wf, "pmc_WRITE_SIZE.csv")):
        if src:
            shutil.copy(src[0], os.path.join(prof_dir, dst))
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        if not name.startswith("rtmi_"):
            continue
        f_kb = min(fetch.get(name, [0.0]))
        w_kb = min(write.get(name, [0.0]))
        kernels[name] = {"fetch_size_kb_raw": f_kb, "write_size_kb": w_kb, "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0,
                         "launches_seen": len(fetch.get(name, []))}
    render = [k for k in kernels if k.startswith("rtmi_render")]
    dom = max(render, key=lambda k: kernels[k]["hbm_bytes_per_launch"]) if render else None
    summary = {
        "workload": workload,
        "dominant_kernel": dom,
        "hbm_bytes_per_launch": kernels[dom]["hbm_bytes_per_launch"] if dom else None,
        "kernels": kernels,
        "source": prof_dir,
        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KB); FETCH_SIZE doubled per MI355X_MICROARCH.md "
                "(gfx950 reports 1/2 of wide reads); min over the launches of the pass (steady state)",
    }
    json.dump(summary, open(os.path.join(os.path.dirname(prof_dir.rstrip("/")), "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
