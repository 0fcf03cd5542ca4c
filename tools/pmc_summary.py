#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes of tools/gpu_profile.sh into profiles/pmc_summary.json.

usage: python tools/pmc_summary.py gpurun_out/TAG profiles/NAME "workload name"

Per workload (the key bench.py looks up: "<scene> <nx>x<ny>x<spp>spp") for the dominant render kernel:
  hbm_bytes_per_launch = FETCH_SIZE x 2 + WRITE_SIZE (both reported in KB; gfx950 reports half of wide reads, see
                         /opt/skills/guides/MI355X_MICROARCH.md), each counter from its own pass; the minimum over the
                         launches of a pass is taken (steady state: the first launch also pays first-touch reads)
  sq                   = sums over the chip of the SQ counters of ONE launch (the last of the pass) and that launch's
                         duration; bench.py turns them into issue_frac / lane_util / wait_frac
                         (raytracing_rust_amd/roofline.py: sq_fractions)
Copies the small CSVs next to the summary."""
import csv
import glob
import json
import os
import shutil
import sys

SQ_COUNTERS = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU",
               "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY", "SQ_INSTS_LDS"]


def newest_csv(tag_dir, sub):
    files = glob.glob(os.path.join(tag_dir, sub, "**", "*counter_collection.csv"), recursive=True)
    return sorted(files, key=os.path.getmtime)[-1:]  # gpurun merges into existing directories: newest run only


def kname(row):
    return row["Kernel_Name"].split("(")[0].replace("void ", "")


def per_kernel(tag_dir, counter):
    files = newest_csv(tag_dir, "pmc_" + counter)
    out = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            out.setdefault(kname(row), {}).setdefault(row["Dispatch_Id"], 0.0)
            out[kname(row)][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {k: sorted(v.values()) for k, v in out.items()}, files


def sq_per_kernel(tag_dir):
    """kernel -> {counter sums of its LAST dispatch, launch_ns}"""
    files = newest_csv(tag_dir, "pmc_SQ")
    disp = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            key = (kname(row), int(row["Dispatch_Id"]))
            d = disp.setdefault(key, {"launch_ns": float(row["End_Timestamp"]) - float(row["Start_Timestamp"])})
            d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    out = {}
    for (name, did), d in sorted(disp.items(), key=lambda kv: kv[0][1]):
        out[name] = dict(d, dispatch_id=did)  # later dispatches overwrite: the last one stays
    return out, files


def build_hash():
    """rtmi_build_hash() of the in-tree library — the one that travelled to the GPU box and produced these counters (run this
    tool before touching csrc/ again; it refuses when the sources no longer hash to what the library carries)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from raytracing_rust_amd import abi, build

    lib_hash = abi.load_rtmi().rtmi_build_hash().decode()
    if lib_hash != build.source_hash():
        raise SystemExit("pmc_summary: librtmi.so is build %s but the sources hash to %s — rebuild, re-profile" % (lib_hash, build.source_hash()))
    return lib_hash


def main():
    tag_dir, prof_dir = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else "final_scene 1920x1080x1000spp"
    os.makedirs(prof_dir, exist_ok=True)
    fetch, ff = per_kernel(tag_dir, "FETCH_SIZE")
    write, wf = per_kernel(tag_dir, "WRITE_SIZE")
    sq, sf = sq_per_kernel(tag_dir)
    for src, dst in ((ff, "pmc_FETCH_SIZE.csv"), (wf, "pmc_WRITE_SIZE.csv"), (sf, "pmc_SQ.csv")):
        if src:
            shutil.copy(src[0], os.path.join(prof_dir, dst))
    kernels = {}
    for name in sorted(set(fetch) | set(write) | set(sq)):
        if not name.startswith("rtmi_"):
            continue
        f_kb = min(fetch.get(name, [0.0]))
        w_kb = min(write.get(name, [0.0]))
        kernels[name] = {"fetch_size_kb_raw": f_kb, "write_size_kb": w_kb, "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0,
                         "launches_seen": len(fetch.get(name, [])), "sq": sq.get(name)}
    render = [k for k in kernels if k.startswith("rtmi_render")]
    dom = max(render, key=lambda k: kernels[k]["hbm_bytes_per_launch"]) if render else None
    entry = {
        "dominant_kernel": dom,
        "hbm_bytes_per_launch": kernels[dom]["hbm_bytes_per_launch"] if dom else None,
        "sq": kernels[dom]["sq"] if dom else None,
        "kernels": kernels,
        "source": prof_dir,
        "build_hash": build_hash(),
        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes; FETCH_SIZE (KB) doubled per "
                "MI355X_MICROARCH.md (gfx950 reports 1/2 of wide reads); HBM bytes = min over the launches of the pass; "
                "SQ counters = chip-wide sums of the last launch of their pass, launch_ns = that launch's duration",
    }
    path = os.path.join(os.path.dirname(prof_dir.rstrip("/")), "pmc_summary.json")
    try:
        summary = json.load(open(path))
    except Exception:
        summary = {"workloads": {}}
    summary["workloads"][workload] = entry
    json.dump(summary, open(path, "w"), indent=1)
    print(json.dumps({workload: entry}, indent=1))


if __name__ == "__main__":
    main()
