#!/usr/bin/env python3
"""dev tool: folds rocprofv3 --pmc passes (one counter group per pass, csv output) into the JSON kept
under profiles/.  usage: summarise_pmc.py <dir with the passes> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

KERNEL = "stage1_kernel<true"          # the emitting stage-1 kernel, any DBG / DIALECT arguments
LAUNCH_BYTES = 8589934592              # bench default: 8 GiB of the 64x31 corpus
TAPE_BYTES = LAUNCH_BYTES // 32 * 8


def main():
    global LAUNCH_BYTES, TAPE_BYTES
    root, out = sys.argv[1], sys.argv[2]
    workload = sys.argv[5] if len(sys.argv) > 5 else "64x31_noquote 8 GiB"
    if len(sys.argv) > 4:  # other workloads: launch bytes and tape bytes per launch
        LAUNCH_BYTES, TAPE_BYTES = int(sys.argv[3]), int(sys.argv[4])
    per_counter = collections.defaultdict(lambda: collections.defaultdict(float))
    name_seen = set()
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                kn = row.get("Kernel_Name", "")
                flat = kn.replace(" ", "")
                if KERNEL not in flat or not any(t in flat for t in ("<true,0,0,false,false>", "<true,0,0,false,true>", "<true,0,0,false>",
                                                                   "<true,0,0>", "<true,0>", "<true>")):
                    continue  # only the reference-dialect, non-probe instantiations (round 4: default or dense geometry)
                name_seen.add(kn)
                key = (os.path.basename(os.path.dirname(path)), row.get("Dispatch_Id"))
                per_counter[row["Counter_Name"]][key] += float(row["Counter_Value"])
    counters = {c: {"dispatches": len(v), "avg": sum(v.values()) / len(v)} for c, v in sorted(per_counter.items())}
    d = {"command": "rocprofv3 --pmc <one counter group per pass> --output-format csv -- python3 bench.py --steps 10 "
                    "--warmup 2 --no-extra --no-cpu-baseline --no-verify --no-q10-check --no-ingest "
                    "(scripts/collect_profiles_r02.sh)",
         "kernel": sorted(name_seen), "workload": workload, "counters": counters}
    g = lambda c: counters[c]["avg"] if c in counters else None
    der = {}
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        der["FETCH_SIZE_bytes_raw"] = g("FETCH_SIZE") * 1024
        der["FETCH_SIZE_bytes_x2_gfx950_correction"] = g("FETCH_SIZE") * 2048
        der["WRITE_SIZE_bytes"] = g("WRITE_SIZE") * 1024
        der["algorithmic_read_bytes"] = LAUNCH_BYTES
        der["algorithmic_tape_bytes"] = TAPE_BYTES
        der["hbm_traffic_bytes_per_launch"] = der["FETCH_SIZE_bytes_x2_gfx950_correction"] + der["WRITE_SIZE_bytes"]
    if g("SQ_INSTS_VALU") is not None and g("GRBM_GUI_ACTIVE") is not None:
        der["valu_issue_utilisation"] = g("SQ_INSTS_VALU") * 4 / (1024 * g("GRBM_GUI_ACTIVE") / 8)
    der["note"] = ("FETCH_SIZE on gfx950 reports exactly half of a wide coalesced stream (MI355X_MICROARCH.md §HBM): "
                   "doubled. WRITE_SIZE is exact for 16-B/lane streaming stores. VALU utilisation = instructions x 4 "
                   "cycles / (1024 SIMDs x GRBM_GUI_ACTIVE/8).")
    d["derived"] = der
    with open(out, "w") as f:
        json.dump(d, f, indent=1)
    print(json.dumps(der, indent=1))


if __name__ == "__main__":
    main()
