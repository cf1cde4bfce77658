"""rocprofv3 --kernel-trace of `bench.py --workload shapes` -> one row per (shape, kernel): the product sweep's dispatches grouped by
kernel symbol, grid and dynamic LDS (which together identify the shape: tools/sweep_shapes.py's table carries the same triple),
their count and average duration, and the fraction of the 8 TB/s HBM peak that duration is for the shape's M*N*sizeof(T) bytes.
The --stats summary merges shapes that share a template instance; this keeps them apart.

    python tools/shape_profile.py <rocprof output dir> <shapes.json>  >  profiles/r05_sweep_shapes_kernel_stats.csv
"""
import csv
import glob
import json
import os
import sys

HBM_PEAK = 8.0e12


def main():
    prof, table = sys.argv[1], json.load(open(sys.argv[2]))
    groups = {}
    for f in glob.glob(os.path.join(prof, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if "k_sweep_gen" not in name:
                continue
            wg = int(row.get("Workgroup_Size_X", row.get("Workgroup_Size", 256)) or 256)
            grid = int(row.get("Grid_Size_X", row.get("Grid_Size", 0)) or 0) // max(wg, 1)
            lds = int(row.get("LDS_Block_Size", row.get("LDS_Block_Size_v", 0)) or 0)
            dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            groups.setdefault((name.split("(")[0], grid, lds), []).append(dur)
    w = csv.writer(sys.stdout)
    w.writerow(["M", "N", "dtype", "Name", "Workgroups", "LDS_Block_Size", "Calls", "AverageNs", "MinNs", "MaxNs", "algorithmic_bytes", "frac_of_8TBps"])
    used = set()
    for r in table:
        tname = "float" if r["dtype"] == "f32" else "double"
        want = "k_sweep_gen<%s, %d, %d, %s>" % (tname, r["unit_loads"], 32 // r["unit_loads"], "true" if r["phases"] > 1 else "false")
        hits = [k for k in groups if want in k[0] and k[1] == r["workgroups"] and k not in used]
        if not hits:
            continue
        key = min(hits, key=lambda k: abs(k[2] - r["lds_bytes"]))  # (the profiler reports the LDS allocation in its own granules)
        used.add(key)
        durs = sorted(groups[key])
        avg = sum(durs) / len(durs)
        w.writerow([r["M"], r["N"], r["dtype"], key[0], key[1], key[2], len(durs), round(avg, 1), durs[0], durs[-1], r["bytes"],
                    round(r["bytes"] / (avg * 1e-9) / HBM_PEAK, 4)])
    for key in sorted(groups):
        if key not in used:  # (dispatches no row of the table claimed: the check sweeps of another grid, if any)
            durs = sorted(groups[key])
            w.writerow(["", "", "", key[0], key[1], key[2], len(durs), round(sum(durs) / len(durs), 1), durs[0], durs[-1], "", ""])


if __name__ == "__main__":
    main()
