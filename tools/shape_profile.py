"""rocprofv3 --kernel-trace of `bench.py --workload shapes` -> one row per (shape, kernel): the product sweep's dispatches grouped by
kernel symbol, grid and dynamic LDS (which together identify the shape: tools/sweep_shapes.py's table carries the same triple),
their count and average duration, and the fraction of the 8 TB/s HBM peak that duration is for the shape's M*N*sizeof(T) bytes.
The --stats summary merges shapes that share a template instance; this keeps them apart.

    python tools/shape_profile.py <rocprof output dir> <shapes.json>  >  profiles/r06_sweep_shapes_kernel_stats.csv
"""
import csv
import glob
import json
import os
import sys

HBM_PEAK = 8.0e12


def main():
    prof, table = sys.argv[1], json.load(open(sys.argv[2]))
    disp = []
    for f in glob.glob(os.path.join(prof, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if not any(k in name for k in ("k_sweep_gen", "k_sweep_short", "k_sweep_ph")):
                continue
            wg = int(row.get("Workgroup_Size_X", row.get("Workgroup_Size", 256)) or 256)
            grid = int(row.get("Grid_Size_X", row.get("Grid_Size", 0)) or 0) // max(wg, 1)
            disp.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), name.split("(")[0], grid))
    disp.sort()
    # one segment per dictionary: the sweeps of a shape run back to back, the next dictionary's generation (seconds of other
    # kernels) lies between two shapes.  (The profiler reports a dynamic LDS allocation as 0 and several shapes share a template
    # instance and a grid: neither tells them apart.)
    segs = []
    for st, en, name, grid in disp:
        if not segs or st - segs[-1]["end"] > 20_000_000 or segs[-1]["name"] != name:
            segs.append({"name": name, "grid": set(), "durs": [], "end": en})
        segs[-1]["durs"].append(en - st)
        segs[-1]["grid"].add(grid)
        segs[-1]["end"] = en
    # a dictionary's first sweep may stand apart from the rest (the first launch of a template instance pays its set-up): neighbours
    # with the same kernel and grid are one shape as long as there are more segments than shapes
    i = 0
    while len(segs) > len(table) and i + 1 < len(segs):
        if segs[i]["name"] == segs[i + 1]["name"] and segs[i]["grid"] == segs[i + 1]["grid"] and min(len(segs[i]["durs"]), len(segs[i + 1]["durs"])) <= 2:
            segs[i]["durs"] += segs[i + 1]["durs"]
            segs[i]["end"] = segs[i + 1]["end"]
            del segs[i + 1]
        else:
            i += 1
    w = csv.writer(sys.stdout)
    w.writerow(["M", "N", "dtype", "Name", "Workgroups", "Calls", "AverageNs", "MedianNs", "MinNs", "MaxNs", "algorithmic_bytes", "frac_of_8TBps_at_median"])
    ok = len(segs) == len(table)
    for i, seg in enumerate(segs):
        r = table[i] if ok else None
        if r is not None:
            tname = "float" if r["dtype"] == "f32" else "double"
            if r["phases"] > 1:  # a residual longer than the LDS (round 6: its own kernel)
                want = "k_sweep_ph<%s, 8, 4>" % tname
            elif r.get("columns_per_unit", 1) > 1:  # short columns, several to a unit: <element type, chunks per column, columns per reduction>
                cpu = r["columns_per_unit"]
                want = "k_sweep_short<%s, %d, %d>" % (tname, 8 // cpu, 4 if cpu >= 4 else 2)
            else:
                want = "k_sweep_gen<%s, %d, %d>" % (tname, r["unit_loads"], 32 // r["unit_loads"])
            if want not in seg["name"]:
                ok, r = False, None
        durs = sorted(seg["durs"])
        avg = sum(durs) / len(durs)
        med = durs[len(durs) // 2]
        grids = "/".join(str(g) for g in sorted(seg["grid"]))
        if r is not None:
            w.writerow([r["M"], r["N"], r["dtype"], seg["name"], grids, len(durs), round(avg, 1), med, durs[0], durs[-1], r["bytes"],
                        round(r["bytes"] / (med * 1e-9) / HBM_PEAK, 4)])
        else:
            w.writerow(["", "", "", seg["name"], grids, len(durs), round(avg, 1), med, durs[0], durs[-1], "", ""])


if __name__ == "__main__":
    main()
