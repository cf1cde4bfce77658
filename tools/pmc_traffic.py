"""Parse a rocprofv3 --pmc csv (counter_collection.csv) -> average FETCH_SIZE / WRITE_SIZE per
k_sweep dispatch, with the gfx950 corrections of MI355X_MICROARCH.md section HBM:
FETCH_SIZE (KiB units) reports exactly 1/2 of the bytes of a wide coalesced 16-B/lane streaming
read -> doubled; WRITE_SIZE is exact for 16-B stores.  Usage: pmc_traffic.py [--kernel SUBSTRING] <dir> [<dir> ...]
(default: the exact sweeps, k_tick< and k_sweep_pf / k_sweep<; --kernel k_sweep_bf16 for the screened sweep)"""
import csv, glob, json, os, sys
out = {}
args = sys.argv[1:]
only = None
if args and args[0] == "--kernel":
    only = args[1]
    args = args[2:]
for d in args:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            kn = row.get("Kernel_Name", "")
            if only is not None:
                if only not in kn:
                    continue
            elif "k_tick<" not in kn and ("k_sweep" not in kn or "k_sweep_bf16" in kn):
                continue
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for name, vals in acc.items():
            out[name] = {"dispatches": len(vals), "avg_raw": sum(vals) / len(vals)}
if "FETCH_SIZE" in out:
    out["fetch_bytes_per_launch_corrected"] = out["FETCH_SIZE"]["avg_raw"] * 1024 * 2
if "WRITE_SIZE" in out:
    out["write_bytes_per_launch"] = out["WRITE_SIZE"]["avg_raw"] * 1024
if "fetch_bytes_per_launch_corrected" in out:
    out["hbm_bytes_per_launch"] = out["fetch_bytes_per_launch_corrected"] + out.get("write_bytes_per_launch", 0.0)
if only is not None:
    out["kernel"] = only
print(json.dumps(out, indent=1))
