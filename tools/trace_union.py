"""Bandwidth of OVERLAPPING launches from a rocprofv3 kernel trace (the per-dispatch start / end timestamps of
`rocprofv3 --kernel-trace --output-format csv`): for the dispatches whose kernel name contains <substr>, the number of launches, the
mean duration of one, the time during which AT LEAST ONE of them runs (the union of their intervals), and algorithmic bytes per
launch x launches / that time.  csmp_omp_batch runs two pipelines side by side from two signals on: two sweep launches overlap,
and `bytes / mean duration` of the stats table is then no bandwidth -- this is the figure the kernel trace itself supports, and the
one bench.py's `roofline.achieved` (HIP events on both streams, csmp_profile_window) must agree with.

    python tools/trace_union.py <rocprof output dir> "<kernel substr>" <bytes per launch> [skip_first]  >  profiles/r06_bench_kernel_union.json
"""
import csv
import glob
import json
import os
import sys


def main():
    d, sub, nbytes = sys.argv[1], sys.argv[2], float(sys.argv[3])
    skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    iv = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if sub in row.get("Kernel_Name", ""):
                iv.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
    if not iv:
        sys.stderr.write(f"trace_union.py: no dispatch of a kernel containing {sub!r} under {d}\n")
        sys.exit(2)
    iv.sort()
    iv = iv[skip:]  # (the warm-up batch's launches)
    union, cur_s, cur_e, peak, active = 0, iv[0][0], iv[0][1], 0, []
    for s, e in iv[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    # how many run at once, time-weighted
    ev = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
    depth, last, area = 0, ev[0][0], 0
    for t, dlt in ev:
        area += depth * (t - last)
        depth += dlt
        last = t
    mean = sum(e - s for s, e in iv) / len(iv)
    out = {"kernel": sub, "launches": len(iv), "skipped_first": skip, "mean_launch_duration_us": mean / 1e3, "union_ms": union / 1e6,
           "span_ms": (max(e for _, e in iv) - iv[0][0]) / 1e6, "mean_launches_in_flight": area / union,
           "us_per_launch_over_the_union": union / len(iv) / 1e3, "algorithmic_bytes_per_launch": nbytes,
           "achieved_GBps": nbytes * len(iv) / union, "frac_of_8_TBps": nbytes * len(iv) / union / 8000.0,
           "note": "achieved = bytes per launch x launches / time in which at least one such launch runs (rocprofv3 --kernel-trace timestamps)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
