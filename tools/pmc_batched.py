"""HBM traffic of the batched path's kernels from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, separate runs of
`bench.py --workload batched --steps 1 --warmup 0`), per launch and kernel, beside the ALGORITHMIC bytes of configs[2]
(1024 signals sharing A 4096 x 65536 f32, k = 128; averages over the 128 steps of a batch, j = support size at the step):
  k_b_screen256p  the bf16 dictionary once + the bf16 residual images, candidates out
  k_b_pick        per signal: candidates (2048 x 8 B), residual (M x 8 B), the window's columns (counted from the run: unknown here)
  k_b_append      per signal: 2 j columns (pass 1, pass 2; 1 j with the Gram option) + the new column + residual in and out +
                  bf16 image out + T and T' (j^2 / 2 x 8 B each)
gfx950 corrections as tools/pmc_traffic.py (FETCH_SIZE in KiB and counting half the bytes of 16-B-per-lane reads).
The WRITE_SIZE pass of this workload hung under rocprofv3 on this pool (round 3: 45 minutes, 564 incomplete dispatches, killed;
the FETCH_SIZE pass of the same command takes seconds), so `<write dir>` may be `-`: the kernels' writes are then taken as
their algorithmic size (residual, bf16 image, T column: 2 % of the append kernel's bytes) and flagged as such.
    pmc_batched.py <fetch dir> <write dir | -> [gram]"""
import csv, glob, json, os, sys

M, N, B, K = 4096, 65536, 1024, 128
gram = len(sys.argv) > 3 and sys.argv[3] == "gram"


KEYS = ("k_b_screen256p", "k_b_pick", "k_b_append")
seen_names, seen_files = set(), []


def collect(d):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen_files.append(f)
        for row in csv.DictReader(open(f)):
            kn = row.get("Kernel_Name", "")
            seen_names.add(kn.split("(")[0][:80])
            for key in KEYS:
                if key in kn:
                    acc.setdefault((key, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    return acc


acc = collect(sys.argv[1])
have_writes = sys.argv[2] != "-"
if have_writes:
    acc.update(collect(sys.argv[2]))
alg_writes = {"k_b_screen256p": B * (N // 128) * 4 * 8, "k_b_pick": B * 64, "k_b_append": B * (M * 8 + M * 2 + 2 * (K / 2) * 8)}
j_avg = (K - 1) / 2.0
j2_avg = sum(j * j for j in range(K)) / K
alg = {
    "k_b_screen256p": N * M * 2 + B * M * 2 + B * (N // 128) * 4 * 8,
    "k_b_pick": B * ((N // 128) * 4 * 8 + M * 8),
    "k_b_append": B * ((1 if gram else 2) * j_avg * M * 4 + M * 4 + 2 * M * 8 + M * 2 + j2_avg * 8),
}
out = {"note": "bytes per launch; fetch corrected x2 (gfx950, 16-B-per-lane reads: an upper bound where a kernel's reads are narrower)",
       "gram_option": gram}
for key in ("k_b_screen256p", "k_b_pick", "k_b_append"):
    f = acc.get((key, "FETCH_SIZE"))
    w = acc.get((key, "WRITE_SIZE"))
    if not f:
        continue
    fb = sum(f) / len(f) * 1024 * 2
    wb = sum(w) / len(w) * 1024 if w else alg_writes[key]
    out[key] = {"dispatches": len(f), "fetch_bytes": fb, "write_bytes": wb, "write_bytes_measured": bool(w), "hbm_bytes": fb + wb,
                "algorithmic_bytes": alg[key], "ratio": (fb + wb) / alg[key]}
    if key == "k_b_pick":
        out[key]["note"] = "algorithmic = candidates + residual only; the rest is the window's rescored columns (16 KiB each)"
        out[key]["rescored_columns_per_signal_and_step"] = max(0.0, (fb + wb - alg[key]) / B / (M * 4))
missing = [key for key in KEYS if key not in out]
if missing:
    # (round 5 committed a file with no kernel rows at all: an empty collection must not look like a result)
    sys.stderr.write("pmc_batched.py: no FETCH_SIZE rows for %s in %s (%d counter file(s); kernels seen: %s)\n"
                     % (", ".join(missing), sys.argv[1], len(seen_files), ", ".join(sorted(seen_names)[:12]) or "none"))
    sys.exit(2)
print(json.dumps(out, indent=1))
