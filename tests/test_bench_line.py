"""The driver keeps only ~8 KB of bench.py's stdout tail: the LAST stdout line must be the compact headline (round 3's single
30 KB line was cut mid-way and its driver record did not parse).  These CPU tests feed bench.headline()/emit() the largest
line the bench has ever produced (profiles/r03_bench_line.json: headline + 22 secondary blocks with prose notes)."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _fat_line():
    out = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_line.json")))
    assert len(json.dumps(out)) > 25000 and len(out["secondary"]) >= 20
    return out


def test_headline_is_compact_and_complete():
    out = _fat_line()
    line = bench.headline(out)
    assert len(line) <= bench.LINE_BUDGET < 4096, len(line)
    o = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in o, k
    assert o["config"]["workload"].startswith("configs[1]")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches_timed", "avg_launch_us", "algorithmic_bytes_per_launch"):
        assert k in o["roofline"], k
    assert abs(o["roofline"]["frac"] - o["roofline"]["achieved"] / o["roofline"]["peak"]) < 1e-4
    for k in ("value", "unit", "cores", "kind", "sample", "selection_order_matches_gpu"):
        assert k in o["cpu_baseline"], k
    # one scalar per secondary workload, no nested blocks, no prose
    derived = {k for k in o["secondary"] if k.endswith("_three_in_flight") or k.startswith("batched_c3_frac_of_")}
    assert set(o["secondary"]) - derived == set(out["secondary"])
    assert all(v is None or isinstance(v, (int, float)) for v in o["secondary"].values())
    assert "note" not in line


def test_last_4k_of_stdout_parses(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    out = _fat_line()
    buf = io.StringIO()
    with redirect_stdout(buf):
        print("some earlier chatter " * 400)
        bench.emit(out)
    tail = buf.getvalue()[-4096:]
    last = tail.strip().splitlines()[-1]
    o = json.loads(last)
    assert o["value"] == float("%.6g" % out["value"]) and o["secondary_file"] == bench.DETAIL_FILE
    # the full detail went to the file, notes and all
    detail = json.load(open(tmp_path / bench.DETAIL_FILE))
    assert detail["secondary"]["sp_c5"]["roofline"]["note"]


def test_headline_survives_errors_and_missing_blocks():
    o = json.loads(bench.headline({"metric": "m", "value": 1.0, "unit": "atoms/s", "cpu_baseline": {"value": None, "error": "x" * 5000},
                                   "secondary": {"a": {"error": "boom"}, "b": {"value": 2.5}}, "roofline": {"bound": "hbm", "kernel": "k = " + "y" * 900}}))
    assert o["secondary"] == {"a": None, "b": 2.5} and len(o["cpu_baseline"]["error"]) <= 160 and o["roofline"]["kernel"] == "k"


def test_headline_carries_both_ceilings_of_the_batched_step_and_the_device_resident_single_signal():
    """batched_c3 priced against the MFMA-only ceiling AND the serial composite (MFMA flop at 2.5 PF + the
    per-signal kernels' bytes at 8 TB/s); a one-signal figure with b and the results resident in HBM; the roofline's launch
    duration net of the event pair's own reading."""
    comp = bench.composite_roofline(flops=2.0 * 4096 * 65536 * 1024, peak_tf=bench.MFMA_PEAK_TF, hbm_bytes=2.27e9, ms_per_omp_step=0.795, nsig=1024)
    assert abs(comp["mfma_floor_us"] - 219.9) < 0.5 and abs(comp["hbm_floor_us"] - 283.75) < 0.5
    assert abs(comp["frac"] - (219.9 + 283.75) / 795.0) < 2e-3 and 1.9e6 < comp["ceiling_atoms_per_s"] < 2.1e6
    sec = {"batched_c3": {"value": 1.29e6, "roofline": {"whole_step": {"frac": 0.277}}, "roofline_composite": comp},
           "lone_omp_c2_device": {"value": 5900.0}, "lone_omp_c2": {"value": 5750.0},
           "srr_8f2": {"value": 45.0, "three_in_flight": {"solves": 9, "solves_per_s": 62.5}}, "ompr_8f2": {"value": 66.0}}
    o = json.loads(bench.headline({"metric": "m", "value": 1.0, "unit": "atoms/s", "secondary": sec,
                                   "roofline": {"bound": "hbm", "avg_launch_us": 160.0, "event_pair_us": 2.1, "kernel": "k"}}))
    assert o["secondary"]["batched_c3_frac_of_mfma_only_ceiling"] == 0.277
    assert abs(o["secondary"]["batched_c3_frac_of_composite_ceiling"] - comp["frac"]) < 1e-5
    assert o["secondary"]["lone_omp_c2_device"] == 5900.0
    assert o["secondary"]["srr_8f2"] == 45.0 and o["secondary"]["srr_8f2_three_in_flight"] == 62.5 and "ompr_8f2_three_in_flight" not in o["secondary"]
    assert all(v is None or isinstance(v, (int, float)) for v in o["secondary"].values())
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "profile_overhead" in src and "statistical_int8" not in src.split("def main()")[1].split("for name, cert, gram, scr in")[1][:400]
