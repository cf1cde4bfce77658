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
    assert set(o["secondary"]) == set(out["secondary"])
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
