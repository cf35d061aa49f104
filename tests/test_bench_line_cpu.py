"""The ONE stdout line of bench.py must stay small enough for the driver to parse (round 3's 20.9 KB line was not:
BENCH_r03.parsed = null) and must carry `roofline` and `cpu_baseline`.  The line builder runs here on canned full records
(the committed records of earlier rounds): no GPU needed."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _records():
    out = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*bench*.json"))):
        text = open(path).read().strip()
        if not text:
            continue
        try:
            rec = json.loads(text) if text.startswith("{") and "\n{" not in text else json.loads(text.splitlines()[-1])
        except ValueError:
            continue
        if isinstance(rec, dict) and "metric" in rec and "value" in rec:
            out.append((os.path.basename(path), rec))
    return out


RECORDS = _records()


def test_there_are_canned_records():
    assert len(RECORDS) >= 3
    assert any(len(json.dumps(r)) > 15000 for _, r in RECORDS), "the round-3 sized record must be among the canned inputs"


@pytest.mark.parametrize("name,rec", RECORDS, ids=[n for n, _ in RECORDS])
def test_compact_line_is_small_and_complete(name, rec):
    line = bench.compact_line(rec)
    assert "\n" not in line and len(line) < bench.COMPACT_LIMIT <= 6000
    got = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config"):
        assert k in got, k
    assert got["value"] == pytest.approx(rec["value"], rel=1e-4)
    assert got["ms_per_step"] == pytest.approx(rec["ms_per_step"], rel=1e-4)
    assert isinstance(got["config"], dict) and "workload" in got["config"] and "model" not in got["config"]
    if "roofline" in rec:
        r = got["roofline"]
        assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
        assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3)
        assert "traffic" in r and "kernel" in r and "avg_launch_us" in r
    if "cpu_baseline" in rec:
        c = got["cpu_baseline"]
        assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] in ("port", "reference") and len(c["sample"]) <= 200


def test_compact_line_survives_pathological_records():
    rec = dict(RECORDS[-1][1])
    rec["config"] = dict(rec["config"], workload="x" * 5000, dense_gemm_arithmetic="y" * 5000)
    rec["cpu_baseline"] = {"value": 1.0, "unit": "u", "cores": 1, "kind": "port", "sample": "z" * 9000}
    for leg in ("cfg3", "cfg5", "cfg2grid"):
        rec[leg] = {"error": "e" * 9000}
    rec["rccl_ranks"] = 8
    line = bench.compact_line(rec)
    assert len(line) < bench.COMPACT_LIMIT
    got = json.loads(line)
    assert got["rccl_ranks"] == 8 and got["cpu_baseline"]["value"] == 1.0
    rec["value"] = float("nan")
    assert json.loads(bench.compact_line(rec))["value"] is None     # strict JSON: never a bare NaN


def test_roofline_frac_follows_from_the_fields_beside_it():
    """`frac` == algorithmic_per_launch / avg_launch_us / peak, on a canned record re-priced with the
    concurrent (rocprofv3) average the way main() does, and the HIP-event figures stay beside it under *_hip_events; the
    compact line keeps the identity and carries the step-level PMC byte total."""
    name, rec = [(n, r) for n, r in RECORDS if "roofline" in r and r["roofline"].get("algorithmic_per_launch")][-1]
    roof = dict(rec["roofline"])
    if "avg_launch_us_hip_events" in roof:       # a round-5 record: consistent as stored
        assert bench.roofline_identity_error(roof) < 2e-3, name
        roof["avg_launch_us"] = roof.pop("avg_launch_us_hip_events")
    # undo the re-pricing with the concurrent average back to the HIP-event state roofline_of() returns
    if "frac_hip_events" in roof:
        for k in ("achieved", "frac", "frac_dense_counted", "issued_bf16_tflops", "frac_of_executing_pipe"):
            if k + "_hip_events" in roof:
                roof[k] = roof.pop(k + "_hip_events")
        roof.pop("avg_launch_us_concurrent", None)
    assert bench.roofline_identity_error(roof) < 2e-3, name
    prefix = bench._kernel_prefix(roof["kernel"])
    views = {"traffic": {prefix + " a>": 1.0e8, prefix + " b>": 2.0e8},
             "kernels_concurrent": [{"kernel": prefix + " a>", "launches_per_step": 4, "avg_us": 0.9 * roof["avg_launch_us"], "per_step_ms": 0.1},
                                    {"kernel": prefix + " b>", "launches_per_step": 12, "avg_us": 0.95 * roof["avg_launch_us"], "per_step_ms": 0.3},
                                    {"kernel": "something_else", "launches_per_step": 3, "avg_us": 1.0, "per_step_ms": 0.003}]}
    hip_avg, hip_frac = roof["avg_launch_us"], roof["frac"]
    bench.apply_concurrent(roof, views)
    assert roof["avg_launch_us_hip_events"] == hip_avg and roof["frac_hip_events"] == hip_frac
    assert roof["avg_launch_us"] == pytest.approx((4 * 0.9 + 12 * 0.95) / 16 * hip_avg)
    assert bench.roofline_identity_error(roof) < 2e-3 and roof["frac"] > hip_frac
    assert roof["traffic"] == pytest.approx(1.5e8) and "avg_launch_us_concurrent" not in roof
    roof["traffic_per_step"], roof["algorithmic_bytes_per_step"] = 9.5e9, 3.2e9
    line = json.loads(bench.compact_line(dict(rec, roofline=roof)))["roofline"]
    assert bench.roofline_identity_error(line) < 2e-3
    assert line["traffic_per_step"] == pytest.approx(9.5e9, rel=1e-4) and line["algorithmic_bytes_per_step"] == pytest.approx(3.2e9, rel=1e-4)
    assert line["avg_launch_us_hip_events"] == pytest.approx(hip_avg, rel=1e-4)
