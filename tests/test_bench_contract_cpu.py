"""bench.py's contract, checked without a GPU: the arithmetic of the committed record line (the driver's own line has
the same fields) and the CPU-baseline leg on a tiny problem.  The oracle is used here as the checker only."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORD = os.path.join(ROOT, "profiles", "r04_bench_line.json")


@pytest.fixture(scope="module")
def line():
    if not os.path.exists(RECORD):
        pytest.skip("no committed record line")
    return json.load(open(RECORD))


def test_record_line_has_the_contract_fields(line):
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["metric"].startswith("ADI shift-solves/sec") and base["metric"].startswith("ADI shift-solves/sec")
    assert line["unit"] == "shift-solves/s" and line["higher_is_better"] is True
    assert line["n_gpus"] == 1 and line["dtype"] == "f64" and line["data"] == "synthetic"
    assert line["vs_baseline"] is None and base["published"] == {}        # no published number for this metric
    cfg = line["config"]
    assert "workload" in cfg and cfg["workload"].startswith("cfg2") and "model" not in cfg
    # whole-job throughput = the step's shift-solves over the step's wall-clock time
    assert line["value"] == pytest.approx(cfg["shift_solves_per_step"] / (line["ms_per_step"] * 1e-3), rel=2e-3)
    assert cfg["K_rel_diff_vs_oracle"] < 1e-6                              # north_star's parity bar


@pytest.mark.parametrize("key", ["roofline", "roofline_cfg5"])
def test_roofline_object_arithmetic(line, key):
    if key not in line:
        pytest.skip(key + " not in this line")
    r = line[key]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # achieved = algorithmic bytes per launch / the launch's duration; frac = achieved / peak
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes"] / (r["us_per_launch"] * 1e-6) / 1e9, rel=2e-3)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], abs=2e-4)
    # one launch serves 16 shifts of one pattern: the line's frac is the batched-form fraction, not the per-panel one
    assert r["units_per_launch"] == 16
    assert r["algorithmic_bytes"] == r["algorithmic_bytes_batched_form"] < r["algorithmic_bytes_per_panel_model"]
    assert r["frac"] == r["frac_batched_form"] < r["frac_per_panel_model"]
    # bytes per unit = 12 nnz + 4 (n + 1) + 16 n m (values + pattern of one shift, FP64 panels in and out)
    n, m, nnz = r["n"], r["m"], r["nnz"]
    assert r["algorithmic_bytes_per_unit"] == 12 * nnz + 4 * (n + 1) + 16 * n * m
    if r.get("traffic") is not None:
        assert r["traffic_over_batched_form"] == pytest.approx(r["traffic"] / r["algorithmic_bytes"], abs=2e-3)


def test_cpu_baseline_object(line):
    c = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["unit"] == "shift-solves/s" and c["cores"] >= 1
    assert line["value"] > c["value"] > 0


def test_cpu_baseline_leg_runs_on_a_tiny_problem():
    """The checker leg itself (oracle SaddleLU timed on the host cores), N = 6: finite, labelled with its own size."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from optconpy_amd import problems as pb
    pr = pb.ricc_problem(6, 0.1, NU=2, NY=2)
    out = bench.cpu_baseline(pr, pb.logshifts(1.0, 100.0, 4), 4, 8)
    one = out.get("single_core", out)
    assert np.isfinite(out["value"]) and out["value"] > 0 and one["cores"] == 1
    assert "n = %d" % (pr.NV + pr.J.shape[0]) in one["sample"]
    assert one["step_seconds"] == pytest.approx(4 * one["lu_seconds"] + 8 * one["solve_seconds"], rel=0.05, abs=0.02)
