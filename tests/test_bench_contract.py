"""The committed bench line (profiles/r2e_bench_10M.json, written by `python bench.py` on an MI355X) keeps
the driver's contract: one JSON object with the metric / config / roofline / cpu_baseline fields, and numbers
that are consistent with each other."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(HERE, "..", "profiles", "r2e_bench_10M.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "records/s" and d["dtype"] == "u8" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and d["config"]["records_per_gpu"] == 10_000_000
    # value = records of all ranks over the timed steps / time
    assert abs(d["value"] - d["config"]["records_per_gpu"] * d["n_gpus"] / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] / 1e3) / 1e9) / r["achieved"] < 1e-2
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == "records/s" and c["sample"]
