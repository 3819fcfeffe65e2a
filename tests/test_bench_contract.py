"""The committed bench line (profiles/r4_bench_10M.json, written by `python bench.py` on an MI355X) keeps
the driver's contract: one JSON object with the metric / config / roofline / cpu_baseline fields, and numbers
that are consistent with each other."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(HERE, "..", "profiles", "r4_bench_10M.json")))
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
    # round 3: the timed output is checked at the timed size, the traffic comes from a PMC pass of the same kernel sources, and the
    # rocprofv3 summary of the same command agrees with the HIP-event time of the dominant kernel
    v = d["verify"]
    assert v["fused_equals_unfused"] is True and v["oracle_sample_matches"] is True and v["oracle_sample_rows"] >= 1000
    assert r["traffic"] is not None and r["traffic"] > r["algorithmic_bytes_per_launch"]
    import csv
    rows = list(csv.DictReader(open(os.path.join(HERE, "..", "profiles", "r4_kernel_stats_bench_10M.csv"))))
    reg = [x for x in rows if "k_parser_reg<false" in x["Name"]]
    assert reg and abs(float(reg[0]["AverageNs"]) / 1e6 - r["avg_launch_ms"]) / r["avg_launch_ms"] < 0.05
    for k in ("parser_only", "mixed_shapes", "multiline", "config2_ndjson_grep32"):
        assert k in d["secondary"] and "error" not in d["secondary"][k], k
    assert d["secondary"]["multiline"]["prefix_matches_oracle"] is True
    # round 4: the mixed-shapes chunk is checked whole (hash of the fused output against the unfused kernels') + 4 000 rows against the oracle
    m = d["secondary"]["mixed_shapes"]
    assert m["fused_equals_unfused"] is True and m["oracle_sample_matches"] is True and m["oracle_sample_rows"] >= 4000 and len(m["fused_sha256"]) == 64
    # ... and the host-level call reports where its time goes
    for k in ("chunk_2MB", "chunk_28MB"):
        ph = d["secondary"]["host_level"][k]["phases_us_median"]
        assert set(ph) >= {"index", "copy_in", "chain", "download", "total"} and ph["total"] > 0
