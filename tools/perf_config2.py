#!/usr/bin/env python3
"""BASELINE configs[2] alone (bench.measure_config2): NDJSON lines -> events -> 16 Regex (OR) -> 16 Exclude (OR), the stage table and the
parity sample against the reference's filter_grep.  usage: perf_config2.py [lines] [nocpu]"""
import json, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import flbamd_loader
import bench as b
import torch

def main():
    g = flbamd_loader.load(); g.init(0)
    args = types.SimpleNamespace(ndjson_lines=int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000, no_cpu="nocpu" in sys.argv)
    out = b.measure_config2(g, torch, g.lib(), 0, 1, args)
    e = out["config2_ndjson_grep32"]
    print(json.dumps({"lines": e["lines"], "seconds_total": e["seconds_total"], "one_by_one": e["seconds_total_instances_called_one_by_one"], "whole_step_frac": e["roofline"]["frac"],
                      "stages": {k: {"ms_per_10M_lines": v["ms_per_10M_lines"], "frac": v["roofline"]["frac"], "kernel_ms": v.get("kernel_ms"), "kept": v.get("kept")}
                                 for k, v in e["stages"].items()},
                      "parity_sample": e.get("parity_sample")}))

if __name__ == "__main__":
    main()
