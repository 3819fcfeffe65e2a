#!/usr/bin/env python3
"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this GPU for the request shapes the filter kernels use
(csrc/calib.hip: kernels whose HBM byte count is known).  Run on the GPU box:

    python3 tools/calib_counters.py run            # (under rocprofv3 --pmc ... by tools/profile_bench.sh)
    python3 tools/calib_counters.py summarize <counter_collection.csv ...> > profiles/r2_counter_calibration.json
"""
import csv, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = 4 << 30                     # bytes per kernel: far beyond the L2s
MODES = {0: "coalesced16", 1: "column4", 2: "lane_line128", 3: "lane_sector16", 4: "write16"}


def run():
    import flbamd_loader
    g = flbamd_loader.load(); g.init(0); L = g.lib()
    L.flbgpu_calib_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int]
    a = L.flbgpu_dev_alloc(N); b = L.flbgpu_dev_alloc(N)
    assert a and b
    for m in MODES:
        for _ in range(2):
            assert L.flbgpu_calib_run(m, a, b, N, L.flbgpu_device_cus()) == 0


def summarize(files):
    acc = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "k_calib" not in k:
                continue
            mode = int(k.split("k_calib<")[1].split(">")[0])
            acc.setdefault((MODES[mode], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    out = {"bytes_per_kernel": N, "counters_KiB": {}, "fetch_factor": {}, "write_factor": {}, "note":
           "factor = known HBM bytes / (counter x 1024).  lane_sector16 reads 16 useful bytes per 128-byte line: its "
           "'fetched_bytes_per_request' is what one isolated 16-byte request pulls from HBM."}
    for (mode, ctr), v in sorted(acc.items()):
        out["counters_KiB"].setdefault(mode, {})[ctr] = sum(v) / len(v)
    for mode, c in out["counters_KiB"].items():
        if "FETCH_SIZE" in c and mode in ("coalesced16", "column4", "lane_line128"):
            out["fetch_factor"][mode] = round(N / (c["FETCH_SIZE"] * 1024), 4)
        if "FETCH_SIZE" in c and mode == "lane_sector16":
            out["fetched_bytes_per_request"] = round(c["FETCH_SIZE"] * 1024 / (N / 128), 2)
        if "WRITE_SIZE" in c and mode == "write16":
            out["write_factor"][mode] = round(N / (c["WRITE_SIZE"] * 1024), 4)
    # the shapes bench.py names
    ff = out["fetch_factor"]
    if "lane_line128" in ff:
        ff["per_lane64"] = ff["lane_line128"]; ff["per_lane16"] = ff["lane_line128"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        summarize(sys.argv[2:])
