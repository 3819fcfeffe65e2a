import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import flbamd_loader
from test_packfmt_oracle import load_kat
g = flbamd_loader.load(); g.init(0)
bad = 0
for i, (cfg, data, want) in enumerate(load_kat()):
    got = g.msgpack_to_json_format(data, cfg["json_format"], cfg["date_format"], cfg["date_key"], cfg["escape_unicode"], cfg["nan_to_null"])
    if got != want:
        bad += 1
        if bad <= 12:
            k = 0
            if got is not None and want is not None:
                while k < min(len(got), len(want)) and got[k] == want[k]: k += 1
            print(i, cfg, "len", None if got is None else len(got), None if want is None else len(want), "diff at", k)
            if got is not None and want is not None:
                print("  want", want[max(0, k - 40):k + 40]); print("  got ", got[max(0, k - 40):k + 40])
            else:
                print("  want", want if want is None else want[:100], "got", got if got is None else got[:100])
print("bad", bad)
