#!/usr/bin/env python3
"""Every regular expression the reference ships in conf/parsers*.conf, as a fixture the tests can read where /root/reference
does not exist (the GPU box): tests/golden/stock_parsers.json.  [PARSER] sections with `Format regex` give their `Regex` value
(the rest of the line, as the reference's config reader takes it: src/config_format/flb_cf_fluentbit.c, one key + one value per
line), their name and time settings; [MULTILINE_PARSER] `rule` lines give their quoted pattern.  tests/test_stock_parsers.py
re-parses the conf files when the reference is present and fails if this fixture is stale.

    python tools/gen_stock_parsers.py [/root/reference] > tests/golden/stock_parsers.json
"""
import glob, json, os, re, sys


def parse_conf(path):
    out = []
    sec = None
    cur = None
    def close():
        nonlocal cur
        if cur and cur.get("regex") is not None:
            out.append(cur)
        cur = None
    for ln, raw in enumerate(open(path, encoding="utf-8", errors="surrogateescape"), 1):
        line = raw.rstrip("\n").rstrip("\r")
        st = line.strip()
        if not st or st.startswith("#"):
            continue
        m = re.match(r"^\[(\w+)\]\s*$", st)
        if m:
            close()
            sec = m.group(1).upper()
            cur = {"file": os.path.basename(path), "line": ln, "section": sec, "name": None, "regex": None,
                   "time_key": None, "time_format": None, "types": None} if sec == "PARSER" else None
            continue
        parts = st.split(None, 1)
        key = parts[0].lower()
        val = parts[1].strip() if len(parts) > 1 else ""
        if sec == "PARSER" and cur is not None:
            if key == "name": cur["name"] = val
            elif key == "regex": cur["regex"] = val; cur["line"] = ln
            elif key == "time_key": cur["time_key"] = val
            elif key == "time_format": cur["time_format"] = val
            elif key == "types": cur["types"] = val
            elif key == "format": cur["format"] = val.lower()
        elif sec == "MULTILINE_PARSER":
            if key == "name":
                mlname = val
            if key == "rule":
                q = re.findall(r'"((?:[^"\\]|\\.)*)"', val)
                if len(q) >= 2:
                    out.append({"file": os.path.basename(path), "line": ln, "section": "MULTILINE_PARSER", "name": "%s:%s" % (mlname, q[0]),
                                "regex": q[1], "time_key": None, "time_format": None, "types": None})
    close()
    return out


def collect(ref):
    items = []
    for p in sorted(glob.glob(os.path.join(ref, "conf", "parsers*.conf"))):
        items += parse_conf(p)
    return items


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    json.dump(collect(ref), sys.stdout, indent=1, ensure_ascii=True)
    sys.stdout.write("\n")
