#!/usr/bin/env python3
"""Writes tests/golden/multiline_vectors.json from the reference's own multiline unit test (tests/internal/multiline.c): the
record_check arrays of the regex-rule cases (java, ruby, python, elastic, go) -- input lines and the concatenated records the test
expects under "log".  Run in the build container (needs /root/reference); the JSON travels with the repo."""
import json, os, re

SRC = "/root/reference/tests/internal/multiline.c"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def c_unescape(s):
    out, i = bytearray(), 0
    simple = {"n": 10, "t": 9, "r": 13, "\\": 92, '"': 34, "'": 39, "0": 0}
    while i < len(s):
        if s[i] == "\\":
            out.append(simple[s[i + 1]])
            i += 2
        else:
            out += s[i].encode()
            i += 1
    return out.decode("latin-1")


def entries(body):
    """top-level {...} items of an array initialiser, each a run of adjacent string literals"""
    items, depth, cur = [], 0, ""
    i = 0
    while i < len(body):
        ch = body[i]
        if ch == '"':
            j = i + 1
            while body[j] != '"':
                j += 2 if body[j] == "\\" else 1
            if depth:
                cur += c_unescape(body[i + 1:j])
            i = j + 1
            continue
        if ch == "/" and body[i + 1] == "*":
            i = body.index("*/", i) + 2
            continue
        if ch == "{":
            depth += 1
            cur = ""
        elif ch == "}":
            depth -= 1
            items.append(cur)
        i += 1
    return items


def main():
    text = open(SRC, encoding="latin-1").read()
    out = {}
    for name in ("java", "ruby", "python", "elastic", "go"):
        v = {}
        for side in ("input", "output"):
            m = re.search(r"struct record_check %s_%s\[\] = \{(.*?)\n\};" % (name, side), text, re.S)
            v[side] = entries(m.group(1))
        out[name] = v
    out["_source"] = "tests/internal/multiline.c (record_check arrays); elastic rules: test_parser_elastic :1028-1036"
    path = os.path.join(ROOT, "tests", "golden", "multiline_vectors.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(path, {k: (len(v["input"]), len(v["output"])) for k, v in out.items() if k[0] != "_"})


if __name__ == "__main__":
    main()
