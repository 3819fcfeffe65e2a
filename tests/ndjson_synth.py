"""BASELINE configs[2] workload: NDJSON lines of a service log and the 32-pattern filter_grep set.

The rule sets are two filter_grep instances (Logical_Op OR needs one rule type per instance, plugins/filter_grep/grep.c:90-98):
16 `Regex` rules -- a record is kept when ANY matches -- then 16 `Exclude` rules -- a record is dropped when ANY matches;
literal-heavy with four class / quantifier patterns in each set.  On these lines the first instance keeps about half of
the records and the second about two thirds of those (tests/test_ndjson_synth.py pins the ratios with the oracle), so
neither instance is a NOTOUCH best case."""
import json
import random

GREP32_REGEX = [("regex", r) for r in (
    "level ^(error|fatal)$", "msg timeout", "msg refused", "$svc['name'] ^db$", "path ^/v1/items/1", "msg request 9", "msg evicted", "path x=7$",
    "msg upstream=auth", "$svc['name'] ^cache-[0-3]$", "path /items/9[0-9]{4}[?]", r"msg ^gc pause \d{5,} us", "level ^w", "path [?]x=[0-9]$",
    "msg closed after 9", r"path ^/v2/\w+/\d*0[?]")]
GREP32_EXCLUDE = [("exclude", r) for r in (
    "msg request 1", "level ^warn$", "path x=1$", "$svc['pod'] ^pod-1", "msg 77", "path /items/4", "level nothing", "msg never",
    "path ^/v3", "$svc['name'] ^$", r"path x=9\d$", "msg [5-6]{3} finished", r"level ^\s", "path items/[1-2]{2}", "msg heap=1", "$svc['name'] b$")]

_LEVELS = ["info"] * 60 + ["debug"] * 22 + ["warn"] * 10 + ["error"] * 7 + ["fatal"]
_SVC = ["api"] * 5 + ["db"] * 1 + ["cache-%d" % i for i in range(8)] + ["auth", "billing", "search", "web"]


def line(rng):
    k = rng.randrange(100)
    if k < 45:
        msg = "request %d finished %s" % (rng.randrange(10 ** 6), rng.choice(["ok"] * 10 + ["timeout", "refused"]))
    elif k < 65:
        msg = "connection from 10.%d.%d.%d closed after %d ms" % (rng.randrange(256), rng.randrange(256), rng.randrange(256), rng.randrange(1, 20000))
    elif k < 80:
        msg = "cache %s key=user:%d" % (rng.choice(["hit", "hit", "hit", "miss", "evicted"]), rng.randrange(10 ** 7))
    elif k < 90:
        msg = "gc pause %d us heap=%dMB" % (int(10 ** (2 + 3.2 * rng.random())), rng.randrange(64, 4096))
    else:
        msg = "retry %d/%d upstream=%s" % (rng.randrange(1, 4), 3, rng.choice(["auth", "billing", "search", "db"]))
    d = {"time": "2026-09-21T10:%02d:%02d.%03dZ" % (rng.randrange(60), rng.randrange(60), rng.randrange(1000)),
         "level": rng.choice(_LEVELS), "msg": msg,
         "code": rng.randrange(200, 600), "latency": round(rng.random() * 100, 3),
         "svc": {"name": rng.choice(_SVC), "pod": "pod-%d" % rng.randrange(1000)},
         "path": "/v%d/items/%d?x=%d" % (rng.choice([1, 1, 1, 2]), rng.randrange(10 ** 5), rng.randrange(100)), "bytes": rng.randrange(10 ** 6)}
    return json.dumps(d).encode() + b"\n"


def lines(n, seed=7):
    """n distinct seeded lines (one bytes object each)"""
    rng = random.Random(seed)
    return [line(rng) for _ in range(n)]
