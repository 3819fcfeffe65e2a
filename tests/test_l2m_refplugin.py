"""filter_log_to_metrics: the oracle restatement (oracle/oflb.c oflb_l2m_*) against the reference's OWN plugin -- plugins/
filter_log_to_metrics/log_to_metrics.c compiled in place over the real cmetrics, record accessor, regex and msgpack-c
(oracle/_ref/ref_filters kind 6; the emitter input and the flush timer are stubs, the filter callback and the metric state are real).
Same configuration, same chunks: return codes, label keys, bucket bounds, the series in list order with every value bit for bit
(histogram sums included: both add sequentially in f64)."""
import math
import random
import struct

import pytest

import oracle_binding as ob
import ref_filters as rf
from synth import v2_record, legacy_record, mp, Raw

pytestmark = pytest.mark.skipif(not rf.available(), reason="oracle/_ref/ref_filters not built")

K8S = {"container_name": "mycontainer", "namespace_name": "k8s-dummy", "docker_id": "abc123", "pod_name": "testpod", "pod_id": "def456"}
LABELS = [("label_field", "color"), ("label_field", "direction")]


def msg(message, direction, duration="20"):
    return v2_record(1448403340, 0, {"message": message, "kubernetes": K8S, "duration": duration, "color": "red", "direction": direction})


def bits(x):
    return struct.pack("<d", x)


def compare(cases):
    """cases: [(mode, props, chunks, kwargs)] -> runs them through the plugin in one process and through the oracle"""
    res = rf.run([rf.l2m_case(m, p, c, **kw) for m, p, c, kw in cases])
    n_series = 0
    for (m, p, chunks, kw), r in zip(cases, res):
        ref = rf.l2m_result(r)
        try:
            o = ob.L2M(m, p, kubernetes_mode=kw.get("kubernetes_mode", False), value_field=kw.get("value_field"), discard_logs=kw.get("discard_logs", False))
        except ValueError:
            assert ref is None, (m, p)
            continue
        assert ref is not None, (m, p, kw)
        rets = [o.filter(c) for c in chunks]
        assert rets == [x[0] for x in ref["rets"]] and all(x[1] == 0 for x in ref["rets"])
        keys, bounds, series = o.snapshot()
        assert keys == ref["keys"], (m, p)
        assert [bits(b) for b in bounds] == [bits(b) for b in ref["bounds"]]
        if not keys and not series:
            # cmt_map_create marks the static metric of a label-less map as set (lib/cmetrics/src/cmt_map.c:141-143): it exists, zeroed,
            # before any record; nothing was handed to the engine (flb_input_metrics_append only follows an update), so no output sees it
            assert ref["appended"] == 0 and len(ref["series"]) == 1
            z = ref["series"][0]
            assert z["value"] == 0.0 and z["count"] == 0 and z["sum"] == 0.0 and not any(z["buckets"])
            continue
        assert [s["labels"] for s in series] == [s["labels"] for s in ref["series"]], (m, p)
        for a, b in zip(series, ref["series"]):
            if m != "histogram":
                assert bits(a["value"]) == bits(b["value"]) or (math.isnan(a["value"]) and math.isnan(b["value"])), (m, p, a, b)
            else:
                assert a["buckets"] == b["buckets"] and a["count"] == b["count"], (m, p, a, b)
                assert bits(a["sum"]) == bits(b["sum"]) or (math.isnan(a["sum"]) and math.isnan(b["sum"])), (m, p, a, b)
        n_series += len(series)
    return n_series


def test_runtime_cases_of_the_reference():
    # tests/runtime/filter_log_to_metrics.c: counter_k8s :232, counter :309, two tuples :382, gauge :464, histogram :530, regex :599-, labels :760-
    m1, m2, m3 = msg("dummy", "right"), msg("dummy", "left"), msg("hello", "left")
    n = compare([
        ("counter", LABELS, [m1] * 5, dict(kubernetes_mode=True)),
        ("counter", LABELS, [m1 * 5], {}),
        ("counter", LABELS, [m1 * 5, m2 * 3], dict(kubernetes_mode=True)),
        ("gauge", LABELS, [m1], dict(value_field="duration")),
        ("histogram", LABELS, [m1] * 5, dict(value_field="duration")),
        ("counter", LABELS + [("regex", "message .*el.*")], [m1 + m2 + m3 * 3], {}),
        ("counter", [("regex", "message .*el.*")], [m1 + m3 * 2], {}),
        ("counter", [("add_label", "pod_name $kubernetes['pod_name']")], [m1 * 2], {}),
        ("counter", [], [m1], dict(discard_logs=True)),
        ("gauge", [], [m1], {}),                                         # value_field missing: cb_init refuses
        ("summary", [], [m1], {}),                                       # not a mode
        ("counter", [("regex", "onlykey")], [m1], {}),                   # a rule without a pattern
    ])
    assert n >= 9


def test_label_formats_and_value_parsing():
    recs = [{"a": "x" * 300}, {"a": b"ab\x00cd"}, {"a": 1.5}, {"a": -7}, {"a": 2 ** 64 - 1}, {"a": True}, {"a": None},
            {"a": {"m": 1}}, {"a": [1, 2]}, {"b": 1}, {"a": 1e300}, {"a": 0.0000004}, {"a": Raw(b"\xca\x3f\xc0\x00\x00")},
            {"a": -0.0}, {"a": float("inf")}, {"a": float("nan")}, {"a": 123456789.987654321}, {"a": 2 ** 63}, {"a": -2 ** 63},
            {"a": 0}, {"a": 10 ** 18}, {"a": 5e-324}, {"a": 0.9999995}, {"a": 0.5000005}, {"a": 1e22}, {"a": 255.0000005}]
    data = b"".join(v2_record(1, 0, r) for r in recs)
    vals = ["5", "abc", " 7e1xyz", "0x", "0x1p4", "1e400", "-3.25", True, None, 50, -2, 0.5, 1e-3, "", "inf", "nanx", "infinit",
            "12345678901234567890123", "0.1", "1e22", "9007199254740993", "4.9e-324", ".5", "+.e1", Raw(b"\xca\x41\x20\x00\x00")]
    rng = random.Random(3)
    chunks = []
    for c in range(3):
        out = []
        for i in range(1500):
            body = {"k": rng.choice(["a", "b", "c"])}
            if rng.random() < 0.95:
                body["v"] = rng.choice(vals)
            out.append(v2_record(1, 0, body))
        chunks.append(b"".join(out))
    c0 = b"".join(v2_record(1, 0, {"k": "a", "v": "zzz"}) for _ in range(10)) + v2_record(1, 0, {"k": "a", "v": "4"}) + v2_record(1, 0, {"k": "b", "v": "zzz"})
    buckets = [("bucket", "10"), ("bucket", "0.5"), ("bucket", "100"), ("bucket", "-1"), ("bucket", "1e21")]
    n = compare([
        ("counter", [("label_field", "a")], [data], {}),
        ("histogram", [("label_field", "k")] + buckets, chunks, dict(value_field="v")),
        ("gauge", [("label_field", "k")], chunks, dict(value_field="v")),
        ("histogram", [("label_field", "k")], [c0, c0], dict(value_field="v")),
        ("gauge", [("label_field", "k")], [c0, c0], dict(value_field="v")),
        ("histogram", [("bucket", "abc")], [c0], dict(value_field="v")),          # not a number: set_buckets fails
        ("histogram", [("bucket", "1"), ("bucket", "1"), ("bucket", "-0")], [chunks[0]], dict(value_field="v")),
    ])
    assert n >= 25


def rand_label(rng):
    t = rng.randrange(10)
    if t < 5:
        return rng.choice(["GET", "POST", "PUT", "a", "", "x" * rng.randrange(1, 40), "é", "200", "404"])
    if t == 5:
        return rng.randrange(-1000, 1000)
    if t == 6:
        return rng.choice([0.5, 2.25, -1.0, 1e6])
    if t == 7:
        return rng.choice([True, None])
    if t == 8:
        return {"in": rng.choice(["p", "q"]), "arr": [1, "z", {"k": "deep"}]}
    return rng.choice([b"\xff\xfe", "tab\there"])


def test_random_records_rules_and_accessors():
    rng = random.Random(11)
    chunks = []
    for c in range(4):
        recs = []
        for i in range(2000):
            body = {}
            if rng.random() < 0.9:
                body["m"] = rand_label(rng)
            if rng.random() < 0.8:
                body["code"] = rand_label(rng)
            if rng.random() < 0.7:
                body["nest"] = {"in": rng.choice(["p", "q", 3]), "arr": [1, rng.choice(["z", "y"]), {"k": "deep"}]}
            body["log"] = rng.choice(["an error here", "DEBUG noise", "fine", "DEBUG error", "érror"])
            if rng.random() < 0.5:
                body["v"] = rng.choice([1, 2.5, "3", "x", -4, 1e3, "0.25"])
            r = rng.random()
            if r < 0.1:
                recs.append(legacy_record(rng.randrange(1, 2 ** 31), body))
            elif r < 0.13:
                recs.append(v2_record(0xFFFFFFFF, 0, body))              # group marker: processed like any record
            elif r < 0.15:
                recs.append(mp(rng.choice([1, "str", {"k": 1}])))         # not an array: skipped
            else:
                recs.append(v2_record(rng.randrange(1, 2 ** 31), rng.randrange(10 ** 9), body))
        chunks.append(b"".join(recs))
    props = [("exclude", "$log ^DEBUG"), ("label_field", "m"), ("add_label", "c $code"), ("add_label", "deep $nest['arr'][2]['k']"),
             ("add_label", "in $nest['in']"), ("label_field", "$TAG"), ("regex", "log err|fine|^.rror")]
    cut = chunks[0][:len(chunks[0]) - 5]                                  # msgpack_unpack_next stops inside the last record
    n = compare([
        ("counter", props, chunks, {}),
        ("histogram", props[:3] + [("bucket", "2"), ("bucket", "100")], chunks, dict(value_field="v")),
        ("gauge", [("label_field", "m"), ("regex", "log e"), ("regex", "m ^[A-Z]")], chunks, dict(value_field="$nest['in']")),
        ("counter", [("label_field", "code")], [cut, chunks[1]], dict(kubernetes_mode=True)),
        ("counter", [("exclude", "log error"), ("regex", "log error")], chunks[:1], dict(discard_logs=True)),
    ])
    assert n > 300
