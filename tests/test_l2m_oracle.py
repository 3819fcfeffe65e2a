"""filter_log_to_metrics: the oracle restatement against the reference's own runtime expectations
(tests/runtime/filter_log_to_metrics.c) and the semantics read off log_to_metrics.c."""
import math
import pytest
import oracle_binding as ob
from synth import v2_record, legacy_record, mp, KV, Raw

K8S = {"container_name": "mycontainer", "namespace_name": "k8s-dummy", "docker_id": "abc123",
       "pod_name": "testpod", "pod_id": "def456"}


def msg(message, direction, duration="20"):
    # JSON_MSG1/2/3 of tests/runtime/filter_log_to_metrics.c:69-115 as in_lib encodes them
    return v2_record(1448403340, 0, {"message": message, "kubernetes": K8S, "duration": duration,
                                     "color": "red", "direction": direction})

MSG1, MSG2, MSG3 = msg("dummy", "right"), msg("dummy", "left"), msg("hello", "left")
K8S_VALUES = (b"k8s-dummy", b"testpod", b"mycontainer", b"abc123", b"def456")
LABELS = [("label_field", "color"), ("label_field", "direction")]


def test_counter_k8s():
    # flb_test_log_to_metrics_counter_k8s :232-307 -> "value":5.0,"labels":["k8s-dummy","testpod","mycontainer","abc123","def456","red","right"]
    f = ob.L2M("counter", LABELS, kubernetes_mode=True)
    for _ in range(5):
        assert f.filter(MSG1) == ob.NOTOUCH
    keys, _, s = f.snapshot()
    assert keys == ["namespace_name", "pod_name", "container_name", "docker_id", "pod_id", "color", "direction"]
    assert s == [dict(labels=K8S_VALUES + (b"red", b"right"), value=5.0, buckets=[0], count=0, sum=0.0)]


def test_counter():
    # :309-380 -> "value":5.0,"labels":["red","right"]
    f = ob.L2M("counter", LABELS)
    f.filter(MSG1 * 5)
    assert [(x["labels"], x["value"]) for x in f.snapshot()[2]] == [((b"red", b"right"), 5.0)]


def test_counter_two_tuples():
    # :382-462 -> 5.0 for (red,right), 3.0 for (red,left)
    f = ob.L2M("counter", LABELS, kubernetes_mode=True)
    f.filter(MSG1 * 5)
    f.filter(MSG2 * 3)
    got = {x["labels"][5:]: x["value"] for x in f.snapshot()[2]}
    assert got == {(b"red", b"right"): 5.0, (b"red", b"left"): 3.0}


def test_gauge():
    # :464-528 -> "value":20.0,"labels":["red","right"]
    f = ob.L2M("gauge", LABELS, value_field="duration")
    f.filter(MSG1)
    assert [(x["labels"], x["value"]) for x in f.snapshot()[2]] == [((b"red", b"right"), 20.0)]


def test_histogram():
    # :530-597 -> "buckets":[0,0,0,0,0,0,0,0,0,0,0,5],"sum":100.0,"count":5 with the default bounds
    f = ob.L2M("histogram", LABELS, value_field="duration")
    for _ in range(5):
        f.filter(MSG1)
    keys, bounds, s = f.snapshot()
    assert bounds == [0.005, 0.01, 0.025, 0.05, 0.1, 0.25, 0.5, 1.0, 2.5, 5.0, 10.0]
    assert s == [dict(labels=(b"red", b"right"), value=0.0, buckets=[0] * 11 + [5], count=5, sum=100.0)]


def test_regex_gate():
    # :599-668: regex "message .*el.*" (field WITHOUT '$': a plain top-level key) -> 3.0 for (red,left)
    f = ob.L2M("counter", LABELS + [("regex", "message .*el.*")])
    for _ in range(3):
        f.filter(MSG1)
        f.filter(MSG3)
    assert [(x["labels"], x["value"]) for x in f.snapshot()[2]] == [((b"red", b"left"), 3.0)]


def test_regex_no_labels():
    # :670-734 -> "value":3.0 on the static (label-less) series
    f = ob.L2M("counter", [("regex", "message .*el.*")])
    f.filter(MSG3 * 3 + MSG1)
    assert [(x["labels"], x["value"]) for x in f.snapshot()[2]] == [((), 3.0)]


def test_add_label():
    # :736-800: add_label "pod_name $kubernetes['pod_name']" -> label name pod_name, value testpod, 2.0
    f = ob.L2M("counter", [("add_label", "pod_name $kubernetes['pod_name']")])
    f.filter(MSG1 * 2)
    keys, _, s = f.snapshot()
    assert keys == ["pod_name"]
    assert [(x["labels"], x["value"]) for x in s] == [((b"testpod",), 2.0)]


def test_label_formats_and_truncation():
    # :1019-1035: STRING -> "%s" cut at MAX_LABEL_LENGTH-2 = 251 chars and at the first NUL;
    # FLOAT -> "%f"; INT -> "%ld" (u64 above INT64_MAX wraps); anything else -> ""
    f = ob.L2M("counter", [("label_field", "a")])
    recs = [
        {"a": "x" * 300}, {"a": b"ab\x00cd"}, {"a": 1.5}, {"a": -7}, {"a": 2 ** 64 - 1}, {"a": True}, {"a": None},
        {"a": {"m": 1}}, {"a": [1, 2]}, {"b": 1}, {"a": 1e300}, {"a": 0.0000004}, {"a": Raw(b"\xca\x3f\xc0\x00\x00")},
    ]
    f.filter(b"".join(v2_record(1, 0, r) for r in recs))
    got = [(x["labels"][0], x["value"]) for x in f.snapshot()[2]]
    assert got[0] == (b"x" * 251, 1.0)
    assert got[1] == (b"ab", 1.0)
    assert got[2] == (b"1.500000", 2.0)          # the f64 and the f32 1.5
    assert got[3] == (b"-7", 1.0)
    assert got[4] == (b"-1", 1.0)
    assert got[5] == (b"", 5.0)                  # bool, nil, map, array, missing
    assert got[6][0] == ("%f" % 1e300).encode()[:251] and len(got[6][0]) == 251
    assert got[7] == (b"0.000000", 1.0)


def test_value_parsing_and_stale_value():
    # :1060-1105: sscanf("%lf") leaves the previous value of THIS call in place when it fails;
    # non-numeric types and missing fields make no observation
    f = ob.L2M("histogram", [("bucket", "10"), ("bucket", "1"), ("bucket", "100")], value_field="v")
    recs = [{"v": "abc"}, {"v": "5"}, {"v": "abc"}, {"v": True}, {"x": 1}, {"v": 50}, {"v": 0.5}, {"v": " 7e1xyz"}, {"v": "0x"}]
    f.filter(b"".join(v2_record(1, 0, r) for r in recs))
    keys, bounds, s = f.snapshot()
    assert bounds == [1.0, 10.0, 100.0]
    # observed: 0 (stale initial), 5, 5 (stale), 50, 0.5, 70, 70 (stale)
    assert s[0]["count"] == 7 and s[0]["sum"] == 0 + 5 + 5 + 50 + 0.5 + 70 + 70
    assert s[0]["buckets"] == [2, 4, 7, 7]
    # a new call starts from 0 again
    f2 = ob.L2M("gauge", [], value_field="v")
    f2.filter(v2_record(1, 0, {"v": "9"}))
    f2.filter(v2_record(1, 0, {"v": "zzz"}))
    assert f2.snapshot()[2][0]["value"] == 0.0


def test_legacy_records_group_markers_and_garbage():
    # the callback walks raw msgpack objects: legacy [ts, map] works, group markers are NOT skipped,
    # non-array objects are skipped, decoding stops at the first malformed object
    f = ob.L2M("counter", [("label_field", "k")])
    data = (legacy_record(5, {"k": "a"}) + mp({"k": "zz"}) + mp(7) + v2_record(0xFFFFFFFF, 0, {"k": "g"}) +
            v2_record(1, 0, {"k": "a"}) + b"\xc1" + v2_record(1, 0, {"k": "a"}))
    f.filter(data)
    got = {x["labels"][0]: x["value"] for x in f.snapshot()[2]}
    assert got == {b"a": 2.0, b"g": 1.0}


def test_exclude_and_rule_order():
    f = ob.L2M("counter", [("exclude", "$log ^DEBUG"), ("regex", "$log error")])
    recs = [{"log": "DEBUG error"}, {"log": "an error"}, {"log": "fine"}, {"nolog": 1}]
    f.filter(b"".join(v2_record(1, 0, r) for r in recs))
    assert [x["value"] for x in f.snapshot()[2]] == [1.0]


def test_discard_logs_and_bad_config():
    assert ob.L2M("counter", [], discard_logs=True).filter(MSG1) == ob.MODIFIED
    with pytest.raises(ValueError):
        ob.L2M("summary")
    with pytest.raises(ValueError):
        ob.L2M("gauge")                       # value_field missing
    with pytest.raises(ValueError):
        ob.L2M("counter", [("regex", "onlyfield")])
