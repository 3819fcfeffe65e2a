"""The drop-in boundary as the engine sees it: flb-filter_<x>_gpu.so files (fluent-bit_amd/plugin/build.sh, compiled
against the reference's headers) are dlopen'd by a minimal host that restates src/flb_plugin.c:194-320's loading and
src/flb_filter.c's cb_init / cb_filter / cb_exit sequence (tests/plugin_host.c), and their output is the oracle's.  The same objects inside the
reference's own engine: tests/test_engine.py (oracle/engine/engine_host.c)."""
import os, re, subprocess, tempfile
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PDIR = os.path.join(ROOT, "fluent-bit_amd", "plugin")
B = os.path.join(PDIR, "_build")
HOST = os.path.join(B, "plugin_host")
APACHE2 = r'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>.*)")?$'
TF = "%d/%b/%Y:%H:%M:%S %z"


def _built():
    if os.path.isdir("/root/reference/include/fluent-bit"):
        subprocess.run(["bash", os.path.join(PDIR, "build.sh")], check=True, capture_output=True)
    return os.path.exists(HOST)


def _host(*args):
    return subprocess.run([HOST] + list(args), capture_output=True, text=True, timeout=300)


def test_plugins_load_like_flb_plugin_load():
    """file name flb-filter_<x>.so, data symbol filter_<x>_plugin, callbacks set, property names = the reference's"""
    if not _built():
        pytest.skip("plugins not built (needs the reference headers once)")
    want = {"grep": ["regex", "exclude", "logical_op"],
            "parser": ["Key_Name", "Parser", "Preserve_Key", "Reserve_Data", "Unescape_key"],
            "log_to_metrics": ["regex", "exclude", "metric_mode", "value_field", "metric_name", "metric_namespace", "metric_subsystem",
                               "metric_description", "kubernetes_mode", "add_label", "label_field", "bucket", "tag", "emitter_name",
                               "emitter_mem_buf_limit", "flush_interval_sec", "flush_interval_nsec", "discard_logs", "sum_order"]}
    for x, props in want.items():
        so = os.path.join(B, "flb-filter_%s_gpu.so" % x)
        r = _host(so, "filter_%s_gpu_plugin" % x, "inspect")
        assert r.returncode == 0, r.stderr
        assert "name=%s_gpu" % x in r.stdout and "cb_init=1 cb_filter=1 cb_exit=1" in r.stdout
        got = re.findall(r"^config_map (\S+) ", r.stdout, re.M)
        assert got == props, (x, got)
        # the reference's own property names (when its sources are here): every one of ours exists there
        src = {"grep": "plugins/filter_grep/grep.c", "parser": "plugins/filter_parser/filter_parser.c",
               "log_to_metrics": "plugins/filter_log_to_metrics/log_to_metrics.c"}[x]
        path = os.path.join("/root/reference", src)
        if os.path.exists(path):
            text = open(path).read()
            ref = set(m.lower() for m in re.findall(r'FLB_CONFIG_MAP_\w+,\s*"([^"]+)"', text))
            # (sum_order is this plugin's own: the reference's order by default)
            assert set(p.lower() for p in props) - {"sum_order"} <= ref, (x, set(p.lower() for p in props) - ref)


def test_cb_init_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    if not _built():
        pytest.skip("plugins not built")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "in.mp"), "wb").close()
        r = _host(os.path.join(B, "flb-filter_grep_gpu.so"), "filter_grep_gpu_plugin", "run", os.path.join(d, "in.mp"), os.path.join(d, "out.mp"), "regex=log x")
        assert r.returncode == 10 and "cb_init=-1" in r.stdout and "no CPU path" in r.stderr     # as the reference aborts startup on cb_init -1


@pytest.mark.gpu
def test_plugins_run_through_the_host_match_the_oracle():
    import oracle_binding as ob
    import synth
    if not os.path.exists(HOST):
        pytest.skip("plugins not built (fluent-bit_amd/plugin/build.sh needs the reference headers)")
    data, off, ep = synth.apache_records(3000)
    blob = bytes(data)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.mp"), os.path.join(d, "out.mp")
        open(fin, "wb").write(blob)
        # filter_parser_gpu: Key_Name log, Parser apache2 (the registry entry = conf/parsers.conf:8-13)
        r = _host(os.path.join(B, "flb-filter_parser_gpu.so"), "filter_parser_gpu_plugin", "run", fin, fout, "Key_Name=log", "Parser=apache2",
                  "--parser", "apache2|%s|%s|time" % (APACHE2, TF))
        assert r.returncode == 0 and "cb_init=0" in r.stdout and "cb_filter=1" in r.stdout and "cb_exit=0" in r.stdout, r.stdout + r.stderr
        po = ob.Parser(APACHE2, time_fmt=TF, time_key="time")
        q, want = ob.FilterParser("log", [po]).filter(blob)
        parsed = open(fout, "rb").read()
        assert q == ob.MODIFIED and parsed == want
        # Reserve_Data / Preserve_Key through the config map offsets
        r = _host(os.path.join(B, "flb-filter_parser_gpu.so"), "filter_parser_gpu_plugin", "run", fin, fout, "Key_Name=log", "Parser=apache2",
                  "Reserve_Data=On", "Preserve_Key=true", "--parser", "apache2|%s|%s|time" % (APACHE2, TF))
        assert r.returncode == 0, r.stdout + r.stderr
        assert open(fout, "rb").read() == ob.FilterParser("log", [po], True, True).filter(blob)[1]
        # filter_grep_gpu on the parsed chunk: repeated properties in configuration order, logical_op
        open(fin, "wb").write(parsed)
        for props, rules, op in (([r"regex=code ^5\d\d$"], [("regex", r"code ^5\d\d$")], None),
                                 (["regex=code ^2", "regex=agent curl", "logical_op=AND"], [("regex", "code ^2"), ("regex", "agent curl")], "AND"),
                                 (["exclude=method GET", "regex=code ^4"], [("exclude", "method GET"), ("regex", "code ^4")], None),
                                 (["regex=host ."], [("regex", "host .")], None)):
            r = _host(os.path.join(B, "flb-filter_grep_gpu.so"), "filter_grep_gpu_plugin", "run", fin, fout, *props)
            assert r.returncode == 0 and "cb_init=0" in r.stdout, r.stdout + r.stderr
            q, want = ob.Grep(rules, op).filter(parsed)
            assert ("cb_filter=%d " % q) in r.stdout, (props, r.stdout)
            if q == ob.MODIFIED:
                assert open(fout, "rb").read() == want, props
        # flb_parser_do's signature on the GPU path
        line = blob[int(off[7]) + 21:int(off[8])]
        open(fin, "wb").write(line)
        r = _host(os.path.join(B, "flb-filter_parser_gpu.so"), "filter_parser_gpu_plugin", "parser_do", fin, fout, "--parser", "apache2|%s|%s|time" % (APACHE2, TF))
        ret, out, tm = po.do(line)
        m = re.search(r"flb_parser_do=(-?\d+) out_size=(\d+) sec=(\d+) nsec=(\d+)", r.stdout)
        assert m, r.stdout + r.stderr
        assert (int(m.group(1)), int(m.group(3)), int(m.group(4))) == (ret, tm[0], tm[1]) and open(fout, "rb").read() == out


@pytest.mark.gpu
def test_log_to_metrics_plugin_runs_through_the_host():
    """flb-filter_log_to_metrics_gpu.so executed: cb_init (cmetrics context, the hidden emitter, the flush timer), cb_filter,
    l2m_gpu_publish -> the struct cmt handed to the emitter's flb_input_metrics_append (the host prints it: the real cmetrics,
    oracle/_ref/libcmetrics_ref.so, holds the series) is the oracle's state; timer mode publishes on the tick, once."""
    import oracle_binding as ob
    import synth
    if not os.path.exists(HOST) or "timer created" not in open(HOST, "rb").read().decode("latin1"):
        pytest.skip("plugin_host built without cmetrics (oracle/_ref/libcmetrics_ref.so was absent at build time)")
    data, off, ep = synth.apache_records(3000)
    po = ob.Parser(APACHE2, time_fmt=TF, time_key="time")
    q, parsed = ob.FilterParser("log", [po]).filter(bytes(data))
    so = os.path.join(B, "flb-filter_log_to_metrics_gpu.so")

    def series(text, want_appends):
        blocks = text.split("metrics_append ")[1:]
        assert len(blocks) == want_appends, text
        out = []
        for ln in blocks[-1].splitlines():
            m = re.match(r"\s+series((?: \[[^\]]*\])*)(.*)$", ln)
            if m:
                labels = tuple(x.encode() for x in re.findall(r"\[([^\]]*)\]", m.group(1)))
                kv = dict(p.split("=") for p in m.group(2).split())
                out.append((labels, kv))
        return out

    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.mp"), os.path.join(d, "out.mp")
        open(fin, "wb").write(parsed)
        # counter, two labels, published after the call (no flush interval)
        r = _host(so, "filter_log_to_metrics_gpu_plugin", "run", fin, fout, "metric_mode=counter", "metric_name=requests", "metric_description=n",
                  "tag=metrics", "label_field=method", "label_field=code")
        assert r.returncode == 0 and "cb_init=0" in r.stdout and "cb_filter=2" in r.stdout and "cb_exit=0" in r.stdout, r.stdout + r.stderr
        assert "emitter property alias=emitter_for_log_to_metrics_gpu.0" in r.stdout and "tag=metrics" in r.stdout
        assert "counter log_metric_counter_requests labels=2" in r.stdout
        o = ob.L2M("counter", [("label_field", "method"), ("label_field", "code")])
        assert o.filter(parsed) == ob.NOTOUCH
        want = o.snapshot()[2]
        got = series(r.stdout, 1)
        assert [g[0] for g in got] == [w["labels"] for w in want]
        assert [float(g[1]["value"]) for g in got] == [w["value"] for w in want]
        # histogram of `size` with explicit buckets, timer mode: nothing at cb_filter, one append on the first tick, none on the second
        props = ["metric_mode=histogram", "metric_name=size", "metric_description=b", "tag=m2", "value_field=size", "label_field=code",
                 "bucket=100", "bucket=10000", "bucket=1000000", "flush_interval_sec=1"]
        r = _host(so, "filter_log_to_metrics_gpu_plugin", "run", fin, fout, *props)
        assert r.returncode == 0 and "timer created ms=1000" in r.stdout and "cb_filter=2" in r.stdout, r.stdout + r.stderr
        assert "bounds 100 10000 1000000" in r.stdout
        o = ob.L2M("histogram", [("label_field", "code"), ("bucket", "100"), ("bucket", "10000"), ("bucket", "1000000")], value_field="size")
        o.filter(parsed)
        want = o.snapshot()[2]
        got = series(r.stdout, 1)
        assert [g[0] for g in got] == [w["labels"] for w in want]
        for g, w in zip(got, want):
            assert [int(g[1]["b%d" % b]) for b in range(4)] == list(w["buckets"]) and int(g[1]["count"]) == w["count"] and float(g[1]["sum"]) == w["sum"]
        # discard_logs: MODIFIED with an empty output (log_to_metrics.c:1143-1150)
        r = _host(so, "filter_log_to_metrics_gpu_plugin", "run", fin, fout, "metric_mode=counter", "metric_name=n", "metric_description=n", "tag=t", "discard_logs=true")
        assert r.returncode == 0 and "cb_filter=1 out_size=0" in r.stdout, r.stdout + r.stderr
        # a rule that is not a regular expression (a look-ahead): cb_init does not fail, the rule runs on the host
        r = _host(so, "filter_log_to_metrics_gpu_plugin", "run", fin, fout, "metric_mode=counter", "metric_name=errs", "metric_description=n",
                  "tag=metrics", "regex=code ^(?=5)\\d+$", "label_field=method")
        assert r.returncode == 0 and "cb_init=0" in r.stdout and "cb_filter=2" in r.stdout, r.stdout + r.stderr
        o = ob.L2M("counter", [("regex", "code ^5\\d*$"), ("label_field", "method")])
        assert o.filter(parsed) == ob.NOTOUCH
        want = o.snapshot()[2]
        got = series(r.stdout, 1)
        assert want and [g[0] for g in got] == [w["labels"] for w in want]
        assert [float(g[1]["value"]) for g in got] == [w["value"] for w in want]
