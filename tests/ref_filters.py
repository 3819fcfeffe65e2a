"""Binding of oracle/_ref/ref_filters: the REAL cb_filter of the reference's filter_grep / filter_parser plugins (built by
oracle/Makefile from /root/reference, see oracle/ref_filters_shim.c).  Test infrastructure only."""
import os, struct, subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_filters")
MODIFIED, NOTOUCH = 1, 2


def available():
    return os.path.exists(EXE)


def _s(x):
    b = x if isinstance(x, bytes) else str(x).encode()
    return struct.pack("<I", len(b)) + b


def _parser_fields(name, p):
    """p: the keyword arguments of oracle_binding.Parser / flbgpu Parser"""
    fmt = p.get("format") or "regex"
    return [name, fmt, p.get("regex") or "", p.get("time_fmt") or "", p.get("time_key") or "", p.get("time_offset") or "", p.get("types") or "",
            "1" if p.get("skip_empty", True) else "0", "%d %d" % (1 if p.get("time_keep") else 0, 1 if p.get("time_strict", True) else 0) +
            (" tz=%s" % p["time_zone"] if p.get("time_zone") else "") + (" systz" if p.get("time_system_timezone") else "") +
            "".join("|%s %s %s%s" % ("decode_field_as" if d[0] else "decode_field", d[1], d[2], " " + d[3] if len(d) > 3 and d[3] else "")
                    for d in (p.get("decoders") or []))]


def _case(kind, props, parsers, data, iters=None):
    out = struct.pack("<I", kind)
    if kind == 3:
        out += struct.pack("<I", iters)
    out += struct.pack("<I", len(props))
    for k, v in props:
        out += _s(k) + _s(v)
    out += struct.pack("<I", len(parsers))
    for name, p in parsers:
        for f in _parser_fields(name, p):
            out += _s(f)
    return out + struct.pack("<Q", len(data)) + data


def run(cases, timeout=600):
    """cases: list of bytes built by grep_case / parser_case / bench_case; returns [(ret, out bytes)]"""
    r = subprocess.run([EXE], input=b"".join(cases), capture_output=True, timeout=timeout)
    res, o, buf = [], 0, r.stdout
    while o + 12 <= len(buf):
        ret, n = struct.unpack_from("<iQ", buf, o)
        o += 12
        res.append((ret, buf[o:o + n]))
        o += n
    assert len(res) == len(cases), (len(res), len(cases), r.returncode, r.stderr[-400:])
    return res


def grep_case(rules, logical_op, data):
    props = [(k, v) for k, v in rules]
    if logical_op:
        props.append(("logical_op", logical_op))
    return _case(1, props, [], data)


def parser_case(key_name, parsers, data, reserve=False, preserve=False):
    named = [("p%d" % i, p) for i, p in enumerate(parsers)]
    props = [("key_name", key_name)] + [("parser", n) for n, _ in named]
    if reserve:
        props.append(("reserve_data", "on"))
    if preserve:
        props.append(("preserve_key", "on"))
    return _case(2, props, named, data)


def bench_pair_case(key_name, parser, rules, data, iters):
    props = [("key_name", key_name), ("parser", "p0"), ("--", "")] + [(k, v) for k, v in rules]
    return _case(3, props, [("p0", parser)], data, iters)


def bench_result(res, with_output=False):
    """-> (seconds, records in, records kept) [+ the pair's output bytes of the first pass]"""
    ret, out = res
    assert ret == 0 and len(out) >= 24, (ret, len(out))
    secs, rin, rkept = struct.unpack("<dQQ", out[:24])
    return (secs, rin, rkept, out[24:]) if with_output else (secs, rin, rkept)


def tail_case(text, key="log", path_key=None, path="", offset_key=None, stream_offset=0, skip_empty_lines=True, sec=0, nsec=0):
    props = [("key", key), ("skip_empty_lines", "1" if skip_empty_lines else "0"), ("sec", sec), ("nsec", nsec), ("stream_offset", stream_offset), ("path", path)]
    if path_key is not None:
        props.append(("path_key", path_key))
    if offset_key is not None:
        props.append(("offset_key", offset_key))
    return _case(4, props, [], text)


def tail_result(res):
    lines, out = res
    (processed,) = struct.unpack("<Q", out[-8:])
    return lines, out[:-8], processed


def ml_case(frames, rules=None, builtin=None, type="regex", match_string=None, negate=False, key_content=None, buffer_limit=None,
            skip_empty_lines=False, final_flush=False, now=(1600000000, 77)):
    """the multiline core (src/multiline/*.c) behind in_tail's line loop.  frames: [(sec, nsec, text)] -- what successive reads append to
    the file's buffer; rules: [(from_states, regex, to_state)]"""
    props = [("type", type), ("negate", 1 if negate else 0), ("skip_empty_lines", 1 if skip_empty_lines else 0), ("final_flush", 1 if final_flush else 0)]
    props += [("now_sec", now[0]), ("now_nsec", now[1])]
    if builtin:
        props.append(("builtin", builtin))
    if match_string is not None:
        props.append(("match_string", match_string))
    if key_content:
        props.append(("key_content", key_content))
    if buffer_limit is not None:
        props.append(("buffer_limit", buffer_limit))
    for fs, rx, to in (rules or []):
        props.append(("rule", b"\x1f".join(x if isinstance(x, bytes) else x.encode() for x in (fs, rx, to or ""))))
    data = b"".join(struct.pack("<III", s, ns, len(t)) + t for s, ns, t in frames)
    return _case(5, props, [], data)


def l2m_case(mode, props, chunks, kubernetes_mode=False, value_field=None, discard_logs=False, extra=()):
    """filter_log_to_metrics: the reference's own plugin (plugins/filter_log_to_metrics/log_to_metrics.c over the real cmetrics).
    `props` as oracle_binding.L2M takes them (regex / exclude / label_field / add_label / bucket in configuration order); one
    cb_filter call per chunk on one instance"""
    p = [("metric_mode", mode), ("metric_name", "m"), ("metric_description", "d"), ("tag", "metrics")] + [(k, v) for k, v in props]
    if kubernetes_mode:
        p.append(("kubernetes_mode", "true"))
    if value_field is not None:
        p.append(("value_field", value_field))
    if discard_logs:
        p.append(("discard_logs", "true"))
    p += list(extra)
    data = b"".join(struct.pack("<Q", len(c)) + bytes(c) for c in chunks)
    return _case(6, p, [], data)


def l2m_result(res):
    """-> None when cb_init refused, else dict(rets=[(ret, out_size)], mode, timer_mode, appended, keys, bounds, series) with series as
    oracle_binding.L2M.snapshot() lists them"""
    ret, b = res
    if ret == -100:
        return None
    o = 0
    def u32():
        nonlocal o
        v = struct.unpack_from("<I", b, o)[0]; o += 4
        return v
    def s():
        nonlocal o
        n = u32()
        v = b[o:o + n]; o += n
        return v
    nch = u32()
    rets = []
    for _ in range(nch):
        r, sz = struct.unpack_from("<iQ", b, o); o += 12
        rets.append((r, sz))
    mode, timer_mode, appended = u32(), u32(), u32()
    keys = [s().decode("latin1") for _ in range(u32())]
    nb = u32()
    bounds = list(struct.unpack_from("<%dd" % nb, b, o)); o += 8 * nb
    series = []
    for _ in range(u32()):
        labels = tuple(s() for _ in keys)
        if mode != 2:
            v = struct.unpack_from("<d", b, o)[0]; o += 8
            series.append(dict(labels=labels, value=v, buckets=[0], count=0, sum=0.0))
        else:
            bk = list(struct.unpack_from("<%dQ" % (nb + 1), b, o)); o += 8 * (nb + 1)
            cnt, sm = struct.unpack_from("<Qd", b, o); o += 16
            series.append(dict(labels=labels, value=0.0, buckets=bk, count=cnt, sum=sm))
    assert o == len(b), (o, len(b))
    return dict(rets=rets, mode=mode, timer_mode=timer_mode, appended=appended, keys=keys, bounds=bounds, series=series)
