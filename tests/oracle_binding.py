"""ctypes binding of the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes, os
from ctypes import c_char_p, c_int, c_void_p, c_size_t, c_int64, c_double, POINTER, byref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODIFIED, NOTOUCH = 1, 2

_L = None

def lib():
    global _L
    if _L is None:
        L = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        L.oflb_parser_create.restype = c_void_p
        L.oflb_parser_create.argtypes = [c_char_p, c_int, c_char_p, c_char_p, c_char_p, c_int, c_int, c_char_p]
        L.oflb_parser_create_kv.restype = c_void_p
        L.oflb_parser_create_kv.argtypes = [c_int, c_char_p, c_char_p, c_char_p, c_int, c_int, c_int]
        L.oflb_unescape_utf8.argtypes = [c_char_p, c_int, c_char_p]
        L.oflb_parser_destroy.argtypes = [c_void_p]
        L.oflb_parser_do.argtypes = [c_void_p, c_char_p, c_size_t, POINTER(c_void_p), POINTER(c_size_t),
                                     POINTER(c_int64), POINTER(c_int64)]
        L.oflb_grep_create.restype = c_void_p
        L.oflb_grep_create.argtypes = [c_int, POINTER(c_char_p), POINTER(c_char_p), c_char_p]
        L.oflb_grep_destroy.argtypes = [c_void_p]
        L.oflb_grep_filter.argtypes = [c_void_p, c_char_p, c_size_t, POINTER(c_void_p), POINTER(c_size_t)]
        L.oflb_fparser_create.restype = c_void_p
        L.oflb_fparser_create.argtypes = [c_char_p, c_int, c_int, c_int, POINTER(c_void_p)]
        L.oflb_fparser_destroy.argtypes = [c_void_p]
        L.oflb_fparser_filter.argtypes = [c_void_p, c_char_p, c_size_t, POINTER(c_void_p), POINTER(c_size_t)]
        L.oflb_free.argtypes = [c_void_p]
        L.oflb_count_records.argtypes = [c_char_p, c_size_t]
        L.oflb_repack.argtypes = [c_char_p, c_size_t, POINTER(c_void_p), POINTER(c_size_t)]
        L.oflb_bench_fparser.restype = c_double
        L.oflb_bench_fparser.argtypes = [c_void_p, c_char_p, c_size_t, c_int, POINTER(c_size_t)]
        L.oflb_bench_grep.restype = c_double
        L.oflb_bench_grep.argtypes = [c_void_p, c_char_p, c_size_t, c_int, POINTER(c_size_t)]
        L.oflb_regex_create.restype = c_void_p
        L.oflb_regex_create.argtypes = [c_char_p]
        L.oflb_regex_match.argtypes = [c_void_p, c_char_p, c_size_t]
        L.oflb_time_lookup.argtypes = [c_void_p, c_char_p, c_size_t, c_int64, c_int, POINTER(c_int64), POINTER(c_double)]
        L.oflb_l2m_create.restype = c_void_p
        L.oflb_l2m_create.argtypes = [c_char_p, c_int, POINTER(c_char_p), POINTER(c_char_p), c_int, c_char_p, c_int]
        L.oflb_l2m_destroy.argtypes = [c_void_p]
        L.oflb_l2m_filter.argtypes = [c_void_p, c_char_p, c_size_t]
        L.oflb_l2m_info.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_double)]
        L.oflb_l2m_label_key.restype = c_char_p
        L.oflb_l2m_label_key.argtypes = [c_void_p, c_int]
        L.oflb_l2m_series_label.restype = c_char_p
        L.oflb_l2m_series_label.argtypes = [c_void_p, c_int, c_int]
        L.oflb_l2m_series_get.argtypes = [c_void_p, c_int, POINTER(c_double), POINTER(ctypes.c_uint64),
                                          POINTER(ctypes.c_uint64), POINTER(c_double)]
        _L = L
    return _L

def _take(ptr, size):
    data = ctypes.string_at(ptr, size.value) if ptr.value else b""
    if ptr.value:
        lib().oflb_free(ptr)
    return data

def set_time_now(now):
    """pins the time(NULL) of year-less Time_Formats (0: the wall clock again)"""
    lib().oflb_set_time_now(c_int64(int(now)))


class Parser:
    """mirrors flb_parser_create(name, "regex", regex, skip_empty, time_fmt, time_key, time_offset,
    time_keep, time_strict, ..., types) -- include/fluent-bit/flb_parser.h:99-110.
    Defaults are the conf-file defaults (src/flb_parser.c:1277-1304)."""
    def __init__(self, regex=None, time_fmt=None, time_key=None, time_offset=None, time_keep=False,
                 time_strict=True, skip_empty=True, types=None, format="regex", no_bare_keys=False, decoders=None,
                 time_zone=None, time_system_timezone=False):
        e = lambda s: s.encode() if isinstance(s, str) else s
        if format == "json":
            regex = None                      # Format json (src/flb_parser_json.c)
        if format in ("logfmt", "ltsv"):      # src/flb_parser_logfmt.c, src/flb_parser_ltsv.c
            lib().oflb_parser_create_kv2.restype = c_void_p
            lib().oflb_parser_create_kv2.argtypes = [c_int, c_char_p, c_char_p, c_char_p, c_int, c_int, c_int, c_char_p]
            self.h = lib().oflb_parser_create_kv2(1 if format == "logfmt" else 2, e(time_fmt), e(time_key), e(time_offset),
                                                  int(time_keep), int(time_strict), int(no_bare_keys), e(types))
        else:
            self.h = lib().oflb_parser_create(e(regex), int(skip_empty), e(time_fmt), e(time_key), e(time_offset),
                                              int(time_keep), int(time_strict), e(types))
        if not self.h:
            raise ValueError("oracle: parser create failed")
        # decoders: [(as: bool, backend, field[, action])] = the parser's Decode_Field / Decode_Field_As lines in order
        lib().oflb_parser_add_decoder.argtypes = [c_void_p, c_int, c_char_p, c_char_p, c_char_p]
        for d in decoders or []:
            if lib().oflb_parser_add_decoder(self.h, int(bool(d[0])), e(d[1]), e(d[2]), e(d[3]) if len(d) > 3 and d[3] else None) != 0:
                raise ValueError("oracle: unknown decoder backend")
        # Time_Zone / Time_System_Timezone (flb_parser_create_with_time_zone, src/flb_parser.c:986-1022)
        if time_zone or time_system_timezone:
            lib().oflb_parser_set_time_zone.argtypes = [c_void_p, c_char_p, c_int, c_int]
            if lib().oflb_parser_set_time_zone(self.h, e(time_zone), int(bool(time_system_timezone)), int(bool(time_offset))) != 0:
                raise ValueError("oracle: time_zone refused")
    def do(self, buf):
        out = c_void_p(); sz = c_size_t(); sec = c_int64(); nsec = c_int64()
        r = lib().oflb_parser_do(self.h, buf, len(buf), byref(out), byref(sz), byref(sec), byref(nsec))
        if r < 0:
            return r, None, None
        return r, _take(out, sz), (sec.value, nsec.value)
    def time_lookup(self, s, now=0, time_offset=None):
        sec = c_int64(); frac = c_double()
        r = lib().oflb_time_lookup(self.h, s, len(s), now, -1 if time_offset is None else time_offset + (1 << 20),
                                   byref(sec), byref(frac))
        return r, sec.value, frac.value

class Grep:
    """rules: list of ("regex"|"exclude", "<key> <pattern>") in config order (plugins/filter_grep/grep.c:56-164)"""
    def __init__(self, rules, logical_op=None):
        n = len(rules)
        kinds = (c_char_p * max(n, 1))(*[k.encode() if isinstance(k, str) else k for k, _ in rules])
        vals = (c_char_p * max(n, 1))(*[v.encode() if isinstance(v, str) else v for _, v in rules])
        self.h = lib().oflb_grep_create(n, kinds, vals, logical_op.encode() if logical_op else None)
        if not self.h:
            raise ValueError("oracle: grep create failed")
    def filter(self, data):
        out = c_void_p(); sz = c_size_t()
        r = lib().oflb_grep_filter(self.h, data, len(data), byref(out), byref(sz))
        return r, (_take(out, sz) if r == MODIFIED else None)
    def bench(self, data, iters):
        ob = c_size_t()
        return lib().oflb_bench_grep(self.h, data, len(data), iters, byref(ob)), ob.value

class FilterParser:
    """mirrors filter_parser's Key_Name / Parser / Reserve_Data / Preserve_Key
    (plugins/filter_parser/filter_parser.c:460-489)"""
    def __init__(self, key_name, parsers, reserve_data=False, preserve_key=False):
        self.parsers = parsers
        arr = (c_void_p * len(parsers))(*[p.h for p in parsers])
        k = key_name.encode() if isinstance(key_name, str) else key_name
        self.h = lib().oflb_fparser_create(k, int(reserve_data), int(preserve_key), len(parsers), arr)
        if not self.h:
            raise ValueError("oracle: filter_parser create failed")
    def filter(self, data):
        out = c_void_p(); sz = c_size_t()
        r = lib().oflb_fparser_filter(self.h, data, len(data), byref(out), byref(sz))
        return r, (_take(out, sz) if r == MODIFIED else None)
    def bench(self, data, iters):
        ob = c_size_t()
        return lib().oflb_bench_fparser(self.h, data, len(data), iters, byref(ob)), ob.value

def msgpack_to_json_format(data, json_format, date_format, date_key, escape_unicode=1, nan_to_null=0):
    """flb_pack_msgpack_to_json_format (src/flb_pack.c:1320): bytes, or None where the reference returns NULL"""
    L = lib()
    L.oflb_msgpack_to_json_format.argtypes = [c_char_p, c_size_t, c_int, c_int, c_char_p, c_int, c_int, c_int,
                                              POINTER(c_void_p), POINTER(c_size_t)]
    out = c_void_p(); n = c_size_t()
    r = L.oflb_msgpack_to_json_format(data, len(data), json_format, date_format, date_key, -1 if date_key is None else len(date_key),
                                      escape_unicode, nan_to_null, byref(out), byref(n))
    if r != 0:
        return None
    return _take(out, n)


def count_records(data):
    return lib().oflb_count_records(data, len(data))

def repack(data):
    out = c_void_p(); sz = c_size_t()
    lib().oflb_repack(data, len(data), byref(out), byref(sz))
    return _take(out, sz)


class L2M:
    """filter_log_to_metrics (plugins/filter_log_to_metrics/log_to_metrics.c).  `props` is the list of
    (key, value) properties in configuration order: regex / exclude / label_field / add_label / bucket;
    the scalar options keep their property names."""
    def __init__(self, metric_mode="counter", props=(), kubernetes_mode=False, value_field=None, discard_logs=False):
        e = lambda s: s.encode() if isinstance(s, str) else s
        n = len(props)
        keys = (c_char_p * max(n, 1))(*[e(k) for k, _ in props])
        vals = (c_char_p * max(n, 1))(*[e(v) for _, v in props])
        self.h = lib().oflb_l2m_create(e(metric_mode), n, keys, vals, int(kubernetes_mode), e(value_field), int(discard_logs))
        if not self.h:
            raise ValueError("oracle: log_to_metrics create failed")
    def filter(self, data):
        return lib().oflb_l2m_filter(self.h, data, len(data))
    def snapshot(self):
        """-> (label_keys, bounds, [series]) with series = dict(labels=(..), value=, buckets=[..], count=, sum=)
        in insertion order"""
        lc = c_int(); nb = c_int()
        bounds = (c_double * 1024)()
        ns = lib().oflb_l2m_info(self.h, byref(lc), byref(nb), bounds)
        keys = [lib().oflb_l2m_label_key(self.h, i).decode("latin1") for i in range(lc.value)]
        out = []
        for s in range(ns):
            v = c_double(); cnt = ctypes.c_uint64(); sm = c_double()
            bk = (ctypes.c_uint64 * (nb.value + 1))()
            lib().oflb_l2m_series_get(self.h, s, byref(v), bk, byref(cnt), byref(sm))
            out.append(dict(labels=tuple(lib().oflb_l2m_series_label(self.h, s, i) for i in range(lc.value)),
                            value=v.value, buckets=list(bk), count=cnt.value, sum=sm.value))
        return keys, list(bounds[: nb.value]), out


def tail_process(text, key="log", path_key=None, path="", offset_key=None, stream_offset=0, skip_empty_lines=True, sec=0, nsec=0):
    """in_tail's process_content (plain path) + flb_tail_file_pack_line: (lines, records, processed bytes)"""
    L = lib()
    L.oflb_tail_process.argtypes = [c_char_p, c_size_t, c_char_p, c_char_p, c_char_p, c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32,
                                    ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t), ctypes.POINTER(ctypes.c_uint64)]
    out = c_void_p(); sz = c_size_t(); proc = ctypes.c_uint64()
    enc = lambda x: None if x is None else (x if isinstance(x, bytes) else x.encode())
    n = L.oflb_tail_process(text, len(text), enc(key), enc(path_key), enc(path), enc(offset_key), stream_offset, 1 if skip_empty_lines else 0, sec, nsec,
                            byref(out), byref(sz), byref(proc))
    return n, _take(out, sz), int(proc.value)


class Multiline:
    """oracle/oml.c: the multiline core behind in_tail's line loop (one parser, text lines)"""
    TYPES = {"regex": 0, "endswith": 1, "equal": 2, "eq": 2}

    # the built-in parsers with a parser in front (src/multiline/flb_ml_parser_cri.c:24-75, flb_ml_parser_docker.c:25-105)
    SUB_BUILTINS = {
        "cri": dict(type="equal", match_string="F", key_content="log", key_group="stream", key_pattern="_p",
                    subparser=dict(regex=r"^(?<time>.+?) (?<stream>stdout|stderr) (?<_p>F|P) (?<log>.*)$", time_fmt="%Y-%m-%dT%H:%M:%S.%L%z", time_key="time", skip_empty=False)),
        "docker": dict(type="endswith", match_string="\n", key_content="log", key_group="stream", key_pattern=None,
                       subparser=dict(regex=None, time_fmt="%Y-%m-%dT%H:%M:%S.%L", time_key="time", skip_empty=True)),
    }

    def __init__(self, rules=None, builtin=None, type="regex", match_string=None, negate=False, key_content=None, buffer_limit=-1,
                 subparser=None, key_group=None, key_pattern=None):
        if builtin in self.SUB_BUILTINS:
            b = self.SUB_BUILTINS[builtin]
            type, match_string, subparser, key_group, key_pattern = b["type"], b["match_string"], b["subparser"], b["key_group"], b["key_pattern"]
            key_content = key_content or b["key_content"]
            builtin = None
        L = lib()
        L.oml_set_subparser.argtypes = [c_void_p, c_char_p, c_char_p, c_char_p, c_int, c_char_p, c_char_p]
        L.oml_create.restype = c_void_p
        L.oml_create.argtypes = [c_int, c_char_p, c_int, c_char_p, c_int64]
        L.oml_destroy.argtypes = [c_void_p]
        L.oml_add_rule.argtypes = [c_void_p, c_char_p, c_char_p, c_char_p]
        L.oml_init.argtypes = [c_void_p]
        L.oml_builtin.argtypes = [c_void_p, c_char_p]
        L.oml_tail_chunk.argtypes = [c_void_p, c_char_p, c_size_t, c_int, c_int64, c_int64]
        L.oml_append_text.argtypes = [c_void_p, c_int64, c_int64, c_char_p, c_size_t]
        L.oml_flush_pending.argtypes = [c_void_p]
        L.oml_output.restype = c_size_t
        L.oml_output.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_int), POINTER(c_int)]
        L.oml_state.argtypes = [c_void_p, POINTER(c_int), POINTER(c_size_t), POINTER(c_size_t)]
        enc = lambda x: None if x is None else (x if isinstance(x, bytes) else x.encode())
        self.h = L.oml_create(self.TYPES[type.lower()], enc(match_string), 1 if negate else 0, enc(key_content), buffer_limit)
        if builtin:
            if L.oml_builtin(self.h, enc(builtin)) != 0:
                raise ValueError("multiline: built-in parser %r" % builtin)
        else:
            for fs, rx, to in (rules or []):
                if L.oml_add_rule(self.h, enc(fs), enc(rx), enc(to)) != 0:
                    raise ValueError("multiline: rule %r" % ((fs, rx, to),))
            if L.oml_init(self.h) != 0:
                raise ValueError("multiline: to_state not registered")
        if subparser is not None:
            if L.oml_set_subparser(self.h, enc(subparser.get("regex")), enc(subparser.get("time_fmt")), enc(subparser.get("time_key")),
                                   1 if subparser.get("skip_empty") else 0, enc(key_group), enc(key_pattern)) != 0:
                raise ValueError("multiline: sub-parser")

    def __del__(self):
        if getattr(self, "h", None):
            lib().oml_destroy(self.h)
            self.h = None

    def chain(self, other):
        """`other` becomes the next parser of this one's list (in_tail: `multiline.parser a, b`); self stays the handle of the stream"""
        L = lib()
        L.oml_chain_add.argtypes = [c_void_p, c_void_p]
        if L.oml_chain_add(self.h, other.h) != 0:
            raise ValueError("multiline: chain")
        self._chain = getattr(self, "_chain", []) + [other]             # (keeps them alive)
        return self

    def append(self, text, sec, nsec, skip_empty_lines=False):
        """one read of in_tail; returns (records bytes, record count, truncations)"""
        lib().oml_tail_chunk(self.h, text, len(text), 1 if skip_empty_lines else 0, sec, nsec)
        return self._out()

    def flush(self):
        lib().oml_flush_pending(self.h)
        return self._out()

    def _out(self):
        p = c_void_p(); r = c_int(); t = c_int()
        n = lib().oml_output(self.h, byref(p), byref(r), byref(t))
        return (ctypes.string_at(p, n) if n else b""), r.value, t.value

    def state(self):
        a = c_int(); b = c_size_t(); c = c_size_t()
        lib().oml_state(self.h, byref(a), byref(b), byref(c))
        return a.value, b.value, c.value
