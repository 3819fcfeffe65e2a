"""fluent-bit_amd -- MI355X-native Fluent Bit filter hot path (filter_parser / filter_grep /
regex parsers) behind the C ABI of include/flb_gpu.h.

This module is a thin ctypes binding used by tests/, bench.py and __graft_entry__.py.  The
product is csrc/libflbgpu.so (hand-written HIP kernels + C++ host code); nothing here computes
on the CPU and nothing here touches oracle/.  If the shared library is missing, import fails
loudly -- there is no fallback path.

The directory name contains a hyphen, so load it with `flbamd_loader.load()` (repo root) or
importlib; it registers itself as `fluent_bit_amd`.
"""
import ctypes
import os
import subprocess
from ctypes import (POINTER, Structure, byref, c_char_p, c_double, c_int, c_int64, c_size_t, c_uint,
                    c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("FLBGPU_LIB") or os.path.join(CSRC, "libflbgpu.so")

MODIFIED, NOTOUCH = 1, 2


def build(force=False):
    """Compile every HIP/C++ source for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-j8", "-C", CSRC, "all"], check=True)


class DevChunk(Structure):
    _fields_ = [("data", c_void_p), ("row_off", c_void_p), ("n", c_uint64), ("bytes", c_uint64)]


_L = None


def lib():
    global _L
    if _L is not None:
        return _L
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `make -C %s` (hipcc, gfx950). There is no CPU fallback." % (LIB_PATH, CSRC))
    L = ctypes.CDLL(LIB_PATH)
    L.flbgpu_init.argtypes = [c_int]
    L.flbgpu_set_time_now.argtypes = [ctypes.c_int64]
    L.flbgpu_set_time_now.restype = None
    L.flbgpu_last_error.restype = c_char_p
    L.flbgpu_parser_create.restype = c_void_p
    L.flbgpu_parser_create.argtypes = [c_char_p, c_char_p, c_int, c_char_p, c_char_p, c_char_p, c_int, c_int, c_char_p]
    L.flbgpu_parser_create_json.restype = c_void_p
    L.flbgpu_parser_create_json.argtypes = [c_char_p, c_char_p, c_char_p, c_char_p, c_int, c_int]
    L.flbgpu_parser_create_kv.restype = c_void_p
    L.flbgpu_parser_create_kv.argtypes = [c_char_p, c_char_p, c_char_p, c_char_p, c_char_p, c_int, c_int, c_int, c_char_p]
    L.flbgpu_parser_destroy.argtypes = [c_void_p]
    L.flbgpu_parser_add_decoder.argtypes = [c_void_p, c_int, c_char_p, c_char_p, c_char_p]
    L.flbgpu_parser_set_time_zone.argtypes = [c_void_p, c_char_p]
    L.flbgpu_parser_set_system_timezone.argtypes = [c_void_p, c_int]
    L.flbgpu_tz_tm2time.argtypes = [c_char_p, c_int64, POINTER(c_int64)]
    L.flbgpu_parser_do.argtypes = [c_void_p, c_char_p, c_size_t, POINTER(c_void_p), POINTER(c_size_t), POINTER(c_int64), POINTER(c_int64)]
    L.flbgpu_filter_parser_create.restype = c_void_p
    L.flbgpu_filter_parser_create.argtypes = [c_char_p, c_int, c_int, c_int, POINTER(c_void_p)]
    L.flbgpu_filter_grep_create.restype = c_void_p
    L.flbgpu_filter_grep_create.argtypes = [c_int, POINTER(c_char_p), POINTER(c_char_p), c_char_p]
    L.flbgpu_filter_destroy.argtypes = [c_void_p]
    L.flbgpu_filter_run.argtypes = [c_void_p, c_char_p, c_size_t, POINTER(c_void_p), POINTER(c_size_t)]
    L.flbgpu_filter_run_dev.argtypes = [c_void_p, POINTER(DevChunk), POINTER(DevChunk), c_void_p]
    L.flbgpu_filter_last_counts.argtypes = [c_void_p, POINTER(c_uint64), POINTER(c_uint64)]
    L.flbgpu_filter_profile.argtypes = [c_void_p, c_int]
    L.flbgpu_filter_profile_read.argtypes = [c_void_p, c_int, POINTER(c_char_p), POINTER(c_double), POINTER(c_uint64)]
    L.flbgpu_index_host.restype = c_int64
    L.flbgpu_index_host.argtypes = [c_char_p, c_size_t, c_void_p, c_size_t, POINTER(c_size_t)]
    L.flbgpu_indexer_create.restype = c_void_p
    L.flbgpu_indexer_destroy.argtypes = [c_void_p]
    L.flbgpu_index_dev.restype = c_int64
    L.flbgpu_index_dev.argtypes = [c_void_p, c_void_p, c_size_t, POINTER(DevChunk), POINTER(c_size_t)]
    L.flbgpu_indexer_stats.argtypes = [c_void_p, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64)]
    L.flbgpu_tail_clean_host.argtypes = [c_char_p, c_size_t, c_size_t]
    L.flbgpu_dev_alloc.restype = c_void_p
    L.flbgpu_dev_alloc.argtypes = [c_size_t]
    L.flbgpu_dev_free.argtypes = [c_void_p]
    L.flbgpu_memcpy_h2d.argtypes = [c_void_p, c_void_p, c_size_t]
    L.flbgpu_memcpy_d2h.argtypes = [c_void_p, c_void_p, c_size_t]
    L.flbgpu_rx_compile.restype = c_void_p
    L.flbgpu_rx_compile.argtypes = [c_char_p, c_int, c_uint, c_int, c_char_p, c_int]
    L.flbgpu_rx_free.argtypes = [c_void_p]
    L.flbgpu_rx_simulate_capture.argtypes = [c_void_p, c_char_p, c_int, POINTER(c_int), POINTER(c_int)]
    L.flbgpu_rx_simulate_match.argtypes = [c_void_p, c_char_p, c_int]
    L.flbgpu_rx_info.argtypes = [c_void_p, POINTER(c_int)]
    L.flbgpu_rx_names.argtypes = [c_void_p, c_char_p, c_int]
    L.flbgpu_rx_engine.argtypes = [c_void_p, POINTER(c_int), c_char_p, c_int]
    L.flbgpu_rx_corner.argtypes = [c_void_p, c_char_p, c_int, POINTER(c_int)]
    L.flbgpu_l2m_set_sum_order.argtypes = [c_void_p, c_int]
    L.flbgpu_l2m_seq_sums.restype = c_int64
    L.flbgpu_l2m_seq_sums.argtypes = [c_void_p, c_uint64, POINTER(c_double)]
    L.flbgpu_host_phases.restype = ctypes.c_int
    L.flbgpu_host_phases.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int]
    L.flbgpu_filter_host_rules.restype = ctypes.c_int
    L.flbgpu_filter_host_rules.argtypes = [c_void_p, ctypes.POINTER(ctypes.c_uint64)]
    L.flbgpu_filter_paths.restype = ctypes.c_int
    L.flbgpu_filter_paths.argtypes = [c_void_p, ctypes.POINTER(ctypes.c_uint64)]
    L.flbgpu_filter_regex_corners.restype = ctypes.c_uint64
    L.flbgpu_filter_regex_corners.argtypes = [c_void_p]
    L.flbgpu_rx_simulate_fx3.argtypes = [c_void_p, c_char_p, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]
    L.flbgpu_rx_sample.argtypes = [c_char_p, c_int, c_uint, ctypes.c_ulonglong, c_char_p, c_int]
    L.flbgpu_filter_chain_run.argtypes = [POINTER(c_void_p), c_int, c_char_p, c_size_t, POINTER(c_void_p), POINTER(c_size_t), c_void_p]
    L.flbgpu_filter_chain_run_dev.argtypes = [POINTER(c_void_p), c_int, POINTER(DevChunk), POINTER(DevChunk), c_void_p]
    L.flbgpu_pack_json_recs.argtypes = [c_char_p, c_size_t, POINTER(c_void_p), POINTER(c_size_t), POINTER(c_int), POINTER(c_int), POINTER(c_size_t)]
    L.flbgpu_pack_json.argtypes = [c_char_p, c_size_t, POINTER(c_void_p), POINTER(c_size_t), POINTER(c_int), POINTER(c_size_t)]
    L.flbgpu_json_create.restype = c_void_p
    L.flbgpu_json_destroy.argtypes = [c_void_p]
    L.flbgpu_json_run_dev.argtypes = [c_void_p, POINTER(DevChunk), c_int, c_uint, c_uint, POINTER(DevChunk)]
    L.flbgpu_json_row_info.argtypes = [c_void_p, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p]
    L.flbgpu_json_stats.argtypes = [c_void_p, POINTER(c_uint64)]
    L.flbgpu_json_tile_stats.argtypes = [c_void_p, POINTER(c_uint64)]
    L.flbgpu_json_tile_debug.argtypes = [c_void_p, c_int, c_int, POINTER(c_uint64)]
    L.flbgpu_split_lines_host.restype = c_int64
    L.flbgpu_split_lines_host.argtypes = [c_char_p, c_size_t, c_void_p, c_size_t]
    L.flbgpu_filter_l2m_create.restype = c_void_p
    L.flbgpu_filter_l2m_create.argtypes = [c_char_p, c_int, POINTER(c_char_p), POINTER(c_char_p), c_int, c_char_p, c_int]
    L.flbgpu_l2m_info.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]
    L.flbgpu_l2m_label_key.restype = c_char_p
    L.flbgpu_l2m_label_key.argtypes = [c_void_p, c_int]
    L.flbgpu_l2m_bounds.argtypes = [c_void_p, POINTER(c_double)]
    L.flbgpu_l2m_export.restype = c_int64
    L.flbgpu_l2m_export.argtypes = [c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_size_t, POINTER(c_size_t)]
    L.flbgpu_l2m_finalize_row.argtypes = [c_int, c_int, c_void_p, POINTER(c_double), c_void_p, POINTER(c_uint64), POINTER(c_double)]
    L.flbgpu_l2m_set_index_base.argtypes = [c_void_p, c_uint64]
    L.flbgpu_l2m_stats.argtypes = [c_void_p, POINTER(c_uint64)]
    L.flbgpu_jsonfmt_create.restype = c_void_p
    L.flbgpu_jsonfmt_create.argtypes = [c_int, c_int, c_char_p, c_int, c_int, c_int]
    L.flbgpu_jsonfmt_run.argtypes = [c_void_p, c_char_p, c_size_t, POINTER(c_void_p), POINTER(c_size_t)]
    L.flbgpu_jsonfmt_run_dev.argtypes = [c_void_p, POINTER(DevChunk), POINTER(DevChunk)]
    L.flbgpu_pack_msgpack_to_json_format.argtypes = [c_char_p, c_uint64, c_int, c_int, c_char_p, c_int, c_int, c_int,
                                                     POINTER(c_void_p), POINTER(c_size_t)]
    L.flbgpu_nc_scan_double.argtypes = [c_char_p, c_int, c_int, c_int, POINTER(c_double), POINTER(c_int)]
    L.flbgpu_nc_fmt_f6.argtypes = [c_double, c_char_p, c_int]
    L.flbgpu_nc_fmt_ld.argtypes = [ctypes.c_longlong, c_char_p]
    L.flbgpu_nc_scan_double_dev.argtypes = [c_char_p, c_void_p, c_uint, c_int, c_void_p, c_void_p]
    _L = L
    return L


_libc = ctypes.CDLL(None)
_libc.free.argtypes = [c_void_p]


def _b(s):
    return s.encode() if isinstance(s, str) else s


def set_time_now(now):
    """pins the clock year-less Time_Formats read (0: time(NULL) again)"""
    lib().flbgpu_set_time_now(ctypes.c_int64(int(now)))


def last_error():
    return lib().flbgpu_last_error().decode(errors="replace")


_inited = False


def init(device=0):
    """Bind the process to one GPU.  Raises when no HIP device exists (no CPU path)."""
    global _inited
    if lib().flbgpu_init(device) != 0:
        raise RuntimeError("flbgpu_init: " + last_error())
    _inited = True


class Parser:
    """struct flb_parser for Format regex (include/fluent-bit/flb_parser.h:41-70); arguments as
    flb_parser_create.  Defaults are the parsers-file defaults (src/flb_parser.c:1277-1304)."""

    def __init__(self, regex=None, time_fmt=None, time_key=None, time_offset=None, time_keep=False,
                 time_strict=True, skip_empty=True, types=None, name="parser", format="regex", no_bare_keys=False, decoders=None,
                 time_zone=None, time_system_timezone=False):
        if format in ("logfmt", "ltsv"):
            self.h = lib().flbgpu_parser_create_kv(_b(name), _b(format), _b(time_fmt), _b(time_key), _b(time_offset), int(time_keep),
                                                   int(time_strict), int(no_bare_keys), _b(types))
        elif format == "json":
            self.h = lib().flbgpu_parser_create_json(_b(name), _b(time_fmt), _b(time_key), _b(time_offset), int(time_keep),
                                                     int(time_strict))
        else:
            self.h = lib().flbgpu_parser_create(_b(name), _b(regex), int(skip_empty), _b(time_fmt), _b(time_key),
                                                _b(time_offset), int(time_keep), int(time_strict), _b(types))
        if not self.h:
            raise ValueError("flbgpu_parser_create: " + last_error())
        # decoders: [(as: bool, backend, field[, action])] = the parser's Decode_Field / Decode_Field_As lines in order
        for d in decoders or []:
            if lib().flbgpu_parser_add_decoder(self.h, int(bool(d[0])), _b(d[1]), _b(d[2]), _b(d[3] if len(d) > 3 else None)) != 0:
                raise ValueError("flbgpu_parser_add_decoder: " + last_error())
        # Time_Zone <IANA name> / Time_System_Timezone On (src/flb_parser.c:986-1022)
        if time_system_timezone and lib().flbgpu_parser_set_system_timezone(self.h, 1) != 0:
            err = last_error(); self.close()
            raise ValueError("flbgpu_parser_set_system_timezone: " + err)
        if time_zone and lib().flbgpu_parser_set_time_zone(self.h, _b(time_zone)) != 0:
            err = last_error(); self.close()
            raise ValueError("flbgpu_parser_set_time_zone: " + err)

    def do(self, buf):
        out = c_void_p(); sz = c_size_t(); sec = c_int64(); nsec = c_int64()
        r = lib().flbgpu_parser_do(self.h, buf, len(buf), byref(out), byref(sz), byref(sec), byref(nsec))
        if r < 0:
            return r, None, None
        data = ctypes.string_at(out, sz.value)
        _libc.free(out)
        return r, data, (sec.value, nsec.value)

    def close(self):
        if self.h:
            lib().flbgpu_parser_destroy(self.h)
            self.h = None


class _Filter:
    h = None

    def filter(self, data):
        """cb_filter on a host buffer: returns (MODIFIED|NOTOUCH, bytes|None)"""
        out = c_void_p(); sz = c_size_t()
        r = lib().flbgpu_filter_run(self.h, data, len(data), byref(out), byref(sz))
        if r != MODIFIED:
            return r, None
        b = ctypes.string_at(out, sz.value) if sz.value else b""
        if out.value:
            _libc.free(out)
        return r, b

    def filter_dev(self, chunk, stream=None):
        """cb_filter on a device-resident chunk: returns (ret, DevChunk)"""
        out = DevChunk()
        r = lib().flbgpu_filter_run_dev(self.h, byref(chunk), byref(out), stream)
        return r, out

    def counts(self):
        a = c_uint64(); b = c_uint64()
        lib().flbgpu_filter_last_counts(self.h, byref(a), byref(b))
        return a.value, b.value

    def regex_corners(self):
        """values that met one of the reference's optimizer-dependent regex corners since the filter was created (flb_gpu.h)"""
        return int(lib().flbgpu_filter_regex_corners(self.h))

    def host_rules(self):
        """rules / parsers of this filter the host's backtracking matcher answers (patterns that are not regular expressions):
        dict(rules, values, budget_over, unhandled) -- flbgpu_filter_host_rules"""
        o = (ctypes.c_uint64 * 4)()
        lib().flbgpu_filter_host_rules(self.h, o)
        return dict(rules=int(o[0]), values=int(o[1]), budget_over=int(o[2]), unhandled=int(o[3]))

    def paths(self):
        """which builds this filter_parser instance ran last and which are set aside right now (flbgpu_filter_paths)"""
        o = (ctypes.c_uint64 * 8)()
        lib().flbgpu_filter_paths(self.h, o)
        lp = int(o[0])
        return dict(single_pass=bool(lp & 1), three_port=bool(lp & 2), emit_time=bool(lp & 4), plain_emit=bool(lp & 8), rows_by_length=bool(lp & 16),
                    aside=dict(single_pass=bool(o[1]), three_port=bool(o[2]), emit_time=bool(o[3]), plain_emit=bool(o[4])),
                    tries=int(o[5]), returns=int(o[6]), calls=int(o[7]))

    def profile(self, enable=True):
        lib().flbgpu_filter_profile(self.h, int(enable))

    def profile_read(self):
        names = (c_char_p * 16)(); ms = (c_double * 16)(); ln = (c_uint64 * 16)()
        n = lib().flbgpu_filter_profile_read(self.h, 16, names, ms, ln)
        return {names[i].decode(): (ms[i], ln[i]) for i in range(n)}

    def close(self):
        if self.h:
            lib().flbgpu_filter_destroy(self.h)
            self.h = None


class FilterParser(_Filter):
    """filter_parser: Key_Name / Parser* / Reserve_Data / Preserve_Key
    (plugins/filter_parser/filter_parser.c:460-489)"""

    def __init__(self, key_name, parsers, reserve_data=False, preserve_key=False):
        self.parsers = list(parsers)
        arr = (c_void_p * len(parsers))(*[p.h for p in parsers])
        self.h = lib().flbgpu_filter_parser_create(_b(key_name), int(reserve_data), int(preserve_key), len(parsers), arr)
        if not self.h:
            raise ValueError("flbgpu_filter_parser_create: " + last_error())


class FilterGrep(_Filter):
    """filter_grep: rules = [("regex"|"exclude", "<key> <pattern>"), ...] in configuration order,
    logical_op None|"legacy"|"AND"|"OR" (plugins/filter_grep/grep.c:407-424)"""

    def __init__(self, rules, logical_op=None):
        n = len(rules)
        kinds = (c_char_p * max(n, 1))(*[_b(k) for k, _ in rules])
        vals = (c_char_p * max(n, 1))(*[_b(v) for _, v in rules])
        self.h = lib().flbgpu_filter_grep_create(n, kinds, vals, _b(logical_op))
        if not self.h:
            raise ValueError("flbgpu_filter_grep_create: " + last_error())


JSON_FORMAT = {"json": 1, "stream": 2, "lines": 3}                                        # flb_pack_to_json_format_type
JSON_DATE = {"double": 0, "iso8601": 1, "epoch": 2, "java_sql_timestamp": 3, "epoch_ms": 4}    # flb_pack_to_json_date_type


class JsonFormatter(_Filter):
    """flb_pack_msgpack_to_json_format (src/flb_pack.c:1320-1600): a chunk of log events as JSON text."""

    def __init__(self, json_format="lines", date_format="double", date_key=b"date", escape_unicode=True, nan_to_null=False):
        jf = JSON_FORMAT.get(json_format, json_format)
        df = JSON_DATE.get(date_format, date_format)
        dk = None if date_key is None else _b(date_key)
        self.h = lib().flbgpu_jsonfmt_create(jf, df, dk, -1 if dk is None else len(dk), int(bool(escape_unicode)), int(bool(nan_to_null)))
        if not self.h:
            raise ValueError(last_error())

    def format(self, data):
        """host chunk -> bytes, or None where the reference returns NULL"""
        out = c_void_p(); sz = c_size_t()
        r = lib().flbgpu_jsonfmt_run(self.h, data, len(data), byref(out), byref(sz))
        if r != 0:
            if last_error():
                raise RuntimeError(last_error())
            return None
        b = ctypes.string_at(out, sz.value)
        _libc.free(out)
        return b

    def format_dev(self, chunk):
        """device chunk -> (0 | -1, DevChunk of the text in HBM)"""
        out = DevChunk()
        r = lib().flbgpu_jsonfmt_run_dev(self.h, byref(chunk), byref(out))
        if r != 0 and last_error():
            raise RuntimeError(last_error())
        return r, out


class TailLines:
    """in_tail's line packing (plugins/in_tail/tail_file.c:689-1040 process_content, the plain path, + :552-604
    flb_tail_file_pack_line): a file buffer cut at the newlines, every line one log event."""

    def __init__(self, key="log", path_key=None, path="", offset_key=None, skip_empty_lines=True):
        L = lib()
        L.flbgpu_tail_create.restype = c_void_p
        L.flbgpu_tail_create.argtypes = [c_char_p, c_char_p, c_char_p, c_char_p, c_int]
        L.flbgpu_tail_destroy.argtypes = [c_void_p]
        L.flbgpu_tail_run.argtypes = [c_void_p, c_void_p, c_size_t, c_uint64, ctypes.c_uint32, ctypes.c_uint32, POINTER(c_void_p), POINTER(c_size_t),
                                      POINTER(c_uint64), POINTER(c_uint64)]
        L.flbgpu_tail_run_dev.argtypes = [c_void_p, c_void_p, c_uint64, c_uint64, ctypes.c_uint32, ctypes.c_uint32, POINTER(DevChunk), POINTER(c_uint64), POINTER(c_uint64)]
        e = lambda x: None if x is None else _b(x)
        self.h = L.flbgpu_tail_create(e(key), e(path_key), e(path), e(offset_key), int(bool(skip_empty_lines)))
        if not self.h:
            raise ValueError(last_error())

    def process(self, text, stream_offset=0, sec=0, nsec=0):
        """host buffer -> (lines, records, processed bytes)"""
        out = c_void_p(); sz = c_size_t(); proc = c_uint64(); lines = c_uint64()
        r = lib().flbgpu_tail_run(self.h, text, len(text), stream_offset, sec, nsec, byref(out), byref(sz), byref(proc), byref(lines))
        if r != 0:
            raise RuntimeError(last_error())
        b = ctypes.string_at(out, sz.value) if out.value else b""
        if out.value:
            _libc.free(out)
        return int(lines.value), b, int(proc.value)

    def process_dev(self, d_text, nbytes, stream_offset=0, sec=0, nsec=0):
        """text in HBM -> (lines, DevChunk, processed bytes)"""
        out = DevChunk(); proc = c_uint64(); lines = c_uint64()
        r = lib().flbgpu_tail_run_dev(self.h, d_text, nbytes, stream_offset, sec, nsec, byref(out), byref(proc), byref(lines))
        if r != 0:
            raise RuntimeError(last_error())
        return int(lines.value), out, int(proc.value)

    def close(self):
        if self.h:
            lib().flbgpu_tail_destroy(self.h)
            self.h = None


class MultilineParser:
    """a multiline parser definition (src/multiline/flb_ml_parser.c flb_ml_parser_create + flb_ml_rule.c flb_ml_rule_create / _init) with the
    instance's key_content and the context's buffer limit; rules: [(from_states, regex, to_state)] or builtin = java | go | python | ruby"""

    def __init__(self, rules=None, builtin=None, type="regex", match_string=None, negate=False, key_content=None, buffer_limit=-1,
                 subparser=None, key_group=None, key_pattern=None):
        L = lib()
        L.flbgpu_ml_parser_set_subparser.argtypes = [c_void_p, c_void_p, c_char_p, c_char_p]
        self.subparser = subparser                              # (kept alive: the library borrows it)
        L.flbgpu_ml_parser_create.restype = c_void_p
        L.flbgpu_ml_parser_create.argtypes = [c_char_p, c_char_p, c_int, c_char_p, ctypes.c_int64]
        L.flbgpu_ml_parser_add_rule.argtypes = [c_void_p, c_char_p, c_char_p, c_char_p]
        L.flbgpu_ml_parser_builtin.argtypes = [c_void_p, c_char_p]
        L.flbgpu_ml_parser_init.argtypes = [c_void_p]
        L.flbgpu_ml_parser_destroy.argtypes = [c_void_p]
        e = lambda x: None if x is None else _b(x)
        self.h = L.flbgpu_ml_parser_create(e(type), e(match_string), int(bool(negate)), e(key_content), int(buffer_limit))
        if not self.h:
            raise ValueError(last_error())
        ok = True
        if builtin:
            ok = L.flbgpu_ml_parser_builtin(self.h, e(builtin)) == 0
        else:
            for fs, rx, to in (rules or []):
                ok = ok and L.flbgpu_ml_parser_add_rule(self.h, e(fs), e(rx), e(to)) == 0
            if subparser is not None:
                ok = ok and L.flbgpu_ml_parser_set_subparser(self.h, subparser.h, e(key_group), e(key_pattern)) == 0
            ok = ok and L.flbgpu_ml_parser_init(self.h) == 0
        if not ok:
            err = last_error()
            self.close()
            raise ValueError(err)

    def stream(self):
        return MultilineStream(self)

    def product(self):
        """(states, joint classes, non-absorbing states) of the product of the rules' match DFAs; states 0: rules walked one by one"""
        a = ctypes.c_uint32(); b = ctypes.c_uint32(); c = ctypes.c_uint32()
        lib().flbgpu_ml_parser_product.argtypes = [c_void_p, POINTER(ctypes.c_uint32), POINTER(ctypes.c_uint32), POINTER(ctypes.c_uint32)]
        lib().flbgpu_ml_parser_product(self.h, byref(a), byref(b), byref(c))
        return a.value, b.value, c.value

    def close(self):
        if self.h:
            lib().flbgpu_ml_parser_destroy(self.h)
            self.h = None


class MultilineStream:
    """what one tailed file carries through a multiline parser (flb_ml_stream_create): in_tail's line loop + flb_ml_append_text per read"""

    def __init__(self, parser):
        L = lib()
        L.flbgpu_ml_stream_create.restype = c_void_p
        L.flbgpu_ml_stream_create.argtypes = [c_void_p]
        L.flbgpu_ml_stream_destroy.argtypes = [c_void_p]
        L.flbgpu_ml_stream_state.argtypes = [c_void_p, POINTER(c_int), POINTER(c_uint64)]
        L.flbgpu_ml_append.argtypes = [c_void_p, c_void_p, c_size_t, ctypes.c_uint32, ctypes.c_uint32, c_int, c_int, POINTER(c_void_p), POINTER(c_size_t),
                                       POINTER(c_uint64), POINTER(c_uint64)]
        L.flbgpu_ml_append_dev.argtypes = [c_void_p, c_void_p, c_uint64, ctypes.c_uint32, ctypes.c_uint32, c_int, c_int, POINTER(DevChunk), POINTER(c_uint64), POINTER(c_uint64)]
        self.parser = parser
        self.h = L.flbgpu_ml_stream_create(parser.h)
        if not self.h:
            raise ValueError(last_error())
        self.pending = b""

    def append(self, text, sec, nsec, skip_empty_lines=False, flush=False):
        """one read of the file appended to its buffer (what follows the last newline waits, as in_tail keeps it) -> (records bytes, count)"""
        buf = self.pending + text
        out = c_void_p(); sz = c_size_t(); proc = c_uint64(); recs = c_uint64()
        r = lib().flbgpu_ml_append(self.h, buf, len(buf), sec, nsec, int(bool(skip_empty_lines)), int(bool(flush)), byref(out), byref(sz), byref(proc), byref(recs))
        if r != 0:
            raise RuntimeError(last_error())
        b = ctypes.string_at(out, sz.value) if out.value else b""
        if out.value:
            _libc.free(out)
        self.pending = buf[proc.value:]
        return b, int(recs.value)

    def append_dev(self, d_text, nbytes, sec, nsec, skip_empty_lines=False, flush=False):
        """text in HBM -> (DevChunk, records, processed bytes)"""
        out = DevChunk(); proc = c_uint64(); recs = c_uint64()
        r = lib().flbgpu_ml_append_dev(self.h, d_text, nbytes, sec, nsec, int(bool(skip_empty_lines)), int(bool(flush)), byref(out), byref(proc), byref(recs))
        if r != 0:
            raise RuntimeError(last_error())
        return out, int(recs.value), int(proc.value)

    def flush(self, sec=0, nsec=0):
        """the flush timer (flb_ml_flush_pending): the open group leaves; (sec, nsec) = the clock, for a group that never saw a time"""
        keep, self.pending = self.pending, b""
        try:
            return self.append(b"", sec, nsec, flush=True)
        finally:
            self.pending = keep

    def state(self):
        a = c_int(); b = c_uint64()
        lib().flbgpu_ml_stream_state(self.h, byref(a), byref(b))
        return a.value, int(b.value)

    def truncations(self):
        lib().flbgpu_ml_stream_truncations.restype = c_uint64
        lib().flbgpu_ml_stream_truncations.argtypes = [c_void_p]
        return int(lib().flbgpu_ml_stream_truncations(self.h))

    def close(self):
        if self.h:
            lib().flbgpu_ml_stream_destroy(self.h)
            self.h = None


class MultilineList:
    """several multiline parsers on one tailed file -- in_tail's `multiline.parser docker, cri` (flb_ml_append_text's loop over the parser
    instances, src/multiline/flb_ml.c:671-760).  `parsers`: MultilineParser objects in configuration order, each with a parser in front.
    A read whose lines split between parsers raises RuntimeError (the caller keeps it on the CPU)."""

    def __init__(self, parsers):
        L = lib()
        L.flbgpu_ml_list_create.restype = c_void_p
        L.flbgpu_ml_list_create.argtypes = [POINTER(c_void_p), c_int]
        L.flbgpu_ml_list_destroy.argtypes = [c_void_p]
        L.flbgpu_ml_list_lru.argtypes = [c_void_p]
        L.flbgpu_ml_list_append.argtypes = [c_void_p, c_void_p, c_size_t, ctypes.c_uint32, ctypes.c_uint32, c_int, c_int, POINTER(c_void_p), POINTER(c_size_t),
                                            POINTER(c_uint64), POINTER(c_uint64)]
        self.streams = [p.stream() for p in parsers]
        arr = (c_void_p * len(self.streams))(*[s.h for s in self.streams])
        self.h = L.flbgpu_ml_list_create(arr, len(self.streams))
        if not self.h:
            raise ValueError(last_error())
        self.pending = b""

    def append(self, text, sec, nsec, skip_empty_lines=False, flush=False):
        buf = self.pending + text
        out = c_void_p(); sz = c_size_t(); proc = c_uint64(); recs = c_uint64()
        r = lib().flbgpu_ml_list_append(self.h, buf, len(buf), sec, nsec, int(bool(skip_empty_lines)), int(bool(flush)), byref(out), byref(sz), byref(proc), byref(recs))
        if r != 0:
            raise RuntimeError(last_error())
        b = ctypes.string_at(out, sz.value) if out.value else b""
        if out.value:
            _libc.free(out)
        self.pending = buf[proc.value:]
        return b, int(recs.value)

    def flush(self, sec=0, nsec=0):
        keep, self.pending = self.pending, b""
        try:
            return self.append(b"", sec, nsec, flush=True)
        finally:
            self.pending = keep

    @property
    def lru(self):
        """index of the parser that took the last line (-1: none yet)"""
        return int(lib().flbgpu_ml_list_lru(self.h))

    def close(self):
        if self.h:
            lib().flbgpu_ml_list_destroy(self.h)
            self.h = None
        for s in self.streams:
            s.close()
        self.streams = []


class StreamTask:
    """one task of the stream processor (src/stream_processor/flb_sp.c: flb_sp_task_create :433, flb_sp_do :2007, the window
    timer of flb_sp_fd_event :2101): aggregate queries (GROUP BY / COUNT SUM AVG MIN MAX / WHERE / WINDOW TUMBLING | HOPPING) and
    SELECTs without aggregation functions (keys, aliases, `*`, WHERE: sp_process_data :1607)."""

    def __init__(self, sql, str_conv=True, tag=b""):
        L = lib()
        L.flbgpu_sp_create.restype = c_void_p
        L.flbgpu_sp_create.argtypes = [c_char_p, c_int]
        L.flbgpu_sp_destroy.argtypes = [c_void_p]
        L.flbgpu_sp_info.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int64), POINTER(c_int), POINTER(c_char_p), POINTER(c_char_p)]
        L.flbgpu_sp_stream_prop.restype = c_char_p
        L.flbgpu_sp_stream_prop.argtypes = [c_void_p, c_char_p]
        L.flbgpu_sp_key_count.argtypes = [c_void_p]
        L.flbgpu_sp_key_name.restype = c_char_p
        L.flbgpu_sp_key_name.argtypes = [c_void_p, c_int]
        L.flbgpu_sp_do.argtypes = [c_void_p, c_char_p, c_size_t, ctypes.c_uint32, ctypes.c_uint32, POINTER(c_void_p), POINTER(c_size_t), POINTER(c_int64)]
        L.flbgpu_sp_do_dev.argtypes = [c_void_p, POINTER(DevChunk), c_void_p, ctypes.c_uint32, ctypes.c_uint32, POINTER(c_void_p), POINTER(c_size_t),
                                       POINTER(c_int64)]
        L.flbgpu_sp_timer.argtypes = [c_void_p, ctypes.c_uint32, ctypes.c_uint32, POINTER(c_void_p), POINTER(c_size_t)]
        L.flbgpu_sp_set_index_base.argtypes = [c_void_p, c_uint64]
        L.flbgpu_sp_hop.argtypes = [c_void_p]
        L.flbgpu_sp_window_advance.restype = c_int64
        L.flbgpu_sp_window_advance.argtypes = [c_void_p]
        L.flbgpu_sp_profile.argtypes = [c_void_p, c_int, POINTER(c_double), POINTER(c_uint64)]
        self.h = L.flbgpu_sp_create(_b(sql), int(bool(str_conv)))
        if not self.h:
            raise ValueError(last_error())
        wt = c_int(); ws = c_int64(); st = c_int(); src = c_char_p(); name = c_char_p()
        L.flbgpu_sp_info(self.h, byref(wt), byref(ws), byref(st), byref(src), byref(name))
        self.window = "hopping" if wt.value == 2 else "tumbling" if wt.value == 1 else "default"
        self.window_size = int(ws.value)
        self.window_advance = int(L.flbgpu_sp_window_advance(self.h))
        self.source_type = "tag" if st.value == 1 else "stream"
        self.source = src.value.decode()
        self.stream_name = name.value.decode() if name.value else None
        self.key_names = [L.flbgpu_sp_key_name(self.h, i).decode() for i in range(L.flbgpu_sp_key_count(self.h))]
        L.flbgpu_sp_select_only.argtypes = [c_void_p]
        L.flbgpu_sp_set_tag.argtypes = [c_void_p, c_char_p, c_size_t]
        L.flbgpu_sp_set_tag.restype = None
        self.set_tag(tag)
        self.select_only = bool(L.flbgpu_sp_select_only(self.h))     # sp_process_data: do() answers (records that passed WHERE, projected records)

    def set_tag(self, tag):
        """the tag of the chunks this task is fed: what RECORD_TAG() packs"""
        tag = _b(tag)
        lib().flbgpu_sp_set_tag(self.h, tag, len(tag))

    def stream_prop(self, key):
        v = lib().flbgpu_sp_stream_prop(self.h, _b(key))
        return None if v is None else v.decode()

    @staticmethod
    def _take(out, sz):
        b = ctypes.string_at(out, sz.value) if out.value else b""
        if out.value:
            _libc.free(out)
        return b

    def do(self, chunk, now=(1, 0)):
        """one appended chunk in host memory -> (records in the window, packaged records when the query has no WINDOW)"""
        out = c_void_p(); sz = c_size_t(); rec = c_int64()
        r = lib().flbgpu_sp_do(self.h, bytes(chunk), len(chunk), now[0], now[1], byref(out), byref(sz), byref(rec))
        if r != 0:
            raise RuntimeError(last_error())
        return int(rec.value), self._take(out, sz)

    def do_dev(self, chunk, now=(1, 0)):
        """the same for a chunk resident in HBM (DevChunk)"""
        out = c_void_p(); sz = c_size_t(); rec = c_int64()
        r = lib().flbgpu_sp_do_dev(self.h, byref(chunk), None, now[0], now[1], byref(out), byref(sz), byref(rec))
        if r != 0:
            raise RuntimeError(last_error())
        return int(rec.value), self._take(out, sz)

    def select_dev(self, chunk, now=(1, 0)):
        """a SELECT without aggregation functions, device chunk in -> (records that passed WHERE, DevChunk of the projected records)"""
        L = lib()
        L.flbgpu_sp_select_dev.argtypes = [c_void_p, POINTER(DevChunk), c_void_p, ctypes.c_uint32, ctypes.c_uint32, POINTER(DevChunk), POINTER(c_int64)]
        out = DevChunk(); rec = c_int64()
        if L.flbgpu_sp_select_dev(self.h, byref(chunk), None, now[0], now[1], byref(out), byref(rec)) != 0:
            raise RuntimeError(last_error())
        return int(rec.value), out

    def timer(self, now=(1, 0)):
        """the window's timer fires: packaged records of the window, which is then pruned"""
        out = c_void_p(); sz = c_size_t()
        r = lib().flbgpu_sp_timer(self.h, now[0], now[1], byref(out), byref(sz))
        if r != 0:
            raise RuntimeError(last_error())
        return self._take(out, sz)

    def hop(self):
        """the hop timer of a HOPPING window fires (every window_advance seconds): a slot is closed"""
        if lib().flbgpu_sp_hop(self.h) != 0:
            raise RuntimeError(last_error())
        return 0

    def set_index_base(self, base):
        lib().flbgpu_sp_set_index_base(self.h, base)

    def export(self):
        """this shard's window state (opaque bytes: typed key tuples + order-independent row words)"""
        L = lib()
        L.flbgpu_sp_export.restype = c_int64
        L.flbgpu_sp_export.argtypes = [c_void_p, c_void_p, c_size_t]
        n = L.flbgpu_sp_export(self.h, None, 0)
        if n < 0:
            raise RuntimeError(last_error())
        buf = ctypes.create_string_buffer(int(n))
        if L.flbgpu_sp_export(self.h, buf, n) != n:
            raise RuntimeError(last_error())
        return buf.raw

    def package_merged(self, snapshots, now=(1, 0)):
        """package_results over the merge of shard states (flbgpu_sp_package_merged): the records a single task fed
        with all the shards' chunks would package"""
        L = lib()
        L.flbgpu_sp_package_merged.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_size_t), c_int, ctypes.c_uint32, ctypes.c_uint32,
                                               POINTER(c_void_p), POINTER(c_size_t)]
        n = len(snapshots)
        keep = [ctypes.create_string_buffer(b, len(b)) for b in snapshots]
        ptrs = (c_void_p * max(n, 1))(*[ctypes.cast(k, c_void_p) for k in keep])
        sizes = (c_size_t * max(n, 1))(*[len(b) for b in snapshots])
        out = c_void_p(); sz = c_size_t()
        if L.flbgpu_sp_package_merged(self.h, ptrs, sizes, n, now[0], now[1], byref(out), byref(sz)) != 0:
            raise RuntimeError(last_error())
        return self._take(out, sz)

    def timer_all_reduce(self, comm, now=(1, 0)):
        """the timer of a window sharded over ranks: exchange over RCCL, merged records on every rank, window pruned"""
        L = lib()
        L.flbgpu_sp_timer_all_reduce.argtypes = [c_void_p, c_void_p, c_void_p, ctypes.c_uint32, ctypes.c_uint32, POINTER(c_void_p), POINTER(c_size_t)]
        out = c_void_p(); sz = c_size_t()
        if L.flbgpu_sp_timer_all_reduce(self.h, comm.h, None, now[0], now[1], byref(out), byref(sz)) != 0:
            raise RuntimeError(last_error())
        return self._take(out, sz)

    def profile(self, enable=True):
        ms = (c_double * 2)(); n = (c_uint64 * 2)()
        lib().flbgpu_sp_profile(self.h, int(bool(enable)), ms, n)
        if self.select_only:
            return {"k_sp_select<size>": (ms[0], int(n[0])), "k_sp_select<emit>": (ms[1], int(n[1]))}
        return {"k_sp_extract": (ms[0], int(n[0])), "k_sp_aggregate": (ms[1], int(n[1]))}

    def close(self):
        if self.h:
            lib().flbgpu_sp_destroy(self.h)
            self.h = None


def msgpack_to_json_format(data, json_format, date_format, date_key, escape_unicode=1, nan_to_null=0):
    """the one-shot C entry with the reference's argument list"""
    out = c_void_p(); sz = c_size_t()
    r = lib().flbgpu_pack_msgpack_to_json_format(data, len(data), json_format, date_format, date_key, -1 if date_key is None else len(date_key),
                                                 escape_unicode, nan_to_null, byref(out), byref(sz))
    if r != 0:
        return None
    b = ctypes.string_at(out, sz.value)
    _libc.free(out)
    return b


class ChainStat(Structure):
    _fields_ = [("ret", c_int), ("in_records", c_uint64), ("out_records", c_uint64), ("out_bytes", c_uint64)]


class FilterChain:
    """flb_filter_do (src/flb_filter.c:121-325) over GPU filters that match the chunk's tag."""

    def __init__(self, filters):
        self.filters = list(filters)
        self.arr = (c_void_p * max(len(self.filters), 1))(*[f.h for f in self.filters])
        self.stats = (ChainStat * max(len(self.filters), 1))()

    def filter(self, data):
        """-> (MODIFIED|NOTOUCH, bytes|None); NOTOUCH means the engine keeps `data`"""
        out = c_void_p(); sz = c_size_t()
        r = lib().flbgpu_filter_chain_run(self.arr, len(self.filters), data, len(data), byref(out), byref(sz), self.stats)
        if r != MODIFIED:
            return r, None
        b = ctypes.string_at(out, sz.value) if sz.value else b""
        if out.value:
            _libc.free(out)
        return r, b

    def filter_dev(self, chunk):
        out = DevChunk()
        r = lib().flbgpu_filter_chain_run_dev(self.arr, len(self.filters), byref(chunk), byref(out), self.stats)
        return r, out

    def last_stats(self):
        return [dict(ret=s.ret, in_records=s.in_records, out_records=s.out_records, out_bytes=s.out_bytes)
                for s in self.stats[: len(self.filters)]]


HOST_PHASES = ("index", "copy_in", "upload_wait", "chain", "download", "malloc", "total")


def host_phases():
    """microseconds of this thread's last host-level call, by phase (flbgpu_host_phases)"""
    ph = (ctypes.c_double * 8)()
    k = lib().flbgpu_host_phases(ph, 8)
    return dict(zip(HOST_PHASES, list(ph)[:k]))


def index_host(data):
    """record boundaries of a host chunk: (n, offsets list[n+1], consumed)"""
    import numpy as np
    consumed = c_size_t()
    for cap in (len(data) // 64 + 1024, len(data) + 2):        # a record can be one byte long
        off = np.zeros(cap, dtype=np.uint64)
        n = lib().flbgpu_index_host(data, len(data), off.ctypes.data, off.size, byref(consumed))
        if n + 1 < cap:
            break
    return n, off[: n + 1], consumed.value


class Indexer:
    """record boundaries of a raw chunk found on the GPU (flbgpu_index_dev)"""

    def __init__(self):
        self.h = lib().flbgpu_indexer_create()
        if not self.h:
            raise RuntimeError(last_error())

    def __del__(self):
        if getattr(self, "h", None) and _L is not None:
            _L.flbgpu_indexer_destroy(self.h)
            self.h = None

    def index_dev(self, dev_ptr, nbytes):
        """-> (DevChunk with device row offsets, consumed)"""
        ch = DevChunk()
        consumed = c_size_t()
        n = lib().flbgpu_index_dev(self.h, dev_ptr, nbytes, byref(ch), byref(consumed))
        if n < 0:
            raise RuntimeError(last_error())
        return ch, consumed.value

    def index(self, data):
        """host bytes -> (n, offsets[n + 1] as numpy u64, consumed); uploads, indexes on the device, reads back"""
        import numpy as np
        L = lib()
        d = L.flbgpu_dev_alloc(len(data) + 16)
        if not d:
            raise RuntimeError(last_error())
        try:
            L.flbgpu_memcpy_h2d(d, data, len(data))
            ch, consumed = self.index_dev(d, len(data))
            off = np.zeros(ch.n + 1, dtype=np.uint64)
            if len(data):
                L.flbgpu_memcpy_d2h(off.ctypes.data, ch.row_off, off.nbytes)
            return int(ch.n), off, consumed
        finally:
            L.flbgpu_dev_free(d)

    def stats(self):
        a, b, c = c_uint64(), c_uint64(), c_uint64()
        lib().flbgpu_indexer_stats(self.h, byref(a), byref(b), byref(c))
        return dict(candidates=a.value, off_chain_rows=b.value, rounds=c.value)


# ---- filter_log_to_metrics ---------------------------------------------------------------------
COUNTER, GAUGE, HISTOGRAM = 0, 1, 2
# word indices of a series row (csrc/dev.hpp): the first two merge by max, word 2 follows word 1,
# everything else adds
W_FIRST, W_LASTIDX, W_LASTVAL, W_COUNT = 0, 1, 2, 3


def finalize_row(mode, nbuckets, row):
    """one (merged) series row -> dict(value=, buckets=[cumulative.., +Inf], count=, sum=)"""
    import numpy as np
    row = np.ascontiguousarray(row, dtype=np.uint64)
    v = c_double(); cnt = c_uint64(); sm = c_double()
    bk = np.zeros(nbuckets + 1, dtype=np.uint64)
    lib().flbgpu_l2m_finalize_row(mode, nbuckets, row.ctypes.data, byref(v), bk.ctypes.data, byref(cnt), byref(sm))
    return dict(value=v.value, buckets=[int(x) for x in bk], count=cnt.value, sum=sm.value)


class FilterLogToMetrics(_Filter):
    """filter_log_to_metrics (plugins/filter_log_to_metrics/log_to_metrics.c).  `props` = the
    instance properties in configuration order as (key, value): regex / exclude / label_field /
    add_label / bucket; the scalar options keep their property names."""

    def __init__(self, metric_mode="counter", props=(), kubernetes_mode=False, value_field=None, discard_logs=False):
        n = len(props)
        keys = (c_char_p * max(n, 1))(*[_b(k) for k, _ in props])
        vals = (c_char_p * max(n, 1))(*[_b(v) for _, v in props])
        self.h = lib().flbgpu_filter_l2m_create(_b(metric_mode), n, keys, vals, int(kubernetes_mode), _b(value_field),
                                                int(discard_logs))
        if not self.h:
            raise ValueError("flbgpu_filter_l2m_create: " + last_error())
        m = c_int(); lc = c_int(); nb = c_int(); w = c_int()
        lib().flbgpu_l2m_info(self.h, byref(m), byref(lc), byref(nb), byref(w))
        self.mode, self.label_count, self.nbuckets, self.row_words = m.value, lc.value, nb.value, w.value
        self.label_keys = [lib().flbgpu_l2m_label_key(self.h, i).decode("latin1") for i in range(lc.value)]
        b = (c_double * max(nb.value, 1))()
        lib().flbgpu_l2m_bounds(self.h, b)
        self.bounds = list(b[: nb.value])
        self.sum_order = 1                  # the C ABI's default: the reference's order

    def set_sum_order(self, reference=True):
        """sum_order reference: also keep the histogram sum as the reference adds it up (one f64 addition per observation in record order);
        reference=2: the same across ranks -- the interval's observations are kept and the flush folds them rank after rank (l2m_chain)"""
        if lib().flbgpu_l2m_set_sum_order(self.h, 2 if reference == 2 else int(bool(reference))) != 0:
            raise RuntimeError(last_error())
        self.sum_order = 2 if reference == 2 else int(bool(reference))

    def _keys_args(self, keys):
        import numpy as np
        off = np.zeros(len(keys) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(k) for k in keys]) if keys else []
        return off, b"".join(keys)

    def chain_begin(self, keys):
        """the sums the last flush ended on, for the union of the label tuples `keys` (0.0 for a new one)"""
        off, blob = self._keys_args(keys)
        sums = (c_double * max(len(keys), 1))()
        L = lib()
        L.flbgpu_l2m_chain_begin.argtypes = [c_void_p, c_uint64, c_void_p, c_char_p, POINTER(c_double)]
        if L.flbgpu_l2m_chain_begin(self.h, len(keys), off.ctypes.data, blob, sums) != 0:
            raise RuntimeError(last_error())
        return [sums[i] for i in range(len(keys))]

    def seq_replay(self, keys, sums):
        """this rank's turn of the chain: its interval's observations are added, in record order, to `sums` (one per key)"""
        off, blob = self._keys_args(keys)
        buf = (c_double * max(len(keys), 1))(*sums)
        L = lib()
        L.flbgpu_l2m_seq_replay.argtypes = [c_void_p, c_uint64, c_void_p, c_char_p, POINTER(c_double)]
        if L.flbgpu_l2m_seq_replay(self.h, len(keys), off.ctypes.data, blob, buf) != 0:
            raise RuntimeError(last_error())
        return [buf[i] for i in range(len(keys))]

    def chain_end(self, keys, sums):
        off, blob = self._keys_args(keys)
        buf = (c_double * max(len(keys), 1))(*sums)
        L = lib()
        L.flbgpu_l2m_chain_end.argtypes = [c_void_p, c_uint64, c_void_p, c_char_p, POINTER(c_double)]
        if L.flbgpu_l2m_chain_end(self.h, len(keys), off.ctypes.data, blob, buf) != 0:
            raise RuntimeError(last_error())

    def chain_sums(self):
        """the sums of the last flush with sum_order 2 (flbgpu_l2m_all_reduce), in the order of its output"""
        cap = 1 << 16
        buf = (c_double * cap)()
        L = lib()
        L.flbgpu_l2m_chain_sums.restype = c_int64
        L.flbgpu_l2m_chain_sums.argtypes = [c_void_p, c_uint64, POINTER(c_double)]
        n = L.flbgpu_l2m_chain_sums(self.h, cap, buf)
        if n < 0:
            raise RuntimeError("no chain sums: %s" % last_error())
        return [buf[i] for i in range(n)]

    def seq_sums(self):
        """the reference-order sums of the series, in snapshot() / export() order"""
        cap = 1 << 16
        buf = (c_double * cap)()
        n = lib().flbgpu_l2m_seq_sums(self.h, cap, buf)
        if n < 0:
            raise RuntimeError("no reference-order sums: %s" % last_error())
        return [buf[i] for i in range(n)]

    def set_index_base(self, base):
        lib().flbgpu_l2m_set_index_base(self.h, base)

    def stats(self):
        o = (c_uint64 * 5)()
        lib().flbgpu_l2m_stats(self.h, o)
        return dict(observations=o[0], deferred=o[1], stale=o[2], grows=o[3], slots=o[4])

    def export(self):
        """-> (keys: list[bytes], rows: np.uint64[n, row_words]) in first-appearance order; a key is the
        series' label values, each NUL-terminated"""
        import numpy as np
        cap, kcap = 1024, 1 << 16
        while True:
            rows = np.zeros((cap, self.row_words), dtype=np.uint64)
            off = np.zeros(cap + 1, dtype=np.uint64)
            keys = ctypes.create_string_buffer(kcap)
            need = c_size_t()
            n = lib().flbgpu_l2m_export(self.h, cap, rows.ctypes.data, off.ctypes.data, keys, kcap, byref(need))
            if n >= 0:
                raw = keys.raw
                return [raw[int(off[i]): int(off[i + 1])] for i in range(n)], rows[:n].copy()
            if n == -1:
                raise RuntimeError("flbgpu_l2m_export: " + last_error())
            cap = max(cap, -n - 2)
            kcap = max(kcap, need.value)

    def snapshot(self, keys_rows=None):
        """-> list of dict(labels=(bytes,..), value=, buckets=, count=, sum=, sum_exact=) in first-appearance order.  A histogram's `sum`
        is what its sum_order says: the reference's sequential sum (the default) for the filter's own state, the exact sum rounded
        once otherwise (merged rows of several ranks carry only that one; sum_order 2: chain_sums()); `sum_exact` is always the latter"""
        own = keys_rows is None
        keys, rows = keys_rows if keys_rows is not None else self.export()
        seq = self.seq_sums() if (own and self.mode == 2 and self.sum_order == 1) else None
        out = []
        for j, (k, r) in enumerate(zip(keys, rows)):
            d = finalize_row(self.mode, self.nbuckets, r)
            d["sum_exact"] = d["sum"]
            if seq is not None and j < len(seq):
                d["sum"] = seq[j]
            d["labels"] = tuple(k.split(b"\0")[:-1]) if self.label_count else ()
            out.append(d)
        return out


class RcclComm:
    """an ncclComm_t made through libflbgpu (flbgpu_rccl_unique_id / flbgpu_rccl_comm_init): rank 0 creates the id,
    `exchange(id_bytes) -> id_bytes` ships it to the other ranks (e.g. a torch.distributed broadcast)"""

    def __init__(self, nranks, rank, exchange=None):
        L = lib()
        L.flbgpu_rccl_unique_id.argtypes = [c_void_p]
        L.flbgpu_rccl_comm_init.argtypes = [POINTER(c_void_p), c_int, c_void_p, c_int]
        L.flbgpu_rccl_comm_destroy.argtypes = [c_void_p]
        buf = ctypes.create_string_buffer(128)
        if rank == 0 and L.flbgpu_rccl_unique_id(buf) != 0:
            raise RuntimeError("flbgpu_rccl_unique_id: " + last_error())
        raw = buf.raw
        if exchange is not None:
            raw = exchange(raw)
        self.h = c_void_p()
        idb = ctypes.create_string_buffer(raw, 128)
        if L.flbgpu_rccl_comm_init(byref(self.h), nranks, idb, rank) != 0:
            raise RuntimeError("flbgpu_rccl_comm_init: " + last_error())
        self.nranks, self.rank = nranks, rank

    def close(self):
        if self.h:
            lib().flbgpu_rccl_comm_destroy(self.h)
            self.h = None


def l2m_all_reduce_rccl(flt, comm):
    """flbgpu_l2m_all_reduce: the merge of every rank's FilterLogToMetrics over RCCL, inside libflbgpu.so.
    -> (keys, rows) like FilterLogToMetrics.export(), identical on every rank"""
    import numpy as np
    L = lib()
    L.flbgpu_l2m_all_reduce.restype = c_int64
    L.flbgpu_l2m_all_reduce.argtypes = [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_size_t, POINTER(c_size_t)]
    cap, kcap = 1024, 1 << 16
    while True:
        rows = np.zeros((cap, flt.row_words), dtype=np.uint64)
        off = np.zeros(cap + 1, dtype=np.uint64)
        keys = ctypes.create_string_buffer(kcap)
        need = c_size_t(0)
        n = L.flbgpu_l2m_all_reduce(flt.h, comm.h, None, cap, rows.ctypes.data, off.ctypes.data, keys, kcap, byref(need))
        if n >= 0:
            raw = keys.raw
            return [raw[int(off[i]): int(off[i + 1])] for i in range(n)], rows[:n].copy()
        if n == -1:
            raise RuntimeError("flbgpu_l2m_all_reduce: " + last_error())
        cap = max(cap, -n - 2)
        kcap = max(kcap, need.value)


def l2m_all_reduce(flt, dist, device=None):
    """Merges the series state of every rank's FilterLogToMetrics (same configuration on all ranks, each
    fed its own shard of the records) -- SURVEY.md section 8e: one all-reduce of the partial
    aggregates per flush.  Label dictionaries are made identical first (all-gather of the keys);
    then the rows go through all_reduce: MAX for the two index words, SUM for the counts, bucket
    counts and fixed-point sum digits (exact integers, so the merged result does not depend on the
    rank count), and the gauge value is taken from the rank that owns the winning index.
    `dist` is torch.distributed (backend nccl == RCCL over xGMI on the GPUs, gloo in the CPU tests).
    Returns (keys, rows) as FilterLogToMetrics.export() does, identical on every rank."""
    keys, rows = flt.export()
    return l2m_merge(keys, rows, flt.row_words, dist, device)


def l2m_chain(keys, dist, rank_state, device=None):
    """sum_order 2 over torch.distributed: the histogram sums as ONE reference process builds them that is fed the interval's records
    of rank 0, then of rank 1, ... (lib/cmetrics/src/cmt_metric_histogram.c:124-137 adds in record order: no merge of per-rank sums
    has those bits).  `keys`: the merged label tuples (l2m_merge's), the same list on every rank; `rank_state`: this rank's
    FilterLogToMetrics -- or anything with chain_begin(keys) -> sums, seq_replay(keys, sums) -> sums, chain_end(keys, sums).  Rank
    after rank takes its turn on the sums the rank in front ended on (one broadcast per rank).  -> the interval's sums, one per key."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    G = torch.tensor(rank_state.chain_begin(keys), dtype=torch.float64, device=dev)
    for r in range(world):
        if r == rank:
            G = torch.tensor(rank_state.seq_replay(keys, G.cpu().tolist()), dtype=torch.float64, device=dev)
        if world > 1 and len(keys):
            dist.broadcast(G, src=r)
    sums = G.cpu().tolist()
    rank_state.chain_end(keys, sums)
    return sums


def l2m_merge(keys, rows, W, dist, device=None):
    """the collective part of l2m_all_reduce on an exported (keys, rows) pair"""
    import numpy as np
    import torch
    world = dist.get_world_size()
    gathered = [None] * world
    dist.all_gather_object(gathered, keys)
    # union of the label tuples; canonical order fixed after the reduce (first appearance)
    index = {}
    for ks in gathered:
        for k in ks:
            if k not in index:
                index[k] = len(index)
    n = len(index)
    dense = np.zeros((n, W), dtype=np.uint64)
    for k, r in zip(keys, rows):
        dense[index[k]] = r
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    t = torch.from_numpy(dense.view(np.int64)).to(dev)
    # max words: shift to signed order so that MAX on int64 is MAX on uint64
    mx = t[:, :2].contiguous() ^ torch.tensor(-2 ** 63, dtype=torch.int64, device=dev)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    mx = mx ^ torch.tensor(-2 ** 63, dtype=torch.int64, device=dev)
    # the gauge value travels with the winning index: ranks that lost contribute 0
    own = (t[:, 1] == mx[:, 1]) & (t[:, 1] != 0)
    t[:, 2] = torch.where(own, t[:, 2], torch.zeros_like(t[:, 2]))
    sm = t[:, 2:].contiguous()
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    # two ranks can only tie on W_LASTIDX if they were given overlapping index ranges
    merged = torch.cat([mx, sm], dim=1).cpu().numpy().view(np.uint64)
    order = np.argsort(~merged[:, 0], kind="stable")
    inv = sorted(index, key=index.get)
    return [inv[i] for i in order], merged[order]


# ---- JSON -> msgpack -----------------------------------------------------------------------------
def pack_json(js):
    """flb_pack_json_recs (src/flb_pack.c:683-688): -> (ret, msgpack bytes | None, root_type, records, consumed)"""
    out = c_void_p(); sz = c_size_t(); rt = c_int(0); rec = c_int(0); cons = c_size_t(0)
    r = lib().flbgpu_pack_json_recs(js, len(js), byref(out), byref(sz), byref(rt), byref(rec), byref(cons))
    if r != 0:
        return (r, None, 0, 0, 0)
    data = ctypes.string_at(out, sz.value) if out.value else b""
    if out.value:
        _libc.free(out)
    return (0, data, rt.value, rec.value, cons.value)


def split_lines(data):
    """row offsets (np.uint64[n+1]) of an NDJSON buffer, one row per line"""
    import numpy as np
    off = np.zeros(data.count(b"\n") + 3, dtype=np.uint64)
    n = lib().flbgpu_split_lines_host(data, len(data), off.ctypes.data, off.size)
    assert n >= 0
    return off[: n + 1]


class JsonPacker:
    """batched flb_pack_json: rows of JSON text in HBM -> rows of msgpack (or V2 log events) in HBM"""

    def __init__(self):
        self.h = lib().flbgpu_json_create()
        if not self.h:
            raise RuntimeError("flbgpu_json_create: " + last_error())

    def run_dev(self, chunk, events=False, ts=(0, 0)):
        out = DevChunk()
        if lib().flbgpu_json_run_dev(self.h, byref(chunk), int(events), ts[0], ts[1], byref(out)) != 0:
            raise RuntimeError("flbgpu_json_run_dev: " + last_error())
        return out

    def row_info(self, n):
        import numpy as np
        rec = np.zeros(n, dtype=np.uint32); cons = np.zeros(n, dtype=np.uint32)
        rt = np.zeros(n, dtype=np.uint8); st = np.zeros(n, dtype=np.uint8)
        if n and lib().flbgpu_json_row_info(self.h, 0, n, rec.ctypes.data, cons.ctypes.data, rt.ctypes.data, st.ctypes.data) != 0:
            raise RuntimeError(last_error())
        return rec, cons, rt, st

    def stats(self):
        o = (c_uint64 * 3)()
        lib().flbgpu_json_stats(self.h, o)
        return dict(generic_rows=o[0], values=o[1], error_rows=o[2])

    def tile_stats(self):
        o = (c_uint64 * 4)()
        lib().flbgpu_json_tile_stats(self.h, o)
        return dict(tile_rows=o[0], left_rows=o[1], launches=o[2], tokens=o[3])

    def tile_debug(self, prof=False, no_lookback=False):
        """measurement only: phase stamps / no look-back for the next runs; returns the last run's phase cycles"""
        o = (c_uint64 * 8)()
        lib().flbgpu_json_tile_debug(self.h, int(prof), int(no_lookback), o)
        return list(o)

    def run_host(self, rows, events=False, ts=(0, 0)):
        """rows: list of bytes -> (list of output rows, records, consumed, root_type, status)"""
        import numpy as np
        blob = b"".join(rows)
        off = np.zeros(len(rows) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(r) for r in rows])
        L = lib()
        d_data = L.flbgpu_dev_alloc(len(blob) + 16); d_off = L.flbgpu_dev_alloc(off.nbytes)
        L.flbgpu_memcpy_h2d(d_data, blob, len(blob)); L.flbgpu_memcpy_h2d(d_off, off.ctypes.data, off.nbytes)
        try:
            out = self.run_dev(DevChunk(d_data, d_off, len(rows), len(blob)), events, ts)
            ooff = np.zeros(len(rows) + 1, dtype=np.uint64)
            L.flbgpu_memcpy_d2h(ooff.ctypes.data, out.row_off, ooff.nbytes)
            buf = ctypes.create_string_buffer(int(out.bytes) + 1)
            if out.bytes:
                L.flbgpu_memcpy_d2h(buf, out.data, int(out.bytes))
            raw = buf.raw
            outs = [raw[int(ooff[i]): int(ooff[i + 1])] for i in range(len(rows))]
            return (outs,) + self.row_info(len(rows))
        finally:
            L.flbgpu_dev_free(d_data); L.flbgpu_dev_free(d_off)

    def close(self):
        if self.h:
            lib().flbgpu_json_destroy(self.h)
            self.h = None
