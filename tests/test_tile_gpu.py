"""filter_parser's pass 1 in its three builds -- the single pass with the value in registers (k_parser_reg, the
default), the single pass over an LDS tile (k_parser_tile), the phase kernels (locate / rx / finish) -- against the CPU
oracle on inputs chosen for the places where they differ: the end-of-text sentinel (a real 0xFF in the text, empty
values, values that end on every byte of a group), values longer than the registers hold, lines the forward walk from
boundary 0 cannot settle, bodies that are not the one-key layout, bad events; alone and as the pair with filter_grep."""
import os, random
import pytest
import oracle_binding as ob
import synth
import flbamd_loader

pytestmark = pytest.mark.gpu
APACHE2 = r'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>.*)")?$'
APACHE = r'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^\"]*?)(?: +\S*)?)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>[^\"]*)")?$'
TF = "%d/%b/%Y:%H:%M:%S %z"
MODES = {"reg": {}, "tile": {"FLBGPU_TILE_MODE": "tile"}, "phase": {"FLBGPU_NO_TILE": "1"}}


@pytest.fixture(scope="module")
def g():
    m = flbamd_loader.load()
    m.init(0)
    return m


def _set_mode(mode):
    for k in ("FLBGPU_TILE_MODE", "FLBGPU_NO_TILE"):
        os.environ.pop(k, None)
    os.environ.update(MODES[mode])


def _rec(body, sec=1, nsec=0, meta=None):
    return synth.v2_record(sec, nsec, body, meta)


def _hostile_chunk(seed, n=3000, map32_first=False):
    rng = random.Random(seed)
    data, off, _ = synth.apache_records(n)
    blob = bytes(data)
    lines = [blob[int(off[i]) + 21:int(off[i + 1])] for i in range(n)]
    out = []
    for i, ln in enumerate(lines):
        r = rng.random()
        m = bytearray(ln)
        if r < 0.05:
            m[rng.randrange(len(m))] = 0xFF                          # the sentinel's own byte value inside the text
        elif r < 0.10:
            m[rng.randrange(len(m))] = rng.choice(b"\xe9\x80\xc3")   # other bytes >= 0x80
        elif r < 0.15:
            m = m[: rng.randrange(len(m))]                           # cut anywhere: most of these do not match
        elif r < 0.20:
            k = rng.randrange(len(m)); m[k:k] = b"x" * rng.randrange(1, 600)    # values far beyond 272 bytes
        elif r < 0.23:
            m = bytearray(b"")                                       # empty value
        elif r < 0.26:
            m = bytearray(ln.replace(b'"GET ', b'"GET  ', 1))
        elif r < 0.29:
            m += b"\n" + ln                                          # a second line: ^ / $ are line anchors
        elif r < 0.32:
            m = bytearray(b"\n") + m                                 # the match starts behind the first byte
        elif r < 0.35:
            m = m[: 250 + rng.randrange(0, 30)]                      # ends around the register windows' edge
        body = {"log": bytes(m)}
        r2 = rng.random()
        if r2 < 0.04:
            rec = _rec(synth.KV([("stream", "stdout"), ("log", bytes(m)), ("n", i)]))                # not the one-key layout
        elif r2 < 0.06:
            rec = _rec(synth.KV([("log", b"first"), ("log", bytes(m))]))                             # two candidates
        elif r2 < 0.08:
            rec = synth.legacy_record(1700000000 + i, body)                                          # legacy event
        elif r2 < 0.09:
            rec = _rec(body, meta={"k": "v"})                                                        # metadata
        elif r2 < 0.10:
            rec = _rec({"log": i})                                                                   # value is not a string
        elif r2 < 0.105:
            rec = synth.mp([[synth.ext_ts(0xffffffff, 0), {}], {"g": 1}])                            # group marker
        elif r2 < (0.16 if not map32_first else 0.9):
            # in_tail's layout (begin_record + append_body_values: 32-bit map headers, tail_file.c:552-604)
            v = bytes(m)
            sh = bytes([0xa0 | len(v)]) if len(v) < 32 else b"\xd9" + bytes([len(v)]) if len(v) < 256 else b"\xda" + len(v).to_bytes(2, "big") if len(v) < 65536 else b"\xdb" + len(v).to_bytes(4, "big")
            rec = b"\x92\x92\xd7\x00" + (1700000000 + i).to_bytes(4, "big") + (i % 1000).to_bytes(4, "big") + b"\xdf\x00\x00\x00\x00\xdf\x00\x00\x00\x01\xa3log" + sh + v
        else:
            rec = _rec(body, sec=1700000000 + i, nsec=i % 1000)
        out.append(rec)
    return b"".join(out)


@pytest.mark.parametrize("regex", [APACHE2, APACHE])
def test_three_builds_against_the_oracle(g, regex):
    chunk = _hostile_chunk(7 if regex is APACHE2 else 8)
    tailish = _hostile_chunk(17, 1500, map32_first=True)               # mostly in_tail's layout, the first record too
    bad_tail = chunk + b"\x92\x92\xd7\x00"                          # the decoder stops at a broken last event
    pargs = dict(regex=regex, time_fmt=TF, time_key="time")
    rule_sets = [([("regex", r"code ^5\d\d$")], None), ([("exclude", "method GET")], None),
                 ([("regex", "code ^2"), ("regex", "agent curl")], "AND"), ([("regex", "time 2024")], None),
                 ([("regex", "log x")], None)]
    for data in (chunk, bad_tail, chunk[:277 * 5], tailish):
        po = ob.Parser(**pargs)
        want_p = ob.FilterParser("log", [po]).filter(data)
        want_pairs = []
        for rules, op in rule_sets:
            # flb_filter_do (src/flb_filter.c:121-325): a NOTOUCH filter hands its input on; MODIFIED with nothing left ends the chain
            inp = want_p[1] if want_p[0] == ob.MODIFIED else data
            if want_p[0] == ob.MODIFIED and not inp:
                want_pairs.append((want_p[0], ob.NOTOUCH, b""))
                continue
            r2, o2 = ob.Grep(rules, op).filter(inp)
            want_pairs.append((want_p[0], r2, o2 if r2 == ob.MODIFIED else inp))
        for mode in MODES:
            _set_mode(mode)
            p = g.Parser(**pargs)
            fp = g.FilterParser("log", [p])
            got = fp.filter(data)
            assert got[0] == want_p[0] and got[1] == want_p[1], (mode, "parser", got[0], want_p[0])
            for (rules, op), want in zip(rule_sets, want_pairs):
                fg = g.FilterGrep(rules, op)
                ch = g.FilterChain([fp, fg])
                r3, o3 = ch.filter(data)
                exp_ret = ob.MODIFIED if ob.MODIFIED in (want[0], want[1]) else ob.NOTOUCH
                assert r3 == exp_ret, (mode, rules, r3, want[0], want[1])
                if exp_ret == ob.MODIFIED:
                    assert (o3 or b"") == (want[2] or b""), (mode, rules, len(o3 or b""), len(want[2] or b""))
                fg.close()
            fp.close(); p.close()
    _set_mode("reg")


def test_three_builds_with_keep_options(g):
    """Time_Keep / Reserve_Data / Preserve_Key / Types: k_parser_finish takes these rows over from the single pass"""
    chunk = _hostile_chunk(9, 800)
    for extra, reserve, preserve in [(dict(time_keep=True), False, False), (dict(), True, False), (dict(), True, True),
                                     (dict(types="code:integer size:integer"), False, False),
                                     (dict(time_fmt="%d/%b/%Y:%H:%M:%S"), False, False)]:
        pargs = dict(regex=APACHE2, time_fmt=TF, time_key="time")
        pargs.update(extra)
        want = ob.FilterParser("log", [ob.Parser(**pargs)], reserve, preserve).filter(chunk)
        for mode in MODES:
            _set_mode(mode)
            p = g.Parser(**pargs)
            fp = g.FilterParser("log", [p], reserve, preserve)
            got = fp.filter(chunk)
            assert got[0] == want[0] and got[1] == want[1], (mode, extra, reserve, preserve)
            fp.close(); p.close()
    _set_mode("reg")


def test_empty_fields_and_the_two_pair_table_forms(g):
    """round 5: the three-port pair tables (fx5, the default) send a record with two capture writes at one even position -- an EMPTY
    field, `""` -- to the generic kernel; a filter that sees more than 1 row in 64 go that way takes the four-port tables (fx4) from its
    next call on.  Same bytes as the oracle before and after the switch, and with either form forced."""
    rng = random.Random(23)
    data, off, _ = synth.apache_records(4000)
    blob = bytes(data)
    lines = [blob[int(off[i]) + 21:int(off[i + 1])] for i in range(4000)]
    recs = []
    for i, ln in enumerate(lines):
        r = rng.random()
        if r < 0.3:
            # empty referer / agent / user / size, at every alignment (a pad of 0 .. 3 bytes in front)
            q = ln.split(b'"')
            if len(q) >= 6:
                if rng.random() < 0.6: q[3] = b""
                if rng.random() < 0.3: q[5] = b""
            ln = b'"'.join(q)
            ln = ln.replace(b" - - [", b" -  [", 1) if rng.random() < 0.3 else ln
            ln = b"1" * rng.randrange(0, 4) + ln
        recs.append(_rec({"log": ln}, sec=1700000000 + i, nsec=i))
    chunk = b"".join(recs)
    pargs = dict(regex=APACHE2, time_fmt=TF, time_key="time")
    want = ob.FilterParser("log", [ob.Parser(**pargs)]).filter(chunk)
    rules = [("regex", r"code ^[45]\d\d$")]
    want2 = ob.Grep(rules).filter(want[1])
    _set_mode("reg")
    for env in ({}, {"FLBGPU_FX": "4"}, {"FLBGPU_FX": "3"}):
        os.environ.pop("FLBGPU_FX", None)
        os.environ.update(env)
        p = g.Parser(**pargs)
        fp = g.FilterParser("log", [p]); fg = g.FilterGrep(rules); ch = g.FilterChain([fp, fg])
        for call in range(3):                                   # (the default form switches after its first call on this data)
            got = fp.filter(chunk)
            assert got == want, (env, call, "parser")
            r3, o3 = ch.filter(chunk)
            assert r3 == ob.MODIFIED and o3 == want2[1], (env, call, "pair")
        fg.close(); fp.close(); p.close()
    os.environ.pop("FLBGPU_FX", None)


def test_choices_between_builds_are_not_for_good(g):
    """round 6: the choices a filter_parser instance makes between a fast build and the one that takes everything (single pass / phase
    kernels, three / four write ports) are made per call from what the last calls showed (host_int.hpp Probe) -- round 5 latched them: one
    odd chunk moved the filter to the slower build for the life of the process.  Plain chunks, then chunks that make each fast build do
    badly, then plain chunks again: the filter sets the build aside, tries it again after 16 calls and comes back; the oracle's bytes on
    every call."""
    rng = random.Random(5)
    data, off, _ = synth.apache_records(3000)
    blob = bytes(data)
    lines = [blob[int(off[i]) + 21:int(off[i + 1])] for i in range(3000)]
    plain = b"".join(_rec({"log": ln}, sec=1700000000 + i, nsec=i) for i, ln in enumerate(lines))
    # (a) empty referers: the three-port tables hand such lines on (two capture writes at one position)
    odd = []
    for i, ln in enumerate(lines):
        q = ln.split(b'"')
        if i % 3 == 0 and len(q) >= 6: q[3] = b""
        odd.append(_rec({"log": b'"'.join(q)}, sec=1700000000 + i, nsec=i))
    odd = b"".join(odd)
    # (b) bodies of several keys: the single pass leaves them to its fix-up launch (more than one row in four: the phase kernels' data)
    multi = b"".join(_rec({"a": "x", "log": ln, "z": 1} if i % 2 else {"log": ln}, sec=1700000000 + i, nsec=i) for i, ln in enumerate(lines))
    pargs = dict(regex=APACHE2, time_fmt=TF, time_key="time")
    rules = [("regex", r"code ^[45]\d\d$")]
    po = ob.Parser(**pargs)
    want = {}
    for name, chunk in (("plain", plain), ("odd", odd), ("multi", multi)):
        w1 = ob.FilterParser("log", [po]).filter(chunk)
        want[name] = (w1, ob.Grep(rules).filter(w1[1]))
    _set_mode("reg")
    os.environ.pop("FLBGPU_FX", None)
    p = g.Parser(**pargs)
    fp = g.FilterParser("log", [p]); fg = g.FilterGrep(rules); ch = g.FilterChain([fp, fg])
    chunks = {"plain": plain, "odd": odd, "multi": multi}

    def call(name):
        r3, o3 = ch.filter(chunks[name])
        assert r3 == ob.MODIFIED and o3 == want[name][1][1], name
        return fp.paths()

    st = call("plain")
    assert st["single_pass"] and st["three_port"] and not any(st["aside"].values()), st
    st = call("odd")                                            # walked with three ports, handed on a third of the rows
    assert st["aside"]["three_port"] and not st["aside"]["single_pass"], st
    st = call("odd")
    assert st["single_pass"] and not st["three_port"], st      # the four-port tables, still the single pass
    seen_back = None
    for k in range(40):
        st = call("plain")
        if st["three_port"] and not st["aside"]["three_port"]:
            seen_back = k
            break
    assert seen_back is not None and 10 <= seen_back <= 20, (seen_back, st)
    st = call("multi")
    assert st["aside"]["single_pass"], st
    st = call("multi")
    assert not st["single_pass"], st                           # the phase kernels
    seen_back = None
    for k in range(40):
        st = call("plain")
        if st["single_pass"] and not st["aside"]["single_pass"]:
            seen_back = k
            break
    assert seen_back is not None and 10 <= seen_back <= 20, (seen_back, st)
    assert st["returns"] >= 2 and st["tries"] >= 2, st
    # data that stays odd: the tries come at growing distances
    t0 = fp.paths()["tries"]
    for k in range(60):
        call("odd")
    assert 2 <= fp.paths()["tries"] - t0 <= 3, fp.paths()
    fg.close(); fp.close(); p.close()


def test_time_lookup_left_to_the_emit_pass(g):
    """round 5 (dev.hpp TileCfg::defer_time): the pair's single pass leaves the time lookup of a fixed-layout Time_Format to k_pg_emit,
    which only sees the records grep keeps.  Everything a time text can change is compared with the oracle's two filters: the kept
    records' timestamps, a record the encoder refuses because its parsed time is out of range (it never reaches grep: the parser's
    record / byte counts of flb_filter_do), a text the plan does not settle (strptime's own reading: one-digit day, full month name,
    or no time at all), an event whose OWN time is bad and whose parsed time repairs it -- among the records grep keeps (the emit
    pass reports, the call is repeated with the lookup in the single pass) and among those it drops (the year test of the single
    pass).  Same answers with the lookup forced into the single pass (FLBGPU_DEFER_TIME=0)."""
    import re
    rng = random.Random(31)
    data, off, _ = synth.apache_records(3000)
    blob = bytes(data)
    lines = [blob[int(off[i]) + 21:int(off[i + 1])] for i in range(3000)]
    odd = [b"10/Mar/1960:08:34:03 +0900", b"10/Mar/2150:08:34:03 +0900", b"10/Foo/2024:08:34:03 +0900", b"5/March/2024:8:34:03 +0900",
           b"31/Dec/1969:23:59:59 +0000", b"01/Jan/1970:00:00:00 +0100", b"10/Mar/2024:08:34:03 +09x0", b"10/Mar/20x4:08:34:03 +0900",
           b"10/Mar/2106:08:34:03 +0900", b"07/Feb/2106:06:28:15 +0000", b"07/Feb/2106:06:28:16 +0000", b"                          ", b"-"]
    pargs = dict(regex=APACHE2, time_fmt=TF, time_key="time")
    rules = [("regex", r"code ^5\d\d$")]

    markers = (b"31/Dec/1969:23:59:59 +0000", b"07/Feb/2106:06:28:15 +0000")    # -1 s and 0xFFFFFFFF s: group markers to the NEXT decoder

    def build(where, with_markers):
        """where: 'dropped' / 'kept' / 'both' -- which records get the odd time texts"""
        odd_ = [t for t in odd if with_markers or t not in markers]
        recs = []
        for i, ln in enumerate(lines):
            is5 = re.search(rb'" 5\d\d ', ln) is not None
            sec, nsec = 1700000000 + i, i
            if rng.random() < 0.04 and (where == "both" or (where == "kept") == is5):
                ln = re.sub(rb"\[[^\]]*\]", b"[" + rng.choice(odd_) + b"]", ln, count=1)
                if rng.random() < 0.3:
                    sec, nsec = 0xFFFFFFFF, 0                        # (an event time the encoder refuses, next to a parsed one that may not be)
            recs.append(_rec({"log": ln}, sec=sec, nsec=nsec))
        return b"".join(recs)

    _set_mode("reg")
    for where, with_markers in (("dropped", False), ("kept", False), ("both", False), ("both", True)):
        chunk = build(where, with_markers)
        want1 = ob.FilterParser("log", [ob.Parser(**pargs)]).filter(chunk)
        want2 = ob.Grep(rules).filter(want1[1])
        assert want1[0] == ob.MODIFIED and want2[0] == ob.MODIFIED
        # what flb_filter_do counts behind filter_parser: the records the log event decoder shows (src/flb_filter.c:272, src/flb_mp.c:49-71) --
        # a parsed time of ff ff ff ff / ff ff ff fe seconds is a group marker to it
        nrec, offs, _ = g.index_host(want1[1])
        n1 = sum(1 for i in range(nrec) if want1[1][int(offs[i]) + 4:int(offs[i]) + 8] not in (b"\xff\xff\xff\xff", b"\xff\xff\xff\xfe"))
        assert (n1 < nrec) == with_markers                           # (the corner is in the data when asked for: the pair then takes the unfused kernels)
        for env in (None, "0"):
            os.environ.pop("FLBGPU_DEFER_TIME", None)
            if env is not None:
                os.environ["FLBGPU_DEFER_TIME"] = env
            p = g.Parser(**pargs)
            fp, fg = g.FilterParser("log", [p]), g.FilterGrep(rules)
            ch = g.FilterChain([fp, fg])
            for rep in range(2):                                     # (the second call: after a repeat the filter keeps the lookup in the single pass)
                r3, o3 = ch.filter(chunk)
                assert r3 == g.MODIFIED and o3 == want2[1], (where, env, rep, len(o3), len(want2[1]))
                st = ch.last_stats()
                assert int(st[0]["out_bytes"]) == len(want1[1]), (where, env, rep, int(st[0]["out_bytes"]), len(want1[1]))
                assert int(st[1]["out_bytes"]) == len(want2[1]) and int(st[0]["out_records"]) == int(st[1]["in_records"])
                assert int(st[0]["out_records"]) == n1, (where, env, rep, int(st[0]["out_records"]), n1)
            fg.close(); fp.close(); p.close()
    os.environ.pop("FLBGPU_DEFER_TIME", None)


def test_plain_emit_build_hands_on_what_it_does_not_take(g):
    """round 5: k_pg_emit<PLAIN> (no general writer in it, three workgroups per CU) writes the kept records that carry a descriptor and fit its
    staging area; anything else -- a kept record of 12 KB (larger than the staging area), kept rows the single pass left to k_parser_finish
    (a time text of another length: no descriptor) -- is counted and the general build runs over the chunk.  Same bytes as the oracle, with
    the plain build, and with the general one forced (FLBGPU_EMIT_GENERAL=1)."""
    import re
    rng = random.Random(41)
    data, off, _ = synth.apache_records(2000)
    blob = bytes(data)
    lines = [blob[int(off[i]) + 21:int(off[i + 1])] for i in range(2000)]
    recs = []
    for i, ln in enumerate(lines):
        r = rng.random()
        if r < 0.01:
            ln = ln[:-1] + b"x" * 12000 + b'"'                                   # a 12 KB agent
            ln = re.sub(rb'" \d\d\d ', b'" 503 ', ln, count=1)
        elif r < 0.05:
            ln = re.sub(rb"\[[^\]]*\]", b"[5/Mar/2024:08:34:03 +0900]", ln, count=1)   # another length: the strptime interpreter, no descriptor
            ln = re.sub(rb'" \d\d\d ', b'" 500 ', ln, count=1)
        recs.append(_rec({"log": ln}, sec=1700000000 + i, nsec=i))
    chunk = b"".join(recs)
    pargs = dict(regex=APACHE2, time_fmt=TF, time_key="time")
    rules = [("regex", r"code ^5\d\d$")]
    want1 = ob.FilterParser("log", [ob.Parser(**pargs)]).filter(chunk)
    want2 = ob.Grep(rules).filter(want1[1])
    assert want2[0] == ob.MODIFIED and len(want2[1]) > 200000
    _set_mode("reg")
    for env in (None, "1"):
        os.environ.pop("FLBGPU_EMIT_GENERAL", None)
        if env:
            os.environ["FLBGPU_EMIT_GENERAL"] = env
        p = g.Parser(**pargs)
        fp, fg = g.FilterParser("log", [p]), g.FilterGrep(rules)
        ch = g.FilterChain([fp, fg])
        r3, o3 = ch.filter(chunk)
        assert r3 == g.MODIFIED and o3 == want2[1], (env, len(o3), len(want2[1]))
        # the device-level call (no look-ahead launches): the plain build, then the general one
        import numpy as np
        L = g.lib()
        arr = np.frombuffer(chunk, dtype=np.uint8)
        n, offs, _c = g.index_host(chunk)
        d_data = L.flbgpu_dev_alloc(arr.nbytes + 512); d_off = L.flbgpu_dev_alloc(offs.nbytes)
        L.flbgpu_memcpy_h2d(d_data, arr.ctypes.data, arr.nbytes); L.flbgpu_memcpy_h2d(d_off, offs.ctypes.data, offs.nbytes)
        r4, o4 = ch.filter_dev(g.DevChunk(d_data, d_off, n, arr.nbytes))
        got = np.empty(int(o4.bytes), dtype=np.uint8)
        L.flbgpu_memcpy_d2h(got.ctypes.data, o4.data, int(o4.bytes))
        koff = np.empty(n + 1, dtype=np.uint64)
        L.flbgpu_memcpy_d2h(koff.ctypes.data, o4.row_off, koff.nbytes)
        assert r4 == g.MODIFIED and int(koff[n]) == len(want2[1]) and got.tobytes()[:len(want2[1])] == want2[1], (env, int(o4.bytes), len(want2[1]))
        fg.close(); fp.close(); p.close()
    os.environ.pop("FLBGPU_EMIT_GENERAL", None)


def test_rows_walked_in_the_order_of_their_lengths(g):
    """round 6 (dev.hpp ParserMatchArgs::perm, kernels_perm.hip): the register kernel's walk is position-synchronous, a wave steps as far
    as its longest record; on a chunk whose lines differ a lot in length the next calls take the rows in the order of their lengths.
    Nothing observable changes: the chain's output and every instance's counts are the chunk-order call's, the oracle's on blocks of rows,
    and a chunk of even lines brings the chunk order back."""
    import numpy as np
    import bench
    L = g.lib()
    recs = bench.mixed_shape_records()[:40000]                  # lines of 80 .. 600 bytes, 10 % four-key bodies, 1 % legacy events
    assert sum(len(r) for r in recs) > (9 << 20)                # (more than a call launched ahead of its sizes takes)
    data, off, ep = synth.apache_records(40000)                 # ... and even lines: 256 bytes each

    def upload(blob, sizes):
        o = np.zeros(len(sizes) + 1, dtype=np.uint64)
        np.cumsum(np.asarray(sizes, dtype=np.uint64), out=o[1:])
        d, do = L.flbgpu_dev_alloc(len(blob) + 16), L.flbgpu_dev_alloc(o.nbytes)
        L.flbgpu_memcpy_h2d(d, blob, len(blob)); L.flbgpu_memcpy_h2d(do, o.ctypes.data, o.nbytes)
        return g.DevChunk(d, do, len(sizes), len(blob)), o

    mixed_blob = b"".join(recs)
    mixed, moff = upload(mixed_blob, [len(r) for r in recs])
    even_blob = bytes(data)
    even, eoff = upload(even_blob, np.diff(off))
    _set_mode("reg")
    os.environ.pop("FLBGPU_FX", None); os.environ.pop("FLBGPU_SORT_ROWS", None)
    pargs = dict(regex=APACHE2, time_fmt=TF, time_key="time")
    rules = [("regex", r"code ^[45]\d\d$")]
    p = g.Parser(**pargs)
    fp = g.FilterParser("log", [p]); fg = g.FilterGrep(rules); ch = g.FilterChain([fp, fg])

    def call(chunk):
        r, o = ch.filter_dev(chunk)
        assert r == g.MODIFIED
        out = np.empty(int(o.bytes), dtype=np.uint8)
        L.flbgpu_memcpy_d2h(out.ctypes.data, o.data, int(o.bytes))
        oo = np.empty(int(o.n) + 1, dtype=np.uint64)
        L.flbgpu_memcpy_d2h(oo.ctypes.data, o.row_off, oo.nbytes)
        return bytes(out), oo, ch.last_stats(), fp.paths()

    outs = [call(mixed) for _ in range(4)]
    assert not outs[0][3]["rows_by_length"]                     # the first chunk of such data: chunk order, and the counters say so
    assert all(o[3]["rows_by_length"] for o in outs[1:]), [o[3] for o in outs]
    for o in outs[1:]:
        assert o[0] == outs[0][0] and (o[1] == outs[0][1]).all() and o[2] == outs[0][2]
    # the oracle's two filters on blocks of input rows (the output keeps one row per input row)
    po = ob.Parser(**pargs)
    fo, go = ob.FilterParser("log", [po]), ob.Grep(rules)
    got, goff = outs[2][0], outs[2][1]
    for start in (0, 13000, 39000):
        blob = mixed_blob[int(moff[start]): int(moff[start + 1000])]
        r1, w1 = fo.filter(blob)
        r2, w2 = go.filter(w1 if r1 == ob.MODIFIED else blob)
        want = w2 if r2 == ob.MODIFIED else (w1 if r1 == ob.MODIFIED else blob)
        assert got[int(goff[start]): int(goff[start + 1000])] == want, start
    # filter_parser alone takes the same kernel
    fp2 = g.FilterParser("log", [p])
    a = [fp2.filter_dev(mixed) for _ in range(2)]
    assert fp2.paths()["rows_by_length"]
    r1, w1 = fo.filter(mixed_blob[: int(moff[2000])])
    pb = np.empty(int(a[1][1].bytes), dtype=np.uint8)
    L.flbgpu_memcpy_d2h(pb.ctypes.data, a[1][1].data, pb.nbytes)
    poff = np.empty(2001, dtype=np.uint64)
    L.flbgpu_memcpy_d2h(poff.ctypes.data, a[1][1].row_off, poff.nbytes)
    assert bytes(pb[: int(poff[2000])]) == w1
    # even lines: the order by length is still on for the first such chunk (its counters come from the ordering pass), then off
    st = [call(even)[3]["rows_by_length"] for _ in range(3)]
    assert st == [True, False, False], st
    assert call(mixed)[3]["rows_by_length"] is False and call(mixed)[3]["rows_by_length"] is True
    fp2.close(); fg.close(); fp.close(); p.close()
    for c in (mixed, even):
        L.flbgpu_dev_free(c.data); L.flbgpu_dev_free(c.row_off)
