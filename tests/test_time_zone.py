"""Time_Zone / Time_System_Timezone on a parser (flb_parser_create_with_time_zone, /root/reference src/flb_parser.c:805-1049;
tzif_load :452-537, tzif_tm2time :560-590, flb_parser_tm2time_parser :685-696).

CPU part: (1) the oracle's restatement against the reference's OWN flb_parser.c compiled in place (oracle/_ref/ref_filters, built
with FLB_HAVE_TIME_ZONE) through filter_parser on the same records -- zones with and without daylight time, both hemispheres,
half-hour offsets, local times inside spring-forward gaps and fall-back overlaps, instants in front of the first and behind the
last transition of the file; (2) the routine the kernels run (csrc/tzif.hpp through the host hook flbgpu_tz_tm2time) against the
oracle on the same local times; (3) what is refused is what the reference refuses.
GPU part: filter_parser with a zone parser on the device against the oracle, regex / json / logfmt / ltsv.

The zone files are the ones of the `tzdata` Python package of this image (TZDIR points the reference, the oracle and the product
at them: the image has no /usr/share/zoneinfo); the same files are on the GPU box."""
import calendar, ctypes, os, random, sys, time
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob
import ref_filters as rf
import synth


def _tzdir():
    try:
        import tzdata
    except ImportError:
        return None
    d = os.path.join(os.path.dirname(tzdata.__file__), "zoneinfo")
    return d if os.path.isfile(os.path.join(d, "America", "New_York")) else None


TZDIR = _tzdir()
pytestmark = pytest.mark.skipif(TZDIR is None, reason="no zone files (tzdata package)")

ZONES = ["America/New_York", "Europe/Berlin", "Asia/Kolkata", "Australia/Lord_Howe", "Pacific/Auckland", "Asia/Tokyo", "Etc/UTC",
         "America/Sao_Paulo", "Africa/Casablanca", "Pacific/Kiritimati", "Europe/London", "Asia/Kathmandu", "America/St_Johns",
         "Europe/Dublin", "Etc/GMT+5"]
FMT = "%Y-%m-%d %H:%M:%S"


@pytest.fixture(autouse=True)
def _env():
    old = os.environ.get("TZDIR")
    os.environ["TZDIR"] = TZDIR
    yield
    if old is None:
        os.environ.pop("TZDIR", None)
    else:
        os.environ["TZDIR"] = old


def _transitions(zone):
    """the transition instants and offsets of the zone's file (64-bit block), read here a third time -- only to aim the test
    texts at the interesting places"""
    import struct
    b = open(os.path.join(TZDIR, zone), "rb").read()
    def hdr(o):
        return struct.unpack(">6I", b[o + 20:o + 44])
    isut, isstd, leap, timecnt, typecnt, charcnt = hdr(0)
    o = 44
    if b[4:5] in (b"2", b"3", b"4"):
        o = 44 + timecnt * 5 + typecnt * 6 + charcnt + leap * 8 + isstd + isut
        isut, isstd, leap, timecnt, typecnt, charcnt = hdr(o)
        tr = struct.unpack(">%dq" % timecnt, b[o + 44:o + 44 + 8 * timecnt])
        p = o + 44 + 8 * timecnt
    else:
        tr = struct.unpack(">%di" % timecnt, b[44:44 + 4 * timecnt])
        p = 44 + 4 * timecnt
    tt = b[p:p + timecnt]
    p += timecnt
    offs = [struct.unpack(">i", b[p + 6 * i:p + 6 * i + 4])[0] for i in range(typecnt)]
    return list(tr), list(tt), offs


def _texts(zone, rng, n_random=40):
    """local time texts around every kind of place of the zone's table"""
    tr, tt, offs = _transitions(zone)
    locs = set()
    picks = list(range(len(tr)))
    rng.shuffle(picks)
    for i in picks[:12] + ([0, len(tr) - 1] if tr else []):
        for off in set(offs):
            for d in (-7200, -3601, -3600, -1800, -1, 0, 1, 1799, 1800, 3599, 3600, 7200):
                locs.add(tr[i] + off + d)
    for _ in range(n_random):
        locs.add(rng.randrange(-2000000000, 4000000000))
    locs.update([0, 1, -1, 86399, 951782400, 1709164800, 4102444800, -2208988800])
    out = []
    for l in sorted(locs):
        if -62135596800 <= l <= 253402300799:
            out.append(time.strftime(FMT, time.gmtime(l)))
    return out


def _records(texts, key="log", wrap="%s"):
    return b"".join(synth.v2_record(1700000000 + i, i, {key: (wrap % t)}) for i, t in enumerate(texts))


# ----------------------------------------------------------------------------------------------------------- CPU: the pin
@pytest.mark.skipif(not rf.available(), reason="oracle/_ref/ref_filters not built (no reference tree)")
def test_oracle_zone_against_the_reference_parser():
    rng = random.Random(5)
    cases, wants, what = [], [], []
    for zone in ZONES:
        texts = _texts(zone, rng)
        plists = [
            (dict(regex=r"^(?<time>[^|]*)\|(?<rest>.*)$", time_fmt=FMT, time_key="time", time_zone=zone), "%s|x"),
            (dict(regex=r"^(?<time>[^|]*)\|(?<rest>.*)$", time_fmt=FMT, time_key="time", time_zone=zone, time_keep=True, time_strict=False), "%s|x"),
            (dict(format="json", time_fmt=FMT, time_key="time", time_zone=zone), '{"time":"%s","a":1}'),
            (dict(format="logfmt", time_fmt=FMT, time_key="time", time_zone=zone), 'a=1 time="%s" b=2'),
            (dict(format="ltsv", time_fmt=FMT, time_key="time", time_zone=zone), "a:1\ttime:%s\tb:2"),
        ]
        for p, wrap in plists:
            data = _records(texts, wrap=wrap)
            cases.append(rf.parser_case("log", [p], data))
            wants.append(ob.FilterParser("log", [ob.Parser(**p)]).filter(data))
            what.append((zone, p.get("format", "regex"), len(texts)))
    # a format that carries its own zone: the table is not asked (src/flb_parser.c:687)
    p = dict(regex=r"^(?<time>.*)$", time_fmt=FMT + " %z", time_key="time", time_zone="America/New_York")
    data = _records(["2024-03-10 02:30:00 +0530", "2024-11-03 01:30:00 -0800", "2024-07-01 00:00:00 +0000"])
    cases.append(rf.parser_case("log", [p], data)); wants.append(ob.FilterParser("log", [ob.Parser(**p)]).filter(data)); what.append("with %z")
    # Time_System_Timezone in a UTC process: mktime
    for fmt, texts in ((FMT, ["2024-03-10 02:30:00", "1969-12-31 23:59:59"]), (FMT + " %z", ["2024-03-10 02:30:00 +0530"])):
        p = dict(regex=r"^(?<time>.*)$", time_fmt=fmt, time_key="time", time_system_timezone=True)
        data = _records(texts)
        cases.append(rf.parser_case("log", [p], data)); wants.append(ob.FilterParser("log", [ob.Parser(**p)]).filter(data)); what.append("systz " + fmt)
    got = rf.run(cases)
    n_mod = 0
    for (ret, out), (wret, wout), w in zip(got, wants, what):
        assert ret == wret, (w, ret, wret)
        if wret == ob.MODIFIED:
            n_mod += 1
            assert out == (wout or b""), w
    assert n_mod == len(cases)


@pytest.mark.skipif(not rf.available(), reason="oracle/_ref/ref_filters not built (no reference tree)")
def test_reference_refusals_are_the_oracles():
    data = _records(["2024-01-01 00:00:00"])
    bad = [dict(regex=r"^(?<time>.*)$", time_fmt=FMT, time_key="time", time_zone="No/Such_Zone"),
           dict(regex=r"^(?<time>.*)$", time_fmt=FMT, time_key="time", time_zone="Europe/Berlin", time_offset="+0100"),
           dict(regex=r"^(?<time>.*)$", time_fmt=FMT, time_key="time", time_zone="Europe/Berlin", time_system_timezone=True),
           dict(regex=r"^(?<time>.*)$", time_key="time", time_zone="Europe/Berlin"),
           # names with a file that are not in the reference's zone index (src/flb_time_tz.c): refused by name
           dict(regex=r"^(?<time>.*)$", time_fmt=FMT, time_key="time", time_zone="UTC"),
           dict(regex=r"^(?<time>.*)$", time_fmt=FMT, time_key="time", time_zone="Antarctica/Troll")]
    got = rf.run([rf.parser_case("log", [p], data) for p in bad])
    for (ret, _), p in zip(got, bad):
        assert ret == -100, p                         # the parser was not created: cb_init finds no parser of that name
        with pytest.raises(ValueError):
            ob.Parser(**p)


# the encoder takes the group markers' two times from anyone (src/flb_log_event_encoder.c:345-363): a text that parses to -1 s or
# -2 s without a fraction leaves filter_parser with the timestamp ffffffff / fffffffe; -3 s, or -1 s with a fraction, fail the record
EDGE_TEXTS = ["1969-12-31 23:59:59", "1969-12-31 23:59:58", "1969-12-31 23:59:57", "1970-01-01 00:00:00", "1970-01-01 00:00:01",
              "1969-12-31 23:59:59.5", "1969-12-31 23:59:58.000", "1969-12-31 23:59:57.25", "2106-02-07 06:28:15", "2106-02-07 06:28:16",
              "1969-12-31 23:59:59.000000001", "1960-01-01 00:00:00"]
EDGE_PARSERS = [
    (dict(regex=r"^(?<time>[^|]*)\|(?<rest>.*)$", time_fmt=FMT, time_key="time"), "%s|x"),
    (dict(regex=r"^(?<time>[^|]*)\|(?<rest>.*)$", time_fmt=FMT + ".%L", time_key="time", time_keep=True), "%s|x"),
    (dict(regex=r"^(?<time>[^|]*)\|(?<rest>.*)$", time_fmt=FMT, time_key="time", time_offset="+0100"), "%s|x"),
    (dict(regex=r"^(?<time>[^|]*)\|(?<rest>.*)$", time_fmt=FMT, time_key="time", types="rest:integer"), "%s|7"),
    (dict(format="json", time_fmt=FMT + ".%L", time_key="time"), '{"time":"%s","a":1}'),
    (dict(format="logfmt", time_fmt=FMT, time_key="time"), 'a=1 time="%s" b=2'),
    (dict(format="ltsv", time_fmt=FMT + ".%L", time_key="time"), "a:1\ttime:%s\tb:2"),
]


@pytest.mark.skipif(not rf.available(), reason="oracle/_ref/ref_filters not built (no reference tree)")
def test_oracle_group_marker_times_against_the_reference():
    cases, wants = [], []
    for p, wrap in EDGE_PARSERS:
        for reserve in (False, True):
            data = _records(EDGE_TEXTS, wrap=wrap)
            cases.append(rf.parser_case("log", [p], data, reserve, False))
            wants.append(ob.FilterParser("log", [ob.Parser(**p)], reserve, False).filter(data))
    got = rf.run(cases)
    seen = 0
    for (ret, out), (wret, wout) in zip(got, wants):
        assert ret == wret == ob.MODIFIED and out == wout
        seen += out.count(b"\x92\x92\xd7\x00\xff\xff\xff\xff\x00\x00\x00\x00") + out.count(b"\x92\x92\xd7\x00\xff\xff\xff\xfe\x00\x00\x00\x00")
    assert seen >= 10


# --------------------------------------------------------------------------------- CPU: the kernels' routine on the host
def test_device_routine_on_the_host_against_the_oracle():
    import flbamd_loader
    fb = flbamd_loader.load()
    L = fb.lib()
    rng = random.Random(9)
    n = 0
    for zone in ZONES:
        op = ob.Parser(regex=r"^(?<time>.*)$", time_fmt=FMT, time_key="time", time_zone=zone)
        for t in _texts(zone, rng, n_random=200):
            r, sec, _ = op.time_lookup(t.encode())
            assert r == 0
            local = calendar.timegm(time.strptime(t, FMT))
            out = ctypes.c_int64()
            assert L.flbgpu_tz_tm2time(zone.encode(), local, ctypes.byref(out)) == 0, fb.last_error()
            assert out.value == sec, (zone, t, out.value, sec)
            n += 1
    assert n > 5000
    out = ctypes.c_int64()
    assert L.flbgpu_tz_tm2time(b"No/Such_Zone", 0, ctypes.byref(out)) == -1


@pytest.mark.gpu
def test_product_refusals():
    """(a parser's tables are uploaded at create: needs the device)"""
    import flbamd_loader
    fb = flbamd_loader.load()
    fb.init()
    rx = r"^(?<time>.*)$"
    for kw in (dict(time_fmt=FMT, time_zone="No/Such_Zone"), dict(time_fmt=FMT, time_zone="Europe/Berlin", time_offset="+0100"),
               dict(time_fmt=FMT, time_zone="Europe/Berlin", time_system_timezone=True), dict(time_zone="Europe/Berlin"),
               dict(time_fmt=FMT, time_zone="UTC"), dict(time_fmt=FMT, time_zone="Antarctica/Troll")):
        with pytest.raises(ValueError, match="time_zone|time_system_timezone"):
            fb.Parser(regex=rx, time_key="time", **kw)
    fb.Parser(regex=rx, time_key="time", time_fmt=FMT, time_zone="Europe/Berlin").close()
    fb.Parser(regex=rx, time_key="time", time_fmt=FMT, time_system_timezone=True).close()


def test_the_zone_name_directive_is_taken_in_any_process_zone():
    """round 6: %Z in a process whose zone is not UTC is no longer refused at create (rounds 4 and 5 refused it: the zone text's last
    resort is the process's tzname[], src/flb_strptime.c:611-650, which the device now gets from the creating process).  Without a GPU
    the create call gets as far as the device in either zone: the only error left is the missing device."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import flbamd_loader; fb = flbamd_loader.load()\n"
            "import time; out = [str(time.timezone)]\n"
            "for fmt_ in ('json', 'regex', 'logfmt'):\n"
            "    try:\n"
            "        fb.Parser(regex=r'^(?<time>.*)$' if fmt_ == 'regex' else None, format=fmt_, time_key='time', time_fmt='%%Y-%%m-%%d %%H:%%M:%%S %%Z').close(); out.append('ok')\n"
            "    except (ValueError, RuntimeError) as e:\n"
            "        out.append(str(e))\n"
            "print('|'.join(out))\n") % os.path.dirname(HERE)
    seen = {}
    for tz in ("Europe/Berlin", "UTC"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, TZ=tz, TZDIR=TZDIR), timeout=300)
        assert r.returncode == 0, r.stderr[-800:]
        seen[tz] = r.stdout.strip().splitlines()[-1].split("|")
    for tz in seen:
        assert not any("not UTC" in m or "%Z" in m for m in seen[tz][1:]), seen


@pytest.mark.gpu
def test_process_zone_that_is_not_utc():
    """%Z's last resort and Time_System_Timezone read the PROCESS's zone (src/flb_strptime.c:611-650, flb_parser.h:80-94).  In a process
    with TZ=Europe/Berlin: Time_System_Timezone is refused at create, with the reason (mktime's answer inside a zone's overlaps depends on
    the thread's previous call); %Z is taken -- the zone texts of the reference's table, GMT / UTC, the process's own two names
    in any case (CET and CEST: BOTH mean -timezone there), names of other zones and non-names (the parser fails: the record keeps its
    time) -- and the records are the REFERENCE's own filter_parser's, run in the same zone (oracle/_ref/ref_filters)."""
    import subprocess, sys
    code = ("import sys, os; sys.path.insert(0, %r); sys.path.insert(0, %r); import flbamd_loader; fb = flbamd_loader.load(); fb.init()\n"
            "import ref_filters as rf, synth, time\n"
            "rx = r'^(?<time>[^|]*)\\|(?<m>.*)$'\n"
            "out = [str(time.timezone), '/'.join(time.tzname)]\n"
            "try:\n"
            "    fb.Parser(regex=rx, time_key='time', time_fmt='%%Y-%%m-%%d %%H:%%M:%%S', time_system_timezone=True).close(); out.append('ok')\n"
            "except ValueError as e:\n"
            "    out.append('refused: ' + str(e))\n"
            "zones = ['CET', 'CEST', 'cet', 'Cest', 'CETX', 'CEST1', 'UTC', 'GMT', 'utc', 'EST', 'PDT', 'pst', 'WET', 'EET', 'MSK', 'XYZ', 'CE', '', 'Z', '+0100', 'JST', 'IST', 'AEDT']\n"
            "texts = ['2024-0%%d-1%%d 0%%d:1%%d:2%%d %%s' %% (1 + i %% 9, i %% 9, i %% 9, i %% 6, i %% 6, z) for i, z in enumerate(zones * 3)]\n"
            "data = b''.join(synth.v2_record(1700000000 + i, 5, {'log': t + '|x'}) for i, t in enumerate(texts))\n"
            "for fmt in ('%%Y-%%m-%%d %%H:%%M:%%S %%Z', '%%Y-%%m-%%d %%H:%%M:%%S %%Z|', '%%Y-%%m-%%d %%H:%%M:%%S%%n%%Z'):\n"
            "    for keep in (False, True):\n"
            "        pa = dict(regex=rx, time_fmt=fmt, time_key='time', time_keep=keep)\n"
            "        gp = fb.Parser(**pa); gf = fb.FilterParser('log', [gp])\n"
            "        ret, got = gf.filter(data)\n"
            "        res = rf.run([rf.parser_case('log', [pa], data)])\n"
            "        want = res[0][1] if res[0][0] == rf.MODIFIED else None\n"
            "        out.append('same' if (ret == rf.MODIFIED and got == want) or (ret != rf.MODIFIED and want is None) else 'DIFFERENT %%s %%s' %% (fmt, keep))\n"
            "        gf.close(); gp.close()\n"
            "print('|'.join(out))\n") % (os.path.dirname(HERE), HERE)
    import ref_filters as rf
    if not rf.available():
        pytest.skip("oracle/_ref/ref_filters not built")
    env = dict(os.environ, TZ="Europe/Berlin", TZDIR=TZDIR)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    parts = r.stdout.strip().splitlines()[-1].split("|")
    tzsec, names, b = parts[0], parts[1], parts[2]
    if tzsec == "0":
        pytest.skip("TZ=Europe/Berlin has no effect in this image's C library (no zone files where it looks)")
    assert b.startswith("refused") and "Time_System_Timezone" in b, b
    assert names == "CET/CEST" and parts[3:] == ["same"] * 6, parts


# ------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_filter_parser_with_a_zone_on_the_device():
    import flbamd_loader
    fb = flbamd_loader.load()
    fb.init()
    rng = random.Random(21)
    for zone in ZONES[:8]:
        texts = _texts(zone, rng, n_random=300)
        plists = [
            (dict(regex=r"^(?<time>[^|]*)\|(?<rest>.*)$", time_fmt=FMT, time_key="time", time_zone=zone), "%s|x"),
            (dict(regex=r"^(?<time>[^|]*)\|(?<rest>.*)$", time_fmt=FMT, time_key="time", time_zone=zone, time_keep=True, time_strict=False), "%s|x"),
            (dict(regex=r"^(?<time>[^|]*)\|(?<rest>.*)$", time_fmt=FMT + " %z", time_key="time", time_zone=zone), "%s +0530|x"),
            (dict(regex=r"^(?<time>[^|]*)\|(?<rest>.*)$", time_fmt=FMT, time_key="time", time_system_timezone=True), "%s|x"),
            (dict(regex=r"^(?<time>[^|]*)\|(?<rest>.*)$", time_fmt=FMT + " %z", time_key="time", time_system_timezone=True), "%s -0800|x"),
            (dict(format="json", time_fmt=FMT, time_key="time", time_zone=zone), '{"time":"%s","a":1}'),
            (dict(format="logfmt", time_fmt=FMT, time_key="time", time_zone=zone), 'a=1 time="%s" b=2'),
            (dict(format="ltsv", time_fmt=FMT, time_key="time", time_zone=zone), "a:1\ttime:%s\tb:2"),
        ]
        for p, wrap in plists:
            data = _records(texts, wrap=wrap)
            wret, wout = ob.FilterParser("log", [ob.Parser(**p)]).filter(data)
            gp = fb.Parser(**p)
            gf = fb.FilterParser("log", [gp])
            ret, out = gf.filter(data)
            assert ret == wret and out == wout, (zone, p.get("format", "regex"), p["time_fmt"])
            gf.close(); gp.close()
    # the two times of the group markers (and their neighbours) through every parser kind
    for p, wrap in EDGE_PARSERS:
        for reserve in (False, True):
            data = _records(EDGE_TEXTS * 40, wrap=wrap)
            wret, wout = ob.FilterParser("log", [ob.Parser(**p)], reserve, False).filter(data)
            gp = fb.Parser(**p)
            gf = fb.FilterParser("log", [gp], reserve, False)
            ret, out = gf.filter(data)
            assert ret == wret and out == wout, (p.get("format", "regex"), p["time_fmt"], reserve)
            gf.close(); gp.close()
    # the pair [filter_parser, filter_grep] with a zone parser: same bytes as the two filters one after the other on the oracle
    zone = "America/New_York"
    p = dict(regex=r"^(?<time>[^|]*)\|(?<rest>.*)$", time_fmt=FMT, time_key="time", time_zone=zone)
    data = _records(_texts(zone, rng, n_random=2000), wrap="%s|x")
    o1 = ob.FilterParser("log", [ob.Parser(**p)]).filter(data)[1]
    wret, wout = ob.Grep([("regex", "rest x")]).filter(o1)
    gp = fb.Parser(**p)
    chain = fb.FilterChain([fb.FilterParser("log", [gp]), fb.FilterGrep([("regex", "rest x")])])
    ret, out = chain.filter(data)
    assert ret == ob.MODIFIED and out == (wout if wret == ob.MODIFIED else o1)
