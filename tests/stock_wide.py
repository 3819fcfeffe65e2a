"""The stock parsers whose UTF-8 capture automaton needs the WIDE reverse tables (rx.hpp `wide`: more than 0x7FF0
states; 32-bit entries walked from HBM by the generic kernels), their patterns as conf/parsers.conf:102-106 (`envoy`)
and conf/parsers_ambassador.conf:4-8 (`ambassador`) spell them, and a generator of access-log lines for both:
well-formed ones, ones with blanks / quotes / backslashes where the lazy path group has to give way, ill-formed
UTF-8, cut lines.  `istio-envoy-proxy` (conf/parsers.conf:111) stays refused at create: its reverse automaton has
more than 4 M states."""
import random

ENVOY = (r'^\[(?<start_time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^\"]*?)(?: +\S*)?)? (?<protocol>\S+)" (?<code>[^ ]*) '
         r'(?<response_flags>[^ ]*) (?<bytes_received>[^ ]*) (?<bytes_sent>[^ ]*) (?<duration>[^ ]*) '
         r'(?<x_envoy_upstream_service_time>[^ ]*) "(?<x_forwarded_for>[^ ]*)" "(?<user_agent>[^\"]*)" "(?<request_id>[^\"]*)" '
         r'"(?<authority>[^ ]*)" "(?<upstream_host>[^ ]*)"')
AMBASSADOR = (r'^(?<type>\S+) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>(?:[^\"]|\\.)*?)(?: +\S*)?) (?<protocol>\S+)?" '
              r'(?<response_code>\S+) (?<response_flags>\S+) (?<bytes_received>\S+) (?<bytes_sent>\S+) (?<duration>\S+) '
              r'(?<x_envoy_upstream_service_time>\S+) "(?<x_forwarded_for>[^\"]*)" "(?<user_agent>[^\"]*)" "(?<x_request_id>[^\"]*)" '
              r'"(?<authority>[^\"]*)" "(?<upstream_host>[^\"]*)"')
ENVOY_TIME_FMT = "%Y-%m-%dT%H:%M:%S.%L%z"          # conf/parsers.conf:105

FRAG = [b"\xe9", b"\xc3", b"\xa9", b"\xe2\x82", b"\xf0\x9f\x98", b"\xff", b"\xc0\x80", b"\xed\xa0\x80", b"\xc3\xa9", b"\xe2\x82\xac",
        b"\xf4\x90\x80\x80", b"\x80", b" ", b"  ", b'"', b'\\"', b"\\", b"]", b"[", b"\n"]
AGENTS = [b"nsq2http", b"Mozilla/5.0 (X11; Linux x86_64) AppleWebKit/537.36", b"curl/7.58.0", b"caf\xc3\xa9 client/1.0", b"-"]
PATHS = [b"/api/v1/locations", b"/", b"/a b/c", b"/search?q=caf\xc3\xa9&lang=fr", b"/x/" + b"y" * 70, b"", b"/q\\\"uoted"]


def lines(n, seed, prefix=b""):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        ts = b"2016-04-15T20:%02d:%02d.%03dZ" % (rng.randrange(60), rng.randrange(60), rng.randrange(1000))
        req = rng.choice([b"GET", b"POST", b"DELETE"])
        path = rng.choice(PATHS)
        if path or rng.random() < 0.5:
            req += b" " * rng.randint(1, 2) + path
        req += b" " + rng.choice([b"HTTP/1.1", b"HTTP/2", b"-"])
        ln = prefix + b'[' + ts + b'] "' + req + b'" %d %s %d %d %d %s "%s" "%s" "%s" "%s" "%s"' % (
            rng.choice([200, 204, 404, 503]), rng.choice([b"-", b"UF,URX", b"NR"]), rng.randrange(5000), rng.randrange(90000),
            rng.randrange(3000), rng.choice([b"-", b"17"]), rng.choice([b"10.0.35.28", b"-", b"10.0.0.1,10.0.0.2"]), rng.choice(AGENTS),
            b"cc21d9b0-cf5c-432b-8c7e-%012x" % rng.getrandbits(48), rng.choice([b"locations", b"svc.ns:8080"]),
            rng.choice([b"tcp://10.0.2.1:80", b"10.0.2.1:8080", b"-"]))
        r = rng.random()
        if r < 0.55:
            m = bytearray(ln)
            for _ in range(rng.randint(1, 3)):
                k = rng.randrange(len(m) + 1)
                m[k:k] = rng.choice(FRAG)
            ln = bytes(m)
        if r < 0.1:
            ln = ln[:rng.randrange(1, len(ln))] + rng.choice(FRAG[:6])           # cut by the end of the text
        elif r > 0.95:
            ln = ln + b"\n" + ln                                                  # ^ is a line anchor: a second candidate start
        out.append(ln)
    return out
