"""Synthetic multiline cases: rule sets (state graphs over simple patterns) and text whose lines hit them, cut into the frames
successive reads of in_tail would deliver.  Test infrastructure."""
import random

BUILTINS = ("java", "go", "python", "ruby")

# lines that walk the built-in parsers' states (and near misses)
SEED_LINES = {
    "java": [b'Exception in thread "main" java.lang.IllegalStateException: ..null property', b"     at com.example.myproject.Author.getBookIds(xx.java:38)",
             b"Caused by: java.lang.NullPointerException", b"     ... 1 more", b"single line", b"\tat a.b.C(D.java:1)", b"   nested exception is:  ",
             b"javax.servlet.ServletException: Something bad happened", b"", b"\r", b"  --- End of inner exception stack trace ---",
             b"--- End of stack trace from previous location where exception was thrown ---", b"Suppressed: x", b" ... 12 common frames omitted",
             b"    eval at foo", b"xError:", b"Error: boom"],
    "go": [b"panic: my panic", b"", b"goroutine 4 [running]:", b"panic(0x45cb40, 0x47ad70)", b"\t/usr/local/go/src/runtime/panic.go:542 +0x46c",
           b"main.main.func1(0xc420024120)", b"created by main.main", b"\tfoo.go:5 +0x58", b"[signal SIGSEGV: segmentation violation]",
           b"2021/01/01 http: panic serving 127.0.0.1", b"one more line, no multiline", b" x", b"goroutine 1 [chan receive]:", b"runtime.goexit()"],
    "python": [b"Traceback (most recent call last):", b'  File "/base/data/home/runtimes/python27/webapp2.py", line 1535, in __call__',
               b"    rv = self.handle_exception(request, response, e)", b"Exception: ('spam', 'eggs')", b"hello world, not multiline", b"", b"   ",
               b"\tFile x", b"ValueError: bad", b"a.b.c: d"],
    "ruby": [b"/app/config/routes.rb:6:in `/': divided by 0 (ZeroDivisionError)", b"\tfrom /app/config/routes.rb:6:in `block in <main>'",
             b"  from /var/lib/gems/3.0.0/gems/x.rb:428:in `instance_exec'", b"hello world, not multiline", b"", b"x:1:in y", b" from a:2:in b"],
}

ATOMS = [(r"^\d+ ", b"12 "), (r"^\[", b"["), (r"^\s+", b"   "), (r"^\s+at ", b"  at "), (r"ERROR", b"ERROR"), (r"^$", b""), (r"end$", b"end"),
         (r"^[A-Z][a-z]+:", b"Caused:"), (r"\bpanic: ", b"panic: "), (r"^--", b"--"), (r"[^\t ]", b"x"), (r"^\S", b"q"), (r"(?i)^warn", b"WaRn"),
         # the documented "not a first line" idiom: a leading look-ahead behind the line anchor
         # rules that only a character >= 0x80 can satisfy (round 3's advice: the product automaton of the rules must not take such a
         # rule for dead because no ASCII byte leads to its match), and a POSIX bracket with its Unicode members
         ("^错误", "错误".encode()), (r"^\s+é", "  é".encode()), ("ü", "grün".encode()), (r"[^\x00-\x7f]$", "tail ж".encode()),
         (r"^[[:upper:]][[:lower:]]", "Жук".encode()),
         (r"^(?!\d+ ).*", b"zz"), (r"^(?!\[|--)(?:\S+ e|x)", b"qq e"), (r"^(?=\s+at )\s+at [a-z]", b"  at b"), (r"^(?![A-Z][a-z]+:)", b"lower:")]
STATES = ["s1", "s2", "s3", "s4"]


def random_rules(rng):
    """a rule list the reference accepts: the first rule holds start_state, every to_state is some rule's from_state"""
    n = rng.randrange(1, 7)
    rules = []
    for i in range(n):
        froms = set()
        if i == 0 or rng.random() < 0.25:
            froms.add("start_state")
        for _ in range(rng.randrange(0 if froms else 1, 3)):
            froms.add(rng.choice(STATES))
        rules.append([sorted(froms, key=lambda s: rng.random()), rng.choice(ATOMS)[0], None])
    known = sorted({s for r in rules for s in r[0]})
    for r in rules:
        r[2] = rng.choice(known) if rng.random() < 0.9 else None
    return [(", ".join(f) if rng.random() < 0.7 else ",".join(f), "/%s/" % rx, to) for f, rx, to in rules]


def random_text(rng, nlines, seeds=None, crlf=0.05, empty=0.08, long_line=0.01, nul_lead=0.05):
    words = [b"alpha", b"beta", b"  gamma", b"12 delta", b"[x]", b"ERROR", b"end", b"Caused: by", b"panic: now", b"--", b"\tTab", b"WARN", b"warn"]
    out = bytearray()
    if rng.random() < nul_lead:
        out += b"\0" * rng.randrange(1, 5)
    for _ in range(nlines):
        r = rng.random()
        if r < empty:
            line = b""
        elif seeds and r < 0.75:
            line = rng.choice(seeds)
        else:
            k = rng.randrange(1, 4)
            line = (rng.choice(ATOMS)[1] if rng.random() < 0.6 else b"") + b" ".join(rng.choice(words) for _ in range(k))
            if rng.random() < 0.1:
                line += rng.choice([b" end", b"\t", b" \xc3\xa9t\xc3\xa9", b"\xff"])
        if rng.random() < long_line:
            line = line + b" " + bytes(rng.choice(b"abcdefghij ") for _ in range(rng.randrange(300, 3000)))
        out += line
        out += b"\r\n" if rng.random() < crlf else b"\n"
    return bytes(out)


def frames_of(rng, text, max_frames=5, t0=1700000000):
    """cuts anywhere (also inside a line, inside a CR LF); every frame carries its own time"""
    k = rng.randrange(1, max_frames + 1)
    cuts = sorted(rng.randrange(0, len(text) + 1) for _ in range(k - 1)) if text else []
    parts, a = [], 0
    for c in cuts + [len(text)]:
        parts.append(text[a:c])
        a = c
    return [(t0 + 7 * i, (i * 1000003 + 5) % 1000000000, p) for i, p in enumerate(parts)]


def random_case(rng, nlines=None):
    """keyword arguments shared by ref_filters.ml_case / oracle_binding.Multiline / the product + frames"""
    cfg = {}
    seeds = None
    r = rng.random()
    if r < 0.35:
        b = rng.choice(BUILTINS)
        cfg["builtin"] = b
        seeds = SEED_LINES[b]
    elif r < 0.85:
        cfg["rules"] = random_rules(rng)
    elif r < 0.93:
        cfg.update(type="endswith", match_string=rng.choice(["end", "\\", "d", ""]), negate=rng.random() < 0.3)
    else:
        cfg.update(type="equal", match_string=rng.choice(["", "--", "end"]), negate=rng.random() < 0.3)
    if rng.random() < 0.4:
        cfg["key_content"] = rng.choice(["log", "message", "k" * 40])
    text = random_text(rng, nlines if nlines is not None else rng.randrange(0, 60), seeds)
    if rng.random() < 0.2:
        text = text[:-1] if text else text                    # the last line is not complete yet
    return cfg, frames_of(rng, text), dict(skip_empty_lines=rng.random() < 0.4, final_flush=rng.random() < 0.7)


def java_service_log(rng, nlines):
    """what a JVM service writes: timestamped log lines of 80-160 bytes, now and then an exception with its stack trace (frames of
    60-110 bytes, Caused by sections, "... n more"); the shape the built-in `java` parser exists for"""
    pk = ["com.example.orders", "org.springframework.web.servlet", "io.netty.channel", "java.util.concurrent", "com.fasterxml.jackson.databind", "org.hibernate.engine.jdbc"]
    cl = ["OrderService", "DispatcherServlet", "AbstractChannelHandlerContext", "ThreadPoolExecutor$Worker", "ObjectMapper", "SqlExceptionHelper", "HttpRequestHandlerAdapter"]
    ex = ["java.lang.IllegalStateException", "java.lang.NullPointerException", "java.sql.SQLTransientConnectionException", "javax.servlet.ServletException", "java.util.concurrent.TimeoutException"]
    out = []
    n = 0
    while n < nlines:
        if rng.random() < 0.06:
            out.append(("2024-03-10 10:%02d:%02d,%03d ERROR [http-nio-8080-exec-%d] %s.%s - request failed" % (rng.randrange(60), rng.randrange(60), rng.randrange(1000), rng.randrange(40), rng.choice(pk), rng.choice(cl))).encode())
            out.append(("%s: %s while handling order %d for customer %d" % (rng.choice(ex), rng.choice(["timeout", "connection is not available", "..null property", "unexpected state"]), rng.randrange(10 ** 7), rng.randrange(10 ** 5))).encode())
            n += 2
            for sect in range(rng.randrange(1, 4)):
                if sect:
                    out.append(("Caused by: %s: %s" % (rng.choice(ex), rng.choice(["upstream closed the stream", "pool exhausted after 30000ms", "value was null"]))).encode())
                    n += 1
                for _ in range(rng.randrange(4, 28)):
                    out.append(("\tat %s.%s.%s(%s.java:%d)" % (rng.choice(pk), rng.choice(cl), rng.choice(["invoke", "doDispatch", "run", "fireChannelRead", "handle", "lambda$process$3", "readValue"]), rng.choice(cl).split("$")[0], rng.randrange(20, 2000))).encode())
                    n += 1
                if rng.random() < 0.7:
                    out.append(("\t... %d more" % rng.randrange(3, 60)).encode())
                    n += 1
        else:
            out.append(("2024-03-10 10:%02d:%02d,%03d %s [http-nio-8080-exec-%d] %s.%s - %s order=%d customer=%d items=%d total=%d.%02d took %d ms" % (
                rng.randrange(60), rng.randrange(60), rng.randrange(1000), rng.choice(["INFO", "INFO", "INFO", "DEBUG", "WARN"]), rng.randrange(40), rng.choice(pk), rng.choice(cl),
                rng.choice(["accepted", "validated", "persisted", "shipped", "rejected"]), rng.randrange(10 ** 7), rng.randrange(10 ** 5), rng.randrange(1, 30), rng.randrange(10 ** 4), rng.randrange(100),
                rng.randrange(1, 900))).encode())
            n += 1
    return b"\n".join(out[:nlines]) + b"\n"


def cri_text(rng, nlines, damage=0.08, bad_times=True, ascii_only=False, long_lines=0.02):
    """containerd's log format (time stream P|F log) with partial lines on two streams interleaved, and lines the cri parser refuses"""
    out = []
    for i in range(nlines):
        r = rng.random()
        if r < damage:
            out.append(rng.choice([b"not a cri line", b"", b"2021-05-17T17:35:01Z stdout X bad flag", b"2021-05-17T17:35:01Z stdin F other stream",
                                   b"2021-05-17T17:35:01.1Z stdout F", b" stdout F leading blank"] + ([b"\xff\xfe stdout F x"] if bad_times else [])))
            continue
        t = "2021-05-17T17:%02d:%02d.%09dZ" % (rng.randrange(60), rng.randrange(60), rng.randrange(10 ** 9)) if rng.random() < 0.9 or not bad_times else rng.choice(["garbage-time", "2021-05-17T17:35:01+02:00", "1"])
        stream = rng.choice(["stdout", "stdout", "stderr"])
        flag = "F" if rng.random() < 0.6 else "P"
        log = rng.choice([b"", b"x", b"[DEBUG] start multiline - ", b"part of a long line that was split by the runtime at 16 KB ", b'{"json":"inside"}',
                          b"cafe \t tab" if ascii_only else b"caf\xc3\xa9 \t tab", b"ends with space "])
        if rng.random() < long_lines:
            log = bytes(rng.choice(b"abc ") for _ in range(rng.randrange(1000, 20000)))
        out.append(("%s %s %s " % (t, stream, flag)).encode() + log)
    return b"".join(l + (b"\r\n" if rng.random() < 0.03 else b"\n") for l in out)


def docker_text(rng, nlines, damage=0.08):
    """docker's json-file lines ({"log": "...", "stream": "...", "time": "..."}; a message is complete when log ends with a newline)"""
    import json
    out = []
    for i in range(nlines):
        r = rng.random()
        if r < damage:
            out.append(rng.choice([b"plain text", b"", b'{"log": 5, "stream": "stdout"}', b'{"stream":"stdout","time":"2021-02-01T01:40:03.5Z"}', b'["log","x"]', b'{"log":"unterminated',
                                   b'{"log":"a\\n","stream":7,"time":"2021-02-01T01:40:03.5Z"}', b'{"log":"a\\n","extra":{"k":[1,2.5,null]},"stream":"stdout"}']))
            continue
        log = rng.choice(["one, ", "two, ", "three\n", "\n", "", "caf\u00e9\n", "tab\there\n", "x" * rng.randrange(1, 300) + ("\n" if rng.random() < 0.5 else "")])
        d = {"log": log, "stream": rng.choice(["stdout", "stdout", "stderr"]), "time": "2021-02-01T01:%02d:%02d.%06dZ" % (rng.randrange(60), rng.randrange(60), rng.randrange(10 ** 6))}
        if rng.random() < 0.1:
            d = dict(reversed(list(d.items())))
        if rng.random() < 0.05:
            d["attrs"] = {"tag": "t%d" % i}
        out.append(json.dumps(d).encode())
    return b"\n".join(out) + b"\n"


def random_sub_case(rng, nlines=None, bad_times=True):
    """the built-in parsers with a parser in front: (cfg, frames, kwargs) like random_case"""
    b = rng.choice(["cri", "docker"])
    n = nlines if nlines is not None else rng.randrange(0, 80)
    text = cri_text(rng, n, bad_times=bad_times) if b == "cri" else docker_text(rng, n)
    if rng.random() < 0.2 and text:
        text = text[:-1]
    return {"builtin": b}, frames_of(rng, text), dict(skip_empty_lines=rng.random() < 0.4, final_flush=rng.random() < 0.7)
