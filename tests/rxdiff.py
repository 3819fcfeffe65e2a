"""Differential helpers: oracle regex (oracle/orx.c) and product compiler vs the REAL Onigmo
(oracle/_ref/libonig_ref.so, built from /root/reference/lib/onigmo by oracle/Makefile)."""
import ctypes, os, random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def load_ref():
    p = os.path.join(ROOT, "oracle", "_ref", "libonig_ref.so")
    if not os.path.exists(p):
        return None
    L = ctypes.CDLL(p)
    L.ref_onig_new.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint, ctypes.POINTER(ctypes.c_void_p)]
    L.ref_onig_search.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    L.ref_onig_match.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    L.ref_onig_names.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    L.ref_onig_free.argtypes = [ctypes.c_void_p]
    return L

class RefRegex:
    def __init__(self, L, pat, options=0):
        self.L = L
        self.reg = ctypes.c_void_p()
        self.rc = L.ref_onig_new(pat, len(pat), options, ctypes.byref(self.reg))
        self.ok = self.rc == 0
    def search(self, s):
        beg = (ctypes.c_int * 64)(); end = (ctypes.c_int * 64)()
        n = self.L.ref_onig_search(self.reg, s, len(s), beg, end, 64)
        if n < 0:
            return None
        return [(beg[i], end[i]) for i in range(n)]
    def names(self):
        buf = ctypes.create_string_buffer(8192)
        self.L.ref_onig_names(self.reg, buf, 8192)
        out = []
        for line in buf.value.decode().splitlines():
            n, g = line.rsplit("=", 1)
            out.append((n, int(g)))
        return out

def load_orx(path=None):
    p = path or os.path.join(ROOT, "oracle", "liboracle.so")
    L = ctypes.CDLL(p)
    L.orx_compile.restype = ctypes.c_void_p
    L.orx_compile.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint, ctypes.c_char_p, ctypes.c_int]
    L.orx_search.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    L.orx_free.argtypes = [ctypes.c_void_p]
    L.orx_num_names.argtypes = [ctypes.c_void_p]
    L.orx_name.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.orx_name.restype = ctypes.c_char_p
    L.orx_name_ngroups.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.orx_name_group.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    return L

class OrxRegex:
    def __init__(self, L, pat, options=0):
        self.L = L
        err = ctypes.create_string_buffer(256)
        self.rx = L.orx_compile(pat, len(pat), options, err, 256)
        self.ok = bool(self.rx)
        self.err = err.value.decode()
    def search(self, s):
        beg = (ctypes.c_int * 64)(); end = (ctypes.c_int * 64)()
        n = self.L.orx_search(self.rx, s, len(s), beg, end, 64)
        if n < 0:
            return None
        return [(beg[i], end[i]) for i in range(n)]
    def names(self):
        out = []
        for i in range(self.L.orx_num_names(self.rx)):
            nm = self.L.orx_name(self.rx, i).decode()
            for k in range(self.L.orx_name_ngroups(self.rx, i)):
                out.append((nm, self.L.orx_name_group(self.rx, i, k)))
        return out

# pattern corpus: parsers.conf-style patterns + syntax coverage
PATTERNS = [
    rb'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^ ]*) +\S*)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>.*)")?$',
    rb'^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \[(?<time>[^\]]*)\] "(?<method>\S+)(?: +(?<path>[^\"]*?)(?: +\S*)?)?" (?<code>[^ ]*) (?<size>[^ ]*)(?: "(?<referer>[^\"]*)" "(?<agent>[^\"]*)")?$',
    rb'^\[[^ ]* (?<time>[^\]]*)\] \[(?<level>[^\]]*)\](?: \[pid (?<pid>[^\]]*)\])?( \[client (?<client>[^\]]*)\])? (?<message>.*)$',
    rb'^(?<time>[^ ]* {1,2}[^ ]* [^ ]*) (?<host>[^ ]*) (?<ident>[a-zA-Z0-9_\/\.\-]*)(?:\[(?<pid>[0-9]+)\])?(?:[^\:]*\:)? *(?<message>.*)$',
    rb'^\<(?<pri>[0-9]+)\>(?<time>[^ ]* {1,2}[^ ]* [^ ]*) (?<host>[^ ]*) (?<ident>[a-zA-Z0-9_\/\.\-]*)(?:\[(?<pid>[0-9]+)\])?(?:[^\:]*\:)? *(?<message>.*)$',
    rb'^(?<time>.+) (?<stream>stdout|stderr) (?<logtag>[^ ]*) (?<message>.*)$',
    rb'(?<tag>[^.]+)?\.?(?<pod_name>[a-z0-9](?:[-a-z0-9]*[a-z0-9])?(?:\.[a-z0-9]([-a-z0-9]*[a-z0-9])?)*)_(?<namespace_name>[^_]+)_(?<container_name>.+)-(?<docker_id>[a-z0-9]{64})\.log$',
    rb'^(?<INT>[^ ]+) (?<FLOAT>[^ ]+) (?<BOOL>[^ ]+) (?<STRING>.+)$',
    rb'^(?<key1>[^ ]*) (?<key2>[^ ]*) (?<time>.+)$',
    rb'a', rb'abc', rb'a|b', rb'ab|cd|ef', rb'a*', rb'a+', rb'a?', rb'a*?b', rb'a+?', rb'a??b',
    rb'(a|ab)(c|bcd)', rb'(a|ab)(c|bcd)(d*)', rb'(a*)*', rb'(a*)+', rb'(a|b)*c', rb'(?:a|b)*?c',
    rb'a{2}', rb'a{2,}', rb'a{2,3}', rb'a{,3}', rb'a{2,3}?', rb'(ab){2,3}c', rb'a{', rb'a{x}', rb'{a}',
    rb'^a', rb'a$', rb'^$', rb'^', rb'$', rb'\Aa', rb'a\z', rb'a\Z', rb'\bfoo\b', rb'\Bfoo', rb'foo\B',
    rb'[abc]', rb'[^abc]', rb'[a-c]+', rb'[^a-c]+', rb'[a\-c]', rb'[]a]', rb'[^]a]', rb'[a-]', rb'[-a]',
    rb'\d+', rb'\D+', rb'\w+', rb'\W+', rb'\s+', rb'\S+', rb'\h+', rb'\H+', rb'[\d\s]+', rb'[^\d\s]+', rb'[\D]+',
    rb'[[:alpha:]]+', rb'[[:digit:][:space:]]+', rb'[[:^alpha:]]+', rb'[^[:alpha:]]+', rb'[[:punct:]]+', rb'[[:xdigit:]]+',
    rb'[a[bc]d]+', rb'[a-c[x-z]]+', rb'.', rb'.*', rb'.+', rb'a.c', rb'a.*c', rb'a.*?c', rb'(?m)a.c', rb'(?m:a.)c',
    rb'(?i)abc', rb'(?i:a)bc', rb'a(?i)b|c', rb'(?i)[a-c]x', rb'(?i)[^a-c]x', rb'(?-i)abc', rb'(?i)a(?-i)b',
    rb'(?x) a b c # comment', rb'(?x)a\ b', rb'(?#comment)ab',
    rb'(a)(b)?', rb'(a)|(b)', rb'(?<x>a)|(?<y>b)', rb'(?<x>a)(b)', rb'(?<x>a)(?<x>b)', rb'(?<n>a)*', rb'(?<n>a|b)+',
    rb'(?:(?<a>x)|(?<b>y))+', rb'(a+)+b', rb'(a|b|ab)*c', rb'x*y*z*', rb'(x*)(y*)(z*)', rb'(x+x+)+y',
    rb'\x41', rb'\x{41}', rb'\101', rb'\t\n', rb'\.', rb'\\', rb'\/', rb'a\|b', rb'A',
    rb'(?=a)ab', rb'(?!a)b', rb'a(?=b)', rb'a(?!b)', rb'(?>a+)b', rb'(?>a|ab)c', rb'a*+a', rb'a++', rb'a?+a', rb'a{2,3}+',
    rb'^.* 5\d\d ', rb'error|warn|fatal', rb'GET|POST', rb'^\d+\.\d+\.\d+\.\d+', rb'(\d+)-(\d+)', rb'[0-9]{4}-[0-9]{2}-[0-9]{2}',
    rb'^[^ ]+ [^ ]+', rb'"[^"]*"', rb'\[([^\]]*)\]', rb'=(\S*)', rb'(?<k>\w+)=(?<v>\w*)', rb'https?://[^/ ]+/', rb'\.(gif|png|jpe?g)$',
    rb'^(?<a>a*)(?<b>a*)$', rb'^(?<a>a*?)(?<b>a*)$', rb'^(?<a>a|ab)(?<b>bc|c)?$', rb'(?<a>.*) (?<b>.*)', rb'(?<a>.*?) (?<b>.*?)',
    "é+".encode(), "[é-ü]+".encode(), "[^é]+".encode(), "(?<w>\\S+) é".encode(), "日本(?<x>.)".encode(),
    rb'a**', rb'a+*', rb'(a*)*b', rb'(a*?)*b', rb'(|a)*b', rb'(a|)*b', rb'()*', rb'(a?)*?b', rb'(?:a?){3}', rb'(?:a?){2,}b', rb'(a|b*)*c',
    rb'\n', rb'a\nb', rb'^b', rb'a$\nb', rb'(?m).*', rb'\s', rb'[\n]', rb'[^\n]+',
    # how each class form treats input that is not well-formed UTF-8 (bit set only / code-range part / mixed,
    # positive and negated, the word opcodes, folded letters)
    rb'[\S]+', rb'[\W]+', rb'[^\w]+', rb'[^\s]+x', rb'[\D]+', rb'[\H]x', rb'[[:^alpha:]]+', rb'[[:^digit:]]+x', rb'(?i)[^a]+', rb'(?i)k+', rb'(?i)[r-t]+',
    rb'[a\x{e9}]+', rb'[\x{80}-\x{ff}]+', rb'[^\x{80}-\x{ff}]+', rb'[a-\x{e9}]+', rb'\x{e9}', rb'[^\x{e2}]+', rb'[a[^b]]+', rb'[^a[^b]]+', rb'\W\w', rb'x.y', rb'(?m)x.y',
    rb'^(?<a>[^ ]*) (?<b>\S+) (?<c>.*)$', rb'^(?<a>[^"]*)"(?<b>[\w]*)', rb'(?<a>\W+)(?<b>\w+)',
]

ALPH_ASCII = b'abcxyz019 _-.:/"[]=\n\tAB5'

def rand_input_illformed(rng, pat=None, maxlen=24):
    """text with stray bytes >= 0x80: Latin-1 letters, lone leads, stray continuations, sequences cut by
    the next character or by the end of the text, overlongs, surrogates, bytes that never occur"""
    n = rng.randint(1, maxlen)
    out = bytearray()
    lits = bytes(c for c in (pat or b'') if c < 0x80 and (chr(c).isalnum() or c in b' _-.:/"=')) or b'a'
    frag = [b'\xe9', b'\xc3', b'\xa9', b'\x80', b'\xbf', b'\xe2\x82', b'\xf0\x9f', b'\xf0\x9f\x98', b'\xff', b'\xfe', b'\xc0\x80', b'\xc1',
            b'\xed\xa0\x80', b'\xf4\x90\x80\x80', b'\xf5', b'\xe0\x80\x80', b'\xc3\xa9', b'\xe2\x82\xac', b'\xf0\x9f\x98\x80', b'\xc5\xbf', b'\xe2\x84\xaa',
            b'\xe2', b'\xf0', b'\xdf']
    for _ in range(n):
        r = rng.random()
        if r < 0.3:
            out += rng.choice(frag)
        elif r < 0.4:
            out.append(rng.randrange(0x80, 0x100))
        elif r < 0.7:
            out.append(rng.choice(lits))
        else:
            out.append(rng.choice(ALPH_ASCII))
    return bytes(out[:maxlen + 4])

def rand_input(rng, pat=None, maxlen=24, utf8=False):
    n = rng.randint(0, maxlen)
    alph = ALPH_ASCII
    out = bytearray()
    # bias with literal bytes from the pattern
    lits = bytes(c for c in (pat or b'') if c < 0x80 and (chr(c).isalnum() or c in b' _-.:/"=')) or b'a'
    for _ in range(n):
        r = rng.random()
        if utf8 and r < 0.15:
            # letters of both cases, a digit, a superscript (a word character for \b, not for [[:word:]]), punctuation, a
            # blank, a full-width letter, the Kelvin sign, symbols outside every class
            out += rng.choice(["é", "ü", "日", "本", "ß", "€", "😀", "É", "Ж", "ж", "٣", "²", "¡", "«", "\u00a0", "\u3000", "Ａ", "\u212a", "ǅ", "\u0301",
                               "\u00ad", "\u2028"]).encode()
        elif r < 0.5:
            out.append(rng.choice(lits))
        else:
            out.append(rng.choice(alph))
    return bytes(out)
