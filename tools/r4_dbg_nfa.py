"""debug aid (GPU): per-subject filter_parser outputs, device vs oracle, for the KAT patterns that fail through the NFA engine"""
import base64, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import oracle_binding as ob, synth, flbamd_loader
g = flbamd_loader.load(); g.init(0)
force = sys.argv[1] if len(sys.argv) > 1 else "1"
os.environ["FLBGPU_RX_FORCE_NFA"] = force
kat = json.load(open(os.path.join(ROOT, "tests", "golden", "regex_kat.json")))
nbad = 0
for ent in kat:
    pat = base64.b64decode(ent["pattern"])
    if not ent["compiles"] or pat.startswith(b"/") or b"\x00" in pat or not ent["names"]:
        continue
    subjects = [base64.b64decode(c[0]) for c in ent["cases"]]
    try:
        po = ob.Parser(pat, skip_empty=False); pg = g.Parser(pat, skip_empty=False)
    except ValueError:
        continue
    fpo = ob.FilterParser("log", [po]); fpg = g.FilterParser("log", [pg])
    blob = b"".join(synth.mp([[synth.ext_ts(7, i), {}], {"log": s}]) for i, s in enumerate(subjects))
    a0, b0 = fpo.filter(blob), fpg.filter(blob)
    if a0 != b0:
        for i, s in enumerate(subjects):
            one = synth.mp([[synth.ext_ts(7, i), {}], {"log": s}])
            a, b = fpo.filter(one), fpg.filter(one)
            if a != b:
                nbad += 1
                if nbad <= 12:
                    print("PAT", pat[:100]); print(" SUBJ", s); print("  oracle", a); print("  gpu   ", b)
        # whole-blob only differences (row interplay)
        if nbad == 0:
            print("blob-level difference only for", pat[:80])
            import msgpack
            a, b = a0, b0
            ra = list(msgpack.Unpacker(__import__("io").BytesIO(a[1]), raw=True, strict_map_key=False))
            rb = list(msgpack.Unpacker(__import__("io").BytesIO(b[1]), raw=True, strict_map_key=False))
            print(" records", len(ra), len(rb), "n subjects", len(subjects), "bytes", len(a[1]), len(b[1]), "rc", a[0], b[0])
            def spans(buf):
                u = msgpack.Unpacker(raw=True, strict_map_key=False); u.feed(buf); out = []; pos = 0
                for _ in u:
                    out.append(buf[pos:u.tell()]); pos = u.tell()
                return out
            sa, sb = spans(a[1]), spans(b[1])
            for i in range(max(len(sa), len(sb))):
                x = sa[i] if i < len(sa) else None; y = sb[i] if i < len(sb) else None
                if x != y:
                    print("  rec bytes", i, "subject", subjects[i] if i < len(subjects) else None, "\n   oracle", x, "\n   gpu   ", y)
            for i in range(max(len(ra), len(rb))):
                x = ra[i] if i < len(ra) else None; y = rb[i] if i < len(rb) else None
                if x != y:
                    print("  rec", i, "\n   oracle", x, "\n   gpu   ", y)
            for sub in (1, 2, 4, 8, 16, 32, 64, 80, 87):
                bl = b"".join(synth.mp([[synth.ext_ts(7, i), {}], {"log": s}]) for i, s in enumerate(subjects[:sub]))
                print("  first", sub, "equal:", fpo.filter(bl) == fpg.filter(bl))
    fpg.close(); pg.close()
print("bad subjects:", nbad)
