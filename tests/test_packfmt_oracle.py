"""The oracle's msgpack -> JSON formatter (oracle/opackfmt.c) against the reference's real
flb_pack_msgpack_to_json_format: the committed known answers (tests/golden/packfmt_kat.json) and, where the
reference build is present (oracle/_ref/ref_packfmt), a fresh random corpus."""
import base64, json, os, struct, subprocess
import pytest
import oracle_binding as ob
import packfmt_cases

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.path.join(ROOT, "oracle", "_ref", "ref_packfmt")


def run_reference(cases):
    """[(cfg, chunk)] -> [bytes | None] through the batch mode of oracle/ref_packfmt_shim.c"""
    inp = bytearray()
    for cfg, data in cases:
        dk = cfg["date_key"]
        inp += struct.pack("<IIIIi", cfg["json_format"], cfg["date_format"], cfg["escape_unicode"], cfg["nan_to_null"], -1 if dk is None else len(dk))
        if dk: inp += dk
        inp += struct.pack("<Q", len(data)) + data
    r = subprocess.run([REF, "batch"], input=bytes(inp), capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr[-500:]
    outs, p, b = [], 0, r.stdout
    while p < len(b):
        (n,) = struct.unpack_from("<q", b, p); p += 8
        if n < 0: outs.append(None)
        else: outs.append(b[p:p + n]); p += n
    assert len(outs) == len(cases)
    return outs


def oracle_out(cfg, data):
    return ob.msgpack_to_json_format(data, cfg["json_format"], cfg["date_format"], cfg["date_key"], cfg["escape_unicode"], cfg["nan_to_null"])


def load_kat():
    kat = json.load(open(os.path.join(HERE, "golden", "packfmt_kat.json")))["cases"]
    out = []
    for k in kat:
        cfg = dict(k["cfg"])
        cfg["date_key"] = None if cfg["date_key"] is None else base64.b64decode(cfg["date_key"])
        out.append((cfg, base64.b64decode(k["in"]), None if k["out"] is None else base64.b64decode(k["out"])))
    return out


def test_oracle_against_golden():
    kat = load_kat()
    assert len(kat) > 1000
    for cfg, data, want in kat:
        assert oracle_out(cfg, data) == want, (cfg, data.hex())


@pytest.mark.skipif(not os.path.exists(REF), reason="reference build absent")
def test_oracle_against_live_reference():
    cases = packfmt_cases.corpus(777, 3000)
    outs = run_reference(cases)
    for (cfg, data), want in zip(cases, outs):
        assert oracle_out(cfg, data) == want, (cfg, data.hex())
