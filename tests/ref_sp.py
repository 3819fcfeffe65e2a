"""Binding to oracle/_ref/ref_sp: the reference's own stream processor (src/stream_processor/*.c compiled in place; the
flex / bison output written by hand in oracle/ref_sp_shim.c).  TEST INFRASTRUCTURE."""
import os
import struct
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_sp")


def available():
    return os.path.exists(BIN)


class RefSp:
    """one task of the reference's stream processor; `now` is what package_results stamps the records with"""

    def __init__(self, sql, str_conv=True, now=(1, 0)):
        self.p = subprocess.Popen([BIN], stdin=subprocess.PIPE, stdout=subprocess.PIPE)
        q = sql.encode() if isinstance(sql, str) else sql
        self.p.stdin.write(struct.pack("<IIII", 1, 1 if str_conv else 0, now[0], now[1]) + struct.pack("<I", len(q)) + q)
        self.p.stdin.flush()
        ret, _ = self._answer()
        self.ok = ret >= 0
        self.select_only = ret == 1          # no aggregation function: do() answers with sp_process_data's records

    def _answer(self):
        h = self.p.stdout.read(12)
        if len(h) != 12:
            raise RuntimeError("ref_sp died (rc %s)" % self.p.poll())
        ret, n = struct.unpack("<iQ", h)
        return ret, self.p.stdout.read(n) if n else b""

    def do(self, chunk):
        """flb_sp_do for one appended chunk: (records in the window, packaged output when the query has no WINDOW)"""
        self.p.stdin.write(struct.pack("<IQ", 2, len(chunk)) + bytes(chunk))
        self.p.stdin.flush()
        return self._answer()

    def timer(self):
        """the window's timer fires: packaged results, window pruned"""
        self.p.stdin.write(struct.pack("<I", 3))
        self.p.stdin.flush()
        return self._answer()[1]

    def hop(self):
        """the hop timer of a HOPPING window fires (sp_process_hopping_slot)"""
        self.p.stdin.write(struct.pack("<I", 5))
        self.p.stdin.flush()
        return self._answer()[0]

    def close(self):
        if self.p:
            try:
                self.p.stdin.close()
                self.p.wait(timeout=5)
            except Exception:
                self.p.kill()
            self.p = None

    def __del__(self):
        self.close()
