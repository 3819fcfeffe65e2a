"""flbgpu_index_host + flbgpu_tail_clean_host (host logic of the product, no GPU needed) against the pinned
msgpack reader of the oracle: the product must agree on where the last whole object ends and on whether
the bytes behind it are a clean end (decoder offset == chunk size, plugins/filter_grep/grep.c:357-360)."""
import flbamd_loader
from test_msgpack_pin import corpus, run_oracle


def test_host_walk_and_clean_tail_rule_match_msgpack_c():
    g = flbamd_loader.load()
    L = g.lib()
    n_clean = n_cut = 0
    for s in corpus(11, 5000):
        codes, ends, _ = run_oracle(s)
        if codes[-1] == -2:
            continue                                     # deeper than msgpack-c's 32-level stack: documented deviation
        nobj = sum(1 for c in codes if c == 2)
        n, off, consumed = g.index_host(s)
        assert n == nobj and consumed == (ends[nobj - 1] if nobj else 0), s[:60]
        want_clean = codes[-1] == 0 and ends[-1] == len(s)
        assert bool(L.flbgpu_tail_clean_host(s, len(s), consumed)) == want_clean, (s[:60], codes, ends)
        n_clean += want_clean and consumed != len(s)
        n_cut += not want_clean
    assert n_clean > 50 and n_cut > 200                  # both sides of the rule were exercised
