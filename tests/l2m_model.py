"""Test-side model of the series row encoding (fluent-bit_amd/csrc/dev.hpp "filter_log_to_metrics"):
turns a list of observations into the integer rows the kernels would have produced, so that the
multi-rank merge (fluent_bit_amd.l2m_merge) and flbgpu_l2m_finalize_row can be exercised without a GPU."""
import struct
import numpy as np

W_FIRST, W_LASTIDX, W_LASTVAL, W_COUNT, W_SPECIAL, W_LIMB, NLIMB = 0, 1, 2, 3, 4, 7, 68
W_BUCKET = W_LIMB + NLIMB
M64 = (1 << 64) - 1


def row_words(mode, nb):
    return W_BUCKET + nb + 1 if mode == 2 else W_SPECIAL


def encode_rows(mode, bounds, observations):
    """observations: [(key_bytes, value_float, global_index)] -> (keys in first-appearance order, rows)"""
    nb = len(bounds)
    W = row_words(mode, nb)
    rows = {}
    for key, v, gidx in observations:
        r = rows.setdefault(key, [0] * W)
        r[W_FIRST] = max(r[W_FIRST], ~gidx & M64)
        bits = struct.unpack("<Q", struct.pack("<d", v))[0]
        if mode == 0:
            r[W_COUNT] += 1
        elif mode == 1:
            if gidx + 1 > r[W_LASTIDX]:
                r[W_LASTIDX] = gidx + 1
                r[W_LASTVAL] = bits
        else:
            b = nb
            for k in range(nb - 1, -1, -1):
                if v > bounds[k]:
                    break
                b = k
            r[W_BUCKET + b] += 1
            mag = bits & ~(1 << 63)
            if mag >= 0x7FF0000000000000:
                r[W_SPECIAL + (0 if mag > 0x7FF0000000000000 else (2 if bits >> 63 else 1))] += 1
            elif mag:
                e = mag >> 52
                m = (mag & ((1 << 52) - 1)) | ((1 << 52) if e else 0)
                s = (e - 1075 if e else -1074) + 1074
                wide = m << (s & 31)
                j = s >> 5
                sign = -1 if bits >> 63 else 1
                for k in range(3):
                    d = (wide >> (32 * k)) & 0xFFFFFFFF
                    r[W_LIMB + j + k] = (r[W_LIMB + j + k] + sign * d) & M64
    keys = sorted(rows, key=lambda k: ~rows[k][W_FIRST] & M64)
    return keys, np.array([rows[k] for k in keys], dtype=np.uint64).reshape(len(keys), W)


class SeqRank:
    """a rank's side of the chain of sum_order 2 (csrc/l2m.cpp "the chain") without a GPU: the interval's observations in record order
    and the sums the last flush ended on -- the same three entry points as FilterLogToMetrics, for fluent_bit_amd.l2m_chain"""

    def __init__(self):
        self.log = []            # (key, value)
        self.sums = {}

    def observe(self, key, value):
        self.log.append((key, float(value)))

    def chain_begin(self, keys):
        return [self.sums.get(k, 0.0) for k in keys]

    def seq_replay(self, keys, sums):
        at = {k: i for i, k in enumerate(keys)}
        acc = list(sums)
        for k, v in self.log:
            acc[at[k]] = acc[at[k]] + v          # one binary64 addition per observation, in record order
        self.log = []
        return acc

    def chain_end(self, keys, sums):
        for k, x in zip(keys, sums):
            self.sums[k] = x
