// numconv.hpp -- text <-> binary64 conversions that must agree bit for bit with the libc calls the
// reference makes on this path:
//   sscanf("%lf")   plugins/filter_log_to_metrics/log_to_metrics.c:1064,1095   (value_field)
//   atof/strtod     src/flb_parser.c:2121 (Types float), src/flb_pack.c reals (JSON -> msgpack)
//   snprintf("%f")  plugins/filter_log_to_metrics/log_to_metrics.c:1027        (float label)
//   snprintf("%ld") plugins/filter_log_to_metrics/log_to_metrics.c:1031        (int label)
// libc is the semantic definition ("device code needs correctly-rounded
// equivalents"); what is written here is integer arithmetic only, shared by the HIP kernels and by a
// host build of the same functions that the CPU tests fuzz against glibc.
//
// decimal -> double: the first 19 significant digits go through the Eisel-Lemire product test
// (D. Lemire, "Number parsing at a gigabyte per second", 2021) against the 128-bit powers of five in
// pow5_table.inc; when digits were dropped and the two bracketing products disagree -- or always, if
// the caller asks for it -- an exact big-integer comparison against the neighbouring doubles decides.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NC_HD __host__ __device__ __forceinline__
#define NC_HD_NOINL __host__ __device__ inline
#else
#define NC_HD inline
#define NC_HD_NOINL inline
#endif

namespace flbgpu {
namespace nc {

constexpr int P5_QMIN = -342, P5_QMAX = 308;

#if defined(__HIP_DEVICE_COMPILE__)
// (static: every translation unit that converts on the device carries its own copy)
static __device__ const uint64_t g_pow5_dev[2 * (P5_QMAX - P5_QMIN + 1)] = {
#include "pow5_table.inc"
};
#define NC_P5(i) (::flbgpu::nc::g_pow5_dev[i])
#else
extern const uint64_t g_pow5_host[2 * (P5_QMAX - P5_QMIN + 1)];
#define NC_P5(i) (::flbgpu::nc::g_pow5_host[i])
#endif

NC_HD int clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long) x);
#else
    return __builtin_clzll(x);
#endif
}

NC_HD void mul64(uint64_t a, uint64_t b, uint64_t &hi, uint64_t &lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    hi = __umul64hi(a, b);
    lo = a * b;
#else
    unsigned __int128 p = (unsigned __int128) a * b;
    hi = (uint64_t) (p >> 64);
    lo = (uint64_t) p;
#endif
}

constexpr uint64_t DBL_INF_BITS = 0x7FF0000000000000ull;
constexpr uint64_t DBL_NAN_BITS = 0x7FF8000000000000ull;     // glibc strtod("nan"): positive quiet NaN, zero payload
constexpr uint64_t DBL_SIGN = 0x8000000000000000ull;

// ---- m * 2^e (m != 0), round to nearest even; `sticky` = bits were already discarded below m
NC_HD uint64_t make_double_bits(uint64_t m, int64_t e, bool sticky) {
    int lz = clz64(m);
    m <<= lz;
    e -= lz;
    int64_t E = e + 63;                       // exponent of the leading bit
    if (E > 1023) return DBL_INF_BITS;
    int shift = 11;
    if (E < -1022) {
        int64_t extra = -1022 - E;
        if (extra > 64) return 0;
        shift += (int) extra;
    }
    uint64_t mant, rem, half;
    if (shift >= 64) {
        // the whole significand lies below the last subnormal bit
        if (shift > 64) return 0;             // < 2^-1075: rounds to zero (shift == 65.. handled above via extra)
        mant = 0; rem = m; half = 1ull << 63;
    }
    else {
        mant = m >> shift;
        rem = m & ((1ull << shift) - 1);
        half = 1ull << (shift - 1);
    }
    bool up = rem > half || (rem == half && (sticky || (mant & 1)));
    mant += up ? 1 : 0;
    if (E < -1022) return mant;               // subnormal (a carry into 2^52 is the smallest normal: same encoding)
    if (mant == (1ull << 53)) { mant >>= 1; E++; if (E > 1023) return DBL_INF_BITS; }
    return ((uint64_t) (E + 1023) << 52) | (mant & ((1ull << 52) - 1));
}

// ---- Eisel-Lemire: w * 10^q -> binary64 bits (w != 0).  Returns false when the product test is
// inconclusive (the caller falls back to the exact comparison).
NC_HD bool eisel_lemire(uint64_t w, int64_t q, uint64_t &bits) {
    if (q < P5_QMIN) { bits = 0; return true; }
    if (q > P5_QMAX) { bits = DBL_INF_BITS; return true; }
    int lz = clz64(w);
    w <<= lz;
    const int idx = 2 * (int) (q - P5_QMIN);
    uint64_t hi, lo;
    mul64(w, NC_P5(idx), hi, lo);
    if ((hi & 0x1FF) == 0x1FF) {              // the 55 bits that decide are followed by all ones: refine
        uint64_t hi2, lo2;
        mul64(w, NC_P5(idx + 1), hi2, lo2);
        lo += hi2;
        if (hi2 > lo) hi++;
    }
    // not separable with 128 bits of 5^q (only possible outside the exactly representable
    // powers): leave it to the exact path
    if (lo == 0xFFFFFFFFFFFFFFFFull && !(q >= -27 && q <= 55)) return false;
    const int upperbit = (int) (hi >> 63);
    const int shift = upperbit + 9;           // 64 - 52 - 3
    uint64_t mant = hi >> shift;
    int64_t p2 = ((217706 * q) >> 16) + 63 + upperbit - lz + 1023;
    if (p2 <= 0) {                            // subnormal
        if (-p2 + 1 >= 64) { bits = 0; return true; }
        mant >>= (int) (-p2 + 1);
        mant += (mant & 1);
        mant >>= 1;
        bits = mant;                          // mant == 2^52 is the smallest normal: same encoding
        return true;
    }
    if (lo <= 1 && q >= -4 && q <= 23 && (mant & 3) == 1) {
        if ((mant << shift) == hi) mant &= ~1ull;      // exactly half way: round to even (down)
    }
    mant += (mant & 1);
    mant >>= 1;
    if (mant >= (2ull << 52)) { mant = 1ull << 52; p2++; }
    mant &= ~(1ull << 52);
    if (p2 >= 0x7FF) { bits = DBL_INF_BITS; return true; }
    bits = ((uint64_t) p2 << 52) | mant;
    return true;
}

// ------------------------------------------------------------------------------------------
// exact path: big integers in 32-bit limbs
// ------------------------------------------------------------------------------------------
constexpr int BIG_LIMBS = 144;                // 4608 bits: 800 digits * 5^1130 never meet in one operand (see cmp_dec_bin)
constexpr int MAX_SIG_DIGITS = 800;           // > 767 = longest exact decimal expansion of a binary64 midpoint

struct Big {
    uint32_t d[BIG_LIMBS];
    int n;
};

NC_HD_NOINL void big_set(Big &b, uint64_t v) {
    b.n = 0;
    if (v) { b.d[b.n++] = (uint32_t) v; if (v >> 32) b.d[b.n++] = (uint32_t) (v >> 32); }
}
NC_HD_NOINL void big_muladd(Big &b, uint32_t m, uint32_t a) {
    uint64_t carry = a;
    for (int i = 0; i < b.n; i++) {
        uint64_t t = (uint64_t) b.d[i] * m + carry;
        b.d[i] = (uint32_t) t;
        carry = t >> 32;
    }
    if (carry && b.n < BIG_LIMBS) b.d[b.n++] = (uint32_t) carry;
}
NC_HD_NOINL void big_mul_pow5(Big &b, int64_t e) {
    while (e >= 13) { big_muladd(b, 1220703125u, 0); e -= 13; }
    uint32_t m = 1;
    while (e-- > 0) m *= 5;
    if (m > 1) big_muladd(b, m, 0);
}
NC_HD_NOINL int64_t big_bitlen(const Big &b) {
    if (b.n == 0) return 0;
    return (int64_t) 32 * (b.n - 1) + (32 - (clz64((uint64_t) b.d[b.n - 1]) - 32));
}
NC_HD_NOINL void big_shl(Big &b, int64_t s) {
    if (b.n == 0 || s == 0) return;
    int ws = (int) (s / 32), bs = (int) (s % 32);
    int nn = b.n + ws + 1;
    if (nn > BIG_LIMBS) nn = BIG_LIMBS;
    for (int i = nn - 1; i >= 0; i--) {
        int src = i - ws;
        uint64_t v = 0;
        if (src >= 0 && src < b.n) v |= (uint64_t) b.d[src] << bs;
        if (bs && src - 1 >= 0 && src - 1 < b.n) v |= (uint64_t) b.d[src - 1] >> (32 - bs);
        b.d[i] = (uint32_t) v;
    }
    b.n = nn;
    while (b.n > 0 && b.d[b.n - 1] == 0) b.n--;
}
NC_HD_NOINL int big_cmp(const Big &a, const Big &b) {
    if (a.n != b.n) return a.n < b.n ? -1 : 1;
    for (int i = a.n - 1; i >= 0; i--) if (a.d[i] != b.d[i]) return a.d[i] < b.d[i] ? -1 : 1;
    return 0;
}

// sign of (D * 10^q [+ epsilon when sticky]) - M * 2^E.  `L` holds D on entry and is consumed.
NC_HD_NOINL int cmp_dec_bin(Big &L, int64_t q, bool sticky, uint64_t M, int64_t E, Big &R) {
    big_set(R, M);
    if (q >= 0) big_mul_pow5(L, q); else big_mul_pow5(R, -q);
    // L * 2^q  ?  R * 2^E   <=>   L * 2^(q - E)  ?  R
    int64_t sh = q - E;
    int64_t bl = big_bitlen(L) + (sh > 0 ? sh : 0), br = big_bitlen(R) + (sh < 0 ? -sh : 0);
    if (R.n == 0) return 1;
    if (bl != br) return bl < br ? -1 : 1;
    if (sh > 0) big_shl(L, sh); else if (sh < 0) big_shl(R, -sh);
    int c = big_cmp(L, R);
    if (c == 0 && sticky) return 1;
    return c;
}

// binary64 with encoding `bits` (finite, >= 0) as M * 2^E; `mid` selects the midpoint to bits + 1
NC_HD void bits_to_me(uint64_t bits, bool mid, uint64_t &M, int64_t &E) {
    uint64_t f = bits & ((1ull << 52) - 1);
    int64_t e = (int64_t) (bits >> 52);
    if (e == 0) { M = f; E = -1074; }
    else { M = f | (1ull << 52); E = e - 1075; }
    if (mid) { M = 2 * M + 1; E -= 1; }
}

// ------------------------------------------------------------------------------------------
// scanner
// ------------------------------------------------------------------------------------------
enum { MODE_STRTOD = 0, MODE_SSCANF = 1 };
enum { NC_FAIL = 0, NC_OK = 1, NC_NEED_EXACT = 2 };

NC_HD bool is_space(uint32_t c) { return c == ' ' || (c >= 9 && c <= 13); }
NC_HD uint32_t lower(uint32_t c) { return (c >= 'A' && c <= 'Z') ? c + 32 : c; }
NC_HD int hexval(uint32_t c) {
    if (c >= '0' && c <= '9') return (int) c - '0';
    c = lower(c);
    if (c >= 'a' && c <= 'f') return (int) c - 'a' + 10;
    return -1;
}

// decimal significand [mb, me) (digits and at most one '.') with decimal exponent `ex` -> bits.
// EXACT selects the big-integer resolution of the doubtful cases; without it they are reported.
template <bool EXACT, class Src>
NC_HD int dec_to_bits(const Src &s, uint32_t mb, uint32_t me, int64_t ex, uint64_t &bits) {
    uint64_t w = 0;
    int nd = 0;
    int64_t fracz = 0, dropped = 0;
    bool trunc = false, seen_dot = false;
    for (uint32_t i = mb; i < me; i++) {
        uint32_t c = s[i];
        if (c == '.') { seen_dot = true; continue; }
        uint32_t d = c - '0';
        if (nd == 0 && d == 0) { if (seen_dot) fracz++; continue; }
        if (nd < 19) { w = w * 10 + d; nd++; if (seen_dot) fracz++; }
        else { if (d) trunc = true; if (!seen_dot) dropped++; }
    }
    if (w == 0) { bits = 0; return NC_OK; }
    int64_t q = ex + dropped - fracz;
    uint64_t b0 = 0;
    bool ok = eisel_lemire(w, q, b0);
    if (ok && !trunc) { bits = b0; return NC_OK; }
    if (ok && trunc) {
        uint64_t b1 = 0;
        if (eisel_lemire(w + 1, q, b1) && b1 == b0) { bits = b0; return NC_OK; }
    }
    if (!EXACT) return NC_NEED_EXACT;
    // ---- exact: D = up to MAX_SIG_DIGITS significant digits (+ sticky), V = D * 10^qd
    Big D, L, R;
    D.n = 0;
    int nsig = 0;
    int64_t fz = 0, dr = 0;
    bool sticky = false;
    seen_dot = false;
    for (uint32_t i = mb; i < me; i++) {
        uint32_t c = s[i];
        if (c == '.') { seen_dot = true; continue; }
        uint32_t d = c - '0';
        if (nsig == 0 && d == 0) { if (seen_dot) fz++; continue; }
        if (nsig < MAX_SIG_DIGITS) { big_muladd(D, 10, d); nsig++; if (seen_dot) fz++; }
        else { if (d) sticky = true; if (!seen_dot) dr++; }
    }
    int64_t qd = ex + dr - fz;
    int64_t sci = qd + nsig - 1;              // decimal exponent of the leading digit
    if (sci > 309) { bits = DBL_INF_BITS; return NC_OK; }
    if (sci < -326) { bits = 0; return NC_OK; }
    uint64_t b = ok ? b0 : 0;
    if (b >= DBL_INF_BITS) b = DBL_INF_BITS - 1;
    uint64_t M;
    int64_t E;
    // largest double <= V
    for (;;) {
        if (b == 0) break;
        bits_to_me(b, false, M, E);
        L = D;
        if (cmp_dec_bin(L, qd, sticky, M, E, R) >= 0) break;
        b--;
    }
    for (;;) {
        if (b + 1 >= DBL_INF_BITS) break;
        bits_to_me(b + 1, false, M, E);
        L = D;
        if (cmp_dec_bin(L, qd, sticky, M, E, R) < 0) break;
        b++;
    }
    bits_to_me(b, true, M, E);
    L = D;
    int c = cmp_dec_bin(L, qd, sticky, M, E, R);
    if (c > 0 || (c == 0 && (b & 1))) b++;
    bits = b;
    return NC_OK;
}

struct ScanResult {
    int status;          // NC_*
    uint64_t bits;       // binary64 encoding (sign included)
    uint32_t consumed;   // bytes of the longest valid prefix (strtod's endptr); 0 = no conversion
};

// strtod() / sscanf("%lf") over s[0 .. len).  Reading stops at len or at a NUL byte.
template <bool EXACT, class Src>
NC_HD ScanResult scan_double(const Src &s, uint32_t len, int mode) {
    ScanResult r;
    r.status = NC_FAIL; r.bits = 0; r.consumed = 0;
    uint32_t i = 0;
    // effective length: C strings end at the first NUL
    {
        uint32_t z = 0;
        while (z < len && s[z] != 0) z++;
        len = z;
    }
    while (i < len && is_space(s[i])) i++;
    uint64_t sign = 0;
    if (i < len && (s[i] == '-' || s[i] == '+')) { if (s[i] == '-') sign = DBL_SIGN; i++; }
    if (i >= len) return r;
    uint32_t c0 = lower(s[i]);
    if (c0 == 'i') {
        if (i + 3 <= len && lower(s[i + 1]) == 'n' && lower(s[i + 2]) == 'f') {
            uint32_t j = i + 3;
            bool full = j + 5 <= len && lower(s[j]) == 'i' && lower(s[j + 1]) == 'n' && lower(s[j + 2]) == 'i' &&
                        lower(s[j + 3]) == 't' && lower(s[j + 4]) == 'y';
            // vfscanf commits to "infinity" once it has seen the second 'i' (stdio-common/vfscanf-internal.c)
            if (mode == MODE_SSCANF && !full && j < len && lower(s[j]) == 'i') return r;
            r.status = NC_OK; r.bits = sign | DBL_INF_BITS; r.consumed = full ? j + 5 : j;
        }
        return r;
    }
    if (c0 == 'n') {
        if (i + 3 <= len && lower(s[i + 1]) == 'a' && lower(s[i + 2]) == 'n') {
            uint32_t j = i + 3;
            if (j < len && s[j] == '(') {      // nan(n-char-sequence): accepted only when closed
                uint32_t k = j + 1;
                while (k < len) {
                    uint32_t c = s[k];
                    bool alnum = (c >= '0' && c <= '9') || (lower(c) >= 'a' && lower(c) <= 'z') || c == '_';
                    if (!alnum) break;
                    k++;
                }
                if (k < len && s[k] == ')') j = k + 1;
            }
            r.status = NC_OK; r.bits = sign | DBL_NAN_BITS; r.consumed = j;
        }
        return r;
    }
    // hexadecimal
    if (s[i] == '0' && i + 1 < len && lower(s[i + 1]) == 'x') {
        uint32_t j = i + 2;
        uint64_t m = 0;
        bool sticky = false, any = false, dot = false;
        int64_t e2 = 0;
        uint32_t k = j;
        for (; k < len; k++) {
            uint32_t c = s[k];
            if (c == '.' && !dot) { dot = true; continue; }
            int h = hexval(c);
            if (h < 0) break;
            any = true;
            if (m >> 60) { if (h) sticky = true; if (!dot) e2 += 4; }
            else { m = (m << 4) | (uint64_t) h; if (dot) e2 -= 4; }
        }
        if (any) {
            uint32_t endp = k;
            if (k < len && lower(s[k]) == 'p') {
                uint32_t p = k + 1;
                bool eneg = false;
                if (p < len && (s[p] == '-' || s[p] == '+')) { eneg = s[p] == '-'; p++; }
                if (p < len && s[p] >= '0' && s[p] <= '9') {
                    int64_t ev = 0;
                    while (p < len && s[p] >= '0' && s[p] <= '9') { if (ev < 100000000) ev = ev * 10 + (s[p] - '0'); p++; }
                    e2 += eneg ? -ev : ev;
                    endp = p;
                }
            }
            r.status = NC_OK; r.consumed = endp;
            r.bits = sign | (m ? make_double_bits(m, e2, sticky) : 0);
            return r;
        }
        // "0x" without hex digits: strtod converts the "0"; vfscanf reports a matching failure
        // unless a '.' follows (its buffer is then "0x.", which strtod reads as 0)
        if (mode == MODE_SSCANF && !(j < len && s[j] == '.')) return r;
        r.status = NC_OK; r.bits = sign; r.consumed = i + 1;
        return r;
    }
    // decimal
    uint32_t mb = i, k = i;
    bool dot = false, any = false;
    for (; k < len; k++) {
        uint32_t c = s[k];
        if (c == '.' && !dot) { dot = true; continue; }
        if (c < '0' || c > '9') break;
        any = true;
    }
    if (!any) return r;
    uint32_t me = k, endp = k;
    int64_t ex = 0;
    if (k < len && lower(s[k]) == 'e') {
        uint32_t p = k + 1;
        bool eneg = false;
        if (p < len && (s[p] == '-' || s[p] == '+')) { eneg = s[p] == '-'; p++; }
        if (p < len && s[p] >= '0' && s[p] <= '9') {
            int64_t ev = 0;
            while (p < len && s[p] >= '0' && s[p] <= '9') { if (ev < 100000000) ev = ev * 10 + (s[p] - '0'); p++; }
            ex = eneg ? -ev : ev;
            endp = p;
        }
    }
    uint64_t b = 0;
    int st = dec_to_bits<EXACT>(s, mb, me, ex, b);
    r.status = st; r.bits = sign | b; r.consumed = endp;
    return r;
}

// ------------------------------------------------------------------------------------------
// formatting
// ------------------------------------------------------------------------------------------
// "%ld": writes at most 20 characters, returns the count
template <class Dst>
NC_HD int fmt_ld(int64_t v, Dst &out) {
    char tmp[20];
    int n = 0;
    uint64_t u = v < 0 ? (uint64_t) 0 - (uint64_t) v : (uint64_t) v;
    do { tmp[n++] = (char) ('0' + u % 10); u /= 10; } while (u);
    int w = 0;
    if (v < 0) { out.put('-'); w++; }
    while (n > 0) { out.put((uint32_t) (uint8_t) tmp[--n]); w++; }
    return w;
}

// "%f" (precision 6) of a binary64, exactly rounded like glibc's printf: the integer part is
// produced from the exact significand, the fraction from (f * 10^6) / 2^k with round-half-even on
// the exact remainder.  out.put() receives at most `cap` characters (snprintf truncation); the
// return value is the number written.
template <class Dst>
NC_HD_NOINL int fmt_f6(uint64_t bits, Dst &out, int cap) {
    int w = 0;
    auto put = [&](uint32_t c) { if (w < cap) { out.put(c); w++; } };
    if (bits & DBL_SIGN) put('-');
    bits &= ~DBL_SIGN;
    if (bits >= DBL_INF_BITS) {
        const char *t = bits == DBL_INF_BITS ? "inf" : "nan";
        for (int i = 0; i < 3; i++) put((uint32_t) t[i]);
        return w;
    }
    uint64_t M;
    int64_t E;
    bits_to_me(bits, false, M, E);
    if (E >= 0 || M == 0) {
        // integer: digits of M * 2^E
        Big B;
        big_set(B, M);
        if (E > 0) big_shl(B, E);
        // repeated division by 10^9, most significant chunk first => collect chunks
        uint32_t chunks[40];
        int nc = 0;
        while (B.n > 0) {
            uint64_t rem = 0;
            for (int i = B.n - 1; i >= 0; i--) {
                uint64_t cur = (rem << 32) | B.d[i];
                B.d[i] = (uint32_t) (cur / 1000000000u);
                rem = cur % 1000000000u;
            }
            while (B.n > 0 && B.d[B.n - 1] == 0) B.n--;
            chunks[nc++] = (uint32_t) rem;
        }
        if (nc == 0) put('0');
        for (int ci = nc - 1; ci >= 0; ci--) {
            char t[9];
            uint32_t v = chunks[ci];
            for (int k = 8; k >= 0; k--) { t[k] = (char) ('0' + v % 10); v /= 10; }
            int k0 = 0;
            if (ci == nc - 1) while (k0 < 8 && t[k0] == '0') k0++;
            for (int k = k0; k < 9; k++) put((uint32_t) t[k]);
        }
        put('.');
        for (int k = 0; k < 6; k++) put('0');
        return w;
    }
    // E < 0: value = M / 2^k, k = -E in 1 .. 1074
    const int64_t k = -E;
    uint64_t ip = k >= 64 ? 0 : (M >> k);
    uint64_t f = k >= 64 ? M : (M & ((1ull << k) - 1));          // fraction numerator over 2^k
    // q = floor(f * 10^6 / 2^k) with the remainder compared against 1/2
    uint64_t q6;
    int cmp;   // remainder vs half: -1, 0, +1
    {
        uint64_t hi, lo;
        mul64(f, 1000000ull, hi, lo);                             // f < 2^53 => product < 2^73
        if (k >= 128) { q6 = 0; cmp = -1; }                       // product < 2^73 < 2^(k-1)
        else if (k >= 64) {
            int s = (int) (k - 64);                               // divide the 128-bit product by 2^k
            q6 = s == 0 ? hi : (hi >> s);
            uint64_t rhi = s == 0 ? 0 : (hi & ((1ull << s) - 1)), rlo = lo;
            // half = 2^(k-1): bit (k-1) of the 128-bit remainder space
            uint64_t hhi = s == 0 ? 0 : (1ull << (s - 1)), hlo = s == 0 ? (1ull << 63) : 0;
            cmp = rhi != hhi ? (rhi < hhi ? -1 : 1) : (rlo != hlo ? (rlo < hlo ? -1 : 1) : 0);
        }
        else {
            int s = (int) k;                                      // 1 .. 63
            q6 = (hi << (64 - s)) | (lo >> s);
            uint64_t rlo = lo & ((1ull << s) - 1), hlo = 1ull << (s - 1);
            cmp = rlo != hlo ? (rlo < hlo ? -1 : 1) : 0;
        }
    }
    if (cmp > 0 || (cmp == 0 && (q6 & 1))) q6++;
    if (q6 >= 1000000ull) { q6 -= 1000000ull; ip++; }
    {
        char t[20];
        int n = 0;
        uint64_t u = ip;
        do { t[n++] = (char) ('0' + u % 10); u /= 10; } while (u);
        while (n > 0) put((uint32_t) t[--n]);
    }
    put('.');
    {
        char t[6];
        uint32_t v = (uint32_t) q6;
        for (int k2 = 5; k2 >= 0; k2--) { t[k2] = (char) ('0' + v % 10); v /= 10; }
        for (int k2 = 0; k2 < 6; k2++) put((uint32_t) t[k2]);
    }
    return w;
}

// ------------------------------------------------------------------------------------------
// "%.16g" / "%.1f": the float rule of the reference's msgpack -> JSON writer (src/flb_pack.c:1020-1034)
// ------------------------------------------------------------------------------------------
NC_HD_NOINL uint32_t big_divsmall(Big &b, uint32_t d) {
    uint64_t rem = 0;
    for (int i = b.n - 1; i >= 0; i--) {
        uint64_t cur = (rem << 32) | b.d[i];
        b.d[i] = (uint32_t) (cur / d);
        rem = cur % d;
    }
    while (b.n > 0 && b.d[b.n - 1] == 0) b.n--;
    return (uint32_t) rem;
}
// b >>= s; returns whether a discarded bit was set
NC_HD_NOINL bool big_shr(Big &b, int64_t s) {
    if (b.n == 0 || s == 0) return false;
    const int64_t ws = s / 32;
    const int bs = (int) (s % 32);
    bool sticky = false;
    if (ws >= b.n) { sticky = true; b.n = 0; return sticky; }      // b != 0 and everything is shifted out
    for (int i = 0; i < (int) ws; i++) if (b.d[i]) sticky = true;
    if (bs && (b.d[ws] & ((1u << bs) - 1))) sticky = true;
    const int nn = b.n - (int) ws;
    for (int i = 0; i < nn; i++) {
        uint64_t v = (uint64_t) b.d[i + ws] >> bs;
        if (bs && i + ws + 1 < b.n) v |= (uint64_t) b.d[i + ws + 1] << (32 - bs);
        b.d[i] = (uint32_t) v;
    }
    b.n = nn;
    while (b.n > 0 && b.d[b.n - 1] == 0) b.n--;
    return sticky;
}

// floor(M * 2^E * 10^s * 2) for M * 2^E > 0, with `sticky` = the floor discarded something.
// Returns false when the result does not fit 63 bits (the caller's decimal exponent is too small).
NC_HD_NOINL bool scaled_floor2(uint64_t M, int64_t E, int64_t s, uint64_t &q2, bool &sticky) {
    sticky = false;
    const int64_t sh = E + s + 1;
    if (s >= 0 && s <= 27) {
        // M * 5^s in 128 bits (M < 2^53, 5^27 < 2^63)
        uint64_t p5 = 1;
        for (int64_t i = 0; i < s; i++) p5 *= 5;
        uint64_t hi, lo;
        mul64(M, p5, hi, lo);
        if (sh >= 0) {
            const int bl = hi ? 128 - clz64(hi) : (lo ? 64 - clz64(lo) : 0);
            if (bl + sh > 63) return false;
            q2 = lo << sh;
            return true;
        }
        const int64_t rs = -sh;
        if (rs >= 128) { q2 = 0; sticky = (hi | lo) != 0; return true; }
        if (rs >= 64) {
            q2 = rs == 64 ? hi : hi >> (rs - 64);
            sticky = lo != 0 || (rs > 64 && (hi & ((1ull << (rs - 64)) - 1)) != 0);
            return (q2 >> 63) == 0;
        }
        if (hi >> rs) return false;                                  // 0 < rs < 64
        q2 = (hi << (64 - rs)) | (lo >> rs);
        sticky = (lo & ((1ull << rs) - 1)) != 0;
        return (q2 >> 63) == 0;
    }
    Big N;
    big_set(N, M);
    if (s >= 0) {
        big_mul_pow5(N, s);
        if (sh >= 0) big_shl(N, sh); else sticky = big_shr(N, -sh);
    }
    else {
        if (sh > 0) big_shl(N, sh);
        int64_t j = -s;
        while (j >= 13) { if (big_divsmall(N, 1220703125u)) sticky = true; j -= 13; }
        uint32_t m = 1;
        while (j-- > 0) m *= 5;
        if (m > 1 && big_divsmall(N, m)) sticky = true;
        if (sh < 0 && big_shr(N, -sh)) sticky = true;
    }
    if (N.n > 2) return false;
    q2 = (N.n > 0 ? N.d[0] : 0) | ((uint64_t) (N.n > 1 ? N.d[1] : 0) << 32);
    return (q2 >> 63) == 0;
}

// "%.16g" of a binary64, exactly rounded like glibc's printf (round-half-even on the exact value)
template <class Dst>
NC_HD_NOINL int fmt_g16(uint64_t bits, Dst &out) {
    int w = 0;
    if (bits & DBL_SIGN) { out.put('-'); w++; }
    bits &= ~DBL_SIGN;
    if (bits >= DBL_INF_BITS) {
        const char *t = bits == DBL_INF_BITS ? "inf" : "nan";
        for (int i = 0; i < 3; i++) { out.put((uint32_t) t[i]); w++; }
        return w;
    }
    if (bits == 0) { out.put('0'); return w + 1; }
    uint64_t M;
    int64_t E;
    bits_to_me(bits, false, M, E);
    const int64_t e2 = (64 - clz64(M)) + E - 1;                    // floor(log2 v)
    int64_t k = (e2 * 78913) >> 18;                                // ~ floor(e2 * log10 2); corrected below
    uint64_t D = 0;
    for (int guard = 0; guard < 8; guard++) {
        uint64_t q2;
        bool sticky;
        if (!scaled_floor2(M, E, 15 - k, q2, sticky)) { k++; continue; }
        const uint64_t pre = q2 >> 1;
        if (pre >= 10000000000000000ull) { k++; continue; }
        if (pre < 1000000000000000ull) { k--; continue; }
        D = pre;
        if ((q2 & 1) && (sticky || (D & 1))) D++;
        if (D == 10000000000000000ull) { D = 1000000000000000ull; k++; }
        break;
    }
    char dg[16];
    { uint64_t u = D; for (int i = 15; i >= 0; i--) { dg[i] = (char) ('0' + u % 10); u /= 10; } }
    int nd = 16;
    while (nd > 1 && dg[nd - 1] == '0') nd--;
    auto put = [&](uint32_t c) { out.put(c); w++; };
    if (k < -4 || k >= 16) {
        put((uint32_t) dg[0]);
        if (nd > 1) { put('.'); for (int i = 1; i < nd; i++) put((uint32_t) dg[i]); }
        put('e');
        int64_t x = k;
        if (x < 0) { put('-'); x = -x; } else put('+');
        if (x >= 100) { put((uint32_t) ('0' + x / 100)); x %= 100; }
        put((uint32_t) ('0' + x / 10));
        put((uint32_t) ('0' + x % 10));
    }
    else if (k >= 0) {
        for (int i = 0; i <= (int) k; i++) put(i < nd ? (uint32_t) dg[i] : '0');
        if (nd > (int) k + 1) { put('.'); for (int i = (int) k + 1; i < nd; i++) put((uint32_t) dg[i]); }
    }
    else {
        put('0'); put('.');
        for (int64_t i = 0; i < -k - 1; i++) put('0');
        for (int i = 0; i < nd; i++) put((uint32_t) dg[i]);
    }
    return w;
}

// the number a msgpack float becomes in the reference's JSON (src/flb_pack.c:1020-1034):
//   f == (double)(long long) f  ->  "%.1f"      (the conversion as x86-64 performs it: out of range -> LLONG_MIN)
//   NaN with json.convert_nan_to_null -> "null"
//   otherwise "%.16g"
template <class Dst>
NC_HD_NOINL int fmt_json_double(uint64_t bits, bool nan_to_null, Dst &out) {
    const uint64_t mag = bits & ~DBL_SIGN;
    bool integral = false;
    uint64_t ival = 0;
    if (mag == 0) integral = true;
    else if (mag < DBL_INF_BITS) {
        uint64_t M;
        int64_t E;
        bits_to_me(mag, false, M, E);
        if (E >= 0) {
            if (E <= 10 && (M << E) >> E == M && ((M << E) >> 63) == 0) { integral = true; ival = M << E; }
            else if (bits == 0xC3E0000000000000ull) { integral = true; ival = 1ull << 63; }          // -2^63
        }
        else if (-E < 53) {
            if ((M & ((1ull << -E) - 1)) == 0) { integral = true; ival = M >> -E; }
        }
    }
    if (integral) {
        int w = 0;
        if (bits & DBL_SIGN) { out.put('-'); w++; }
        char t[20];
        int n = 0;
        do { t[n++] = (char) ('0' + ival % 10); ival /= 10; } while (ival);
        while (n > 0) { out.put((uint32_t) (uint8_t) t[--n]); w++; }
        out.put('.'); out.put('0');
        return w + 2;
    }
    if (nan_to_null && mag > DBL_INF_BITS) {
        out.put('n'); out.put('u'); out.put('l'); out.put('l');
        return 4;
    }
    return fmt_g16(bits, out);
}

}  // namespace nc
}  // namespace flbgpu
