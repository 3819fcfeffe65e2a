// tzif.hpp -- a parser's Time_Zone (an IANA name) on the device and on the host: the two lookups the reference makes
// over the zone's TZif table when it turns a parsed, zone-less `struct tm` into seconds
// (/root/reference src/flb_parser.c:539-590: tzif_type_at_utc, tzif_tm2time; reached from
// flb_parser_tm2time_parser :685-696 when the parser has a time_zone and its Time_Format carries no zone of its own).
//
// The table is what tzif_parse_data (:359-450) keeps of the file: the transition instants (64-bit block of a version
// 2+ file, the 32-bit block of a version 1 file), the type index of each, each type's offset from UTC, and the
// default type (the first that is not daylight time, 0 when all are).  The POSIX TZ string behind the 64-bit block is
// NOT read by the reference: an instant behind the last transition keeps the last transition's type -- the same here.
//
// Host and device share this text (the host side is the CPU test hook flbgpu_tz_tm2time and nothing else).
#ifndef FLBGPU_TZIF_HPP
#define FLBGPU_TZIF_HPP
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TZ_HD __host__ __device__ inline
#else
#define TZ_HD inline
#endif

namespace flbgpu { namespace tz {

// tzif_type_at_utc: the type in force at `utc` = the type of the last transition that is <= utc
TZ_HD int type_at_utc(const int64_t *trans, const uint8_t *ttype, int timecnt, int default_type, int64_t utc) {
    if (timecnt == 0 || utc < trans[0]) return default_type;
    int lo = 0, hi = timecnt - 1;
    while (lo <= hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (trans[mid] <= utc) lo = mid + 1;
        else hi = mid - 1;
    }
    return ttype[hi];
}

// tzif_tm2time: `local_epoch` = timegm() of the parsed fields.  The first type (in file order) whose offset, taken
// away, lands on an instant where a type of that same offset is in force wins; when none does (a local time inside a
// spring-forward gap) the offset in force at `local_epoch` read as UTC is used.
TZ_HD int64_t tm2time(const int64_t *trans, const uint8_t *ttype, const int32_t *gmtoff, int timecnt, int typecnt, int default_type,
                      int64_t local_epoch) {
    for (int i = 0; i < typecnt; i++) {
        const int64_t cand = local_epoch - (int64_t) gmtoff[i];
        const int ty = type_at_utc(trans, ttype, timecnt, default_type, cand);
        if (ty >= 0 && ty < typecnt && gmtoff[ty] == gmtoff[i]) return cand;
    }
    const int ty = type_at_utc(trans, ttype, timecnt, default_type, local_epoch);
    if (ty < 0 || ty >= typecnt) return -1;
    return local_epoch - (int64_t) gmtoff[ty];
}

} }
#endif
