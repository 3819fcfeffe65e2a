"""(format, text) pairs for the strptime pins: the formats of conf/parsers.conf and of the reference's parser
tests, every directive the device interpreter implements, and mutated texts around them."""
import random

FORMATS = ["%d/%b/%Y:%H:%M:%S %z", "%Y-%m-%dT%H:%M:%S", "%Y-%m-%d %H:%M:%S", "%b %d %H:%M:%S", "%Y %b %d %H:%M:%S", "%Y-%m-%dT%H:%M:%S %z",
           "%Y-%m-%dT%H:%M:%SZ", "%d/%b/%Y:%H:%M:%S", "%s", "%y%m%d %H%M%S", "%A, %d %B %Y %I:%M:%S %p", "%a %b %e %T %Y", "%D %R", "%F %T",
           "%C%y-%j", "%Y-%m-%d%n%H:%M%t%S", "%Y %U %w", "%Y-%m-%d %H:%M:%S %Z", "%Ey %Om %Od %Y", "%%%Y%%", "%G-%m-%d %Y", "%h %d %Y %k:%M %l",
           "%x %X %Y", "%c", "%r %Y-%m-%d", "%u %V %g %Y", "  %Y  -  %m", "%Y-%m-%d %H:%M:%S +0000", "%H:%M:%S %Y", "%Q %Y"]
TEXTS = ["10/Oct/2000:13:55:36 -0700", "2017-11-01T22:25:21", "2017-11-01 22:25:21", "Feb 28 11:12:13", "2024 Feb 29 23:59:60", "2017-11-01T22:25:21 +0530",
         "2017-11-01T22:25:21Z", "1509575121", "171101 222521", "Wednesday, 01 November 2017 10:25:21 PM", "wed nov  1 22:25:21 2017", "11/01/17 22:25",
         "2017-11-01 22:25:21", "20-17-305", "2017-11-01\n22:25\t21", "2017 44 3", "2017-11-01 22:25:21 UTC", "2017-11-01 22:25:21 EST", "2017-11-01 22:25:21 PDT",
         "2017-11-01 22:25:21 GMT", "2017-11-01 22:25:21 -08", "2017-11-01 22:25:21 +05:30", "2017-11-01 22:25:21 Z", "17 11 01 2017", "%2017%", "2016-11-01 2017",
         "Nov 1 2017 7:05 11", "11/01/17 22:25:21 2017", "Wed Nov  1 22:25:21 2017", "10:25:21 PM 2017-11-01", "3 44 17 2017", "  2017  -  11", "12:00:00 AM 2017-01-01",
         "12:30:00 pm 2017-01-01", "13:00:00 PM 2017-01-01", "", " ", "9999999999999999999", "99999999999999999", "67767976233532800", "-5", "2017-02-30 25:61:61",
         "0000-01-01T00:00:00", "9999-12-31T23:59:59", "69-01-01", "68-12-31", "2017-1-1 1:1:1", "2017-11-01T22:25:21 +9999", "2017-11-01T22:25:21 -0", "Sept 1 2017"]


# %Z (round 4: on the device too): every abbreviation of the reference's table in three spellings, abbreviations with a letter or a digit
# behind them, the exact-case GMT / UTC last resort, names no table holds (the tzname[] resort depends on the process's TZ: UTC here)
ZONE_FORMATS = ["%Y-%m-%d %H:%M:%S %Z", "%Z %Y", "%H:%M %Z|%Y", "%Y%Z"]
ZONES = ["GMT", "UTC", "Z", "UT", "EST", "EDT", "CST", "CDT", "MST", "MDT", "PST", "PDT", "AKST", "AKDT", "HST", "HADT", "AST", "ADT", "NST", "NDT", "WET", "WEST",
         "CET", "CEST", "EET", "EEST", "MSK", "MSD", "ART", "BRT", "BRST", "CLT", "CLST", "AEST", "AEDT", "ACST", "ACDT", "AWST", "NZST", "NZDT", "JST", "KST", "SGT", "IST",
         "GST", "ICT", "WIB", "WITA", "WIT", "MYT", "BDT", "NPT", "WAT", "CAT", "EAT", "SAST", "A", "B", "J", "K", "M", "N", "Y", "XYZ", "GMTx", "UTC5", "GMT+1", "UTCZ",
         "ESTx", "EST5", "WITAX", "WITA.", "", "é", "Zulu", "gm", "ut", "utcx"]


def zone_cases():
    out = []
    for f in ZONE_FORMATS:
        for z in ZONES:
            for sp in (z, z.lower(), z.capitalize()):
                t = f.replace("%Y-%m-%d %H:%M:%S", "2017-11-01 22:25:21").replace("%H:%M", "22:25").replace("%Y", "2017").replace("%Z", sp)
                out.append((f, t))
    return sorted(set(out))


def corpus(seed=20260921, extra=4000):
    rng = random.Random(seed)
    out = [(f, t) for f in FORMATS for t in TEXTS] + zone_cases()
    alpha = "0123456789 :-/+TZaApPmMeEoOcCtTnNvVdDbBuUgGsS.%\t"
    for _ in range(extra):
        f = rng.choice(FORMATS)
        t = list(rng.choice(TEXTS))
        for _ in range(rng.randrange(0, 3)):
            if t and rng.random() < 0.7:
                t[rng.randrange(len(t))] = rng.choice(alpha)
            else:
                t.insert(rng.randrange(len(t) + 1), rng.choice(alpha))
        out.append((f, "".join(t)))
    return out
