"""The host's backtracking matcher (csrc/rxbt.inc) against the real Onigmo (oracle/_ref/libonig_ref.so) on one core: microseconds per
search over 20 000 synthetic access-log lines (256 B), for patterns of the kind that reach the host matcher (look-around, atomic groups,
back-references).  Needs no GPU; needs /root/reference to have been there when oracle/_ref was built.  A small C++ driver is compiled
to /tmp so that no ctypes call sits in the loop.
    python tools/perf_rxbt.py            (FLBGPU_BT_NO_PREFILTER=1: without the start-position analysis)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import msgpack, synth

SRC = r'''
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <dlfcn.h>
extern "C" {
void *flbgpu_rxbt_compile(const char *pattern, int len, unsigned options, char *err, int errlen);
int flbgpu_rxbt_search(void *h, const char *s, int len, int *beg, int *end);
}
typedef int (*onew_t)(const char *, int, unsigned, void **);
typedef int (*osearch_t)(void *, const char *, int, int *, int *, int);
int main(int argc, char **argv) {
    std::vector<std::string> lines;
    FILE *f = fopen(argv[1], "rb");
    static char buf[70000];
    while (fgets(buf, sizeof buf, f)) { size_t n = strlen(buf); if (n && buf[n - 1] == '\n') n--; lines.emplace_back(buf, n); }
    fclose(f);
    void *ol = dlopen(argv[2], RTLD_NOW);
    if (!ol) { printf("no reference engine: %s\n", dlerror()); return 1; }
    onew_t onew = (onew_t) dlsym(ol, "ref_onig_new"); osearch_t osearch = (osearch_t) dlsym(ol, "ref_onig_search");
    const char *pats[] = { "^(?!.*error).*$", "(?<=GET )/\\S+",
      "^(?<host>[^ ]*) [^ ]* (?<user>[^ ]*) \\[(?<time>[^\\]]*)\\] \"(?<method>\\S+)(?: +(?<path>[^ ]*) +\\S*)?\" (?<code>[^ ]*) (?<size>[^ ]*)(?: \"(?<referer>[^\\\"]*)\" \"(?<agent>.*)\")?$(?<!x)",
      "\"(?<m>\\w+) (?=/)", "(\\d+)\\.\\1", "(?>\\d+) (?=\\d)", "(?i)mozilla(?!/4)", "\\b5\\d\\d\\b(?= )", "^x(?=y)", "(?=zzz)q", "\\Anope(?!x)" };
    int beg[64], end[64];
    for (const char *pat : pats) {
        char err[256];
        void *h = flbgpu_rxbt_compile(pat, (int) strlen(pat), 0, err, 256);
        void *reg = nullptr; onew(pat, (int) strlen(pat), 0, &reg);
        if (!h || !reg) { printf("compile failed %s %s\n", pat, err); continue; }
        const int reps = 5; long m1 = 0, m2 = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) for (auto &l : lines) m1 += flbgpu_rxbt_search(h, l.data(), (int) l.size(), beg, end) > 0;
        auto t1 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) for (auto &l : lines) m2 += osearch(reg, l.data(), (int) l.size(), beg, end, 64) > 0;
        auto t2 = std::chrono::steady_clock::now();
        const double a = std::chrono::duration<double>(t1 - t0).count() / (reps * lines.size()) * 1e6, b = std::chrono::duration<double>(t2 - t1).count() / (reps * lines.size()) * 1e6;
        printf("%-44.44s  host matcher %6.2f us   Onigmo %6.2f us   ratio %5.2f   matches %ld / %ld\n", pat, a, b, a / b, m1, m2);
    }
}
'''

def main():
    data, off, ep = synth.apache_records(20000)
    with open("/tmp/perf_rxbt_lines.txt", "wb") as f:
        for i in range(20000):
            f.write(msgpack.unpackb(bytes(data[off[i]:off[i + 1]]), raw=True)[1][b"log"] + b"\n")
    open("/tmp/perf_rxbt.cpp", "w").write(SRC)
    lib = os.path.join(ROOT, "fluent-bit_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-o", "/tmp/perf_rxbt", "/tmp/perf_rxbt.cpp", "-L" + lib, "-lflbgpu", "-ldl", "-Wl,-rpath," + lib])
    subprocess.check_call(["/tmp/perf_rxbt", "/tmp/perf_rxbt_lines.txt", os.path.join(ROOT, "oracle", "_ref", "libonig_ref.so")])

if __name__ == "__main__":
    main()
