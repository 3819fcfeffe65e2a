// rx_capi.cpp -- C entry points over the regex table compiler (diagnostics / self-test part of
// the C-ABI, declared in include/flb_gpu.h).  The simulate_* calls execute the compiled TABLES on
// the host so the CPU-only unit tests can check them against the golden vectors; the filters
// never call them.
#include <cstring>
#include <string>
#include "rx.hpp"

extern "C" {

void *flbgpu_rx_compile(const char *pattern, int len, unsigned options, int want_captures, char *err, int errlen)
{
    auto *p = new rx::Program();
    std::string e;
    if (!rx::compile(pattern, (size_t) len, options, want_captures != 0, *p, e)) {
        if (err && errlen > 0) { strncpy(err, e.c_str(), errlen - 1); err[errlen - 1] = 0; }
        delete p;
        return nullptr;
    }
    if (err && errlen > 0) err[0] = 0;
    return p;
}

void flbgpu_rx_free(void *h) { delete (rx::Program *) h; }

int flbgpu_rx_simulate_capture(void *h, const char *s, int len, int *beg, int *end)
{
    return rx::simulate_capture(*(rx::Program *) h, (const uint8_t *) s, len, beg, end);
}

int flbgpu_rx_simulate_match(void *h, const char *s, int len)
{
    return rx::simulate_match(*(rx::Program *) h, (const uint8_t *) s, len);
}

/* info[0..11] = ascii{ncls, nD, nR, nX, NK, list entries}, utf8{ncls, nR, nX, NK, list entries}, ngroups */
void flbgpu_rx_info(void *h, int *info)
{
    auto *p = (rx::Program *) h;
    info[0] = p->ascii.ncls; info[1] = p->ascii.nD; info[2] = p->ascii.nR; info[3] = p->ascii.nX;
    info[4] = p->ascii.NK; info[5] = (int) p->ascii.list_ent.size();
    info[6] = p->utf8.ncls; info[7] = p->utf8.nR; info[8] = p->utf8.nX; info[9] = p->utf8.NK;
    info[10] = (int) p->utf8.list_ent.size(); info[11] = p->ngroups;
}

void flbgpu_rx_debug_stats(long *out3) { rx::debug_stats(out3); }

/* "name=group\n" lines in onig_foreach_name order */
int flbgpu_rx_names(void *h, char *buf, int cap)
{
    auto *p = (rx::Program *) h;
    std::string o;
    for (size_t i = 0; i < p->names.size(); i++)
        for (int g : p->name_groups[i]) o += p->names[i] + "=" + std::to_string(g) + "\n";
    if ((int) o.size() + 1 > cap) return -1;
    memcpy(buf, o.c_str(), o.size() + 1);
    return (int) o.size();
}

}
