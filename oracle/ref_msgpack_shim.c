/*
 * oracle/ref_msgpack_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Drives the REAL msgpack-c of the reference (lib/msgpack-c/src/{unpack,objectc,zone,vrefbuffer,version}.c,
 * compiled in place by oracle/Makefile into _ref/libmsgpack_ref.so) the way the log event decoder and
 * filter_parser use it: msgpack_unpack_next over a buffer (src/flb_log_event_decoder.c:296-333) and
 * msgpack_pack_object of what came out (plugins/filter_parser/filter_parser.c:403-409), so that
 * oracle/omp.c -- the restatement -- can be pinned on the real thing.
 *
 * ref_msgpack_roundtrip: every object of `data` is unpacked and re-packed into *out (malloc'd); codes[i]
 * receives the return value of the i-th msgpack_unpack_next call (the last one is the call that stopped
 * the loop), ends[i] the offset after it.  Returns the number of calls recorded.
 */
#include <stdlib.h>
#include <string.h>
#include <msgpack.h>

int ref_msgpack_roundtrip(const char *data, size_t len, char **out, size_t *out_size, int *codes, size_t *ends, int max_calls)
{
    msgpack_unpacked result;
    msgpack_sbuffer sbuf;
    msgpack_packer pck;
    size_t off = 0;
    int n = 0;
    msgpack_unpacked_init(&result);
    msgpack_sbuffer_init(&sbuf);
    msgpack_packer_init(&pck, &sbuf, msgpack_sbuffer_write);
    while (n < max_calls) {
        int r = msgpack_unpack_next(&result, data, len, &off);
        codes[n] = r;
        ends[n] = off;
        n++;
        if (r != MSGPACK_UNPACK_SUCCESS) break;
        msgpack_pack_object(&pck, result.data);
    }
    msgpack_unpacked_destroy(&result);
    *out = malloc(sbuf.size ? sbuf.size : 1);
    memcpy(*out, sbuf.data, sbuf.size);
    *out_size = sbuf.size;
    msgpack_sbuffer_destroy(&sbuf);
    return n;
}
