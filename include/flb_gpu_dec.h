/* flb_gpu_dec.h -- diagnostics entry point of libflbgpu.so for the decoders of a parser (Decode_Field_As <backend> <key>,
 * /root/reference/src/flb_parser_decoder.c:85-147).  The product does not take parsers with decoders yet (round 2): this runs
 * the device-ready string backends (csrc/dec.hpp) on the HOST, for unit tests only.
 *
 * flbgpu_dec_simulate: backend 2 = escaped_utf8 (flb_unescape_string_utf8, src/flb_unescape.c:186-277, as decode_escaped_utf8
 * calls it, src/flb_parser_decoder.c:100-112; 102 = the same followed by logfmt's strlen(), src/flb_parser_logfmt.c:186-196),
 * 1 = escaped (replaces flb_unescape_string, src/flb_unescape.c:278-335, as decode_escaped calls
 * it, src/flb_parser_decoder.c:85-98), 3 = mysql_quoted (decode_mysql_quoted :114-147 over flb_mysql_unquote_string,
 * src/flb_unescape.c:338-388).  Writes at most cap bytes to out (NULL: size only) and returns the decoded length; -1 for any
 * other backend. */
#ifndef FLB_GPU_DEC_H
#define FLB_GPU_DEC_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int64_t flbgpu_dec_simulate(int backend, const void *in, size_t n, void *out, size_t cap);
#ifdef __cplusplus
}
#endif
#endif
