/*
 * flb_gpu.h -- C ABI of libflbgpu.so: the MI355X-native implementation of Fluent Bit's per-record
 * filter hot path (flb_filter_do -> cb_filter of filter_parser / filter_grep / filter_log_to_metrics,
 * flb_parser_do for regex parsers).  Plain pointers and sizes only; every entry point names the reference
 * interface it replaces (paths relative to the fluent-bit source tree).
 *
 * Two call levels are offered for each filter:
 *   - host level   : same contract as the reference callback (borrowed input buffer in host
 *                    memory, malloc()'d output handed to the caller, FLB_FILTER_MODIFIED /
 *                    FLB_FILTER_NOTOUCH return) -- this is what a plugin shim binds;
 *   - device level : chunk bytes + row offsets already resident in HBM, output left in HBM --
 *                    used to chain filters without a PCIe round trip and by bench.py.
 *
 * There is NO CPU fallback: a pattern/option the tables cannot express makes *_create() fail
 * with a message, exactly like the reference fails cb_init on an invalid regex.
 */
#ifndef FLB_GPU_H
#define FLB_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* include/fluent-bit/flb_filter.h:41-42 */
#define FLBGPU_FILTER_MODIFIED 1
#define FLBGPU_FILTER_NOTOUCH  2

typedef struct flbgpu_parser flbgpu_parser;
typedef struct flbgpu_filter flbgpu_filter;

/* A chunk resident in device memory: the reference wire format (concatenated msgpack log events,
 * src/flb_log_event_encoder.c:195-217) as one byte column + the row-offset column (n+1 entries,
 * row_off[n] == bytes). */
typedef struct flbgpu_dev_chunk {
    const void *data;
    const uint64_t *row_off;
    uint64_t n;
    uint64_t bytes;
} flbgpu_dev_chunk;

/* ---- library -------------------------------------------------------------------------------- */
/* Selects the HIP device for the calling process (one process per GPU). 0 on success. */
int flbgpu_init(int device);
/* Last error text of the calling thread ("" if none). */
const char *flbgpu_last_error(void);
/* Number of compute units of the selected device (0 before init). */
int flbgpu_device_cus(void);

/* ---- parsers: replaces flb_parser_create() / flb_parser_do() / flb_parser_destroy() ------------
 * include/fluent-bit/flb_parser.h:99-149, src/flb_parser.c:805-1049, src/flb_parser_regex.c:114-227.
 * Only Format regex is on this path.  Arguments keep the reference meaning; `types` is the
 * "key:type key:type" string of the parsers file (src/flb_parser.c:1130-1182).  Conf-file defaults:
 * skip_empty=1, time_keep=0, time_strict=1 (src/flb_parser.c:1277-1304). */
flbgpu_parser *flbgpu_parser_create(const char *name, const char *regex, int skip_empty,
                                    const char *time_fmt, const char *time_key, const char *time_offset,
                                    int time_keep, int time_strict, const char *types);
/* Format json (src/flb_parser_json.c:28-247): the value is one JSON object; Time_Key (default "time") /
 * Time_Format / Time_Keep as in the parsers file.  Types do not apply; Decode_Field: flbgpu_parser_add_decoder. */
flbgpu_parser *flbgpu_parser_create_json(const char *name, const char *time_fmt, const char *time_key,
                                         const char *time_offset, int time_keep, int time_strict);

/* Format logfmt / Format ltsv parsers for filter_parser: replaces flb_parser_create(name, "logfmt" | "ltsv",
 * NULL, ...) + flb_parser_logfmt_do / flb_parser_ltsv_do (src/flb_parser_logfmt.c:63-325,
 * src/flb_parser_ltsv.c:82-268; quoted logfmt values are decoded like flb_unescape_string_utf8,
 * src/flb_unescape.c:186-277).  logfmt_no_bare_keys = the Logfmt_No_Bare_Keys property
 * (src/flb_parser.c:1319-1324).  `types` = "key:type ..." as for regex parsers: with Types every kept pair goes
 * through flb_parser_typecast on the raw value text (src/flb_parser_logfmt.c:176-182, src/flb_parser_ltsv.c:149-155).
 * Decode_Field: flbgpu_parser_add_decoder. */
flbgpu_parser *flbgpu_parser_create_kv(const char *name, const char *format, const char *time_fmt, const char *time_key,
                                       const char *time_offset, int time_keep, int time_strict, int logfmt_no_bare_keys,
                                       const char *types);
/* One Decode_Field / Decode_Field_As line of the parser's section -- `Decode_Field[_As] <backend> <key> [try_next | do_next]`
 * (conf/parsers.conf:44-58; src/flb_parser_decoder.c:593-776 flb_parser_decoder_list_create builds struct flb_parser.decoders from
 * them): as = 1 for Decode_Field_As, backend "json" | "escaped" | "escaped_utf8" | "mysql_quoted", action NULL / "" | "try_next" |
 * "do_next".  Call in configuration order, before the parser is handed to flbgpu_filter_parser_create; the rules then run inside
 * the filter exactly where flb_parser_decoder_do (:215-550) runs inside flb_parser_do (regex / logfmt / ltsv: on the packed
 * map; Format json: before the time lookup).  0, or -1 with flbgpu_last_error(). */
int flbgpu_parser_add_decoder(flbgpu_parser *p, int as, const char *backend, const char *key, const char *action);
/* The parser section's Time_Zone line -- an IANA name, the `time_zone` argument of flb_parser_create_with_time_zone
 * (src/flb_parser.c:805-1049, :988-1022): a Time_Format without %z / %Z reads its fields as local time of that zone,
 * the seconds come from the zone's TZif table under $TZDIR (default /usr/share/zoneinfo) exactly as tzif_load :452-537
 * reads it and tzif_tm2time :560-590 walks it (flb_parser_tm2time_parser :685-696).  NULL / "": nothing.  Refused like the
 * reference refuses: without a Time_Format, together with Time_Offset or Time_System_Timezone, for a zone whose file is not
 * there.  Call before the parser is handed to a filter.  0, or -1 with flbgpu_last_error(). */
int flbgpu_parser_set_time_zone(flbgpu_parser *p, const char *iana_zone);
/* Time_System_Timezone On (`time_system_timezone`, src/flb_parser.c:986; include/fluent-bit/flb_parser.h:80-94: mktime() of the
 * parsed fields).  Accepted when the process's zone is UTC (mktime is timegm then; the offset a text names is ignored as mktime
 * ignores it), refused for any other process zone.  0, or -1 with flbgpu_last_error(). */
int flbgpu_parser_set_system_timezone(flbgpu_parser *p, int on);
/* Test hook without a device: tzif_tm2time (src/flb_parser.c:560-590) of local_epoch = timegm() of the parsed fields in the
 * named zone -- the routine the kernels run, on the host.  0 with *out set, -1 when the zone's file does not load. */
int flbgpu_tz_tm2time(const char *iana_zone, int64_t local_epoch, int64_t *out);
void flbgpu_parser_destroy(flbgpu_parser *p);
/* flb_parser_do(): one value in host memory -> malloc()'d msgpack map.  Returns the last byte
 * consumed (>= 0: the end of the last named group that took part in the match, src/flb_regex.c:52-54)
 * or -1.  Runs the same kernels on a batch of one. */
int flbgpu_parser_do(flbgpu_parser *p, const char *buf, size_t length, void **out_buf, size_t *out_size,
                     int64_t *out_sec, int64_t *out_nsec);

/* Year-less Time_Formats ("%b %d %H:%M:%S", src/flb_parser.c:922-941) read the current year, month and day
 * from the clock at every lookup (flb_parser_time_lookup with now == 0 -> time(NULL), :1966-1971).  Tests pin
 * that clock: now > 0 replaces time(NULL) for every later run of this process, 0 restores it. */
void flbgpu_set_time_now(int64_t now);

/* ---- filter_parser: replaces cb_parser_init / cb_parser_filter / cb_parser_exit ----------------
 * plugins/filter_parser/filter_parser.c:96-149,174-442.  Properties Key_Name / Parser (repeated) /
 * Reserve_Data / Preserve_Key keep their names and meaning (:460-489). */
flbgpu_filter *flbgpu_filter_parser_create(const char *key_name, int reserve_data, int preserve_key,
                                           int nparsers, flbgpu_parser **parsers);

/* ---- filter_grep: replaces cb_grep_init / cb_grep_filter / cb_grep_exit ------------------------
 * plugins/filter_grep/grep.c:56-164,196-248,286-392.  kinds[i] is the property name ("regex" or
 * "exclude", case-insensitive) and values[i] its value "<key> <pattern>" in configuration order;
 * logical_op is NULL/"legacy"/"AND"/"OR". */
flbgpu_filter *flbgpu_filter_grep_create(int nrules, const char *const *kinds, const char *const *values,
                                         const char *logical_op);

/* ---- filter_log_to_metrics: replaces cb_log_to_metrics_init / cb_log_to_metrics_filter -----------
 * plugins/filter_log_to_metrics/log_to_metrics.c:655-968,970-1156.  (keys[i], values[i]) are the
 * instance's properties in configuration order; the ones read are regex / exclude (set_rules
 * :216-312 -- the field is used verbatim, a name without '$' is a top-level key), label_field /
 * add_label (set_labels :355-497) and bucket (set_buckets :540-595); metric_mode, kubernetes_mode,
 * value_field and discard_logs keep their property names.  metric_name / namespace / subsystem /
 * description, the tag, the emitter and the flush timer only label and ship the cmetrics context:
 * they stay with the plugin shim.
 * flbgpu_filter_run / flbgpu_filter_run_dev on this filter return NOTOUCH, or MODIFIED with an empty
 * output when discard_logs is set (:1141-1145); the side effect is the series state below. */
flbgpu_filter *flbgpu_filter_l2m_create(const char *metric_mode, int nprops, const char *const *keys,
                                        const char *const *values, int kubernetes_mode, const char *value_field,
                                        int discard_logs);
/* mode 0 counter / 1 gauge / 2 histogram (log_to_metrics.h:41-43); row_words = 64-bit words of one
 * series row in flbgpu_l2m_export */
int flbgpu_l2m_info(flbgpu_filter *f, int *mode, int *label_count, int *nbuckets, int *row_words);
const char *flbgpu_l2m_label_key(flbgpu_filter *f, int i);      /* ctx->label_keys[i] */
int flbgpu_l2m_bounds(flbgpu_filter *f, double *bounds);        /* ascending upper bounds; returns nbuckets */
/* Series state in first-appearance order (the order cmt_map keeps its metrics in,
 * lib/cmetrics/src/cmt_map.c:377-452).  rows[n][row_words] are exact integer words that merge by
 * max (words 0,1; word 2 follows word 1) or add (the rest) -- across chunks and across GPUs; keys
 * holds each series' label values as NUL-terminated strings back to back, key_off[n+1] delimits them.
 * Returns n, or -(n) - 2 when max_series / keys_cap are too small (*keys_needed = bytes wanted),
 * -1 on error. */
int64_t flbgpu_l2m_export(flbgpu_filter *f, uint64_t max_series, uint64_t *rows, uint64_t *key_off, char *keys,
                          size_t keys_cap, size_t *keys_needed);
/* One (possibly merged) row -> what cmetrics holds: counter/gauge value, or cumulative buckets
 * [nbuckets + 1] (+Inf last), count and sum (lib/cmetrics/src/cmt_histogram.c:328-361).  Host
 * arithmetic on integers only; the sum is the exact sum of the observations rounded once. */
int flbgpu_l2m_finalize_row(int mode, int nbuckets, const uint64_t *row, double *value, uint64_t *buckets, uint64_t *count,
                            double *sum);
/* sum_order.  A histogram keeps its sum twice: exactly (fixed-point digits that merge by addition; flbgpu_l2m_finalize_row rounds them
 * once) and -- sum_order reference, THE DEFAULT since round 6 -- as the reference builds it: cmt_metric_hist_sum_add,
 * lib/cmetrics/src/cmt_metric_histogram.c:124-137, one binary64 addition per observation in record order, bit for bit (the price: one
 * wave per series walks the call's observations in order).  flbgpu_l2m_set_sum_order(f, 0) before the first record leaves only the exact
 * sum (the reference's own sequential sum is hundreds of ULP from it after 10 M observations); flbgpu_l2m_seq_sums fills
 * one sum per series in flbgpu_l2m_export's order and returns their number (-1: the filter does not keep them).
 * flbgpu_l2m_set_sum_order(f, 2): the same ACROSS RANKS.  The order the bits belong to is "the records of the interval (since the last
 * flush) of rank 0, then of rank 1, ..." -- the index ranges flbgpu_l2m_set_index_base hands out --, i.e. one reference process fed the
 * ranks' shards one after the other; every rank keeps its interval's observations (12 bytes each) and flbgpu_l2m_all_reduce folds them
 * rank after rank, each continuing from the sums the rank in front ended on (one ncclBroadcast of the sums per rank): bit-identical
 * to that process for ANY number of ranks.  flbgpu_l2m_chain_sums: the last flush's sums in the order of its output.  The three steps
 * of the chain are entry points of their own for a merge that runs over another transport (the Python helper over torch.distributed):
 * chain_begin fills `sums` with what the last flush ended on for the union of the label tuples (key_off / keys as flbgpu_l2m_export
 * writes them), seq_replay is this rank's turn (in / out), chain_end keeps the final sums on every rank. */
int flbgpu_l2m_set_sum_order(flbgpu_filter *f, int reference);
int64_t flbgpu_l2m_seq_sums(flbgpu_filter *f, uint64_t max_series, double *sums);
int flbgpu_l2m_chain_begin(flbgpu_filter *f, uint64_t n_keys, const uint64_t *key_off, const char *keys, double *sums);
int flbgpu_l2m_seq_replay(flbgpu_filter *f, uint64_t n_keys, const uint64_t *key_off, const char *keys, double *sums);
int flbgpu_l2m_chain_end(flbgpu_filter *f, uint64_t n_keys, const uint64_t *key_off, const char *keys, const double *sums);
int64_t flbgpu_l2m_chain_sums(flbgpu_filter *f, uint64_t max_series, double *sums);
/* ---- multi-GPU: the collective of the log_to_metrics aggregates (one process per GPU, records sharded) --------
 * Same configuration on every rank; each rank runs the filter on its own shard (flbgpu_l2m_set_index_base gives
 * the ranks disjoint record index ranges).  flbgpu_l2m_all_reduce makes the label dictionaries identical
 * (all-gather of the tuples) and merges the rows over RCCL -- MAX for the two index words, SUM for counts, bucket
 * counts and fixed-point sum digits, on device buffers over xGMI -- and returns the merged state in
 * flbgpu_l2m_export's format, identical on every rank and independent of the rank count.  rccl_comm is an
 * ncclComm_t (the caller's, or one made with the helpers below, which exist so that a C engine needs no RCCL
 * headers: rank 0 calls flbgpu_rccl_unique_id, ships the 128 bytes to the others by its own means, every rank calls
 * flbgpu_rccl_comm_init); stream a hipStream_t or NULL.  librccl is loaded on first use. */
int flbgpu_rccl_unique_id(void *id128);
int flbgpu_rccl_comm_init(void **rccl_comm, int nranks, const void *id128, int rank);
int flbgpu_rccl_comm_destroy(void *rccl_comm);
int64_t flbgpu_l2m_all_reduce(flbgpu_filter *f, void *rccl_comm, void *stream, uint64_t max_series, uint64_t *rows,
                              uint64_t *key_off, char *keys, size_t keys_cap, size_t *keys_needed);
/* Global index of the next record (orders first-appearance / last-writer across shards). */
void flbgpu_l2m_set_index_base(flbgpu_filter *f, uint64_t base);
/* last run: observations, rows sent to the exact-arithmetic kernel, rows with a stale value; total
 * dictionary growths; dictionary slots */
void flbgpu_l2m_stats(flbgpu_filter *f, uint64_t *out5);

void flbgpu_filter_destroy(flbgpu_filter *f);

/* cb_filter (include/fluent-bit/flb_filter.h:57-81): `data` is borrowed host memory.  On
 * FLBGPU_FILTER_MODIFIED *out_buf is malloc()'d (release with free(), i.e. flb_free) and
 * *out_size may be 0 (every record dropped).  On FLBGPU_FILTER_NOTOUCH they are left untouched. */
int flbgpu_filter_run(flbgpu_filter *f, const void *data, size_t bytes, void **out_buf, size_t *out_size);

/* Device-level cb_filter: `in` lives in HBM.  On MODIFIED, *out describes filter-owned device
 * buffers that stay valid until the next call on the same filter (or its destruction).  `stream`
 * is a hipStream_t (NULL = the filter's own stream); the call returns after the stream work it
 * enqueued has completed (sizes are needed on the host to size the output).
 * in->row_off == NULL means "raw chunk bytes": the records are found on the device first
 * (flbgpu_index_dev below), exactly as the decoder loop at the top of every cb_filter would. */
int flbgpu_filter_run_dev(flbgpu_filter *f, const flbgpu_dev_chunk *in, flbgpu_dev_chunk *out, void *stream);

/* ---- flb_filter_do: replaces the filter loop of src/flb_filter.c:121-325 for GPU filters -------------
 * Runs filters[0..n) in order on one chunk: a MODIFIED output becomes the next filter's input
 * (:235-245), an empty MODIFIED output ends the chain (:247-269), NOTOUCH leaves the working chunk
 * alone.  The chunk crosses PCIe once in each direction; intermediate chunks stay in HBM.  Tag
 * routing (flb_router_match, :179-186) stays with the engine: pass the filters that match.
 * Returns MODIFIED (*out_buf malloc()'d, or NULL with *out_size 0 when every record was dropped) or
 * NOTOUCH (*out_buf / *out_size untouched: the engine keeps `data`).  stats[i] (optional, n entries)
 * is what flb_filter_do feeds its per-filter counters with (:213-222,272-312). */
typedef struct flbgpu_chain_stat {
    int ret;                 /* MODIFIED / NOTOUCH; 0 = not reached */
    uint64_t in_records;
    uint64_t out_records;    /* flb_mp_count_log_records of the output (src/flb_filter.c:272) */
    uint64_t out_bytes;
} flbgpu_chain_stat;
int flbgpu_filter_chain_run(flbgpu_filter *const *filters, int nfilters, const void *data, size_t bytes,
                            void **out_buf, size_t *out_size, flbgpu_chain_stat *stats);
int flbgpu_filter_chain_run_dev(flbgpu_filter *const *filters, int nfilters, const flbgpu_dev_chunk *in,
                                flbgpu_dev_chunk *out, flbgpu_chain_stat *stats);

/* Record accounting of the last run (what flb_filter_do derives with flb_mp_count_log_records,
 * src/flb_filter.c:272): records decoded from the input / records in the output. */
void flbgpu_filter_last_counts(flbgpu_filter *f, uint64_t *in_records, uint64_t *out_records);
/* Values (since the filter was created) that met one of the two documented corners in which the reference's regex answer depends on
 * its search optimizer (lib/onigmo/regexec.c:3559 forward_search_range / regenc.c:107 onigenc_get_prev_char_head; case folds that
 * change the byte length): the product answered leftmost-first, the reference MAY have answered otherwise.  Not silent: the plugin
 * shims warn once when this is non-zero (filter_gpu_plugins.c). */
uint64_t flbgpu_filter_regex_corners(flbgpu_filter *f);
/* Which builds a filter_parser instance runs.  The choices between a fast build and the build that takes everything (the single pass or
 * the phase kernels, three or four capture-write ports in the pair tables, the kept records' time looked up at emit or inside the pass, the
 * plain or the general emit build) are made per call from what the last calls showed; a build that did badly is set aside for 16 calls,
 * then tried again (the interval doubles while the tries fail, up to 1024).  out8: [0] what the last call ran (bit 0 single pass, 1
 * three-port tables, 2 emit-side time lookup, 3 plain emit build, 4 the rows walked in the order of their lengths: chunks whose lines
 * differ a lot in length), [1..4] set aside right now: single pass / three-port tables / emit-side
 * lookup / plain build, [5] tries of a build that was set aside, [6] tries that brought it back, [7] device-level calls so far. */
int flbgpu_filter_paths(flbgpu_filter *f, uint64_t *out8);
/* Rules / parsers whose pattern is NOT a regular expression (look-around, atomic groups, possessive repeats, back-references, \Z \G \K; round 5: the absent operator (?~X), subexpression calls \g<..>, more than 31 groups, (?i) over non-ASCII literals and classes, \X)
 * do not fail the create calls: the device does everything but the search of that pattern, which the product's backtracking matcher
 * (csrc/rxbt.inc) runs on the host over the values the device located -- filter_grep rules, filter_log_to_metrics rules (run as a
 * filter_grep in front of the metric kernels) and single-parser filter_parser lists; the fused pair and multiline rules still refuse them.  out4: [0] host rules / parsers of this filter, [1] values searched
 * on the host so far, [2] searches that ended on the backtrack budget ("no match"), [3] records a host parser did not take (duplicate
 * Key_Name entries, values of 64 KB and more: passed on unparsed).  FLBGPU_NO_HOST_RULES=1: refuse such patterns at create as before. */
int flbgpu_filter_host_rules(flbgpu_filter *f, uint64_t *out4);

/* Kernel timing (HIP events recorded on the stream the kernels run on).  When enabled, every
 * run accumulates the duration of each kernel; names[] is a NUL-separated list. */
void flbgpu_filter_profile(flbgpu_filter *f, int enable);
int flbgpu_filter_profile_read(flbgpu_filter *f, int max, const char **names, double *ms, uint64_t *launches);

/* ---- msgpack -> JSON: replaces flb_pack_msgpack_to_json_format ------------------------------------
 * src/flb_pack.c:1320-1600 (what out_stdout / out_http / out_kafka / out_file ... call on every flushed chunk) with
 * msgpack2json :984-1145 and flb_utils_write_str src/flb_utils.c:877-1368 underneath.  Same arguments:
 * json_format FLB_PACK_JSON_FORMAT_JSON 1 / STREAM 2 / LINES 3, date_format FLB_PACK_JSON_DATE_DOUBLE 0 / ISO8601 1 /
 * EPOCH 2 / JAVA_SQL_TIMESTAMP 3 / EPOCH_MS 4 (include/fluent-bit/flb_pack.h:38-60), date_key NULL or date_key_len < 0
 * = no date entry, escape_unicode = json.escape_unicode, convert_nan_to_null = json.convert_nan_to_null
 * (flb_pack_init, src/flb_pack.c:1740).  Returns 0 with *out_buf malloc()'d (NUL terminated, *out_size bytes;
 * flb_sds_t there) or -1 where the reference returns NULL. */
int flbgpu_pack_msgpack_to_json_format(const char *data, uint64_t bytes, int json_format, int date_format, const char *date_key,
                                       int date_key_len, int escape_unicode, int convert_nan_to_null, char **out_buf, size_t *out_size);
/* The same as an object that keeps its device buffers between chunks; release with flbgpu_filter_destroy.
 * _run_dev: the chunk is in HBM (row_off == NULL: raw bytes) and the text stays there: out->data / out->bytes, and
 * out->row_off[0..n] = where each input row's text starts (rows that print nothing are empty). */
flbgpu_filter *flbgpu_jsonfmt_create(int json_format, int date_format, const char *date_key, int date_key_len,
                                     int escape_unicode, int convert_nan_to_null);
int flbgpu_jsonfmt_run(flbgpu_filter *f, const void *data, size_t bytes, char **out_buf, size_t *out_size);
int flbgpu_jsonfmt_run_dev(flbgpu_filter *f, const flbgpu_dev_chunk *in, flbgpu_dev_chunk *out);

/* ---- JSON -> msgpack: replaces flb_pack_json / flb_pack_json_recs --------------------------------
 * src/flb_pack.c:670-688 -> :389-508 (default backend: the yyjson reader with STOP_WHEN_DONE | INSITU |
 * ALLOW_INVALID_UNICODE | REPLACE_INVALID_UNICODE, then yyjson_val_to_msgpack :328-387).  Same arguments
 * and results: 0 / -1, *buffer malloc()'d (NULL with *size 0 for blank input), *root_type one of jsmn's
 * JSMN_OBJECT 1 / ARRAY 2 / STRING 3 / PRIMITIVE 4, *consumed = offset where the value stream stopped. */
int flbgpu_pack_json(const char *js, size_t len, char **buffer, size_t *size, int *root_type, size_t *consumed);
int flbgpu_pack_json_recs(const char *js, size_t len, char **buffer, size_t *size, int *root_type, int *out_records,
                          size_t *consumed);
/* Batched sibling: `text_rows` is a device-resident byte column + row offsets, every row an independent
 * flb_pack_json_recs call (for NDJSON: one line per row).  events == 0: out row i = the msgpack of row
 * i's values.  events != 0: a row that is exactly one JSON object becomes the V2 log event
 * [[ts, {}], object] (src/flb_log_event_encoder.c:195-217), any other row is left empty -- the output is
 * then a chunk for flbgpu_filter_run_dev / flbgpu_filter_chain_run_dev.  Output buffers belong to the
 * packer and stay valid until its next run. */
typedef struct flbgpu_json flbgpu_json;
flbgpu_json *flbgpu_json_create(void);
void flbgpu_json_destroy(flbgpu_json *j);
int flbgpu_json_run_dev(flbgpu_json *j, const flbgpu_dev_chunk *text_rows, int events, uint32_t ts_sec, uint32_t ts_nsec,
                        flbgpu_dev_chunk *out);
/* per-row results of the last run, rows [first, first + count): values parsed, bytes consumed, root type
 * of the first value, status (0 ok / blank, 1 error: flb_pack_json would return -1) */
int flbgpu_json_row_info(flbgpu_json *j, uint64_t first, uint64_t count, uint32_t *records, uint32_t *consumed,
                         uint8_t *root_type, uint8_t *status);
void flbgpu_json_stats(flbgpu_json *j, uint64_t *out3);   /* rows sent to the generic kernels, values, error rows */
/* the last run's first leg (the tile pass: a wave per tile of rows, csrc/jtile_kernels.inc): rows it wrote, rows it left to the
 * row-per-lane kernels, its launches (0: not used, 2: the output outgrew the first estimate), tokens it saw */
void flbgpu_json_tile_stats(flbgpu_json *j, uint64_t *out4);
/* measurement only (tools/perf_json.py): prof != 0 makes the tile pass stamp its phases (shader cycles summed over the waves, read back
 * into phases8 by the next call of this function: staging, bytes, numbers, rows, tokens, look-back, headers, string bodies);
 * no_lookback != 0 lets every workgroup write at a place of its own -- the output is then NOT the packed chunk, only the time is of use */
void flbgpu_json_tile_debug(flbgpu_json *j, int prof, int no_lookback, uint64_t *phases8);
/* ---- in_tail: a file buffer cut into lines, every line one log event ------------------------------------------------
 * plugins/in_tail/tail_file.c:689-1040 (process_content: leading NULs skipped, lines end at '\n', Skip_Empty_Lines, the CR of
 * a CR LF dropped, what follows the last newline stays in the buffer) + :552-604 (flb_tail_file_pack_line: Path_Key, Offset_Key
 * = stream offset of the line, Key = the line; records in the begin_record / append_body_values layout with map32 headers).
 * The plain path: no multiline, parser, docker mode, truncate_long_lines, encoding conversion.  The reference stamps every
 * record with flb_time_get(); the caller passes the timestamp of the call.  *processed = bytes consumed. */
typedef struct flbgpu_tail flbgpu_tail;
flbgpu_tail *flbgpu_tail_create(const char *key, const char *path_key, const char *path, const char *offset_key, int skip_empty_lines);
void flbgpu_tail_destroy(flbgpu_tail *t);
int flbgpu_tail_run(flbgpu_tail *t, const void *text, size_t bytes, uint64_t stream_offset, uint32_t ts_sec, uint32_t ts_nsec,
                    void **out_buf, size_t *out_size, uint64_t *processed, uint64_t *lines);
/* text already in HBM -> a device chunk (one row per newline, skipped lines as empty rows) the filters take as it is */
int flbgpu_tail_run_dev(flbgpu_tail *t, const void *d_text, uint64_t bytes, uint64_t stream_offset, uint32_t ts_sec, uint32_t ts_nsec,
                        flbgpu_dev_chunk *out, uint64_t *processed, uint64_t *lines);

/* ---- multiline in front of the path: in_tail with `multiline.parser` ----------------------------------------------------------
 * plugins/in_tail/tail_file.c:840-898 (the line loop; every line goes to flb_ml_append_text instead of being packed) over
 * src/multiline/flb_ml.c:685-762 (flb_ml_append_text), :197-364 (package_content), src/multiline/flb_ml_rule.c:245-436 (the regex
 * rule state machine), flb_ml_group.c:87-122 (flb_ml_group_cat), flb_ml.c:1590-1790 (flb_ml_flush_stream_group: one record
 * [[ts, {}], {key_content | "log": concatenated lines}] per group).  One multiline parser per context, types regex / endswith /
 * equal; a parser in front (the built-in docker / cri parsers, flbgpu_ml_parser_set_subparser) for endswith / equal types.
 * The buffer limit (flb_ml_group_cat's truncation, the "multiline_truncated" metadata of a cut group) is reproduced.
 *
 * flbgpu_ml_parser  = flb_ml_parser_create (src/multiline/flb_ml_parser.c:46-140) + the instance's key_content + flb_ml_create's
 *                     buffer limit (< 0: the 2 MB default, 0: none); rules: flb_ml_rule_create (flb_ml_rule.c:48-118),
 *                     flbgpu_ml_parser_init = flb_ml_parser_init / flb_ml_rule_init (:279-299);
 *                     flbgpu_ml_parser_builtin: the rule tables of flb_ml_parser_java.c / _go.c / _python.c / _ruby.c, and cri / docker
 *                     (flb_ml_parser_cri.c, flb_ml_parser_docker.c: the parser in front is created with them) (calls init)
 * flbgpu_ml_stream  = flb_ml_stream_create: what one tailed file carries between reads (rule_to_state, the open group, its time)
 * flbgpu_ml_append  = one read: the buffer is cut into lines (leading NULs, Skip_Empty_Lines, CR LF as in flbgpu_tail_run), every
 *                     line runs through the parser; the reference stamps flb_time_get() per line, the caller passes the time of
 *                     the call.  flush != 0: the group still open afterwards leaves too (the flush timer, flb_ml_flush_pending).
 *                     *processed = bytes consumed (the caller keeps what follows the last newline), *records = records produced. */
typedef struct flbgpu_ml_parser flbgpu_ml_parser;
typedef struct flbgpu_ml_stream flbgpu_ml_stream;
flbgpu_ml_parser *flbgpu_ml_parser_create(const char *type, const char *match_string, int negate, const char *key_content, int64_t buffer_limit);
int flbgpu_ml_parser_add_rule(flbgpu_ml_parser *p, const char *from_states, const char *regex, const char *to_state);
int flbgpu_ml_parser_builtin(flbgpu_ml_parser *p, const char *name);
/* `parser` + key_group + key_pattern of a [MULTILINE_PARSER] of type endswith / equal (the shape of the built-in cri / docker parsers):
 * every line is parsed first (ml_append_try_parser_type_text, flb_ml.c:505-532), the buffers live per key_group value, a flush re-packs the
 * first line's map with the concatenation under key_content.  `sub` stays the caller's. */
int flbgpu_ml_parser_set_subparser(flbgpu_ml_parser *p, flbgpu_parser *sub, const char *key_group, const char *key_pattern);
int flbgpu_ml_parser_init(flbgpu_ml_parser *p);
void flbgpu_ml_parser_destroy(flbgpu_ml_parser *p);
/* diagnostics: size of the product of the rules' match-only DFAs (states 0: too large for LDS, the rules are walked one by one) */
void flbgpu_ml_parser_product(const flbgpu_ml_parser *p, uint32_t *states, uint32_t *classes, uint32_t *live);
flbgpu_ml_stream *flbgpu_ml_stream_create(flbgpu_ml_parser *p);
void flbgpu_ml_stream_destroy(flbgpu_ml_stream *s);
void flbgpu_ml_stream_state(const flbgpu_ml_stream *s, int *rule_to_state, uint64_t *buffered);
uint64_t flbgpu_ml_stream_truncations(const flbgpu_ml_stream *s);   /* lines that returned FLB_MULTILINE_TRUNCATED so far */
int flbgpu_ml_append(flbgpu_ml_stream *s, const void *text, size_t bytes, uint32_t ts_sec, uint32_t ts_nsec, int skip_empty_lines, int flush,
                     void **out_buf, size_t *out_size, uint64_t *processed, uint64_t *records);
/* text already in HBM -> a device chunk (one row per group, groups without content as empty rows) the filters take as it is */
int flbgpu_ml_append_dev(flbgpu_ml_stream *s, const void *d_text, uint64_t bytes, uint32_t ts_sec, uint32_t ts_nsec, int skip_empty_lines, int flush,
                         flbgpu_dev_chunk *out, uint64_t *processed, uint64_t *records);
/* A list of parsers on one stream -- in_tail's `multiline.parser docker, cri` (flb_ml_append_text's loop over the instances, src/multiline/
 * flb_ml.c:671-760: the parser that took the stream's last line first, then the others in order; a line nobody takes flushes every parser's
 * groups and leaves alone).  Every parser of the list needs a parser in front (taking a line is then stateless) and the same key_content.
 * A read in which ONE parser takes every line that is taken at all behaves exactly like that parser alone and runs here; a read whose lines
 * split between parsers, or a change of parsers while a group is open, makes the call fail (-1, flbgpu_last_error): the caller keeps it on
 * the CPU.  The streams stay the caller's (one per parser, created from it, used by this list only). */
typedef struct flbgpu_ml_list flbgpu_ml_list;
flbgpu_ml_list *flbgpu_ml_list_create(flbgpu_ml_stream **streams, int n);
void flbgpu_ml_list_destroy(flbgpu_ml_list *l);
int flbgpu_ml_list_lru(const flbgpu_ml_list *l);     /* which parser took the last line (-1: none yet): flb_ml_group.lru_parser */
int flbgpu_ml_list_append(flbgpu_ml_list *l, const void *text, size_t bytes, uint32_t ts_sec, uint32_t ts_nsec, int skip_empty_lines, int flush,
                          void **out_buf, size_t *out_size, uint64_t *processed, uint64_t *records);
int flbgpu_ml_list_append_dev(flbgpu_ml_list *l, const void *d_text, uint64_t bytes, uint32_t ts_sec, uint32_t ts_nsec, int skip_empty_lines, int flush,
                              flbgpu_dev_chunk *out, uint64_t *processed, uint64_t *records);

/* row offsets of an NDJSON buffer (each line with its '\n'); returns the row count or -1 if cap is short */
int64_t flbgpu_split_lines_host(const void *data, size_t bytes, uint64_t *row_off, size_t cap);

/* ---- stream processor: replaces flb_sp_task_create / flb_sp_do / the window timer of flb_sp_fd_event -------------------------
 * src/stream_processor/flb_sp.c:433-560 (task), :2007-2097 -> sp_process_data_aggr :1435-1601 (one appended chunk),
 * :2101-2160 -> package_results :1161-1278 + flb_sp_window_prune (timer).  `sql` is the task's Exec string, parsed with the
 * token rules of parser/sql.l and the grammar of parser/sql.y:
 *   [CREATE STREAM name [WITH (k='v', ...)] AS] SELECT key | COUNT(*) | COUNT|SUM|AVG|MIN|MAX(key) [AS alias], ...
 *   FROM STREAM:name | TAG:'pattern' [WINDOW TUMBLING (n SECOND|MINUTE|HOUR)] [WHERE condition] [GROUP BY key, ...];
 * keys may carry sub-keys (k['a']['b']); conditions: key|@record.time()|@record.contains(key) =,!=,<>,<,<=,>,>= constant,
 * key IS [NOT] NULL, NOT / AND / OR / parentheses with the reference's (precedence-less, right-associative) binding.
 * str_conv = the engine's stream_processor_str_conv (src/flb_config.c:482, default on): numeric strings count as numbers.
 * NULL (flbgpu_last_error says why) for what flb_sp_task_create rejects and for what this path does not take:
 * TIMESERIES_FORECAST, snapshots, a HOPPING window that advances by its size or more.  NOW() / UNIX_TIMESTAMP() / RECORD_TAG() /
 * RECORD_TIME() are select keys of both kinds of task; the `now` of the calls below stands for the reference's time(NULL).
 * A SELECT without aggregation functions -- SELECT key [AS alias] | key['sub'] | *, ... [WHERE condition] -- is flb_sp_do's other
 * branch, sp_process_data (flb_sp.c:1607-1850): every appended chunk answers with the projected records, [record's own time
 * element, {selected pairs}], in *out_buf, *records = the records that passed WHERE; WINDOW / GROUP BY are ignored there as in
 * the reference (flb_sp_info reports window_type 0), and the map-header quirk of :1801-1815 (a fixmap header patched with more
 * than 15 entries) is kept byte for byte.
 * State lives in HBM as order-independent integer words per group (counts, wrapping int64 sums, exact fixed-point sums,
 * min / max): chunks and GPUs can be visited in any order; float SUM / AVG are the exact sum rounded once where the
 * reference adds sequentially (both leave as float32: msgpack_pack_float). */
typedef struct flbgpu_sp flbgpu_sp;
flbgpu_sp *flbgpu_sp_create(const char *sql, int str_conv);
void flbgpu_sp_destroy(flbgpu_sp *t);
/* window_type 0 none (results are packaged per chunk) / 1 tumbling (the caller's timer calls flbgpu_sp_timer every
 * window_sec seconds) / 2 hopping (see flbgpu_sp_hop); source_type 0 STREAM: / 1 TAG:; stream_name NULL unless CREATE STREAM */
int flbgpu_sp_info(const flbgpu_sp *t, int *window_type, int64_t *window_sec, int *source_type, const char **source, const char **stream_name);
const char *flbgpu_sp_stream_prop(const flbgpu_sp *t, const char *key);        /* WITH (tag='...') */
void flbgpu_sp_set_tag(flbgpu_sp *t, const char *tag, size_t len);   /* what RECORD_TAG() packs: the tag flb_sp_do is called with (default "") */
int flbgpu_sp_select_only(const flbgpu_sp *t);      /* 1: no aggregation function (task->aggregate_keys off, flb_sp.c:491): flbgpu_sp_do hands back records */
int flbgpu_sp_key_count(const flbgpu_sp *t);
const char *flbgpu_sp_key_name(const flbgpu_sp *t, int i);                     /* output name: alias, "AVG(k)", "k['a']" */
/* flb_sp_do for one chunk (host memory / device resident).  *records = task->window.records after the chunk.  Without a
 * WINDOW the result records -- one [now, {key: value, ...}] per group in first-seen order -- come back in *out_buf
 * (malloc()'d, release with free(); NULL when no group exists) and the window is pruned; with one they wait for the timer.
 * 0 ok, -1 error: device failure, or input on which the reference's own answer depends on arrival order / tree shape
 * (a GROUP BY column mixing ints, floats and strings inside one window, NaN group keys, NaN under MIN / MAX). */
int flbgpu_sp_do(flbgpu_sp *t, const void *data, size_t bytes, uint32_t now_sec, uint32_t now_nsec, void **out_buf, size_t *out_size,
                 int64_t *records);
int flbgpu_sp_do_dev(flbgpu_sp *t, const flbgpu_dev_chunk *in, void *stream, uint32_t now_sec, uint32_t now_nsec, void **out_buf,
                     size_t *out_size, int64_t *records);
/* a SELECT without aggregation functions whose result stays in HBM: *out = the projected records as a device chunk (one row per incoming
 * row, empty where nothing leaves; valid until the task's next call), what flbgpu_jsonfmt_run_dev / a further filter chain takes as it is */
int flbgpu_sp_select_dev(flbgpu_sp *t, const flbgpu_dev_chunk *in, void *stream, uint32_t now_sec, uint32_t now_nsec, flbgpu_dev_chunk *out,
                         int64_t *records);
/* the window's timer fired: package (if the window saw records) and prune */
int flbgpu_sp_timer(flbgpu_sp *t, uint32_t now_sec, uint32_t now_nsec, void **out_buf, size_t *out_size);
/* WINDOW HOPPING (n unit, ADVANCE BY m unit) -- replaces flb_sp_fd_event's window.fd_hop branch (src/stream_processor/flb_sp.c:2170-2185
 * -> sp_process_hopping_slot :1852-2004) and flb_sp_window_prune's HOPPING branch (flb_sp_window.c:57-104).  The caller arms the
 * two timers the reference arms (flb_sp.c:517-545, :2119-2140): flbgpu_sp_hop every flbgpu_sp_window_advance() seconds, and
 * flbgpu_sp_timer first after window_sec seconds, then every advance seconds.  The hop closes a slot (what the window gained
 * since the previous one); the timer packages the window and drops its oldest slot: counts and SUM / AVG lose the slot's
 * share, MIN / MAX and the int / float type of a sum stay as long as the group's node lives -- as in the reference.
 * -1: a string GROUP BY value (the reference frees the key twice and dies), or an int-typed node that meets a float-typed slot
 * of an earlier life of its group (the reference's answer then depends on arrival order). */
int flbgpu_sp_hop(flbgpu_sp *t);
int64_t flbgpu_sp_window_advance(const flbgpu_sp *t);
/* the SQL front end alone (needs no device): 0 and a canonical text of the plan in desc, -1 when the query is refused */
int flbgpu_sp_parse_check(const char *sql, char *desc, size_t cap);
/* global index of the next record (first-seen order of groups across shards) */
void flbgpu_sp_set_index_base(flbgpu_sp *t, uint64_t base);
/* ---- multi-GPU: one window sharded over ranks (one process per GPU, records sharded by batch; no exchange on the data path).
 * flbgpu_sp_export: this rank's window state -- per group the typed key tuple and the order-independent row words -- as an
 * opaque blob (returns its size; written when cap suffices).  flbgpu_sp_package_merged: package_results over the word-wise
 * merge (max / add) of such blobs, a pure host function.  flbgpu_sp_timer_all_reduce: the timer of a sharded window --
 * all-gather of the blobs over RCCL (rccl_comm: an ncclComm_t, see flbgpu_rccl_comm_init), the same merged records on every
 * rank, window pruned.  The merged result does not depend on the number of ranks or on which rank saw which record. */
int64_t flbgpu_sp_export(flbgpu_sp *t, void *buf, size_t cap);
int flbgpu_sp_package_merged(flbgpu_sp *t, const void *const *snaps, const size_t *sizes, int n, uint32_t now_sec, uint32_t now_nsec,
                             void **out_buf, size_t *out_size);
int flbgpu_sp_timer_all_reduce(flbgpu_sp *t, void *rccl_comm, void *stream, uint32_t now_sec, uint32_t now_nsec, void **out_buf,
                               size_t *out_size);
/* event-timed kernel milliseconds / launches since the last call: [0] k_sp_extract, [1] k_sp_aggregate (+ normalize) */
void flbgpu_sp_profile(flbgpu_sp *t, int enable, double *ms2, uint64_t *launches2);

/* ---- record boundary discovery (flb_log_event_decoder_next loop, src/flb_log_event_decoder.c:342) --
 * Walks concatenated msgpack objects in host memory; fills row_off[0..n] (capacity cap entries).
 * Returns n; *consumed is the offset where decoding stopped (== bytes for a clean chunk). */
int64_t flbgpu_index_host(const void *data, size_t bytes, uint64_t *row_off, size_t cap, size_t *consumed);
/* 1 when the bytes behind `consumed` end exactly on a msgpack field boundary (or there are none): msgpack-c's
 * executor then leaves the decoder offset at the end of the chunk and filter_grep treats the chunk as clean
 * (lib/msgpack-c/include/msgpack/unpack_template.h:242-247,439-447; src/flb_log_event_decoder.c:334-342;
 * plugins/filter_grep/grep.c:357-360); 0 when a field is cut short, at 0xc1, or past 32 open containers. */
int flbgpu_tail_clean_host(const void *data, size_t bytes, size_t consumed);

/* The same boundaries found on the device from the raw bytes of a chunk that is already in HBM
 * (what msgpack_unpack_next yields object by object, lib/msgpack-c/src/unpack.c via
 * src/flb_log_event_decoder.c:296-333): every 0x92 byte is a candidate start, one lane skips the
 * object behind each candidate, and the boundaries are the candidates reachable from byte 0; records
 * that do not start with a 2-element fixarray are walked one by one.  out->row_off (n + 1 entries,
 * row_off[n] == *consumed) belongs to the indexer and stays valid until its next call.  Returns n or
 * -1 (flbgpu_last_error()).  dev_data should be 16-byte aligned (hipMalloc is). */
typedef struct flbgpu_indexer flbgpu_indexer;
flbgpu_indexer *flbgpu_indexer_create(void);
void flbgpu_indexer_destroy(flbgpu_indexer *ix);
int64_t flbgpu_index_dev(flbgpu_indexer *ix, const void *dev_data, size_t bytes, flbgpu_dev_chunk *out, size_t *consumed);
/* last call: candidate bytes, rows found off the candidate chain, chain rounds */
void flbgpu_indexer_stats(const flbgpu_indexer *ix, uint64_t *candidates, uint64_t *off_chain_rows, uint64_t *rounds);

/* ---- device memory helpers for callers that have no HIP binding of their own ------------------ */
void *flbgpu_dev_alloc(size_t bytes);
void flbgpu_dev_free(void *p);
int flbgpu_memcpy_h2d(void *dst, const void *src, size_t bytes);
int flbgpu_memcpy_d2h(void *dst, const void *src, size_t bytes);
int flbgpu_sync(void);

/* ---- diagnostics: regex table compiler ------------------------------------------------------
 * flbgpu_rx_compile mirrors onig_new() as called by src/flb_regex.c:142-145 (options = ONIG_OPTION_*
 * bits i=1,x=2,m=4).  The simulate_* calls execute the compiled tables on the host (self-test of
 * the compiler on machines without a GPU); filters never use them. */
void *flbgpu_rx_compile(const char *pattern, int len, unsigned options, int want_captures, char *err, int errlen);
void flbgpu_rx_free(void *h);
int flbgpu_rx_simulate_capture(void *h, const char *s, int len, int *beg, int *end);
int flbgpu_rx_simulate_match(void *h, const char *s, int len);
void flbgpu_rx_info(void *h, int *info12);
int flbgpu_rx_names(void *h, char *buf, int cap);
/* where the wall time of this thread's last host-level call (flbgpu_filter_run / flbgpu_filter_chain_run) went, microseconds:
 * [0] record boundaries on the host, [1] caller's buffer -> page-locked slab, [2] waiting for the upload (0 for a small chunk: nothing
 * waits there), [3] the device chain (launches, kernels, the wait), [4] output -> caller's buffer, [5] malloc of the output, [6] the call.
 * Returns the number of phases (7); out may hold fewer. */
int flbgpu_host_phases(double *out, int cap);

/* The backtracking matcher behind rules / parsers that are not regular expressions (look-around, atomic groups, possessive repeats,
 * back-references, \Z \G \K; csrc/rxbt.inc -- semantics of lib/onigmo/regexec.c:1431 match_at).  The filters use it by themselves
 * (flbgpu_filter_host_rules); these entry points are for the differential tests against the real engine.
 * search: groups + 1 on a match (beg / end may be NULL), -1 no match, -4 the backtrack budget was spent ("no match" to the callers). */
void *flbgpu_rxbt_compile(const char *pattern, int len, unsigned options, char *err, int errlen);
void flbgpu_rxbt_free(void *h);
int flbgpu_rxbt_search(void *h, const char *s, int len, int *beg, int *end);
int flbgpu_rx_is_nonregular(const char *pattern, int len, unsigned options);
void flbgpu_diag_copy(void *dst, const void *src, size_t n);   /* the threaded slab copy of the host-level calls (unit test) */
uint64_t flbgpu_diag_fused_failures(void);                       /* single passes over a [parser, grep] pair that failed on the device (the chain then answers NOTOUCH) */
int flbgpu_rx_simulate_fx(void *h, const char *s, int len, int *beg, int *end);   /* compact tables of the tile kernel, host execution */
int flbgpu_rx_simulate_fx2(void *h, const char *s, int len, int *beg, int *end);  /* ... with a cell per pair of byte classes (two positions per read) */
int flbgpu_rx_simulate_fx_walk_all(void *h, const char *s, int len, int *beg, int *end);   /* flbgpu_rx_simulate_fx without the tail skip */
int flbgpu_rx_simulate_fx3(void *h, const char *s, int len, int *beg, int *end, int *info2);  /* ... the tables without special entries (8-byte cells, two writes per step) */
int flbgpu_rx_fx_tail(void *h, const char *s, int len, int *nkill, unsigned char *kill4, int *first);   /* the tables' tail (rows, kill bytes; where a text enters it) */
int flbgpu_rx_fx_profile(void *h, const char *s, int len, long *out);              /* table sizes, look-ahead / double-write steps over a text */
void flbgpu_rx_debug_stats(long *out3);   /* forward-walk steps since last call: fast, lookahead, slow */
/* which engine answers: bit 0 = the bit-parallel NFA engine for values with a byte >= 0x80, bit 1 = for every value; info6 = its
 * positions, words per set, character classes, context kinds, list entries, code point intervals; why = the table compiler's reason */
int flbgpu_rx_engine(void *h, int *info6, char *why, int whylen);
/* 1: for this text the reference's own answer depends on its search optimizer (a match start behind stray continuation bytes under
 * ^ \b \B; (?i) and a character whose case fold changes the UTF-8 length): the filters count such values, see
 * flbgpu_filter_regex_corners; flags (may be NULL): which corners the pattern can meet (1 line anchor, 2 word anchor, 4 case folds) */
int flbgpu_rx_corner(void *h, const char *s, int len, int *flags);
/* test aid: a random text drawn from the pattern's own syntax tree (length, cut to cap; -1: the pattern does not parse) */
int flbgpu_rx_sample(const char *pattern, int len, unsigned options, unsigned long long seed, char *out, int cap);

/* ---- diagnostics: text <-> binary64 (csrc/numconv.hpp) ----------------------------------------
 * strtod() (mode 0) / sscanf("%lf") (mode 1) and printf("%f") / ("%ld") as the kernels compute them;
 * the host entry points run the host instantiation of the same header (CPU tests fuzz them against
 * glibc), flbgpu_nc_scan_double_dev runs a batch on the device.  status: 0 no conversion, 1 ok,
 * 2 = needs the exact path (only when exact == 0). */
int flbgpu_nc_scan_double(const char *s, int len, int mode, int exact, double *out, int *consumed);
int flbgpu_nc_fmt_f6(double v, char *buf, int cap);
int flbgpu_nc_fmt_json_double(double v, int nan_to_null, char *buf);   /* src/flb_pack.c:1020-1034: "%.1f" / "null" / "%.16g"; <= 32 chars */
int flbgpu_nc_fmt_ld(long long v, char *buf);
int flbgpu_nc_scan_double_dev(const char *strs, const uint32_t *off, uint32_t n, int mode, uint64_t *bits, int *consumed);

/* ---- diagnostics: counter calibration (csrc/calib.hip, tools/calib_counters.py) ------------------------------
 * Runs one kernel whose HBM byte count is known (mode 0 coalesced 16 B/lane reads, 1 coalesced 4 B/lane reads,
 * 2 one whole 128 B line per lane, 3 16 B of a 128 B line per lane, 4 coalesced 16 B/lane writes) over n bytes of
 * dev_in / dev_out, so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be scaled per request shape. */
int flbgpu_calib_run(int mode, void *dev_in, void *dev_out, uint64_t n, int cus);

#ifdef __cplusplus
}
#endif
#endif
