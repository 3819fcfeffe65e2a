/* ref_sp_shim.c -- TEST INFRASTRUCTURE.  Drives the REAL stream processor of the reference -- src/stream_processor/flb_sp.c
 * (flb_sp_task_create, sp_process_data_aggr :1435, sp_process_aggregate_data :1280, package_results :1161, flb_sp_window_prune),
 * flb_sp_aggregate_func.c, flb_sp_groupby.c, flb_sp_key.c, flb_sp_window.c, flb_sp_func_*.c, parser/flb_sp_parser.c and
 * lib/rbtree -- compiled from where they lie (oracle/Makefile, _ref/ref_sp: an executable for the reason given at ref_filters).
 *
 * What is NOT the reference's: (1) the two files flex / bison generate from parser/sql.l / sql.y (neither tool is in this image):
 * the grammar is written out by hand below -- same token rules (caseless keywords, INTEGER through atoi into an int, FLOATING
 * through atof into a *float*, ''-escaped strings, identifiers [_A-Za-z][A-Za-z0-9_.]*), same builder calls in the same order
 * (an alias / sub-key list is handed over before flb_sp_cmd_key_add, as the mid-rule reductions do), and bison's default
 * conflict resolution for the precedence-less condition rules: shift, i.e. AND / OR associate to the right and NOT takes
 * everything after it; (2) the engine around a task: the two drivers are the ones tests/internal/stream_processor.c:60-150
 * defines for itself (flb_sp_do_test / flb_sp_fd_event_test), the timer fd is a constant, flb_time_get() is pinned
 * (-Wl,--wrap) so that the packaged records are reproducible.
 *
 * Protocol (stdin -> stdout, binary): ops until EOF
 *   op 1: u32 str_conv, u32 sec, u32 nsec, str sql      -> i32 0 / -1 (task not created)
 *   op 2: u64 len, chunk bytes                         -> i32 ret (window.records), u64 out_len, out (DEFAULT window: packaged now)
 *   op 3: (timer of the window fires)                  -> i32 0, u64 out_len, out
 *   op 4: destroy the task                             -> i32 0
 *   op 5: (the hop timer of a HOPPING window fires)    -> i32 ret of sp_process_hopping_slot (flb_sp_fd_event, the window.fd_hop
 *         branch, src/stream_processor/flb_sp.c:2170-2185) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <ctype.h>
#include <signal.h>
#include <execinfo.h>
#include <unistd.h>
#include <fluent-bit/flb_info.h>
#include <fluent-bit/flb_mem.h>
#include <fluent-bit/flb_sds.h>
#include <fluent-bit/flb_str.h>
#include <fluent-bit/flb_config.h>
#include <fluent-bit/flb_log.h>
#include <fluent-bit/flb_worker.h>
#include <fluent-bit/flb_slist.h>
#include <fluent-bit/flb_time.h>
#include <fluent-bit/stream_processor/flb_sp.h>
#include <fluent-bit/stream_processor/flb_sp_parser.h>
#include <fluent-bit/stream_processor/flb_sp_window.h>

/* ---- the logger: the worker context stays NULL and the print hooks do nothing */
FLB_TLS_DEFINE(struct flb_worker, flb_worker_ctx);
void flb_log_print(int type, const char *file, int line, const char *fmt, ...) { (void) type; (void) file; (void) line; (void) fmt; }
int flb_log_is_truncated(int type, const char *file, int line, const char *fmt, ...) { (void) type; (void) file; (void) line; (void) fmt; return 0; }
int flb_errno_print(int errnum, const char *file, int line) { (void) errnum; (void) file; (void) line; return 0; }
struct flb_worker *flb_worker_get(void) { return NULL; }
int flb_worker_log_level(struct flb_worker *worker) { (void) worker; return 0; }
int flb_log_cache_check_suppress(struct flb_log_cache *cache, char *msg_buf, size_t msg_size) { (void) cache; (void) msg_buf; (void) msg_size; return 0; }

/* ---- the engine's timer: a constant fd, nothing to arm */
int mk_event_timeout_create(struct mk_event_loop *loop, time_t sec, long nsec, void *data) { (void) loop; (void) sec; (void) nsec; (void) data; return 1000; }
int mk_event_timeout_destroy(struct mk_event_loop *loop, void *data) { (void) loop; (void) data; return 0; }
int mk_event_closesocket(int fd) { (void) fd; return 0; }

/* ---- CREATE STREAM: the in_stream_processor instance the result would be appended to is the engine's; the packaged bytes are
 * what this driver returns (task->stream stays NULL) */
int flb_sp_stream_create(const char *name, struct flb_sp_task *task, struct flb_sp *sp) { (void) name; (void) task; (void) sp; return 0; }
void flb_sp_stream_destroy(struct flb_sp_stream *stream, struct flb_sp *sp) { (void) stream; (void) sp; }

/* ---- the clock of package_results */
static struct flb_time g_now;
int __wrap_flb_time_get(struct flb_time *tm) { *tm = g_now; return 0; }
/* NOW() / UNIX_TIMESTAMP() read time(NULL) (flb_sp_func_time.c:54,75): the same given time */
time_t __wrap_time(time_t *t) { if (t) *t = g_now.tm.tv_sec; return g_now.tm.tv_sec; }

/* ---- parser/sql.l + sql.y by hand (see the header of this file) */
#include "grammar_sql.inc"

/* ---- the driver */
static void on_segv(int sig)
{
    void *bt[32];
    int n = backtrace(bt, 32);
    (void) sig;
    backtrace_symbols_fd(bt, n, 2);
    _exit(139);
}
static int rd(void *p, size_t n) { return fread(p, 1, n, stdin) == n; }
static char *rd_str(void)
{
    uint32_t n;
    char *s;
    if (!rd(&n, 4)) return NULL;
    s = malloc(n + 1);
    if (n && !rd(s, n)) return NULL;
    s[n] = 0;
    return s;
}
static void wr_answer(int32_t ret, const void *out, uint64_t n)
{
    fwrite(&ret, 4, 1, stdout); fwrite(&n, 8, 1, stdout);
    if (n) fwrite(out, 1, n, stdout);
    fflush(stdout);
}

/* src/stream_processor/flb_sp.c (not in a header) */
int sp_process_data_aggr(const char *buf_data, size_t buf_size, const char *tag, int tag_len, struct flb_sp_task *task, struct flb_sp *sp,
                         int convert_str_to_num);
void package_results(const char *tag, int tag_len, char **out_buf, size_t *out_size, struct flb_sp_task *task);
int sp_process_hopping_slot(const char *tag, int tag_len, struct flb_sp_task *task);
int sp_process_data(const char *tag, int tag_len, const char *buf_data, size_t buf_size, char **out_buf, size_t *out_size, struct flb_sp_task *task,
                    struct flb_sp *sp);

int main(void)
{
    struct flb_config *config = flb_calloc(1, sizeof(struct flb_config));
    struct flb_sp *sp = flb_calloc(1, sizeof(struct flb_sp));
    struct flb_sp_task *task = NULL;
    uint32_t str_conv = 1;
    signal(SIGSEGV, on_segv);
    mk_list_init(&config->inputs);
    mk_list_init(&config->stream_processor_tasks);
    sp->config = config;
    mk_list_init(&sp->tasks);
    for (;;) {
        uint32_t op;
        if (!rd(&op, 4)) break;
        if (op == 1) {
            uint32_t sec, nsec;
            char *sql;
            if (!rd(&str_conv, 4) || !rd(&sec, 4) || !rd(&nsec, 4)) break;
            sql = rd_str();
            if (!sql) break;
            g_now.tm.tv_sec = sec; g_now.tm.tv_nsec = nsec;
            if (task) { flb_sp_task_destroy(task); task = NULL; }
            task = flb_sp_task_create(sp, "t", sql);
            if (task && task->stream) task->stream = NULL;
            /* 0: an aggregate task, 1: a task of flb_sp_do's other branch (sp_process_data), -1: refused */
            wr_answer(!task ? -1 : task->aggregate_keys == FLB_TRUE ? 0 : 1, NULL, 0);
            free(sql);
        }
        else if (op == 2) {
            uint64_t n;
            char *data, *out = NULL;
            size_t out_size = 0;
            int ret;
            if (!rd(&n, 8)) break;
            data = malloc(n + 1);
            if (n && !rd(data, n)) break;
            if (!task) { wr_answer(-1, NULL, 0); free(data); continue; }
            if (task->aggregate_keys != FLB_TRUE) {
                /* flb_sp.c:2059-2069, the other branch of flb_sp_do: the answer is sp_process_data's return value and buffer */
                ret = sp_process_data("t", 1, data, n, &out, &out_size, task, sp);
                wr_answer(ret, ret > 0 ? out : NULL, ret > 0 ? out_size : 0);
                if (ret > 0 && out) flb_free(out);
                free(data);
                continue;
            }
            /* tests/internal/stream_processor.c:92-150 (flb_sp_do_test), the aggregate branch */
            ret = sp_process_data_aggr(data, n, "t", 1, task, sp, (int) str_conv);
            if (ret != -1 && flb_sp_window_populate(task, data, n) != -1 && task->window.type == FLB_SP_WINDOW_DEFAULT) {
                package_results("t", 1, &out, &out_size, task);
                flb_sp_window_prune(task);
            }
            wr_answer(ret, out, out_size);
            if (out) flb_free(out);
            free(data);
        }
        else if (op == 3) {
            char *out = NULL;
            size_t out_size = 0;
            /* tests/internal/stream_processor.c:60-90 (flb_sp_fd_event_test), the window.fd branch */
            if (task && task->window.records > 0) package_results("t", 1, &out, &out_size, task);
            if (task) flb_sp_window_prune(task);
            wr_answer(0, out, out_size);
            if (out) flb_free(out);
        }
        else if (op == 4) {
            if (task) { flb_sp_task_destroy(task); task = NULL; }
            wr_answer(0, NULL, 0);
        }
        else if (op == 5) {
            wr_answer(task && task->window.type == FLB_SP_WINDOW_HOPPING ? sp_process_hopping_slot("t", 1, task) : -1, NULL, 0);
        }
        else break;
    }
    return 0;
}
