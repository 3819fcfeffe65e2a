// sp.cpp -- the stream processor's aggregate queries behind the C ABI (include/flb_gpu.h, flbgpu_sp_*):
//   SELECT keys / COUNT SUM AVG MIN MAX FROM STREAM:x | TAG:'m' [WINDOW TUMBLING (n unit) | HOPPING (n unit, ADVANCE BY m unit)]
//   [WHERE cond] [GROUP BY keys];
// replaces, for those queries, flb_sp_task_create (src/stream_processor/flb_sp.c:433-560), flb_sp_do's aggregate branch
// (:2007-2097 -> sp_process_data_aggr :1435-1601) and the window timer of flb_sp_fd_event (:2101-2160 -> package_results
// :1161-1278, flb_sp_window_prune flb_sp_window.c:26-104) and, for HOPPING windows, its hop timer (sp_process_hopping_slot :1852-2004).
//
// Host side: the SQL front end (the token rules of parser/sql.l and the grammar of parser/sql.y, restated -- flex / bison
// resolve the precedence-less AND / OR / NOT rules by shifting: right-associative, NOT takes everything after it), the plan
// the kernels interpret, and package_results over the order-independent group rows the kernels maintain (dev.hpp, SpArgs).
// A SELECT without aggregation functions (keys, aliases, `*`) is flb_sp_do's other branch, sp_process_data (:1607-1850): records in,
// projected records out per appended chunk (sp_select.inc); WINDOW / GROUP BY are then never looked at, as in the reference.
// NOW() / UNIX_TIMESTAMP() / RECORD_TAG() / RECORD_TIME() are select keys of both branches (the caller's `now` stands for time(NULL)).
// Queries outside that set (TIMESERIES_FORECAST, snapshots) are refused at create time; inputs on
// which the reference's own result depends on the rb-tree's shape (a GROUP BY column mixing numbers and strings, NaN keys)
// make the call fail instead of answering something else.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <deque>
#include <map>
#include <string>
#include <vector>
#include "../../include/flb_gpu.h"
#include "dev.hpp"
#include "host_int.hpp"
#include "spsel.hpp"
#include "numconv.hpp"

using namespace flbgpu;

namespace {

enum { F_NOP = 0, F_AVG = 1, F_SUM = 2, F_COUNT = 3, F_MIN = 4, F_MAX = 5 };
const char *FUNC_NAME[] = {"", "AVG", "SUM", "COUNT", "MIN", "MAX"};
enum { TF_NONE = 0, TF_NOW = 1, TF_UNIX = 2, TF_TAG = 3, TF_TIME = 4 };
const char *TFUNC_NAME[] = {"", "NOW()", "UNIX_TIMESTAMP()", "RECORD_TAG()", "RECORD_TIME()"};

// ------------------------------------------------------------------------------------------ sql.l
enum { TK_EOF, TK_IDENT, TK_INT, TK_FLOAT, TK_STR, TK_BOOL, TK_KW, TK_CH, TK_OP, TK_BAD };
struct Token { int t = TK_EOF; std::string s; int i = 0; float f = 0; };

const char *KEYWORDS[] = {"CREATE", "FLUSH", "STREAM", "SNAPSHOT", "WITH", "SELECT", "AS", "FROM", "WHERE", "AND", "OR", "NOT", "WINDOW", "LIMIT",
                          "IS", "NULL", "SUM", "AVG", "COUNT", "MIN", "MAX", "TIMESERIES_FORECAST", "CONTAINS", "TIME", "TUMBLING", "HOPPING",
                          "HOUR", "MINUTE", "SECOND", "NOW", "UNIX_TIMESTAMP", "RECORD_TAG", "RECORD_TIME", nullptr};

bool ci_prefix(const char *p, const char *w) {
    for (; *w; p++, w++) if (toupper((unsigned char) *p) != *w) return false;
    return true;
}
bool ident_char(int c) { return isalnum(c) || c == '_' || c == '.'; }

bool tokenize(const char *sql, std::vector<Token> &out, std::string &why) {
    const char *p = sql;
    for (;;) {
        while (*p == ' ' || *p == '\t' || *p == '\n') p++;
        Token t;
        if (!*p) { out.push_back(t); return true; }
        static const char *two[] = {"GROUP BY", "ADVANCE BY", "STREAM:", "TAG:", "@RECORD", nullptr};
        bool hit = false;
        for (int k = 0; two[k]; k++) {
            if (ci_prefix(p, two[k])) { t.t = TK_KW; t.s = two[k]; p += strlen(two[k]); hit = true; break; }
        }
        if (hit) { out.push_back(t); continue; }
        if (*p == '_' || isalpha((unsigned char) *p)) {
            const char *q = p;
            while (ident_char((unsigned char) *q)) q++;
            std::string w(p, q), u = w;
            for (auto &c : u) c = (char) toupper((unsigned char) c);
            p = q;
            for (int k = 0; KEYWORDS[k]; k++) if (u == KEYWORDS[k]) { t.t = TK_KW; t.s = u; hit = true; break; }
            if (!hit) {
                if (u == "TRUE" || u == "FALSE") { t.t = TK_BOOL; t.i = u == "TRUE"; }
                else { t.t = TK_IDENT; t.s = w; }
            }
            out.push_back(t);
            continue;
        }
        if (isdigit((unsigned char) *p) || (*p == '-' && p[1] >= '1' && p[1] <= '9')) {
            const char *q = p;
            if (*q == '-') q++;
            if (*q == '0') q++;
            else while (isdigit((unsigned char) *q)) q++;
            if (*q == '.' && isdigit((unsigned char) q[1])) {
                q++;
                while (isdigit((unsigned char) *q)) q++;
                t.t = TK_FLOAT; t.f = (float) atof(std::string(p, q).c_str());       // yylval->fval is a float
            }
            else {
                std::string w(p, q);
                long long v = strtoll(w.c_str(), nullptr, 10);
                if (v < INT32_MIN || v > INT32_MAX) { why = "integer literal outside int"; return false; }
                t.t = TK_INT; t.i = (int) v;
            }
            p = q;
            out.push_back(t);
            continue;
        }
        if (*p == '\'') {
            const char *q = p + 1;
            for (;;) {
                if (!*q) { why = "unterminated string"; return false; }
                if (*q == '\'') { if (q[1] == '\'') { q += 2; continue; } break; }
                q++;
            }
            for (const char *r = p + 1; r < q; r++) { t.s.push_back(*r); if (*r == '\'') r++; }
            t.t = TK_STR;
            p = q + 1;
            out.push_back(t);
            continue;
        }
        if ((p[0] == '!' && p[1] == '=') || (p[0] == '<' && p[1] == '>')) { t.t = TK_OP; t.s = "!="; p += 2; }
        else if (p[0] == '<' && p[1] == '=') { t.t = TK_OP; t.s = "<="; p += 2; }
        else if (p[0] == '>' && p[1] == '=') { t.t = TK_OP; t.s = ">="; p += 2; }
        else if (p[0] == '<' || p[0] == '>') { t.t = TK_OP; t.s = std::string(1, *p); p++; }
        else if (strchr("*,=()[].;", *p)) { t.t = TK_CH; t.s = std::string(1, *p); p++; }
        else { why = std::string("bad input character '") + *p + "'"; return false; }
        out.push_back(t);
    }
}

// ------------------------------------------------------------------------------------------ sql.y
struct KeyName { std::string name; std::vector<std::string> sub; };
struct SelKey {
    int func = F_NOP;
    bool star = false;
    KeyName k;
    std::string out_name;
    bool has_alias = false;
    std::string alias;
    int tfunc = 0;              // TF_*: NOW() / UNIX_TIMESTAMP() / RECORD_TAG() / RECORD_TIME() (flb_sp_func_time.c, flb_sp_func_record.c)
    int gb = -1;
};
struct Node {                   // condition tree
    enum Kind { LEAF, OP } kind = LEAF;
    SpLeaf leaf = {};           // LEAF (key index filled at compile time)
    KeyName key;                // LEAF of kind SPL_KEY / SPL_CONTAINS
    std::string str;            // LEAF of kind SPL_STR
    int op = 0;                 // OP: SPO_* (comparisons, NOT, AND, OR) or -1 for parentheses
    int l = -1, r = -1;
};
struct Query {
    std::vector<SelKey> keys;
    std::vector<KeyName> gb;
    std::vector<Node> nodes;
    int cond = -1;
    int window = 0;             // 0 default, 1 tumbling, 2 hopping
    int64_t window_sec = 0, advance_sec = 0;
    int source_type = 0;        // 0 stream, 1 tag
    std::string source, stream_name;
    std::vector<std::pair<std::string, std::string>> props;
    bool select_only = false;   // no aggregation function, no GROUP BY: sp_process_data (flb_sp.c:1607-1850), records out per appended chunk
};

struct Parser {
    std::vector<Token> t;
    size_t i = 0;
    std::string why;
    Query q;
    const Token &cur() const { return t[i]; }
    bool is(int kind, const char *v = nullptr) const { return t[i].t == kind && (!v || t[i].s == v); }
    bool eat(int kind, const char *v = nullptr) { if (is(kind, v)) { i++; return true; } return false; }
    bool fail(const std::string &m) { if (why.empty()) why = m; return false; }
    bool need(int kind, const char *v = nullptr) { return eat(kind, v) || fail(std::string("syntax error near token ") + std::to_string(i) + (v ? std::string(" (expected ") + v + ")" : "")); }

    bool subkeys(std::vector<std::string> &out) {
        while (eat(TK_CH, "[")) {
            if (!is(TK_STR)) return fail("sub-key must be a string");
            out.push_back(cur().s); i++;
            if (!need(TK_CH, "]")) return false;
        }
        return true;
    }
    bool alias(std::string &a, bool &has) {
        has = false;
        if (eat(TK_KW, "AS")) {
            if (!is(TK_IDENT)) return fail("alias expected");
            a = cur().s; i++; has = true;
        }
        return true;
    }
    // flb_sp_key_create (parser/flb_sp_parser.c:120-290): the output name
    static std::string out_name(const SelKey &k, bool has_alias, const std::string &al) {
        if (has_alias) return al;
        std::string base = k.star ? "*" : k.k.name;
        for (auto &s : k.k.sub) base += "['" + s + "']";
        if (k.func) return std::string(FUNC_NAME[k.func]) + "(" + base + ")";
        return base;
    }
    bool record_key() {
        SelKey k;
        std::string al;
        bool has = false;
        if (eat(TK_CH, "*")) {
            // flb_sp_key_create (flb_sp_parser.c:172-184): the wildcard only as the first select key
            if (!q.keys.empty()) return fail("wildcard after other select keys");
            k.star = true; k.out_name = "*"; q.keys.push_back(k); return true;
        }
        if (is(TK_IDENT)) {
            k.k.name = cur().s; i++;
            if (!subkeys(k.k.sub) || !alias(al, has)) return false;
        }
        else if (is(TK_KW)) {
            const std::string f = cur().s;
            k.tfunc = f == "NOW" ? TF_NOW : f == "UNIX_TIMESTAMP" ? TF_UNIX : f == "RECORD_TAG" ? TF_TAG : f == "RECORD_TIME" ? TF_TIME : 0;
            if (k.tfunc) {
                // sql.y: time_record_func '(' ')' key_alias
                i++;
                if (!need(TK_CH, "(") || !need(TK_CH, ")") || !alias(al, has)) return false;
                k.out_name = has ? al : TFUNC_NAME[k.tfunc];
                k.has_alias = has; k.alias = al;
                q.keys.push_back(k);
                return true;
            }
            k.func = f == "AVG" ? F_AVG : f == "SUM" ? F_SUM : f == "COUNT" ? F_COUNT : f == "MIN" ? F_MIN : f == "MAX" ? F_MAX : 0;
            if (!k.func) return fail(f + ": not supported in a select key here");
            i++;
            if (!need(TK_CH, "(")) return false;
            if (k.func == F_COUNT && eat(TK_CH, "*")) k.star = true;
            else {
                if (!is(TK_IDENT)) return fail("key expected");
                k.k.name = cur().s; i++;
                if (!subkeys(k.k.sub)) return false;
            }
            if (!need(TK_CH, ")") || !alias(al, has)) return false;
        }
        else return fail("select key expected");
        k.out_name = out_name(k, has, al);
        // flb_sp_key_create :206-222: a key with sub-keys and no alias gets "k['a']['b']" as its alias -- that is what sp_process_data packs
        k.has_alias = has || !k.k.sub.empty(); k.alias = has ? al : k.out_name;
        q.keys.push_back(k);
        return true;
    }
    int add(const Node &n) { q.nodes.push_back(n); return (int) q.nodes.size() - 1; }
    int op_node(int op, int l, int r) { Node n; n.kind = Node::OP; n.op = op; n.l = l; n.r = r; return add(n); }
    int key_leaf(int kind) {
        Node n;
        n.leaf.kind = (uint8_t) kind;
        if (!is(TK_IDENT)) { fail("key expected"); return -1; }
        n.key.name = cur().s; i++;
        if (!subkeys(n.key.sub)) return -1;
        return add(n);
    }
    bool is_value() const { return t[i].t == TK_INT || t[i].t == TK_FLOAT || t[i].t == TK_STR || t[i].t == TK_BOOL; }
    int value_leaf() {
        Node n;
        const Token &c = cur();
        if (c.t == TK_INT) { n.leaf.kind = SPL_INT; n.leaf.v = (uint64_t) (int64_t) c.i; }
        else if (c.t == TK_FLOAT) { n.leaf.kind = SPL_FLOAT; double d = (double) c.f; memcpy(&n.leaf.v, &d, 8); }
        else if (c.t == TK_STR) { n.leaf.kind = SPL_STR; n.str = c.s; }
        else if (c.t == TK_BOOL) { n.leaf.kind = SPL_BOOL; n.leaf.v = c.i ? 1 : 0; }
        else { fail("value expected"); return -1; }
        i++;
        return add(n);
    }
    int const_leaf(int kind, uint64_t v) { Node n; n.leaf.kind = (uint8_t) kind; n.leaf.v = v; return add(n); }
    // comparison | key | value | '(' condition ')'
    int primary() {
        if (eat(TK_CH, "(")) {
            int e = condition();
            if (e < 0 || !need(TK_CH, ")")) return -1;
            return op_node(-1, e, -1);
        }
        if (is_value()) { int v = value_leaf(); return v < 0 ? -1 : op_node(SPO_TRUTH, v, -1); }
        int left;
        bool plain = false;
        if (eat(TK_KW, "@RECORD")) {
            if (!need(TK_CH, ".")) return -1;
            if (eat(TK_KW, "CONTAINS")) {
                if (!need(TK_CH, "(")) return -1;
                left = key_leaf(SPL_CONTAINS);
                if (left < 0 || !need(TK_CH, ")")) return -1;
            }
            else if (eat(TK_KW, "TIME")) {
                if (!need(TK_CH, "(") || !need(TK_CH, ")")) return -1;
                left = const_leaf(SPL_TIME, 0);
            }
            else { fail("record function expected"); return -1; }
        }
        else { left = key_leaf(SPL_KEY); plain = true; }
        if (left < 0) return -1;
        if (plain && eat(TK_KW, "IS")) {
            const bool neg = eat(TK_KW, "NOT");
            if (!need(TK_KW, "NULL")) return -1;
            int c = op_node(SPO_EQ, left, const_leaf(SPL_NULL, 0));
            return neg ? op_node(SPO_NOT, c, -1) : c;
        }
        int op = -2;
        bool neg = false;
        if (is(TK_CH, "=")) op = SPO_EQ;
        else if (is(TK_OP)) {
            const std::string &o = cur().s;
            if (o == "!=") { op = SPO_EQ; neg = true; }
            else op = o == "<" ? SPO_LT : o == "<=" ? SPO_LTE : o == ">" ? SPO_GT : SPO_GTE;
        }
        if (op == -2) {
            // a bare key is "condition: key" (an OR with nothing: its truth value); a bare record function compares with true
            if (plain) return op_node(SPO_TRUTH, left, -1);
            return op_node(SPO_EQ, left, const_leaf(SPL_BOOL, 1));
        }
        i++;
        int v = value_leaf();
        if (v < 0) return -1;
        int c = op_node(op, left, v);
        return neg ? op_node(SPO_NOT, c, -1) : c;
    }
    int condition() {
        if (eat(TK_KW, "NOT")) { int e = condition(); return e < 0 ? -1 : op_node(SPO_NOT, e, -1); }
        int left = primary();
        if (left < 0) return -1;
        if (is(TK_KW, "AND") || is(TK_KW, "OR")) {
            const int op = is(TK_KW, "AND") ? SPO_AND : SPO_OR;
            i++;
            int right = condition();
            return right < 0 ? -1 : op_node(op, left, right);
        }
        return left;
    }
    bool time_unit(int64_t &mult) {
        if (eat(TK_KW, "SECOND")) { mult = 1; return true; }
        if (eat(TK_KW, "MINUTE")) { mult = 60; return true; }
        if (eat(TK_KW, "HOUR")) { mult = 3600; return true; }
        return fail("time unit expected");
    }
    bool select() {
        if (!need(TK_KW, "SELECT")) return false;
        do { if (!record_key()) return false; } while (eat(TK_CH, ","));
        if (!need(TK_KW, "FROM")) return false;
        if (eat(TK_KW, "STREAM:")) {
            if (!is(TK_IDENT)) return fail("stream name expected");
            q.source_type = 0; q.source = cur().s; i++;
        }
        else if (eat(TK_KW, "TAG:")) {
            if (!is(TK_STR)) return fail("tag pattern expected");
            q.source_type = 1; q.source = cur().s; i++;
        }
        else return fail("STREAM: or TAG: expected");
        if (eat(TK_KW, "WINDOW")) {
            // sql.y:269-278: TUMBLING '(' INTEGER time ')' | HOPPING '(' INTEGER time ',' ADVANCE_BY INTEGER time ')'
            const bool hopping = eat(TK_KW, "HOPPING");
            if (!hopping && !need(TK_KW, "TUMBLING")) return false;
            if (!need(TK_CH, "(")) return false;
            if (!is(TK_INT)) return fail("window size expected");
            int64_t n = cur().i, m = 1;
            i++;
            if (!time_unit(m)) return false;
            q.window = 1; q.window_sec = n * m;
            if (hopping) {
                if (!need(TK_CH, ",") || !need(TK_KW, "ADVANCE BY")) return false;
                if (!is(TK_INT)) return fail("ADVANCE BY size expected");
                int64_t an = cur().i, am = 1;
                i++;
                if (!time_unit(am)) return false;
                q.window = 2; q.advance_sec = an * am;
                // flb_sp_cmd_window returns -1 for this (parser/flb_sp_parser.c:552) but the grammar action drops the result and the
                // task would run with a window that never prunes
                if (q.advance_sec >= q.window_sec) return fail("HOPPING window that advances by its size or more");
            }
            if (!need(TK_CH, ")")) return false;
        }
        if (eat(TK_KW, "WHERE")) { q.cond = condition(); if (q.cond < 0) return fail("condition expected"); }
        if (eat(TK_KW, "GROUP BY")) {
            do {
                KeyName k;
                if (!is(TK_IDENT)) return fail("GROUP BY key expected");
                k.name = cur().s; i++;
                if (!subkeys(k.sub)) return false;
                q.gb.push_back(k);
            } while (eat(TK_CH, ","));
        }
        if (eat(TK_KW, "LIMIT")) { if (!is(TK_INT)) return fail("limit expected"); i++; }
        return need(TK_CH, ";");
    }
    bool statement() {
        if (eat(TK_KW, "CREATE")) {
            if (!eat(TK_KW, "STREAM")) return fail("snapshots are not supported");
            if (!is(TK_IDENT)) return fail("stream name expected");
            q.stream_name = cur().s; i++;
            if (eat(TK_KW, "WITH")) {
                if (!need(TK_CH, "(")) return false;
                do {
                    if (!is(TK_IDENT)) return fail("property name expected");
                    std::string k = cur().s; i++;
                    if (!need(TK_CH, "=")) return false;
                    if (!is(TK_STR)) return fail("property value expected");
                    q.props.emplace_back(k, cur().s); i++;
                } while (eat(TK_CH, ","));
                if (!need(TK_CH, ")")) return false;
            }
            if (!need(TK_KW, "AS")) return false;
        }
        if (!select()) return false;
        if (!is(TK_EOF)) return fail("trailing input");
        // sp_cmd_aggregated_keys (flb_sp.c:201-262)
        int aggr = 0;
        for (auto &k : q.keys) if (k.func) aggr++;
        if (!aggr) {
            // sp_cmd_aggregated_keys returns 0: flb_sp_task_create leaves aggregate_keys off and never looks at WINDOW / GROUP BY again
            q.select_only = true; q.window = 0; q.gb.clear();
            return true;
        }
        for (auto &k : q.keys) if (!k.func && k.star) return fail("SELECT * next to aggregation functions");
        aggr = 0;
        for (auto &k : q.keys) {
            if (k.tfunc) continue;                  // neither kind (flb_sp.c:243-245)
            if (k.func) { aggr++; continue; }
            for (size_t g = 0; g < q.gb.size(); g++) {
                if (k.k.name == q.gb[g].name && k.k.sub == q.gb[g].sub) { k.gb = (int) g; break; }
            }
            if (k.gb < 0) return fail("aggregated query cannot include the aggregated keys");
        }
        if (!aggr) return fail("not an aggregate query");
        return true;
    }
};

struct SpMisc { unsigned long long first_bad; unsigned long long counts[2]; unsigned int col_class[SP_MAX_GB]; unsigned int flags; unsigned int pad; };

}  // namespace

struct flbgpu_sp {
    Query q;
    SpPlan plan;
    int str_conv = 1;
    std::vector<int> key_src;           // select key -> aggregated source index (-1: none)
    std::vector<SpSelKey> sel;          // a plain SELECT: its keys in order (sp_select.inc)
    DevBuf d_slen, d_soff, d_sout;
    void *sel_out = nullptr;            // what the last appended chunk left: a malloc()'d buffer the D2H copy filled (finish_do hands it over)
    size_t sel_bytes = 0;
    std::string tag;                    // RECORD_TAG(): the tag of the chunks this task sees (flbgpu_sp_set_tag)
    std::string sel_const;              // the packed constant pairs of the current call (time / record functions)
    DevBuf d_sconst;
    hipStream_t stream = nullptr;
    L2mState tab;                       // group dictionary + rows (the table part of the log_to_metrics state)
    DevBuf d_plan, d_gid, d_val, d_vt, d_misc, d_in, d_off;
    uint64_t idx_base = 0;
    uint64_t records = 0;               // task->window.records
    unsigned int col_class[SP_MAX_GB] = {0, 0, 0, 0};
    // HOPPING (flb_sp.c:1852-2004 sp_process_hopping_slot, flb_sp_window.c:57-104): the device rows hold what a group gained
    // since its node was created; what the pruned slots took away again lives here, and the slots themselves
    struct HopNum { bool f = false; int64_t i64 = 0; double f64 = 0; };       // aggregate_num of a SUM / AVG key: type, i64, f64
    struct HopNode { int64_t records = 0; std::vector<HopNum> src; };
    struct HopSlot { std::map<std::string, HopNode> nodes; int64_t records = 0; };
    struct HopLife { int64_t rm_records = 0; std::vector<int64_t> rm_i; std::vector<double> rm_f; };
    std::deque<HopSlot> hop_slots;                                            // task->window.hopping_slot, oldest first
    std::map<std::string, HopLife> hop_life;
    KernelProf kp[2] = {{"k_sp_extract"}, {"k_sp_aggregate"}};
    bool prof = false;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
};

namespace {

int plan_key(flbgpu_sp *t, const KeyName &k, std::string &why) {
    SpPlan &pl = t->plan;
    for (int i = 0; i < pl.nkeys; i++) {
        const SpKeyRef &r = pl.keys[i];
        if (std::string(pl.blob + r.name_off, r.name_len) != k.name || r.nsub != k.sub.size()) continue;
        bool same = true;
        for (size_t j = 0; j < k.sub.size(); j++) if (std::string(pl.blob + r.sub_off[j], r.sub_len[j]) != k.sub[j]) same = false;
        if (same) return i;
    }
    if (pl.nkeys >= SP_MAX_KEYS) { why = "too many distinct keys in the query"; return -1; }
    if (k.sub.size() > (size_t) SP_MAX_SUB) { why = "too many sub-key levels"; return -1; }
    return -2;
}

struct Blob {
    SpPlan &pl;
    size_t used = 0;
    explicit Blob(SpPlan &p) : pl(p) {}
    bool put(const std::string &s, uint16_t &off, uint16_t &len) {
        if (used + s.size() > (size_t) SP_BLOB) return false;
        memcpy(pl.blob + used, s.data(), s.size());
        off = (uint16_t) used; len = (uint16_t) s.size();
        used += s.size();
        return true;
    }
};

int key_ref(flbgpu_sp *t, Blob &b, const KeyName &k, std::string &why) {
    int r = plan_key(t, k, why);
    if (r != -2) return r;
    SpPlan &pl = t->plan;
    SpKeyRef &ref = pl.keys[pl.nkeys];
    memset(&ref, 0, sizeof(ref));
    if (!b.put(k.name, ref.name_off, ref.name_len)) { why = "query text too long"; return -1; }
    ref.nsub = (uint16_t) k.sub.size();
    for (size_t j = 0; j < k.sub.size(); j++) if (!b.put(k.sub[j], ref.sub_off[j], ref.sub_len[j])) { why = "query text too long"; return -1; }
    return pl.nkeys++;
}

// postfix emission of the condition tree; parentheses are the identity on a boolean
bool emit(flbgpu_sp *t, Blob &b, int ni, std::string &why) {
    SpPlan &pl = t->plan;
    const Node &n = t->q.nodes[ni];
    auto leaf = [&](int li) -> int {
        if (li < 0) return -1;
        const Node &ln = t->q.nodes[li];
        if (pl.nleaf >= SP_MAX_LEAF) { why = "condition too long"; return -1; }
        SpLeaf lf = ln.leaf;
        if (lf.kind == SPL_KEY || lf.kind == SPL_CONTAINS) {
            int k = key_ref(t, b, ln.key, why);
            if (k < 0) return -1;
            lf.key = (uint8_t) k;
        }
        else if (lf.kind == SPL_STR) { if (!b.put(ln.str, lf.str_off, lf.str_len)) { why = "query text too long"; return -1; } }
        pl.leaf[pl.nleaf] = lf;
        return pl.nleaf++;
    };
    auto push = [&](int op, int l, int r) -> bool {
        if (pl.nops >= SP_MAX_OPS) { why = "condition too long"; return false; }
        SpOp o;
        o.op = (uint8_t) op; o.l = (uint8_t) (l < 0 ? 0 : l); o.r = (uint8_t) (r < 0 ? 0 : r); o.pad = 0;
        pl.ops[pl.nops++] = o;
        return true;
    };
    if (n.kind != Node::OP) { why = "internal: leaf outside a comparison"; return false; }
    if (n.op == -1) return emit(t, b, n.l, why);
    if (n.op <= SPO_GTE) {
        int l = leaf(n.l), r = leaf(n.r);
        return l >= 0 && r >= 0 && push(n.op, l, r);
    }
    if (n.op == SPO_TRUTH) { int l = leaf(n.l); return l >= 0 && push(SPO_TRUTH, l, -1); }
    if (n.op == SPO_NOT) return emit(t, b, n.l, why) && push(SPO_NOT, -1, -1);
    return emit(t, b, n.l, why) && emit(t, b, n.r, why) && push(n.op, -1, -1);
}

bool compile(flbgpu_sp *t, std::string &why) {
    SpPlan &pl = t->plan;
    memset(&pl, 0, sizeof(pl));
    pl.str_conv = t->str_conv;
    Blob b(pl);
    if (t->q.gb.size() > (size_t) SP_MAX_GB) { why = "too many GROUP BY keys"; return false; }
    for (auto &g : t->q.gb) {
        int k = key_ref(t, b, g, why);
        if (k < 0) return false;
        pl.gb_key[pl.ngb++] = (uint8_t) k;
    }
    t->key_src.assign(t->q.keys.size(), -1);
    t->sel.clear();
    if (t->q.select_only) {
        for (const SelKey &k : t->q.keys) {
            SpSelKey sk;
            memset(&sk, 0, sizeof(sk));
            if (k.star) sk.star = 1;
            else if (k.tfunc) sk.star = k.tfunc == TF_TIME ? 3 : 2;      // constant bytes per call (run_select fills alias_off / alias_len)
            else {
                const int kr = key_ref(t, b, k.k, why);
                if (kr < 0) return false;
                sk.key = (uint8_t) kr;
                if (k.has_alias) { sk.has_alias = 1; if (!b.put(k.alias, sk.alias_off, sk.alias_len)) { why = "query text too long"; return false; } }
            }
            if (t->sel.size() >= (size_t) SP_MAX_KEYS) { why = "too many select keys"; return false; }
            t->sel.push_back(sk);
        }
    }
    for (size_t i = 0; i < t->q.keys.size(); i++) {
        const SelKey &k = t->q.keys[i];
        if (!k.func || k.star || k.tfunc) continue;
        int kr = key_ref(t, b, k.k, why);
        if (kr < 0) return false;
        int src = -1;
        for (int s = 0; s < pl.nsrc; s++) if (pl.src_key[s] == kr) src = s;
        if (src < 0) {
            if (pl.nsrc >= SP_MAX_SRC) { why = "too many aggregated keys"; return false; }
            src = pl.nsrc;
            pl.src_key[pl.nsrc++] = (uint8_t) kr;
        }
        t->key_src[i] = src;
    }
    if (t->q.cond >= 0) {
        if (!emit(t, b, t->q.cond, why)) return false;
        // the stack is a 32-bit mask; right-nested conditions need one level per pending operator
        int depth = 0, maxd = 0;
        for (int i = 0; i < pl.nops; i++) {
            const int op = pl.ops[i].op;
            if (op <= SPO_TRUTH) depth++;
            else if (op != SPO_NOT) depth--;
            maxd = std::max(maxd, depth);
        }
        if (maxd > 30) { why = "condition nested too deeply"; return false; }
    }
    return true;
}

struct Profile {
    flbgpu_sp *t;
    int k;
    hipStream_t st;
    Profile(flbgpu_sp *t_, int k_, hipStream_t st_) : t(t_), k(k_), st(st_) { if (t->prof) (void) hipEventRecord(t->ev[0], st); }
    ~Profile() {
        if (!t->prof) return;
        (void) hipEventRecord(t->ev[1], st);
        (void) hipEventSynchronize(t->ev[1]);
        float ms = 0;
        if (hipEventElapsedTime(&ms, t->ev[0], t->ev[1]) == hipSuccess) { t->kp[k].ms += ms; t->kp[k].launches++; }
    }
};

void reset_window(flbgpu_sp *t) {
    L2mState &s = t->tab;
    (void) hipMemset(s.d_slot_hash.p, 0, s.cap * 8);
    (void) hipMemset(s.d_rows.p, 0, (size_t) s.max_series * s.W * 8);
    L2mCtr c;
    memset(&c, 0, sizeof(c));
    if (t->plan.ngb == 0) c.n_series = 1;
    (void) hipMemcpy(s.d_ctr.p, &c, sizeof(c), hipMemcpyHostToDevice);
    t->records = 0;
    memset(t->col_class, 0, sizeof(t->col_class));
}

// msgpack writers (msgpack-c's pack templates: shortest encodings)
void pk_str(std::string &o, const char *s, size_t n) {
    if (n < 32) o.push_back((char) (0xa0 | n));
    else if (n < 256) { o.push_back((char) 0xd9); o.push_back((char) n); }
    else if (n < 65536) { o.push_back((char) 0xda); o.push_back((char) (n >> 8)); o.push_back((char) n); }
    else { o.push_back((char) 0xdb); for (int i = 3; i >= 0; i--) o.push_back((char) (n >> (8 * i))); }
    o.append(s, n);
}
void pk_be(std::string &o, uint64_t v, int bytes) { for (int i = bytes - 1; i >= 0; i--) o.push_back((char) (v >> (8 * i))); }
void pk_int64(std::string &o, int64_t v) {
    if (v >= 0) {
        if (v < 128) o.push_back((char) v);
        else if (v < 256) { o.push_back((char) 0xcc); pk_be(o, (uint64_t) v, 1); }
        else if (v < 65536) { o.push_back((char) 0xcd); pk_be(o, (uint64_t) v, 2); }
        else if (v < (1ll << 32)) { o.push_back((char) 0xce); pk_be(o, (uint64_t) v, 4); }
        else { o.push_back((char) 0xcf); pk_be(o, (uint64_t) v, 8); }
    }
    else if (v >= -32) o.push_back((char) v);
    else if (v >= -128) { o.push_back((char) 0xd0); pk_be(o, (uint64_t) v, 1); }
    else if (v >= -32768) { o.push_back((char) 0xd1); pk_be(o, (uint64_t) v, 2); }
    else if (v >= -(1ll << 31)) { o.push_back((char) 0xd2); pk_be(o, (uint64_t) v, 4); }
    else { o.push_back((char) 0xd3); pk_be(o, (uint64_t) v, 8); }
}
void pk_uint64(std::string &o, uint64_t v) {
    if (v < 128) o.push_back((char) v);
    else if (v < 256) { o.push_back((char) 0xcc); pk_be(o, v, 1); }
    else if (v < 65536) { o.push_back((char) 0xcd); pk_be(o, v, 2); }
    else if (v < (1ull << 32)) { o.push_back((char) 0xce); pk_be(o, v, 4); }
    else { o.push_back((char) 0xcf); pk_be(o, v, 8); }
}
// the value half of flb_sp_func_time / flb_sp_func_record (flb_sp_func_time.c:39-83, flb_sp_func_record.c:39-63); `now` stands for
// time(NULL) (NOW, UNIX_TIMESTAMP) and, where the reference hands over the package time, for that (RECORD_TIME of an aggregate)
void pk_tfunc_value(std::string &o, int tfunc, const std::string &tag, uint32_t now_sec, uint32_t now_nsec) {
    if (tfunc == TF_NOW) {
        time_t now = (time_t) now_sec;
        struct tm local;
        char buf[32];
        localtime_r(&now, &local);
        const size_t len = strftime(buf, sizeof(buf) - 1, "%Y-%m-%d %H:%M:%S", &local);
        pk_str(o, buf, len);
    }
    else if (tfunc == TF_UNIX) pk_uint64(o, now_sec);
    else if (tfunc == TF_TAG) pk_str(o, tag.data(), tag.size());
    else {
        const double d = (double) now_sec + (double) now_nsec / 1000000000.0;      // flb_time_to_double
        uint64_t b;
        memcpy(&b, &d, 8);
        o.push_back((char) 0xcb); pk_be(o, b, 8);
    }
}
void pk_float(std::string &o, double d) {              // msgpack_pack_float: the value leaves as binary32
    float f = (float) d;
    uint32_t b;
    memcpy(&b, &f, 4);
    o.push_back((char) 0xca);
    pk_be(o, b, 4);
}
double bits_double(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
int64_t unord_i(uint64_t o) { return (int64_t) (o ^ 0x8000000000000000ull); }
uint64_t unord_f(uint64_t o) { return (o >> 63) ? (o & 0x7FFFFFFFFFFFFFFFull) : ~o; }

// ---- the window's state on the host: live groups (typed key tuple as stored in the arena + the row words), the record count
// and the class mask of every GROUP BY column.  Snapshots of shards merge word by word (max / add), which is what the
// multi-GPU exchange ships.
struct Group { std::string key; std::vector<uint64_t> row; uint32_t idx = 0; };
struct Snapshot {
    uint64_t records = 0;
    unsigned int col_class[SP_MAX_GB] = {0, 0, 0, 0};
    std::vector<Group> groups;
};

bool snapshot(flbgpu_sp *t, Snapshot &sn) {
    L2mState &s = t->tab;
    const SpPlan &pl = t->plan;
    L2mCtr c;
    HIPOK(hipMemcpy(&c, s.d_ctr.p, sizeof(c), hipMemcpyDeviceToHost));
    const uint32_t ng = c.n_series;
    const size_t W = (size_t) s.W;
    std::vector<uint64_t> rows((size_t) ng * W);
    std::vector<unsigned long long> koff(ng);
    std::vector<uint32_t> klen(ng);
    std::vector<uint8_t> arena(c.arena_used);
    if (ng) {
        HIPOK(hipMemcpy(rows.data(), s.d_rows.p, rows.size() * 8, hipMemcpyDeviceToHost));
        if (pl.ngb) {
            HIPOK(hipMemcpy(koff.data(), s.d_key_off.p, (size_t) ng * 8, hipMemcpyDeviceToHost));
            HIPOK(hipMemcpy(klen.data(), s.d_key_len.p, (size_t) ng * 4, hipMemcpyDeviceToHost));
            if (!arena.empty()) HIPOK(hipMemcpy(arena.data(), s.d_arena.p, arena.size(), hipMemcpyDeviceToHost));
        }
    }
    sn.records = t->records;
    memcpy(sn.col_class, t->col_class, sizeof(sn.col_class));
    sn.groups.clear();
    for (uint32_t g = 0; g < ng; g++) {
        if (rows[g * W] == 0) continue;                 // a dictionary entry no counted record stands behind
        Group gr;
        if (pl.ngb) gr.key.assign((const char *) arena.data() + koff[g], klen[g]);
        gr.row.assign(rows.begin() + g * W, rows.begin() + (g + 1) * W);
        gr.idx = g;
        sn.groups.push_back(std::move(gr));
    }
    return true;
}

// word-wise merge: max for the first 1 + 4 nsrc words, add for the rest
void merge_into(Snapshot &dst, const Snapshot &src, int nsrc) {
    const size_t nmax = 1 + (size_t) SP_SRC_MAX * nsrc;
    dst.records += src.records;
    for (int g = 0; g < SP_MAX_GB; g++) dst.col_class[g] |= src.col_class[g];
    for (const Group &g : src.groups) {
        Group *hit = nullptr;
        for (Group &d : dst.groups) if (d.key == g.key) { hit = &d; break; }
        if (!hit) { dst.groups.push_back(g); continue; }
        for (size_t w = 0; w < g.row.size(); w++) {
            if (w < nmax) hit->row[w] = std::max(hit->row[w], g.row[w]);
            else hit->row[w] += g.row[w];
        }
    }
}

void put_u64(std::string &o, uint64_t v) { o.append((const char *) &v, 8); }
std::string serialize(const Snapshot &sn, int W) {
    std::string o;
    put_u64(o, 0x5350534e41503031ull);                  // "SPSNAP01"
    put_u64(o, sn.records);
    for (int g = 0; g < SP_MAX_GB; g++) put_u64(o, sn.col_class[g]);
    put_u64(o, sn.groups.size());
    put_u64(o, (uint64_t) W);
    for (const Group &g : sn.groups) {
        put_u64(o, g.key.size());
        o.append(g.key);
        o.append((8 - g.key.size() % 8) % 8, '\0');
        o.append((const char *) g.row.data(), g.row.size() * 8);
    }
    return o;
}
bool deserialize(const uint8_t *p, size_t n, int W, Snapshot &sn) {
    auto get = [&](uint64_t &v) -> bool { if (n < 8) return false; memcpy(&v, p, 8); p += 8; n -= 8; return true; };
    uint64_t magic, ng, w, cc;
    if (!get(magic) || magic != 0x5350534e41503031ull || !get(sn.records)) return false;
    for (int g = 0; g < SP_MAX_GB; g++) { if (!get(cc)) return false; sn.col_class[g] = (unsigned int) cc; }
    if (!get(ng) || !get(w) || w != (uint64_t) W) return false;
    if (ng > n / 8) return false;                             // (every group takes at least its length word: a corrupt count ends here)
    sn.groups.clear();
    for (uint64_t i = 0; i < ng; i++) {
        uint64_t kl;
        if (!get(kl)) return false;
        if (kl > n) return false;                             // (before the rounding below can wrap)
        const size_t padded = (size_t) ((kl + 7) & ~7ull);
        if (n < padded + (size_t) W * 8) return false;
        Group g;
        g.key.assign((const char *) p, (size_t) kl);
        p += padded; n -= padded;
        g.row.resize((size_t) W);
        memcpy(g.row.data(), p, (size_t) W * 8);
        p += (size_t) W * 8; n -= (size_t) W * 8;
        sn.groups.push_back(std::move(g));
    }
    return true;
}

// package_results (flb_sp.c:1161-1278) over a window's groups
bool package(flbgpu_sp *t, Snapshot &sn, uint32_t now_sec, uint32_t now_nsec, std::string &out) {
    const SpPlan &pl = t->plan;
    for (int g = 0; g < pl.ngb; g++) {
        const unsigned int m = sn.col_class[g];
        if (m & (m - 1)) { set_err("stream processor: GROUP BY column %d mixes value classes across the shards of one window", g); return false; }
    }
    std::stable_sort(sn.groups.begin(), sn.groups.end(), [](const Group &a, const Group &b) { return ~a.row[0] < ~b.row[0]; });
    const size_t A = 1 + (size_t) SP_SRC_MAX * pl.nsrc;
    for (const Group &grp : sn.groups) {
        const uint64_t *row = grp.row.data();
        // the group's typed key values
        int gcls[SP_MAX_GB] = {0, 0, 0, 0};
        uint64_t gu[SP_MAX_GB] = {0, 0, 0, 0};
        const char *gs[SP_MAX_GB] = {nullptr, nullptr, nullptr, nullptr};
        size_t gl[SP_MAX_GB] = {0, 0, 0, 0};
        if (pl.ngb) {
            const uint8_t *p = (const uint8_t *) grp.key.data(), *e = p + grp.key.size();
            for (int k = 0; k < pl.ngb && p < e; k++) {
                gcls[k] = *p++;
                if (gcls[k] == 's') { gs[k] = (const char *) p; gl[k] = strnlen((const char *) p, (size_t) (e - p)); p += gl[k] + 1; }
                else { memcpy(&gu[k], p, 8); p += 8; }
            }
        }
        // HOPPING: minus what the pruned slots took from this node (aggregate_func_remove_sum; MIN / MAX / the type stay)
        const flbgpu_sp::HopLife *lf = nullptr;
        if (t->q.window == 2) { auto it = t->hop_life.find(grp.key); if (it != t->hop_life.end()) lf = &it->second; }
        const uint64_t records = row[A] - (lf ? (uint64_t) lf->rm_records : 0);
        out.push_back((char) 0x92);
        out.push_back((char) 0xd7); out.push_back((char) 0x00);
        pk_be(out, now_sec, 4); pk_be(out, now_nsec, 4);
        const size_t nk = t->q.keys.size();
        if (nk < 16) out.push_back((char) (0x80 | nk));
        else { out.push_back((char) 0xde); pk_be(out, nk, 2); }
        for (size_t ki = 0; ki < nk; ki++) {
            const SelKey &k = t->q.keys[ki];
            pk_str(out, k.out_name.data(), k.out_name.size());
            if (k.tfunc) { pk_tfunc_value(out, k.tfunc, t->tag, now_sec, now_nsec); continue; }   // flb_sp.c:1199-1206 (RECORD_TIME: the package time)
            if (k.func == F_NOP) {
                const int gi = k.gb;
                if (gcls[gi] == 'i') pk_int64(out, (int64_t) gu[gi]);
                else if (gcls[gi] == 'f') pk_float(out, bits_double(gu[gi]));
                else pk_str(out, gs[gi], gl[gi]);
                continue;
            }
            if (k.func == F_COUNT) { pk_int64(out, (int64_t) records); continue; }
            const int src = t->key_src[ki];
            const uint64_t *mx = row + 1 + (size_t) SP_SRC_MAX * src, *ad = row + A + 1 + (size_t) SP_SRC_ADD * src;
            const uint64_t n_int = ad[SP_A_NINT], n_flt = ad[SP_A_NFLT];
            const bool is_f64 = n_flt > 0;          // a non-zero float turned nums[key].type into FLB_SP_NUM_F64
            if (k.func == F_SUM || k.func == F_AVG) {
                double dsum = 0;
                int64_t isum = (int64_t) ad[SP_A_ISUM];
                if (is_f64) dsum = bits_double(l2m_limbs_bits(ad + SP_A_LIMB, ad[SP_A_NAN], ad[SP_A_PINF], ad[SP_A_NINF]));
                if (lf) {
                    if (is_f64) dsum = dsum - (double) lf->rm_i[src] - lf->rm_f[src];
                    else isum = (int64_t) ((uint64_t) isum - (uint64_t) lf->rm_i[src]);
                }
                if (k.func == F_SUM) { if (is_f64) pk_float(out, dsum); else pk_int64(out, isum); }
                else pk_float(out, (is_f64 ? dsum : (double) isum) / (double) (int64_t) records);
                continue;
            }
            // MIN / MAX (flb_sp_aggregate_func.c:73-157): NaN never replaces a value and is never replaced -- order-dependent
            if (is_f64 && ad[SP_A_NAN] > 0) { set_err("stream processor: NaN under MIN / MAX (the reference's answer depends on arrival order)"); return false; }
            const bool want_min = k.func == F_MIN;
            if (!is_f64) pk_int64(out, n_int ? unord_i(want_min ? ~mx[0] : mx[1]) : 0);
            else {
                const bool has_f = n_flt - ad[SP_A_NAN] > 0;
                double v = 0;
                bool have = false;
                if (n_int) { v = (double) unord_i(want_min ? ~mx[0] : mx[1]); have = true; }
                if (has_f) {
                    const double f = bits_double(unord_f(want_min ? ~mx[2] : mx[3]));
                    if (!have || (want_min ? f < v : f > v)) v = f;
                }
                pk_float(out, v);
            }
        }
    }
    return true;
}

// a plain SELECT over one chunk: size pass, scan, emit; what leaves is copied straight into the buffer finish_do hands over
bool run_select(flbgpu_sp *t, const flbgpu_dev_chunk *in, hipStream_t st, uint32_t now_sec, uint32_t now_nsec, flbgpu_dev_chunk *dev_out = nullptr) {
    uint64_t n = in->n;
    free(t->sel_out); t->sel_out = nullptr; t->sel_bytes = 0;
    if (dev_out) memset(dev_out, 0, sizeof(*dev_out));
    t->records = 0;
    if (n == 0) return true;
    // time / record functions: the pair (RECORD_TIME: its key) packed once per call
    t->sel_const.clear();
    std::vector<SpSelKey> sel = t->sel;
    for (size_t i = 0; i < sel.size(); i++) {
        if (sel[i].star < 2) continue;
        const SelKey &k = t->q.keys[i];
        const size_t at = t->sel_const.size();
        pk_str(t->sel_const, k.out_name.data(), k.out_name.size());
        if (sel[i].star == 2) pk_tfunc_value(t->sel_const, k.tfunc, t->tag, now_sec, now_nsec);
        if (t->sel_const.size() > 60000) { set_err("stream processor: time / record function pairs too long"); return false; }
        sel[i].alias_off = (uint16_t) at; sel[i].alias_len = (uint16_t) (t->sel_const.size() - at);
    }
    if (!t->sel_const.empty()) {
        if (!t->d_sconst.ensure(t->sel_const.size() + 16)) return false;
        HIPOK(hipMemcpyAsync(t->d_sconst.p, t->sel_const.data(), t->sel_const.size(), hipMemcpyHostToDevice, st));
    }
    if (!t->d_slen.ensure(n * 4) || !t->d_soff.ensure((n + 1) * 8) || !t->d_misc.ensure(sizeof(SpMisc)) || !t->d_gid.ensure(scan_tmp_elems(n) * 8)) return false;
    SpMisc *dm = t->d_misc.as<SpMisc>();
    SpSelArgs a;
    memset(&a, 0, sizeof(a));
    a.data = (const uint8_t *) in->data; a.row_off = in->row_off; a.bytes = in->bytes; a.plan = t->d_plan.as<SpPlan>();
    a.nsel = (int) sel.size();
    for (int i = 0; i < a.nsel; i++) a.sel[i] = sel[(size_t) i];
    a.consts = t->d_sconst.as<uint8_t>();
    a.out_len = t->d_slen.as<uint32_t>(); a.out_off = t->d_soff.as<uint64_t>();
    a.first_bad = &dm->first_bad; a.records = &dm->counts[0]; a.flags = &dm->flags;
    SpMisc hm;
    uint64_t total = 0;
    for (int pass = 0; pass < 2; pass++) {
        // (a chunk with an object that does not decode: msgpack_unpack_next stops there -- the rows before it, once more)
        memset(&hm, 0, sizeof(hm));
        hm.first_bad = ~0ull;
        HIPOK(hipMemcpyAsync(t->d_misc.p, &hm, sizeof(hm), hipMemcpyHostToDevice, st));
        a.n = n;
        { Profile pr(t, 0, st); launch_sp_select(a, false, st); }       // (flbgpu_sp_profile: [0] the size pass, [1] the emit pass)
        launch_scan(a.out_len, n, t->d_gid.as<uint64_t>(), t->d_soff.as<uint64_t>(), st);
        HIPOK(hipMemcpyAsync(&hm, t->d_misc.p, sizeof(hm), hipMemcpyDeviceToHost, st));
        HIPOK(hipMemcpyAsync(&total, t->d_soff.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
        if (hm.first_bad >= n) break;
        n = hm.first_bad;
        if (n == 0) { total = 0; hm.counts[0] = 0; break; }
    }
    if (hm.flags & SPF_BAD_RECORD) { set_err("stream processor: a record is not [time, map] / [[time, metadata], map]"); return false; }
    t->records = hm.counts[0];
    if (total == 0) return true;
    if (!t->d_sout.ensure(total + 16)) return false;
    a.out = t->d_sout.as<uint8_t>();
    { Profile pr(t, 1, st); launch_sp_select(a, true, st); }
    if (dev_out) {
        // the projected records stay in HBM: one row per incoming row (empty where nothing leaves), the layout every *_run_dev takes
        HIPOK(hipStreamSynchronize(st));
        dev_out->data = t->d_sout.p; dev_out->row_off = t->d_soff.as<uint64_t>(); dev_out->n = n; dev_out->bytes = total;
        return true;
    }
    t->sel_out = malloc(total);
    if (!t->sel_out) { set_err("out of memory"); return false; }
    t->sel_bytes = total;
    HIPOK(hipMemcpyAsync(t->sel_out, a.out, total, hipMemcpyDeviceToHost, st));
    HIPOK(hipStreamSynchronize(st));
    return true;
}

bool run_dev(flbgpu_sp *t, const flbgpu_dev_chunk *in, hipStream_t st, uint32_t now_sec, uint32_t now_nsec) {
    if (t->q.select_only) return run_select(t, in, st, now_sec, now_nsec);
    L2mState &s = t->tab;
    const SpPlan &pl = t->plan;
    const uint64_t n = in->n;
    if (n == 0) return true;
    const int nsrc = std::max(pl.nsrc, 1);
    if (!t->d_gid.ensure(n * 4) || !t->d_val.ensure(n * 8 * nsrc) || !t->d_vt.ensure(n * nsrc) || !t->d_misc.ensure(sizeof(SpMisc))) return false;
    SpMisc hm;
    L2mCtr hc;
    const int cus = device_cus() > 0 ? device_cus() : 256;
    for (int attempt = 0;; attempt++) {
        if (attempt > 40) { set_err("stream processor: group dictionary keeps overflowing"); return false; }
        memset(&hm, 0, sizeof(hm));
        hm.first_bad = ~0ull;
        memcpy(hm.col_class, t->col_class, sizeof(hm.col_class));
        HIPOK(hipMemcpyAsync(t->d_misc.p, &hm, sizeof(hm), hipMemcpyHostToDevice, st));
        HIPOK(hipMemsetAsync(&s.d_ctr.as<L2mCtr>()->overflow, 0, sizeof(unsigned int), st));
        SpArgs a;
        a.data = (const uint8_t *) in->data; a.row_off = in->row_off; a.n = n; a.bytes = in->bytes;
        a.plan = t->d_plan.as<SpPlan>();
        a.t = l2m_table_of(&s);
        a.gid_col = t->d_gid.as<uint32_t>(); a.val_col = t->d_val.as<uint64_t>(); a.vt_col = t->d_vt.as<uint8_t>();
        SpMisc *dm = t->d_misc.as<SpMisc>();
        a.first_bad = &dm->first_bad; a.counts = dm->counts; a.col_class = dm->col_class; a.flags = &dm->flags;
        { Profile ps(t, 0, st); launch_sp_extract(a, cus, st); }
        HIPOK(hipMemcpyAsync(&hm, t->d_misc.p, sizeof(hm), hipMemcpyDeviceToHost, st));
        HIPOK(hipMemcpyAsync(&hc, s.d_ctr.p, sizeof(hc), hipMemcpyDeviceToHost, st));
        HIPOK(hipStreamSynchronize(st));
        if (!hc.overflow && hm.counts[1] > 0) {
            launch_sp_generic(a, st);
            HIPOK(hipMemcpyAsync(&hm, t->d_misc.p, sizeof(hm), hipMemcpyDeviceToHost, st));
            HIPOK(hipMemcpyAsync(&hc, s.d_ctr.p, sizeof(hc), hipMemcpyDeviceToHost, st));
            HIPOK(hipStreamSynchronize(st));
        }
        if (!hc.overflow) break;
        if (!l2m_table_grow(&s, st)) return false;          // nothing was aggregated yet: the pass runs again
    }
    if (hm.flags) {
        set_err("stream processor: %s", (hm.flags & SPF_BAD_RECORD) ? "a record is not [time, map] / [[time, metadata], map]"
                                        : (hm.flags & SPF_KEY_NUL) ? "NUL inside a string GROUP BY value (the reference compares with strcmp)"
                                        : (hm.flags & SPF_KEY_NAN) ? "NaN GROUP BY value" : "decimal exponent beyond what this path decides (strtold range)");
        return false;
    }
    for (int g = 0; g < pl.ngb; g++) {
        const unsigned int m = hm.col_class[g];
        if (m & (m - 1)) {
            set_err("stream processor: GROUP BY column %d mixes value classes in one window (int / float / string: the reference's "
                    "rb-tree comparator is not an order there, flb_sp_groupby.c:77)", g);
            return false;
        }
        if (t->q.window == 2 && (m & 4)) {
            set_err("stream processor: string GROUP BY value in a HOPPING window (a slot's nodes share the key's string with the window's "
                    "node and both free it, flb_sp.c:1985: the reference dies with a double free)");
            return false;
        }
        t->col_class[g] = m;
    }
    SpAggArgs g;
    g.gid_col = t->d_gid.as<uint32_t>(); g.val_col = t->d_val.as<uint64_t>(); g.vt_col = t->d_vt.as<uint8_t>(); g.n = n;
    g.first_bad = &t->d_misc.as<SpMisc>()->first_bad;
    g.rows = s.d_rows.as<unsigned long long>(); g.W = s.W; g.nsrc = pl.nsrc; g.idx_base = t->idx_base;
    g.n_series = &s.d_ctr.as<L2mCtr>()->n_series;
    { Profile ps(t, 1, st); launch_sp_aggregate(g, cus, st); }
    HIPOK(hipStreamSynchronize(st));
    t->idx_base += n;
    t->records += hm.counts[0];
    return true;
}

// ---- HOPPING windows on the host.  The view of a live node: the device row (everything since the node was created) minus
// what flb_sp_window_prune took away again.  aggregate_num is a struct, not a union: an I64-typed num has f64 == 0.0, and the
// i64 of an F64-typed one is stale (its value when the first non-zero float arrived) -- the one thing the rows cannot give.
// It is only read when an I64-typed node meets an F64-typed slot of an earlier life of the same group: refused.
flbgpu_sp::HopNode hop_view(flbgpu_sp *t, const Group &grp) {
    const SpPlan &pl = t->plan;
    const size_t A = 1 + (size_t) SP_SRC_MAX * pl.nsrc;
    const uint64_t *row = grp.row.data();
    const flbgpu_sp::HopLife *lf = nullptr;
    auto it = t->hop_life.find(grp.key);
    if (it != t->hop_life.end()) lf = &it->second;
    flbgpu_sp::HopNode n;
    n.records = (int64_t) row[A] - (lf ? lf->rm_records : 0);
    n.src.resize((size_t) pl.nsrc);
    for (int s = 0; s < pl.nsrc; s++) {
        const uint64_t *ad = row + A + 1 + (size_t) SP_SRC_ADD * s;
        flbgpu_sp::HopNum &x = n.src[(size_t) s];
        x.f = ad[SP_A_NFLT] > 0;
        const int64_t rmi = lf ? lf->rm_i[(size_t) s] : 0;
        if (x.f) x.f64 = bits_double(l2m_limbs_bits(ad + SP_A_LIMB, ad[SP_A_NAN], ad[SP_A_PINF], ad[SP_A_NINF])) - (double) rmi - (lf ? lf->rm_f[(size_t) s] : 0.0);
        else x.i64 = (int64_t) (ad[SP_A_ISUM] - (uint64_t) rmi);
    }
    return n;
}

// the sources a SUM / AVG key reads (aggregate_func_remove_sum runs for those keys only)
std::vector<char> hop_sum_sources(const flbgpu_sp *t) {
    std::vector<char> u((size_t) std::max(t->plan.nsrc, 1), 0);
    for (size_t ki = 0; ki < t->q.keys.size(); ki++)
        if ((t->q.keys[ki].func == F_SUM || t->q.keys[ki].func == F_AVG) && t->key_src[ki] >= 0) u[(size_t) t->key_src[ki]] = 1;
    return u;
}

bool hop_stale(const char *where) {
    set_err("stream processor: HOPPING window, %s: an int-typed node meets a float-typed slot of an earlier life of the same group "
            "(the reference subtracts the slot's stale i64, which depends on arrival order)", where);
    return false;
}

// sp_process_hopping_slot: the slot = a clone of every live node minus the slots still in the list
bool hop_slot(flbgpu_sp *t) {
    Snapshot sn;
    if (!snapshot(t, sn)) return false;
    const std::vector<char> used = hop_sum_sources(t);
    flbgpu_sp::HopSlot hs;
    for (const Group &grp : sn.groups) {
        flbgpu_sp::HopNode c = hop_view(t, grp);
        for (const flbgpu_sp::HopSlot &prev : t->hop_slots) {
            auto it = prev.nodes.find(grp.key);
            if (it == prev.nodes.end()) continue;
            c.records -= it->second.records;
            for (size_t s = 0; s < c.src.size(); s++) {
                if (!used[s]) continue;
                const flbgpu_sp::HopNum &p = it->second.src[s];
                if (!c.src[s].f) { if (p.f) return hop_stale("closing a slot"); c.src[s].i64 = (int64_t) ((uint64_t) c.src[s].i64 - (uint64_t) p.i64); }
                else c.src[s].f64 -= p.f64;
            }
        }
        if (c.records > 0) hs.nodes.emplace(grp.key, std::move(c));
    }
    hs.records = (int64_t) t->records;
    for (const flbgpu_sp::HopSlot &prev : t->hop_slots) hs.records -= prev.records;
    t->hop_slots.push_back(std::move(hs));
    return true;
}

// flb_sp_window_prune, FLB_SP_WINDOW_HOPPING: the oldest slot leaves the window
bool hop_prune(flbgpu_sp *t, const Snapshot &sn) {
    if (t->hop_slots.empty()) return true;
    const flbgpu_sp::HopSlot &hs = t->hop_slots.front();
    const std::vector<char> used = hop_sum_sources(t);
    L2mState &st = t->tab;
    for (const Group &grp : sn.groups) {
        auto it = hs.nodes.find(grp.key);
        if (it == hs.nodes.end()) continue;
        const flbgpu_sp::HopNode node = hop_view(t, grp);
        if (it->second.records == node.records) {
            // the node is destroyed: its row starts over (a later record creates a new node at the end of the list)
            HIPOK(hipMemset((uint8_t *) st.d_rows.p + (size_t) grp.idx * st.W * 8, 0, (size_t) st.W * 8));
            t->hop_life.erase(grp.key);
            continue;
        }
        flbgpu_sp::HopLife &lf = t->hop_life[grp.key];
        if (lf.rm_i.empty()) { lf.rm_i.assign(node.src.size(), 0); lf.rm_f.assign(node.src.size(), 0.0); }
        lf.rm_records += it->second.records;
        for (size_t s = 0; s < node.src.size(); s++) {
            if (!used[s]) continue;
            const flbgpu_sp::HopNum &p = it->second.src[s];
            if (!node.src[s].f) { if (p.f) return hop_stale("pruning"); lf.rm_i[s] = (int64_t) ((uint64_t) lf.rm_i[s] + (uint64_t) p.i64); }
            else lf.rm_f[s] += p.f64;
        }
    }
    t->records -= (uint64_t) hs.records;
    t->hop_slots.pop_front();
    return true;
}

int finish_do(flbgpu_sp *t, uint32_t now_sec, uint32_t now_nsec, void **out_buf, size_t *out_size, int64_t *records) {
    if (records) *records = (int64_t) t->records;
    if (out_buf) *out_buf = nullptr;
    if (out_size) *out_size = 0;
    if (t->q.select_only) {
        // sp_process_data: "records == 0 -> return 0" (nothing handed on); otherwise the buffer, which may be empty
        if (out_buf && out_size && t->records > 0 && t->sel_out) { *out_buf = t->sel_out; *out_size = t->sel_bytes; t->sel_out = nullptr; }
        free(t->sel_out); t->sel_out = nullptr; t->sel_bytes = 0;
        t->records = 0;
        return 0;
    }
    if (t->q.window == 0) {
        // no WINDOW: packaged per appended chunk (flb_sp.c:2051-2054), then pruned (only a window that saw records is)
        std::string out;
        Snapshot sn;
        if (!snapshot(t, sn) || !package(t, sn, now_sec, now_nsec, out)) return -1;
        if (t->records > 0) reset_window(t);
        if (out_buf && out_size && !out.empty()) {
            *out_buf = malloc(out.size());
            if (!*out_buf) { set_err("out of memory"); return -1; }
            memcpy(*out_buf, out.data(), out.size());
            *out_size = out.size();
        }
    }
    return 0;
}

}  // namespace

extern "C" flbgpu_sp *flbgpu_sp_create(const char *sql, int str_conv) {
    if (!sql) { set_err("stream processor: no query"); return nullptr; }
    auto *t = new flbgpu_sp();
    t->str_conv = str_conv ? 1 : 0;
    Parser p;
    if (!tokenize(sql, p.t, p.why) || !p.statement()) {
        set_err("stream processor: invalid or unsupported query (%s): %s", p.why.c_str(), sql);
        delete t;
        return nullptr;
    }
    t->q = p.q;
    std::string why;
    if (!compile(t, why)) { set_err("stream processor: %s: %s", why.c_str(), sql); delete t; return nullptr; }
    t->tab.W = sp_row_words(t->plan.nsrc);
    bool ok = hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) == hipSuccess && l2m_table_init(&t->tab) && t->d_plan.ensure(sizeof(SpPlan)) &&
              hipMemcpy(t->d_plan.p, &t->plan, sizeof(SpPlan), hipMemcpyHostToDevice) == hipSuccess;
    for (auto &e : t->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    if (!ok) { if (!*flbgpu_last_error()) set_err("stream processor: device setup failed"); flbgpu_sp_destroy(t); return nullptr; }
    reset_window(t);
    return t;
}

// the front end alone (no device): 0 and a canonical text of the plan -- output names, GROUP BY mapping, window, source, the
// condition as the postfix program the kernels run -- or -1 (flbgpu_last_error).  tests compare it with the oracle's parse.
extern "C" int flbgpu_sp_parse_check(const char *sql, char *desc, size_t cap) {
    flbgpu_sp t;
    Parser p;
    if (!sql || !tokenize(sql, p.t, p.why) || !p.statement()) { set_err("stream processor: invalid or unsupported query (%s)", p.why.c_str()); return -1; }
    t.q = p.q;
    std::string why;
    if (!compile(&t, why)) { set_err("stream processor: %s", why.c_str()); return -1; }
    const SpPlan &pl = t.plan;
    auto keyname = [&](int k) {
        const SpKeyRef &r = pl.keys[k];
        std::string o(pl.blob + r.name_off, r.name_len);
        for (int j = 0; j < r.nsub; j++) o += "['" + std::string(pl.blob + r.sub_off[j], r.sub_len[j]) + "']";
        return o;
    };
    auto leaf = [&](int li) {
        const SpLeaf &l = pl.leaf[li];
        char buf[64];
        switch (l.kind) {
        case SPL_KEY: return "K:" + keyname(l.key);
        case SPL_CONTAINS: return "C:" + keyname(l.key);
        case SPL_INT: snprintf(buf, sizeof(buf), "I:%lld", (long long) (int64_t) l.v); return std::string(buf);
        case SPL_FLOAT: snprintf(buf, sizeof(buf), "F:%016llx", (unsigned long long) l.v); return std::string(buf);
        case SPL_BOOL: return std::string(l.v ? "B:1" : "B:0");
        case SPL_NULL: return std::string("N");
        case SPL_TIME: return std::string("T");
        case SPL_STR: {
            std::string o = "S:";
            for (int i = 0; i < l.str_len; i++) { snprintf(buf, sizeof(buf), "%02x", (unsigned char) pl.blob[l.str_off + i]); o += buf; }
            return o;
        }
        default: return std::string("?");
        }
    };
    std::string o = "keys=";
    for (size_t i = 0; i < t.q.keys.size(); i++) {
        o += (i ? "|" : "") + t.q.keys[i].out_name + ":" + std::to_string(t.q.keys[i].func) + ":" + std::to_string(t.q.keys[i].gb);
    }
    o += ";gb=";
    for (int g = 0; g < pl.ngb; g++) o += (g ? "|" : "") + keyname(pl.gb_key[g]);
    o += ";window=" + std::to_string(t.q.window) + ":" + std::to_string((long long) t.q.window_sec);
    if (t.q.window == 2) o += ":" + std::to_string((long long) t.q.advance_sec);
    o += ";source=" + std::to_string(t.q.source_type) + ":" + t.q.source + ";stream=" + t.q.stream_name + ";where=";
    static const char *OPN[] = {"EQ", "LT", "LTE", "GT", "GTE", "TRUTH", "NOT", "AND", "OR"};
    for (int i = 0; i < pl.nops; i++) {
        const SpOp &op = pl.ops[i];
        o += i ? " " : "";
        o += OPN[op.op];
        if (op.op <= SPO_GTE) o += "(" + leaf(op.l) + "," + leaf(op.r) + ")";
        else if (op.op == SPO_TRUTH) o += "(" + leaf(op.l) + ")";
    }
    if (desc && cap) { strncpy(desc, o.c_str(), cap - 1); desc[cap - 1] = 0; }
    return 0;
}

extern "C" void flbgpu_sp_destroy(flbgpu_sp *t) {
    if (!t) return;
    L2mState &s = t->tab;
    DevBuf *all[] = {&s.d_slot_hash, &s.d_slot_sid, &s.d_arena, &s.d_key_off, &s.d_key_len, &s.d_series_hash, &s.d_rows, &s.d_ctr,
                     &t->d_plan, &t->d_gid, &t->d_val, &t->d_vt, &t->d_misc, &t->d_in, &t->d_off, &t->d_slen, &t->d_soff, &t->d_sout, &t->d_sconst};
    for (auto *b : all) b->release();
    free(t->sel_out);
    for (auto &e : t->ev) if (e) (void) hipEventDestroy(e);
    if (t->stream) (void) hipStreamDestroy(t->stream);
    delete t;
}

extern "C" int flbgpu_sp_info(const flbgpu_sp *t, int *window_type, int64_t *window_sec, int *source_type, const char **source,
                              const char **stream_name) {
    if (!t) return -1;
    if (window_type) *window_type = t->q.window;
    if (window_sec) *window_sec = t->q.window_sec;
    if (source_type) *source_type = t->q.source_type;
    if (source) *source = t->q.source.c_str();
    if (stream_name) *stream_name = t->q.stream_name.empty() ? nullptr : t->q.stream_name.c_str();
    return 0;
}
extern "C" const char *flbgpu_sp_stream_prop(const flbgpu_sp *t, const char *key) {
    if (!t || !key) return nullptr;
    for (auto &p : t->q.props) if (p.first == key) return p.second.c_str();
    return nullptr;
}
extern "C" int flbgpu_sp_key_count(const flbgpu_sp *t) { return t ? (int) t->q.keys.size() : -1; }
extern "C" const char *flbgpu_sp_key_name(const flbgpu_sp *t, int i) {
    return (t && i >= 0 && (size_t) i < t->q.keys.size()) ? t->q.keys[i].out_name.c_str() : nullptr;
}
extern "C" void flbgpu_sp_set_tag(flbgpu_sp *t, const char *tag, size_t len) { if (t) t->tag.assign(tag ? tag : "", tag ? len : 0); }
extern "C" int flbgpu_sp_select_only(const flbgpu_sp *t) { return t && t->q.select_only ? 1 : 0; }
extern "C" void flbgpu_sp_set_index_base(flbgpu_sp *t, uint64_t base) { if (t) t->idx_base = base; }

extern "C" int flbgpu_sp_do_dev(flbgpu_sp *t, const flbgpu_dev_chunk *in, void *stream, uint32_t now_sec, uint32_t now_nsec, void **out_buf,
                                size_t *out_size, int64_t *records) {
    if (!t || !in) { set_err("stream processor: null argument"); return -1; }
    hipStream_t st = stream ? (hipStream_t) stream : t->stream;
    if (!run_dev(t, in, st, now_sec, now_nsec)) return -1;
    return finish_do(t, now_sec, now_nsec, out_buf, out_size, records);
}

extern "C" int flbgpu_sp_select_dev(flbgpu_sp *t, const flbgpu_dev_chunk *in, void *stream, uint32_t now_sec, uint32_t now_nsec, flbgpu_dev_chunk *out,
                                    int64_t *records) {
    if (!t || !in || !out) { set_err("stream processor: null argument"); return -1; }
    if (!t->q.select_only) { set_err("stream processor: flbgpu_sp_select_dev is for SELECTs without aggregation functions"); return -1; }
    hipStream_t st = stream ? (hipStream_t) stream : t->stream;
    if (!run_select(t, in, st, now_sec, now_nsec, out)) return -1;
    if (records) *records = (int64_t) t->records;
    t->records = 0;
    return 0;
}

extern "C" int flbgpu_sp_do(flbgpu_sp *t, const void *data, size_t bytes, uint32_t now_sec, uint32_t now_nsec, void **out_buf, size_t *out_size,
                            int64_t *records) {
    if (!t || (!data && bytes)) { set_err("stream processor: null argument"); return -1; }
    // msgpack_unpack_next over the chunk (flb_sp.c:1470): the records up to the first object that does not decode
    std::vector<uint64_t> off(bytes / 16 + 1024);
    size_t consumed = 0;
    int64_t n;
    for (;;) {
        n = flbgpu_index_host(data, bytes, off.data(), off.size(), &consumed);
        if (n < 0) return -1;
        if ((size_t) n + 1 < off.size() || consumed >= bytes) break;     // stopped on an object that does not decode, or done
        off.resize(off.size() * 2);
    }
    if (n > 0) {
        if (!t->d_in.ensure(consumed + 16) || !t->d_off.ensure(((size_t) n + 1) * 8)) return -1;
        HIPOK(hipMemcpyAsync(t->d_in.p, data, consumed, hipMemcpyHostToDevice, t->stream));
        HIPOK(hipMemcpyAsync(t->d_off.p, off.data(), ((size_t) n + 1) * 8, hipMemcpyHostToDevice, t->stream));
        flbgpu_dev_chunk ch;
        ch.data = t->d_in.p; ch.row_off = t->d_off.as<uint64_t>(); ch.n = (uint64_t) n; ch.bytes = consumed;
        if (!run_dev(t, &ch, t->stream, now_sec, now_nsec)) return -1;
    }
    return finish_do(t, now_sec, now_nsec, out_buf, out_size, records);
}

extern "C" int flbgpu_sp_timer(flbgpu_sp *t, uint32_t now_sec, uint32_t now_nsec, void **out_buf, size_t *out_size) {
    if (!t) { set_err("stream processor: null argument"); return -1; }
    if (out_buf) *out_buf = nullptr;
    if (out_size) *out_size = 0;
    if (t->q.window == 2) {
        // flb_sp_fd_event, the window.fd branch: package when the window holds records, then the oldest slot leaves
        std::string out;
        Snapshot sn;
        if (!snapshot(t, sn)) return -1;
        if (t->records > 0 && !package(t, sn, now_sec, now_nsec, out)) return -1;
        if (!hop_prune(t, sn)) return -1;
        if (out_buf && out_size && !out.empty()) {
            *out_buf = malloc(out.size());
            if (!*out_buf) { set_err("out of memory"); return -1; }
            memcpy(*out_buf, out.data(), out.size());
            *out_size = out.size();
        }
        return 0;
    }
    if (t->records > 0) {
        std::string out;
        Snapshot sn;
        if (!snapshot(t, sn) || !package(t, sn, now_sec, now_nsec, out)) return -1;
        reset_window(t);
        if (out_buf && out_size && !out.empty()) {
            *out_buf = malloc(out.size());
            if (!*out_buf) { set_err("out of memory"); return -1; }
            memcpy(*out_buf, out.data(), out.size());
            *out_size = out.size();
        }
    }
    return 0;
}

// the hop timer of a HOPPING window (flb_sp_fd_event, the window.fd_hop branch): fires every ADVANCE BY seconds
extern "C" int flbgpu_sp_hop(flbgpu_sp *t) {
    if (!t) { set_err("stream processor: null argument"); return -1; }
    if (t->q.window != 2) { set_err("stream processor: not a HOPPING window"); return -1; }
    return hop_slot(t) ? 0 : -1;
}
extern "C" int64_t flbgpu_sp_window_advance(const flbgpu_sp *t) { return t ? t->q.advance_sec : -1; }

// ---- multi-GPU: the shards of one window.  Every rank runs flbgpu_sp_do on its own records (flbgpu_sp_set_index_base gives the
// ranks disjoint record index ranges: first-seen order across shards); when the window's timer fires the ranks exchange their
// group states and every rank packages the same merged result.
extern "C" int64_t flbgpu_sp_export(flbgpu_sp *t, void *buf, size_t cap) {
    if (!t) { set_err("stream processor: null argument"); return -1; }
    if (t->q.window == 2) { set_err("stream processor: HOPPING windows are not sharded (their slots live on the host of one task)"); return -1; }
    if (t->q.select_only) { set_err("stream processor: a SELECT without aggregation functions keeps no window state"); return -1; }
    Snapshot sn;
    if (!snapshot(t, sn)) return -1;
    const std::string o = serialize(sn, t->tab.W);
    if (buf && cap >= o.size()) memcpy(buf, o.data(), o.size());
    return (int64_t) o.size();
}

static int package_snaps(flbgpu_sp *t, const void *const *snaps, const size_t *sizes, int n, uint32_t now_sec, uint32_t now_nsec, void **out_buf,
                         size_t *out_size) {
    if (out_buf) *out_buf = nullptr;
    if (out_size) *out_size = 0;
    Snapshot total;
    for (int i = 0; i < n; i++) {
        Snapshot sn;
        if (!deserialize((const uint8_t *) snaps[i], sizes[i], t->tab.W, sn)) { set_err("stream processor: snapshot %d does not belong to this query", i); return -1; }
        merge_into(total, sn, t->plan.nsrc);
    }
    if (total.records == 0) return 0;
    std::string out;
    if (!package(t, total, now_sec, now_nsec, out)) return -1;
    if (out_buf && out_size && !out.empty()) {
        *out_buf = malloc(out.size());
        if (!*out_buf) { set_err("out of memory"); return -1; }
        memcpy(*out_buf, out.data(), out.size());
        *out_size = out.size();
    }
    return 0;
}

extern "C" int flbgpu_sp_package_merged(flbgpu_sp *t, const void *const *snaps, const size_t *sizes, int n, uint32_t now_sec, uint32_t now_nsec,
                                        void **out_buf, size_t *out_size) {
    if (!t || (n > 0 && (!snaps || !sizes))) { set_err("stream processor: null argument"); return -1; }
    return package_snaps(t, snaps, sizes, n, now_sec, now_nsec, out_buf, out_size);
}

extern "C" int flbgpu_sp_timer_all_reduce(flbgpu_sp *t, void *rccl_comm, void *stream, uint32_t now_sec, uint32_t now_nsec, void **out_buf,
                                          size_t *out_size) {
    if (!t || !rccl_comm) { set_err("stream processor: null argument"); return -1; }
    hipStream_t st = stream ? (hipStream_t) stream : t->stream;
    Snapshot sn;
    if (!snapshot(t, sn)) return -1;
    const std::string mine = serialize(sn, t->tab.W);
    std::vector<std::vector<uint8_t>> all;
    if (!rccl_all_gather_bytes(rccl_comm, st, mine.data(), mine.size(), all)) return -1;
    std::vector<const void *> ptrs;
    std::vector<size_t> sizes;
    for (auto &b : all) { ptrs.push_back(b.data()); sizes.push_back(b.size()); }
    const int r = package_snaps(t, ptrs.data(), sizes.data(), (int) ptrs.size(), now_sec, now_nsec, out_buf, out_size);
    if (r == 0) reset_window(t);
    return r;
}

// kernel times since the last call (ms, launches) for k_sp_extract, k_sp_aggregate; enable != 0 switches the event timing on
extern "C" void flbgpu_sp_profile(flbgpu_sp *t, int enable, double *ms2, uint64_t *launches2) {
    if (!t) return;
    for (int i = 0; i < 2; i++) {
        if (ms2) ms2[i] = t->kp[i].ms;
        if (launches2) launches2[i] = t->kp[i].launches;
        t->kp[i].ms = 0; t->kp[i].launches = 0;
    }
    t->prof = enable != 0;
}
