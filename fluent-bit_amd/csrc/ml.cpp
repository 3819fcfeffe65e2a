// ml.cpp -- the multiline core in front of the path, behind the C ABI (include/flb_gpu.h flbgpu_ml_*): what in_tail does with a
// `multiline.parser` -- plugins/in_tail/tail_file.c:840-898 cuts the file buffer into lines and hands each to flb_ml_append_text
// (src/multiline/flb_ml.c:685-762) -- for the text lines of one stream and ONE multiline parser of type regex (rules:
// src/multiline/flb_ml_rule.c), endswith or equal; for the latter two also with a parser in front (cri, docker).  Kernels: ml_kernels.inc, mlo_kernels.inc.  Host side: the parser
// definition (flb_ml_parser_create / flb_ml_rule_create / flb_ml_rule_init :279-299 restated as masks), the stream's carried
// state (rule_to_state, the open group's bytes and time), buffers.  No CPU path: every line is matched and packed on the device.
#include <map>
#include "host_int.hpp"
#include "mlo.hpp"

using namespace flbgpu;
static const int ML_TRUNC_ROUNDS = 4096;     // truncating continuations settled per read (flbgpu_ml_append_dev): ~0.15 ms each, so a read stalls 0.6 s at worst

struct MlRuleSrc { std::vector<std::string> from; std::string regex, to; bool start = false; };
struct MlPending { int rule; std::string pat, shown; unsigned opts; bool neg; };

struct flbgpu_ml_parser {
    MlParserDev dev;
    std::vector<MlRuleSrc> src;
    std::vector<GrepRule> rules;
    std::vector<rx::Program> progs;          // the rules' tables on the host: the product automaton is built from them
    // a rule "^(?!A)B" / "^(?=A)B" (the documented idiom for "a line that does not start like a first line"): on a line -- no '\n' inside,
    // so '^' is position 0 only -- it says [not] match(^(?:A)) and match(^B): two match-only automata, no look-around in the tables.
    // rules / progs hold the B parts at [0, n) and the A parts behind them
    std::vector<int> look_of;                // [rule] index of its A part in rules / progs, -1: none
    std::vector<uint8_t> look_neg;
    std::vector<MlPending> pending;           // A parts waiting for init
    // a parser in front (cri, docker, flbgpu_ml_parser_set_subparser): lines are parsed first, the groups live per key_group value
    flbgpu_parser *sub = nullptr;
    bool sub_owned = false;
    std::string key_content, key_group, key_pattern;
    std::vector<TableBlob *> blobs;
    DevBuf d_rules, d_prod;
    uint32_t prod_bytes = 0, prod_nj = 0, prod_nS = 0, prod_T = 0, prod_init = 0;
    bool inited = false;
    flbgpu_ml_parser() { memset(&dev, 0, sizeof(dev)); }
    ~flbgpu_ml_parser() {
        for (TableBlob *b : blobs) delete b;
        d_rules.release(); d_prod.release();
        if (sub && sub_owned) flbgpu_parser_destroy(sub);
    }
};

struct flbgpu_ml_stream {
    flbgpu_ml_parser *p = nullptr;
    hipStream_t stream = nullptr;
    uint32_t state = 0;                       // rule_to_state: 0 none, r + 1
    DevBuf carry[2];                          // the open group's bytes (flb_ml_stream_group.buf), double-buffered
    int cur = 0;
    uint32_t carry_len = 0, carry_tail = MLT_EMPTY, carry_sec = 0, carry_nsec = 0;       // mp_time of the group
    uint32_t carry_trunc = 0;                 // the open group was cut by the buffer limit (flb_ml_stream_group.truncated)
    uint64_t truncations = 0;                 // lines that truncated a buffer (FLB_MULTILINE_TRUNCATED returns)
    // a parser in front: in_tail's packing + filter_parser(sub) give the rows; one carried buffer + first-line map per group
    flbgpu_tail *tail[2] = {nullptr, nullptr};             // [skip_empty_lines]
    flbgpu_filter *subf = nullptr;
    // between the two halves of a read (mlo_phase1 / mlo_phase2; a list of parsers looks at what every parser's first half says)
    MloArgs ph_a;
    uint64_t ph_n = 0, ph_lines = 0, ph_taken = 0;
    bool ph_empty = true;
    std::vector<std::string> gnames;          // [0] the default group
    DevBuf gc_content[MLO_G][2], gc_map[MLO_G][2];
    int gc_cur[MLO_G] = {0, 0, 0, 0};
    uint32_t gc_content_len[MLO_G] = {0, 0, 0, 0}, gc_map_len[MLO_G] = {0, 0, 0, 0}, gc_sec[MLO_G] = {0, 0, 0, 0}, gc_nsec[MLO_G] = {0, 0, 0, 0};
    DevBuf o_rows, o_st, o_nrec, o_base, o_recs, o_size, o_off, o_idx, o_tmp, o_misc;
    DevBuf d_masks, d_tcnt, d_toff, d_scan_tmp, d_nl, d_keep, d_koff, d_ls, d_ll, d_info, d_F, d_sin, d_act, d_c, d_coff, d_head, d_gidx,
           d_ghead, d_plen, d_po, d_pk, d_gC, d_ovr, d_slow, d_fs_tmp, d_rows, d_out, d_misc, d_in;
    ~flbgpu_ml_stream() {
        DevBuf *all[] = {&carry[0], &carry[1], &d_masks, &d_tcnt, &d_toff, &d_scan_tmp, &d_nl, &d_keep, &d_koff, &d_ls, &d_ll, &d_info, &d_F, &d_sin,
                         &d_act, &d_c, &d_coff, &d_head, &d_gidx, &d_ghead, &d_plen, &d_po, &d_pk, &d_gC, &d_ovr, &d_slow, &d_fs_tmp, &d_rows, &d_out, &d_misc, &d_in};
        for (DevBuf *b : all) b->release();
        for (int g = 0; g < MLO_G; g++) for (int k = 0; k < 2; k++) { gc_content[g][k].release(); gc_map[g][k].release(); }
        DevBuf *oall[] = {&o_rows, &o_st, &o_nrec, &o_base, &o_recs, &o_size, &o_off, &o_idx, &o_tmp, &o_misc};
        for (DevBuf *b : oall) b->release();
        if (tail[0]) flbgpu_tail_destroy(tail[0]);
        if (tail[1]) flbgpu_tail_destroy(tail[1]);
        if (subf) flbgpu_filter_destroy(subf);
        if (stream) (void) hipStreamDestroy(stream);
    }
};

static int ml_type_lookup(const char *s) {        // flb_ml.c:84-100 flb_ml_type_lookup
    if (!s || !strcasecmp(s, "regex")) return ML_REGEX;
    if (!strcasecmp(s, "endswith")) return ML_ENDSWITH;
    if (!strcasecmp(s, "equal") || !strcasecmp(s, "eq")) return ML_EQ;
    return -1;
}

// flb_ml_parser_create (src/multiline/flb_ml_parser.c:46-140) + the instance's key_content (flb_ml_parser_instance_create / _set :281-352)
// + the context's buffer limit (flb_ml_create :878-930; < 0: the default of 2 MB, 0: none)
extern "C" flbgpu_ml_parser *flbgpu_ml_parser_create(const char *type, const char *match_string, int negate, const char *key_content, int64_t buffer_limit) {
    if (flbgpu_device_cus() <= 0) { set_err("flbgpu_init has not run: libflbgpu has no CPU path"); return nullptr; }
    const int ty = ml_type_lookup(type);
    if (ty < 0) { set_err("multiline: unknown parser type '%s'", type); return nullptr; }
    auto *p = new flbgpu_ml_parser();
    p->dev.type = ty; p->dev.negate = negate ? 1 : 0;
    if (match_string) {
        const size_t n = strlen(match_string);
        if (n > sizeof(p->dev.match_str)) { set_err("multiline: match_string too long for the GPU path"); delete p; return nullptr; }
        memcpy(p->dev.match_str, match_string, n);
        p->dev.match_len = (uint32_t) n;
    }
    const char *key = key_content && key_content[0] ? key_content : "log";
    const size_t kn = strlen(key);
    if (kn > 255) { set_err("multiline: key_content too long for the GPU path"); delete p; return nullptr; }
    uint32_t o = 0;
    if (kn < 32) p->dev.key[o++] = (uint8_t) (0xa0 | kn);
    else { p->dev.key[o++] = 0xd9; p->dev.key[o++] = (uint8_t) kn; }
    memcpy(p->dev.key + o, key, kn);
    p->dev.key_len = o + (uint32_t) kn;
    p->dev.has_key_content = key_content && key_content[0] ? 1 : 0;
    p->key_content = key_content && key_content[0] ? key_content : "";
    p->dev.buffer_limit = buffer_limit < 0 ? 2ull * 1024 * 1024 : (uint64_t) buffer_limit;
    return p;
}

extern "C" void flbgpu_ml_parser_destroy(flbgpu_ml_parser *p) { delete p; }
// diagnostics: states / joint classes / non-absorbing states of the product automaton (0 states: the rules are walked one by one)
extern "C" void flbgpu_ml_parser_product(const flbgpu_ml_parser *p, uint32_t *states, uint32_t *classes, uint32_t *live) {
    if (states) *states = p && p->prod_bytes ? p->prod_nS : 0;
    if (classes) *classes = p ? p->prod_nj : 0;
    if (live) *live = p ? p->prod_T : 0;
}

// "^(?!A)B" / "^(?=A)B" -> A, B.  False: not of that form
static bool split_leading_lookahead(const char *ps, const char *pe, std::string &A, std::string &B, bool &neg) {
    if (pe - ps < 5 || ps[0] != '^' || ps[1] != '(' || ps[2] != '?' || (ps[3] != '!' && ps[3] != '=')) return false;
    neg = ps[3] == '!';
    int depth = 1;
    bool cls = false;
    const char *q = ps + 4;
    for (; q < pe; q++) {
        if (*q == '\\') { q++; continue; }
        if (cls) { if (*q == ']') cls = false; continue; }
        if (*q == '[') { cls = true; if (q + 1 < pe && q[1] == '^') q++; if (q + 1 < pe && q[1] == ']') q++; continue; }
        if (*q == '(') depth++;
        else if (*q == ')' && --depth == 0) break;
    }
    if (q >= pe) return false;
    // an alternation at the top level ("^(?!A)B|C") scopes the look-ahead to its own branch: not this form
    depth = 0; cls = false;
    for (const char *t = q + 1; t < pe; t++) {
        if (*t == '\\') { t++; continue; }
        if (cls) { if (*t == ']') cls = false; continue; }
        if (*t == '[') { cls = true; if (t + 1 < pe && t[1] == '^') t++; if (t + 1 < pe && t[1] == ']') t++; continue; }
        if (*t == '(') depth++;
        else if (*t == ')') depth--;
        else if (*t == '|' && depth == 0) return false;
    }
    A.assign(ps + 4, (size_t) (q - (ps + 4)));
    B.assign(q + 1, (size_t) (pe - (q + 1)));
    return true;
}

static bool ml_compile(flbgpu_ml_parser *p, const std::string &pat, unsigned opts, const char *shown) {
    GrepRule gr;
    memset(&gr, 0, sizeof(gr));
    rx::Program prog;
    std::string err;
    if (!rx::compile(pat.data(), pat.size(), opts, false, prog, err)) { set_err("multiline: could not compile regex pattern '%s' for the GPU path: %s", shown, err.c_str()); return false; }
    auto *b1 = new TableBlob(), *b2 = new TableBlob();
    p->blobs.push_back(b1); p->blobs.push_back(b2);
    if (!upload_dfa(prog.ascii, *b1, gr.dfa) || !upload_utf8(prog, *b2, gr.utf8)) return false;
    p->rules.push_back(gr);
    p->progs.push_back(std::move(prog));
    return true;
}

// flb_ml_rule_create (flb_ml_rule.c:48-118): from_states split at ',' with blanks trimmed (flb_slist_split_string), the first rule
// must hold a start_state
extern "C" int flbgpu_ml_parser_add_rule(flbgpu_ml_parser *p, const char *from_states, const char *regex, const char *to_state) {
    if (!p || !from_states || !regex) { set_err("multiline rule: missing argument"); return -1; }
    if (p->inited) { set_err("multiline rule: the parser is already initialised"); return -1; }
    if (p->dev.type != ML_REGEX) { set_err("multiline rule: the parser is not of type regex"); return -1; }
    if ((int) p->src.size() >= ML_MAX_RULES) { set_err("multiline: more than %d rules: not on the GPU path", ML_MAX_RULES); return -1; }
    MlRuleSrc r;
    const char *q = from_states;
    while (*q) {
        const char *e = strchr(q, ',');
        if (!e) e = q + strlen(q);
        const char *a = q, *b = e;
        while (a < b && *a == ' ') a++;
        while (b > a && b[-1] == ' ') b--;
        if (b > a) { r.from.emplace_back(a, (size_t) (b - a)); if (r.from.back() == "start_state") r.start = true; }
        q = *e ? e + 1 : e;
    }
    if (r.from.empty()) { set_err("[multiline] rule is empty or has invalid 'from_states' tokens"); return -1; }
    if (!r.start && p->src.empty()) { set_err("[multiline] rule don't contain a 'start_state'"); return -1; }
    r.regex = regex;
    if (to_state && to_state[0]) r.to = to_state;
    const char *ps, *pe;
    unsigned opts;
    rx::split_flb_pattern(regex, &ps, &pe, &opts);
    std::string A, B;
    bool neg = false;
    if (split_leading_lookahead(ps, pe, A, B, neg)) {
        // the B parts stay at the rules' own indexes: the A part is parked and appended behind all rules at init
        if (!ml_compile(p, "^(?:" + B + ")", opts, regex)) return -1;
        MlPending pd;
        pd.rule = (int) p->src.size(); pd.pat = "^(?:" + A + ")"; pd.opts = opts; pd.neg = neg; pd.shown = regex;
        p->pending.push_back(pd);
    }
    else if (!ml_compile(p, std::string(ps, (size_t) (pe - ps)), opts, regex)) return -1;
    p->src.push_back(r);
    return 0;
}

// the built-in regex parsers: src/multiline/flb_ml_parser_java.c:59-128, _go.c:59-125, _python.c:60-83, _ruby.c:59-71
extern "C" int flbgpu_ml_parser_builtin(flbgpu_ml_parser *p, const char *name) {
    struct R { const char *from, *rx, *to; };
    static const R java[] = {
        {"start_state, java_start_exception", "/(.)(?:Exception|Error|Throwable|V8 errors stack trace)[:\\r\\n]/", "java_after_exception"},
        {"java_after_exception", "/^[\\t ]*nested exception is:[\\t ]*/", "java_start_exception"},
        {"java_after_exception", "/^[\\r\\n]*$/", "java_after_exception"},
        {"java_after_exception, java", "/^[\\t ]+(?:eval )?at /", "java"},
        {"java_after_exception, java", "/^[\\t ]+--- End of inner exception stack trace ---$/", "java"},
        {"java_after_exception, java", "/^--- End of stack trace from previous (?x:)location where exception was thrown ---$/", "java"},
        {"java_after_exception, java", "/^[\\t ]*(?:Caused by|Suppressed):/", "java_after_exception"},
        {"java_after_exception, java", "/^[\\t ]*... \\d+ (?:more|common frames omitted)/", "java"}, {nullptr, nullptr, nullptr}};
    static const R go[] = {
        {"start_state", "/\\bpanic: /", "go_after_panic"},
        {"start_state", "/http: panic serving/", "go_goroutine"},
        {"go_after_panic", "/^$/", "go_goroutine"},
        {"go_after_panic, go_after_signal, go_frame_1", "/^$/", "go_goroutine"},
        {"go_after_panic", "/^\\[signal /", "go_after_signal"},
        {"go_goroutine", "/^goroutine \\d+ \\[[^\\]]+\\]:$/", "go_frame_1"},
        {"go_frame_1", "/^(?:[^\\s.:]+\\.)*[^\\s.():]+\\(|^created by /", "go_frame_2"},
        {"go_frame_2", "/^\\s/", "go_frame_1"}, {nullptr, nullptr, nullptr}};
    static const R python[] = {
        {"start_state", "/^Traceback \\(most recent call last\\):$/", "python"},
        {"python", "/^[\\t ]+File /", "python_code"},
        {"python_code", "/[^\\t ]/", "python"},
        {"python", "/^(?:[^\\s.():]+\\.)*[^\\s.():]+:/", "start_state"}, {nullptr, nullptr, nullptr}};
    static const R ruby[] = {
        {"start_state, ruby_start_exception", "/^.+:\\d+:in\\s+.*/", "ruby_after_exception"},
        {"ruby_after_exception, ruby", "/^\\s+from\\s+.*:\\d+:in\\s+.*/", "ruby"}, {nullptr, nullptr, nullptr}};
    if (!p || !name) { set_err("multiline: missing argument"); return -1; }
    if (!strcasecmp(name, "cri") || !strcasecmp(name, "docker")) {
        // src/multiline/flb_ml_parser_cri.c:24-75: EQ "F" on _p, content log, groups by stream, the regex parser in front;
        // flb_ml_parser_docker.c:25-105: ENDSWITH "\n" on log, groups by stream, docker's json parser in front (both: Time_Keep on)
        const bool cri = !strcasecmp(name, "cri");
        flbgpu_parser *sub = cri ? flbgpu_parser_create("_ml_cri", "^(?<time>.+?) (?<stream>stdout|stderr) (?<_p>F|P) (?<log>.*)$", 0, "%Y-%m-%dT%H:%M:%S.%L%z", "time", nullptr, 1, 0, nullptr)
                                 : flbgpu_parser_create_json("_ml_json_docker", "%Y-%m-%dT%H:%M:%S.%L", "time", nullptr, 1, 0);
        if (!sub) return -1;
        p->dev.type = cri ? ML_EQ : ML_ENDSWITH; p->dev.negate = 0;
        p->dev.match_len = 1; p->dev.match_str[0] = cri ? 'F' : '\n';
        if (p->key_content.empty()) {
            p->key_content = "log";
            p->dev.key[0] = 0xa3; memcpy(p->dev.key + 1, "log", 3); p->dev.key_len = 4; p->dev.has_key_content = 1;
        }
        p->sub = sub; p->sub_owned = true;
        p->key_group = "stream"; p->key_pattern = cri ? "_p" : "";
        return flbgpu_ml_parser_init(p);
    }
    const R *t = !strcasecmp(name, "java") ? java : !strcasecmp(name, "go") ? go : !strcasecmp(name, "python") ? python : !strcasecmp(name, "ruby") ? ruby : nullptr;
    if (!t) { set_err("multiline: built-in parser '%s' needs a sub-parser (docker, cri) or does not exist: not on the GPU path", name); return -1; }
    for (int i = 0; t[i].from; i++) if (flbgpu_ml_parser_add_rule(p, t[i].from, t[i].rx, t[i].to) != 0) return -1;
    return flbgpu_ml_parser_init(p);
}

// The product of the rules' match-only DFAs over the bytes a line can hold (no '\n': the line loop cuts there; a byte >= 0x80 leaves
// the table -- the per-rule UTF-8 tables take the line).  A component is a state of its rule's DFA, ACC (the rule has matched: its
// DFA said D_ACCEPT) or DEAD (no match can follow: nothing accepting is reachable).  States whose components are all ACC / DEAD are
// absorbing and numbered last, so the walk leaves at `state >= T`.  False: too large for LDS -- the kernel then walks rule by rule.
static bool build_product(flbgpu_ml_parser *p, std::vector<uint8_t> &blob) {
    const int R = (int) p->progs.size();
    constexpr uint16_t ACC = 0xFFFF, DEAD = 0xFFFE;
    // joint classes of the bytes < 0x80 except '\n'; class nj - 1 = everything else
    std::map<std::vector<int>, int> sig2cls;
    std::vector<int> jcls(256, -1), rep;
    for (int b = 0; b < 128; b++) {
        if (b == '\n') continue;
        std::vector<int> sig((size_t) R);
        for (int r = 0; r < R; r++) sig[(size_t) r] = p->progs[(size_t) r].ascii.cls[b];
        auto it = sig2cls.find(sig);
        if (it == sig2cls.end()) { it = sig2cls.emplace(sig, (int) rep.size()).first; rep.push_back(b); }
        jcls[(size_t) b] = it->second;
    }
    const int nj = (int) rep.size() + 1;
    for (int b = 0; b < 256; b++) if (jcls[(size_t) b] < 0) jcls[(size_t) b] = nj - 1;
    // per rule: the states from which a match is still possible on the rest of a line.  NOT judged on the bytes < 0x80 alone (ADVICE
    // round 3): a rule like /^\s+原因/ has no accepting path over them at all, its component would be DEAD from the start, the
    // product state absorbing, and the walk would stop before it met the byte >= 0x80 that sends the line to the per-rule tables.
    // rx.cpp's d_live is liveness over texts of any characters (without line feeds).
    std::vector<std::vector<uint8_t>> good((size_t) R);
    for (int r = 0; r < R; r++) {
        const rx::Program &pr = p->progs[(size_t) r];
        if (pr.ascii_stub || pr.ascii.nD == 0 || pr.ascii.d_live.size() != (size_t) pr.ascii.nD) return false;    // (no ASCII automaton: rule by rule)
        good[(size_t) r] = pr.ascii.d_live;
        for (int s = 0; s < pr.ascii.nD; s++) if (pr.ascii.d_final[(size_t) s]) good[(size_t) r][(size_t) s] = 1;
    }
    std::map<std::vector<uint16_t>, int> ids;
    std::vector<std::vector<uint16_t>> states;
    auto canon = [&](int r, uint16_t s) -> uint16_t { return (s == ACC || s == DEAD) ? s : good[(size_t) r][s] ? s : DEAD; };
    std::vector<uint16_t> init((size_t) R);
    for (int r = 0; r < R; r++) init[(size_t) r] = canon(r, (uint16_t) p->progs[(size_t) r].ascii.d_init);
    ids.emplace(init, 0); states.push_back(init);
    std::vector<uint32_t> delta;                  // [state][nj - 1]
    for (size_t si = 0; si < states.size(); si++) {
        if (states.size() > 4096 || states.size() * (size_t) nj * 2 > 60000) return false;
        const std::vector<uint16_t> cur = states[si];
        for (int c = 0; c + 1 < nj; c++) {
            std::vector<uint16_t> nx((size_t) R);
            for (int r = 0; r < R; r++) {
                const rx::TableSet &t = p->progs[(size_t) r].ascii;
                const uint16_t s = cur[(size_t) r];
                if (s == ACC || s == DEAD) { nx[(size_t) r] = s; continue; }
                const uint16_t n = t.ddelta[(size_t) s * t.ncls + t.cls[rep[(size_t) c]]];
                if (n == 0xFFFE) return false;                       // a byte < 0x80 never poisons; be safe
                nx[(size_t) r] = n == 0xFFFF ? ACC : canon(r, n);
            }
            auto it = ids.find(nx);
            if (it == ids.end()) { it = ids.emplace(nx, (int) states.size()).first; states.push_back(nx); }
            delta.push_back((uint32_t) it->second);
        }
    }
    const int nS = (int) states.size();
    if ((size_t) nS * (size_t) nj * 2 + 256 + (size_t) nS * 2 > 60000) return false;
    // absorbing states last
    std::vector<int> order((size_t) nS), newid((size_t) nS);
    int T = 0;
    auto absorbing = [&](int s) { for (uint16_t c : states[(size_t) s]) if (c != ACC && c != DEAD) return false; return true; };
    for (int s = 0; s < nS; s++) if (!absorbing(s)) order[(size_t) T++] = s;
    int q = T;
    for (int s = 0; s < nS; s++) if (absorbing(s)) order[(size_t) q++] = s;
    for (int i = 0; i < nS; i++) newid[(size_t) order[(size_t) i]] = i;
    blob.assign(256 + (size_t) nS * 2 + (size_t) nS * (size_t) nj * 2, 0);
    for (int b = 0; b < 256; b++) blob[(size_t) b] = (uint8_t) jcls[(size_t) b];
    uint16_t *fm = (uint16_t *) (blob.data() + 256), *dl = fm + nS;
    for (int i = 0; i < nS; i++) {
        const int s = order[(size_t) i];
        uint16_t m = 0;
        auto acc = [&](int r) -> bool { const uint16_t c = states[(size_t) s][(size_t) r]; return c == ACC || (c != DEAD && p->progs[(size_t) r].ascii.d_final[c]); };
        for (int r = 0; r < p->dev.nrules; r++) {
            bool ok = acc(r);
            const int la = p->dev.look_idx[r];
            if (ok && la >= 0) ok = ((p->dev.look_neg >> r) & 1) ? !acc(la) : acc(la);
            if (ok) m |= (uint16_t) (1u << r);
        }
        fm[i] = m;
        for (int c = 0; c + 1 < nj; c++) dl[(size_t) i * nj + c] = (uint16_t) newid[(size_t) delta[(size_t) s * (nj - 1) + c]];
        dl[(size_t) i * nj + (nj - 1)] = 0xFFFF;
    }
    p->prod_nj = (uint32_t) nj; p->prod_nS = (uint32_t) nS; p->prod_T = (uint32_t) T; p->prod_init = (uint32_t) newid[0];
    return true;
}

// a [MULTILINE_PARSER] with `parser`, key_group, key_pattern (src/multiline/flb_ml_parser.c:46-140; flb_ml_parsers_init resolves the name):
// ENDSWITH / EQ types only -- a parser in front of a regex state machine is not on the GPU path.  `sub` stays the caller's.
extern "C" int flbgpu_ml_parser_set_subparser(flbgpu_ml_parser *p, flbgpu_parser *sub, const char *key_group, const char *key_pattern) {
    if (!p || !sub) { set_err("multiline: missing argument"); return -1; }
    if (p->inited) { set_err("multiline: the parser is already initialised"); return -1; }
    if (p->dev.type == ML_REGEX) { set_err("multiline: a parser in front of a regex parser is not on the GPU path"); return -1; }
    if (p->key_content.empty()) { set_err("multiline: a parser in front needs key_content"); return -1; }
    p->sub = sub; p->sub_owned = false;
    p->key_group = key_group ? key_group : ""; p->key_pattern = key_pattern ? key_pattern : "";
    return 0;
}

// flb_ml_rule_init (flb_ml_rule.c:279-299): every rule's to_state_map, here as masks over the rules
extern "C" int flbgpu_ml_parser_init(flbgpu_ml_parser *p) {
    if (!p) { set_err("multiline: no parser"); return -1; }
    if (p->inited) return 0;
    const int n = (int) p->src.size();
    if (p->dev.type == ML_REGEX && n == 0) { set_err("multiline: a regex parser without rules"); return -1; }
    p->dev.nrules = n;
    for (int i = 0; i < 16; i++) p->dev.look_idx[i] = -1;
    for (const MlPending &pd : p->pending) {
        if (p->rules.size() >= 32) { set_err("multiline: too many look-ahead rules for the GPU path"); return -1; }
        p->dev.look_idx[pd.rule] = (int8_t) p->rules.size();
        if (pd.neg) p->dev.look_neg |= 1u << pd.rule;
        if (!ml_compile(p, pd.pat, pd.opts, pd.shown.c_str())) return -1;
    }
    p->pending.clear();
    for (int i = 0; i < n; i++) if (p->src[(size_t) i].start) p->dev.start_mask |= 1u << i;
    for (int i = 0; i < n; i++) {
        const MlRuleSrc &r = p->src[(size_t) i];
        if (r.to.empty()) continue;
        uint32_t map = 0;
        for (int j = 0; j < n; j++)
            for (const std::string &f : p->src[(size_t) j].from) if (f == r.to) { map |= 1u << j; break; }
        if (!map) { set_err("[multiline parser] to_state='%s' is not registered", r.to.c_str()); return -1; }
        p->dev.cont_mask[i + 1] = map & ~p->dev.start_mask;             // flb_ml_rule_process skips start rules among the continuations
        if (map & p->dev.start_mask) p->dev.flush_after |= 1u << i;     // try_flushing_buffer: a start rule may follow
    }
    if (n) {
        if (!p->d_rules.ensure(p->rules.size() * sizeof(GrepRule))) return -1;
        if (hipMemcpy(p->d_rules.p, p->rules.data(), p->rules.size() * sizeof(GrepRule), hipMemcpyHostToDevice) != hipSuccess) { set_err("multiline: uploading the rules failed"); return -1; }
        std::vector<uint8_t> blob;
        const char *no = getenv("FLBGPU_ML_NO_PRODUCT");
        if (!(no && no[0] == '1') && build_product(p, blob)) {
            blob.resize((blob.size() + 15) & ~(size_t) 15);
            if (!p->d_prod.ensure(blob.size())) return -1;
            if (hipMemcpy(p->d_prod.p, blob.data(), blob.size(), hipMemcpyHostToDevice) != hipSuccess) { set_err("multiline: uploading the product table failed"); return -1; }
            p->prod_bytes = (uint32_t) blob.size();
        }
    }
    p->inited = true;
    return 0;
}

extern "C" flbgpu_ml_stream *flbgpu_ml_stream_create(flbgpu_ml_parser *p) {
    if (!p || !p->inited) { set_err("multiline: the parser is not initialised"); return nullptr; }
    auto *s = new flbgpu_ml_stream();
    s->p = p;
    if (hipStreamCreate(&s->stream) != hipSuccess) { set_err("hipStreamCreate failed"); delete s; return nullptr; }
    if (p->sub) {
        // the rows: in_tail's packing of the lines (key "log"), then filter_parser(Key_Name log, the parser in front) on them
        flbgpu_parser *arr[1] = {p->sub};
        s->tail[0] = flbgpu_tail_create("log", nullptr, nullptr, nullptr, 0);
        s->tail[1] = flbgpu_tail_create("log", nullptr, nullptr, nullptr, 1);
        s->subf = flbgpu_filter_parser_create("log", 0, 0, 1, arr);
        if (!s->tail[0] || !s->tail[1] || !s->subf) { delete s; return nullptr; }
        s->gnames.push_back("_default");
    }
    return s;
}
extern "C" void flbgpu_ml_stream_destroy(flbgpu_ml_stream *s) { delete s; }

// ---- a parser in front (mlo_kernels.inc)
static void put_name(uint8_t *dst, uint32_t *len, const std::string &v, size_t cap) { *len = (uint32_t) (v.size() < cap ? v.size() : cap); memcpy(dst, v.data(), *len); }

static void mlo_fill_groups(flbgpu_ml_stream *s, MloArgs &a) {
    a.ngroups = (uint32_t) s->gnames.size();
    for (uint32_t g = 0; g < a.ngroups; g++) put_name(a.names[g], &a.name_len[g], s->gnames[g], 32);
    for (int g = 0; g < MLO_G; g++) {
        a.carry[g].content = s->gc_content[g][s->gc_cur[g]].as<uint8_t>(); a.carry[g].map = s->gc_map[g][s->gc_cur[g]].as<uint8_t>();
        a.carry[g].content_len = s->gc_content_len[g]; a.carry[g].map_len = s->gc_map_len[g]; a.carry[g].sec = s->gc_sec[g]; a.carry[g].nsec = s->gc_nsec[g];
    }
}

// The first half of a read: the lines through in_tail's packing and the parser in front, every line classified (k_mlo_extract).  Nothing
// of the stream's state changes; ph_lines / ph_taken say how many lines there are and how many this parser takes (count: only then).
static int mlo_phase1(flbgpu_ml_stream *s, const void *d_text, uint64_t bytes, uint32_t ts_sec, uint32_t ts_nsec, int skip_empty_lines, int flush,
                      uint64_t *processed, bool count) {
    auto fail = [](const char *w) -> int { set_err("multiline: %s", w); return -1; };
    flbgpu_ml_parser *p = s->p;
    hipStream_t st = s->stream;
    s->ph_empty = true; s->ph_n = 0; s->ph_lines = 0; s->ph_taken = 0;
    if (p->key_content.size() > 63 || p->key_group.size() > 63 || p->key_pattern.size() > 63 || p->dev.match_len > 64) return fail("key names / match string too long for the GPU path");
    flbgpu_dev_chunk T, P;
    memset(&T, 0, sizeof(T)); memset(&P, 0, sizeof(P));
    uint64_t lines = 0;
    // (the line packing and the parser in front run on their own streams: what this stream still copies into d_text must have landed)
    if (hipStreamSynchronize(st) != hipSuccess) return fail("upload failed");
    if (bytes && flbgpu_tail_run_dev(s->tail[skip_empty_lines ? 1 : 0], d_text, bytes, 0, ts_sec, ts_nsec, &T, processed, &lines) != 0) return -1;
    const uint64_t n = T.n;
    bool pending = false;
    for (int g = 0; g < MLO_G; g++) pending = pending || s->gc_map_len[g] > 0;
    if (n == 0 && !(flush && pending)) return 0;
    s->ph_empty = false; s->ph_n = n;
    const uint32_t *pinfo = nullptr;
    if (n) {
        const int r = flbgpu_filter_run_dev(s->subf, &T, &P, nullptr);
        pinfo = s->subf->d_info.as<uint32_t>();
        if (r != FLBGPU_FILTER_MODIFIED) { P = T; pinfo = nullptr; }     // nothing was emitted: every row is a skipped line -- or a dropped record (refused)
        if (P.n != n) return fail("the parser in front changed the number of rows");
    }
    const uint64_t N1 = n + 1;
    static const uint64_t zero_row[2] = {0, 0};
    (void) zero_row;
    if (!s->o_rows.ensure(N1 * sizeof(MloRow)) || !s->o_st.ensure(N1 * sizeof(MloState)) || !s->o_nrec.ensure(N1 * 4) || !s->o_base.ensure((N1 + 1) * 8) ||
        !s->o_idx.ensure((size_t) MLO_G * (n ? n : 1) * 4) || !s->o_tmp.ensure(mlo_vscan_tmp_bytes(N1)) || !s->o_misc.ensure(32 * 4) ||
        !s->d_scan_tmp.ensure(scan_tmp_elems(N1 * (MLO_G + 1)) * sizeof(uint64_t))) return -1;
    unsigned int hmisc[32];
    memset(hmisc, 0, sizeof(hmisc));
    hmisc[0] = 0xFFFFFFFFu;
    if (hipMemcpyAsync(s->o_misc.p, hmisc, sizeof(hmisc), hipMemcpyHostToDevice, st) != hipSuccess) return fail("upload failed");
    MloArgs a;
    memset(&a, 0, sizeof(a));
    a.tdata = (const uint8_t *) T.data; a.trow = T.row_off; a.pdata = (const uint8_t *) P.data; a.prow = P.row_off; a.pinfo = pinfo;
    a.n = n; a.flush_all = flush ? 1 : 0;
    a.type = p->dev.type; a.negate = p->dev.negate; a.match_len = p->dev.match_len; memcpy(a.match_str, p->dev.match_str, a.match_len);
    put_name(a.kc, &a.kc_len, p->key_content, 64); put_name(a.kp, &a.kp_len, p->key_pattern, 64); put_name(a.kg, &a.kg_len, p->key_group, 64);
    a.key_len = p->dev.key_len; memcpy(a.key, p->dev.key, a.key_len <= sizeof(a.key) ? a.key_len : sizeof(a.key));
    a.buffer_limit = p->dev.buffer_limit; a.ts_sec = ts_sec; a.ts_nsec = ts_nsec;
    a.rows = s->o_rows.as<MloRow>(); a.st = s->o_st.as<MloState>(); a.nrec = s->o_nrec.as<uint32_t>(); a.rec_base = s->o_base.as<uint64_t>();
    a.idx = s->o_idx.as<uint32_t>(); a.misc = s->o_misc.as<unsigned int>();
    mlo_fill_groups(s, a);
    launch_mlo_extract(a, st);
    s->ph_a = a;
    if (count && n) {
        launch_mlo_kinds(a, st);
        if (hipMemcpyAsync(hmisc, s->o_misc.p, sizeof(hmisc), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail("count pass failed");
        s->ph_lines = hmisc[19]; s->ph_taken = hmisc[20];
    }
    return 0;
}

// The second half: groups, scans, records, the carried buffers -- the stream moves on.
static int mlo_phase2(flbgpu_ml_stream *s, flbgpu_dev_chunk *out, uint64_t *records) {
    auto fail = [](const char *w) -> int { set_err("multiline: %s", w); return -1; };
    hipStream_t st = s->stream;
    if (s->ph_empty) return 0;
    s->ph_empty = true;
    MloArgs a = s->ph_a;
    const uint64_t n = s->ph_n, N1 = n + 1;
    unsigned int hmisc[32];
    memset(hmisc, 0, sizeof(hmisc));
    hmisc[0] = 0xFFFFFFFFu;
    auto fill_groups = [&]() { mlo_fill_groups(s, a); };
    // (a read that fails below leaves the stream as it found it: the names this read added go again -- the carried buffers are only
    // switched at the very end anyway; ADVICE r3.  A caller that then hands the file to the CPU path first takes what the stream still
    // holds with a flush call of no bytes: flbgpu_ml_append(s, NULL, 0, ..., flush = 1).)
    struct NamesBack { std::vector<std::string> &v; size_t n; bool keep = false; ~NamesBack() { if (!keep) v.resize(n); } } names_back{s->gnames, s->gnames.size()};
    // group names the stream has not seen yet join its dictionary in the order they appear (flb_ml_stream_group_get creates them so)
    for (int round = 0;; round++) {
        launch_mlo_gid(a, st);
        if (hipMemcpyAsync(hmisc, s->o_misc.p, sizeof(hmisc), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail("group pass failed");
        if (hmisc[0] == 0xFFFFFFFFu) break;
        if (round > MLO_G || s->gnames.size() >= (size_t) MLO_G) {
            set_err("multiline: more than %d key_group values in one stream: not on the GPU path", MLO_G - 1);
            return -1;
        }
        MloRow hr;
        if (hipMemcpy(&hr, a.rows + hmisc[0], sizeof(hr), hipMemcpyDeviceToHost) != hipSuccess) return fail("group pass failed");
        if (hr.g_len > 31) return fail("a key_group value longer than 31 bytes: not on the GPU path");
        std::string nm(hr.g_len, '\0');
        if (hr.g_len && hipMemcpy(&nm[0], a.pdata + hr.g_off, hr.g_len, hipMemcpyDeviceToHost) != hipSuccess) return fail("group pass failed");
        s->gnames.push_back(nm);
        fill_groups();
        hmisc[0] = 0xFFFFFFFFu;
        if (hipMemcpyAsync(s->o_misc.p, hmisc, 4, hipMemcpyHostToDevice, st) != hipSuccess) return fail("upload failed");
    }
    launch_mlo_vscan(a, s->o_tmp.p, st);
    launch_mlo_count(a, st);
    launch_scan(a.nrec, N1, s->d_scan_tmp.as<uint64_t>(), s->o_base.as<uint64_t>(), st);
    uint64_t R = 0;
    if (hipMemcpyAsync(&R, s->o_base.as<uint64_t>() + N1, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(hmisc, s->o_misc.p, sizeof(hmisc), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail("count pass failed");
    if (hmisc[1] == 1) return fail("a line parsed with a time the event encoder refuses (the reference drops the group's record and keeps its bytes): not on the GPU path");
    if (!s->o_recs.ensure((R ? R : 1) * sizeof(MloRec)) || !s->o_size.ensure((R ? R : 1) * 4) || !s->o_off.ensure((R + 1) * 8) ||
        !s->d_scan_tmp.ensure(scan_tmp_elems(R ? R : 1) * sizeof(uint64_t))) return -1;
    a.recs = s->o_recs.as<MloRec>(); a.rec_size = s->o_size.as<uint32_t>(); a.rec_off = s->o_off.as<uint64_t>();
    if (n) launch_mlo_idx(a, st);
    launch_mlo_recs(a, st);
    launch_mlo_size(a, R, st);
    launch_scan(a.rec_size, R, s->d_scan_tmp.as<uint64_t>(), s->o_off.as<uint64_t>(), st);
    launch_mlo_carry(a, 0, st);
    uint64_t total = 0;
    if (hipMemcpyAsync(&total, s->o_off.as<uint64_t>() + R, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(hmisc, s->o_misc.p, sizeof(hmisc), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail("size pass failed");
    if (hmisc[1] == 2) return fail("the first line of a group holds key_content twice: not on the GPU path");
    if (!s->d_out.ensure(total + 16)) return -1;
    a.out = s->d_out.as<uint8_t>();
    int nxt[MLO_G];
    for (int g = 0; g < MLO_G; g++) {
        nxt[g] = s->gc_cur[g] ^ 1;
        if (!s->gc_content[g][nxt[g]].ensure((size_t) hmisc[2 + g] + 64) || !s->gc_map[g][nxt[g]].ensure((size_t) hmisc[6 + g] + 64)) return -1;
        a.carry_out_content[g] = s->gc_content[g][nxt[g]].as<uint8_t>(); a.carry_out_map[g] = s->gc_map[g][nxt[g]].as<uint8_t>();
    }
    launch_mlo_emit(a, R, st);
    launch_mlo_carry(a, 1, st);
    if (hipStreamSynchronize(st) != hipSuccess) return fail("emit pass failed");
    for (int g = 0; g < MLO_G; g++) {
        s->gc_cur[g] = nxt[g];
        s->gc_content_len[g] = hmisc[2 + g]; s->gc_map_len[g] = hmisc[6 + g]; s->gc_sec[g] = hmisc[10 + g]; s->gc_nsec[g] = hmisc[14 + g];
    }
    s->truncations += hmisc[18];
    names_back.keep = true;
    *records = R;
    out->data = s->d_out.p; out->row_off = s->o_off.as<uint64_t>(); out->n = R; out->bytes = total;
    return 0;
}

static int mlo_append_dev(flbgpu_ml_stream *s, const void *d_text, uint64_t bytes, uint32_t ts_sec, uint32_t ts_nsec, int skip_empty_lines, int flush,
                          flbgpu_dev_chunk *out, uint64_t *processed, uint64_t *records) {
    if (mlo_phase1(s, d_text, bytes, ts_sec, ts_nsec, skip_empty_lines, flush, processed, false) != 0) return -1;
    return mlo_phase2(s, out, records);
}

// what the stream carries: rule_to_state (-1 none), bytes of the open group
extern "C" void flbgpu_ml_stream_state(const flbgpu_ml_stream *s, int *rule_to_state, uint64_t *buffered) {
    if (rule_to_state) *rule_to_state = s ? (int) s->state - 1 : -1;
    if (buffered) { *buffered = s ? s->carry_len : 0; if (s && s->p->sub) for (int g = 0; g < MLO_G; g++) *buffered += s->gc_content_len[g]; }
}
// lines that truncated a buffer so far (every one is a FLB_MULTILINE_TRUNCATED return of flb_ml_append_text: in_tail warns and counts them)
extern "C" uint64_t flbgpu_ml_stream_truncations(const flbgpu_ml_stream *s) { return s ? s->truncations : 0; }

// one read of the file: text in HBM -> records in HBM.  flush != 0: the group still open afterwards leaves too (the flush timer,
// flb_ml_flush_pending :123-139).  *processed: bytes consumed (the file's buffer keeps what follows the last newline).
extern "C" int flbgpu_ml_append_dev(flbgpu_ml_stream *s, const void *d_text, uint64_t bytes, uint32_t ts_sec, uint32_t ts_nsec, int skip_empty_lines, int flush,
                                    flbgpu_dev_chunk *out, uint64_t *processed, uint64_t *records) {
    auto fail = [](const char *w) -> int { set_err("multiline: %s", w); return -1; };
    memset(out, 0, sizeof(*out));
    *processed = 0; *records = 0;
    if (!s) return fail("no stream");
    if (bytes > 0xFFFF0000ull) return fail("more than 4 GB in one call");
    if (s->p->sub) return mlo_append_dev(s, d_text, bytes, ts_sec, ts_nsec, skip_empty_lines, flush, out, processed, records);
    hipStream_t st = s->stream;
    const uint8_t *text = (const uint8_t *) d_text;
    if (!s->d_misc.ensure(sizeof(MlMisc))) return -1;
    MlMisc *dm = s->d_misc.as<MlMisc>();
    if (hipMemsetAsync(dm, 0, sizeof(MlMisc), st) != hipSuccess) return fail("memset failed");
    launch_ml_reset(dm, st);
    uint64_t nl = 0;
    MlMisc hm;
    memset(&hm, 0, sizeof(hm));
    if (bytes) {
        const size_t ntiles = tl_tiles(bytes);
        if (!s->d_masks.ensure((bytes + 63) / 64 * sizeof(uint64_t)) || !s->d_tcnt.ensure(ntiles * sizeof(uint32_t)) ||
            !s->d_toff.ensure((ntiles + 1) * sizeof(uint64_t)) || !s->d_scan_tmp.ensure(scan_tmp_elems(ntiles) * sizeof(uint64_t))) return -1;
        launch_tl_lead(text, bytes, &dm->lead, st);
        launch_tl_count(text, bytes, s->d_masks.as<uint64_t>(), s->d_tcnt.as<uint32_t>(), st);
        launch_scan(s->d_tcnt.as<uint32_t>(), ntiles, s->d_scan_tmp.as<uint64_t>(), s->d_toff.as<uint64_t>(), st);
        if (hipMemcpyAsync(&nl, s->d_toff.as<uint64_t>() + ntiles, sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail("mask pass failed");
        if (nl == 0) *processed = hm.lead;                 // no complete line yet: only the leading NULs are consumed
    }
    if (nl == 0 && (!flush || s->carry_len == 0)) return 0;
    const uint64_t NB = nl + 1;
    if (!s->d_nl.ensure((nl + 1) * sizeof(uint64_t)) || !s->d_keep.ensure(NB * 4) || !s->d_koff.ensure((NB + 1) * 8) || !s->d_ls.ensure(NB * 8) || !s->d_ll.ensure(NB * 4) ||
        !s->d_info.ensure(NB * 4) || !s->d_F.ensure(NB * 8) || !s->d_sin.ensure(NB) || !s->d_act.ensure(NB) || !s->d_c.ensure(NB * 4) || !s->d_coff.ensure((NB + 1) * 8) ||
        !s->d_head.ensure(NB * 4) || !s->d_gidx.ensure((NB + 1) * 8) || !s->d_ghead.ensure((NB + 1) * 8) || !s->d_plen.ensure(NB * 4) || !s->d_po.ensure((NB + 1) * 8) ||
        !s->d_pk.ensure(NB * 4) || !s->d_gC.ensure(NB * 4) || !s->d_ovr.ensure(NB * 4) || !s->d_slow.ensure(NB * 4) || !s->d_rows.ensure((NB + 1) * 8) || !s->d_fs_tmp.ensure(ml_fscan_tmp_bytes(NB)) ||
        !s->d_scan_tmp.ensure(scan_tmp_elems(NB) * sizeof(uint64_t)) || !s->carry[s->cur].ensure(64)) return -1;
    if (nl) launch_tl_fill(s->d_masks.as<uint64_t>(), bytes, s->d_toff.as<uint64_t>(), s->d_nl.as<uint64_t>(), st);
    MlArgs a;
    memset(&a, 0, sizeof(a));
    a.p = s->p->dev; a.rules = s->p->d_rules.as<GrepRule>();
    a.prod = s->p->d_prod.as<uint8_t>(); a.prod_bytes = s->p->prod_bytes; a.prod_nj = s->p->prod_nj; a.prod_nS = s->p->prod_nS; a.prod_T = s->p->prod_T; a.prod_init = s->p->prod_init;
    a.text = text; a.bytes = bytes; a.nl_pos = s->d_nl.as<uint64_t>(); a.nl = nl;
    a.skip_empty_lines = skip_empty_lines ? 1 : 0; a.flush_all = flush ? 1 : 0;
    a.NB = NB; a.keep = s->d_keep.as<uint32_t>(); a.koff = s->d_koff.as<uint64_t>(); a.ls = s->d_ls.as<uint64_t>(); a.ll = s->d_ll.as<uint32_t>();
    a.info = s->d_info.as<uint32_t>(); a.F = s->d_F.as<uint64_t>(); a.sin = s->d_sin.as<uint8_t>(); a.act = s->d_act.as<uint8_t>();
    a.c = s->d_c.as<uint32_t>(); a.coff = s->d_coff.as<uint64_t>(); a.head = s->d_head.as<uint32_t>(); a.gidx = s->d_gidx.as<uint64_t>();
    a.ghead = s->d_ghead.as<uint64_t>(); a.plen = s->d_plen.as<uint32_t>(); a.po = s->d_po.as<uint64_t>(); a.pk = s->d_pk.as<uint32_t>(); a.gC = s->d_gC.as<uint32_t>(); a.ovr = s->d_ovr.as<uint32_t>(); a.slow = s->d_slow.as<uint32_t>();
    a.carry = s->carry[s->cur].as<uint8_t>(); a.carry_len = s->carry_len; a.carry_tail = s->carry_tail; a.carry_state = s->state; a.carry_trunc = s->carry_trunc;
    a.carry_sec = s->carry_sec; a.carry_nsec = s->carry_nsec; a.ts_sec = ts_sec; a.ts_nsec = ts_nsec;
    // a group that is flushed before any line registered a time takes flb_time_get() (flb_ml.c:1619-1624): the clock of this call
    if (a.carry_sec == 0 && a.carry_nsec == 0) { a.carry_sec = ts_sec; a.carry_nsec = ts_nsec; }
    a.misc = dm;
    uint64_t *tmp = s->d_scan_tmp.as<uint64_t>();
    launch_ml_keep(a, st);
    launch_scan(a.keep, nl, tmp, s->d_koff.as<uint64_t>(), st);
    launch_ml_compact(a, st);
    if (hipMemsetAsync(a.ovr, 0xFF, NB * 4, st) != hipSuccess) return fail("memset failed");
    launch_ml_match(a, flbgpu_device_cus(), st);
    uint64_t total = 0, groups = 0, last_nl = 0;
    const bool may_truncate = a.p.type == ML_REGEX && a.p.buffer_limit > 0;
    for (uint64_t round = 0;; round++) {
        // a continuation that does not fit its group's buffer resets rule_to_state: the lines behind it are met in another state.
        // Such lines are pinned one at a time, the first first; a call without one (the rule) runs this body once
        launch_ml_reset(dm, st);
        launch_ml_fscan(a.F, NB, a.p.type == ML_REGEX ? s->state : (uint32_t) MLT_EMPTY, a.sin, s->d_fs_tmp.p, &dm->final_state, st);
        launch_ml_act(a, st);
        launch_scan(a.c, NB, tmp, s->d_coff.as<uint64_t>(), st);
        launch_scan(a.head, NB, tmp, s->d_gidx.as<uint64_t>(), st);
        launch_ml_ghead(a, st);
        if (may_truncate) launch_ml_trunc(a, st);
        launch_ml_piece(a, st);
        launch_scan(a.plen, NB, tmp, s->d_po.as<uint64_t>(), st);
        if (hipMemcpyAsync(&total, s->d_po.as<uint64_t>() + NB, sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(&groups, s->d_gidx.as<uint64_t>() + NB, sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
            (nl && hipMemcpyAsync(&last_nl, s->d_nl.as<uint64_t>() + (nl - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, st) != hipSuccess) ||
            hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail("size pass failed");
        if (hm.trunc_k == ~0ull) break;
        // One truncating continuation is settled per round (the lines behind it are met in another state, so what overflows behind it is
        // only known after the pass), each round a full pass + a wait: a read with K of them costs K passes.  Reads are bounded -- past
        // ML_TRUNC_ROUNDS the call fails like a read the list path turns down, and the caller keeps THIS read on the CPU's flb_ml (a
        // buffer_limit so small that thousands of lines of one read overflow it is a configuration, not a log burst) -- ADVICE r3.
        if (round >= ML_TRUNC_ROUNDS || round > nl) {
            set_err("multiline: more than %d lines of one read overflow buffer_limit (%llu bytes): this read is not taken (keep it on the CPU path)", ML_TRUNC_ROUNDS, (unsigned long long) a.p.buffer_limit);
            return -1;
        }
        launch_ml_override(a, hm.trunc_k, st);
    }
    if (hm.refused) { set_err("multiline: a group of this buffer holds more than 4 GB"); return -1; }
    const int nxt = s->cur ^ 1;
    if (!s->d_out.ensure(total + 16) || !s->carry[nxt].ensure((size_t) hm.new_carry_len + 64)) return -1;
    a.out = s->d_out.as<uint8_t>(); a.carry_out = s->carry[nxt].as<uint8_t>();
    if (total > 0) launch_ml_emit(a, flbgpu_device_cus(), st);
    launch_ml_rows(a, s->d_rows.as<uint64_t>(), st);
    launch_ml_carry(a, st);
    if (hipStreamSynchronize(st) != hipSuccess) return fail("emit pass failed");
    // the stream's state after the call
    if (a.p.type == ML_REGEX) s->state = hm.final_state;
    s->cur = nxt;
    s->carry_len = hm.has_open ? hm.new_carry_len : 0;
    s->carry_tail = hm.has_open ? hm.new_tail : (uint32_t) MLT_EMPTY;
    s->carry_trunc = hm.has_open ? hm.new_carry_trunc : 0;
    s->truncations += hm.truncated;
    if (hm.first_reg != 0xFFFFFFFFu) { s->carry_sec = ts_sec; s->carry_nsec = ts_nsec; }
    if (nl) *processed = last_nl + 1;
    *records = hm.records;
    out->data = s->d_out.p; out->row_off = s->d_rows.as<uint64_t>(); out->n = groups; out->bytes = total;
    return 0;
}

// the same on a host buffer; *out_buf is malloc()'d (nullptr when nothing was flushed)
extern "C" int flbgpu_ml_append(flbgpu_ml_stream *s, const void *text, size_t bytes, uint32_t ts_sec, uint32_t ts_nsec, int skip_empty_lines, int flush,
                                void **out_buf, size_t *out_size, uint64_t *processed, uint64_t *records) {
    *out_buf = nullptr; *out_size = 0;
    if (!s) { set_err("multiline: no stream"); return -1; }
    if (bytes && (!s->d_in.ensure(bytes + 16) || hipMemcpyAsync(s->d_in.p, text, bytes, hipMemcpyHostToDevice, s->stream) != hipSuccess)) { set_err("multiline: upload failed"); return -1; }
    flbgpu_dev_chunk out;
    if (flbgpu_ml_append_dev(s, s->d_in.p, bytes, ts_sec, ts_nsec, skip_empty_lines, flush, &out, processed, records) != 0) return -1;
    if (out.bytes == 0) return 0;
    void *hb = malloc(out.bytes);
    if (!hb) { set_err("out of memory"); return -1; }
    if (hipMemcpy(hb, out.data, out.bytes, hipMemcpyDeviceToHost) != hipSuccess) { free(hb); set_err("multiline: download failed"); return -1; }
    *out_buf = hb; *out_size = out.bytes;
    return 0;
}

// ---- a list of parsers on one stream: in_tail's `multiline.parser docker, cri` (plugins/in_tail/tail_config.c builds one instance per name).
// flb_ml_append_text (src/multiline/flb_ml.c:671-760) offers every line to the parser that took the stream's last line, then to the others
// in order; a line nobody takes flushes every parser's groups and leaves alone through the FIRST parser's default group.  Taking a line is
// stateless for parsers with a parser in front (the parse succeeds, the content key is there): the first half of a read says per parser
// how many lines it takes.  What runs here is the case files are made of -- ONE parser takes every line that is taken at all in a read
// (then the list behaves exactly like that parser alone: the others only ever see lines they turn down, and the lone lines leave with the
// same key) -- and the call FAILS for a read whose lines split between parsers, or that changes parsers while a group is open: the caller
// keeps those on the CPU.
struct flbgpu_ml_list {
    std::vector<flbgpu_ml_stream *> s;
    int lru = -1;
    DevBuf d_in;
};

extern "C" flbgpu_ml_list *flbgpu_ml_list_create(flbgpu_ml_stream **streams, int n) {
    if (!streams || n < 1 || n > 8) { set_err("multiline list: 1 to 8 streams"); return nullptr; }
    for (int i = 0; i < n; i++) {
        if (!streams[i] || !streams[i]->p->sub) { set_err("multiline list: every parser of a list needs a parser in front (docker, cri, or flbgpu_ml_parser_set_subparser)"); return nullptr; }
        if (streams[i]->p->key_content != streams[0]->p->key_content) { set_err("multiline list: the parsers of a list must share key_content"); return nullptr; }
    }
    auto *l = new flbgpu_ml_list();
    l->s.assign(streams, streams + n);
    return l;
}
extern "C" void flbgpu_ml_list_destroy(flbgpu_ml_list *l) { if (l) { l->d_in.release(); delete l; } }
extern "C" int flbgpu_ml_list_lru(const flbgpu_ml_list *l) { return l ? l->lru : -1; }

extern "C" int flbgpu_ml_list_append_dev(flbgpu_ml_list *l, const void *d_text, uint64_t bytes, uint32_t ts_sec, uint32_t ts_nsec, int skip_empty_lines, int flush,
                                         flbgpu_dev_chunk *out, uint64_t *processed, uint64_t *records) {
    memset(out, 0, sizeof(*out));
    *processed = 0; *records = 0;
    if (!l) { set_err("multiline list: no list"); return -1; }
    if (bytes > 0xFFFF0000ull) { set_err("multiline: more than 4 GB in one call"); return -1; }
    const int n = (int) l->s.size();
    auto pending = [](const flbgpu_ml_stream *s) { for (int g = 0; g < MLO_G; g++) if (s->gc_map_len[g] > 0 || s->gc_content_len[g] > 0) return true; return false; };
    auto drop_phases = [&]() { for (auto *s : l->s) s->ph_empty = true; };
    const int A = l->lru >= 0 ? l->lru : 0;
    if (mlo_phase1(l->s[(size_t) A], d_text, bytes, ts_sec, ts_nsec, skip_empty_lines, flush, processed, true) != 0) { drop_phases(); return -1; }
    int chosen = A;
    const uint64_t lines = l->s[(size_t) A]->ph_lines, taken_a = l->s[(size_t) A]->ph_taken;
    if (taken_a < lines) {
        // lines the first parser in turn leaves to the others: does anybody take one?
        int first_other = -1, others = 0;
        for (int j = 0; j < n; j++) {
            if (j == A) continue;
            uint64_t pr = 0;
            if (mlo_phase1(l->s[(size_t) j], d_text, bytes, ts_sec, ts_nsec, skip_empty_lines, flush, &pr, true) != 0) { drop_phases(); return -1; }
            if (l->s[(size_t) j]->ph_taken > 0) { others++; if (first_other < 0) first_other = j; }
        }
        if (first_other >= 0) {
            const uint64_t taken_b = l->s[(size_t) first_other]->ph_taken;
            if (taken_a > 0 || (others > 1 && taken_b < lines)) {
                drop_phases();
                set_err("multiline list: the lines of one read split between parsers (%llu of %llu for parser %d, parser %d takes others): not on the GPU path",
                        (unsigned long long) taken_a, (unsigned long long) lines, A, first_other);
                return -1;
            }
            for (int j = 0; j < n; j++)
                if (j != first_other && pending(l->s[(size_t) j])) { drop_phases(); set_err("multiline list: the stream changes parsers while a group of parser %d is open: not on the GPU path", j); return -1; }
            chosen = first_other;
        }
    }
    flbgpu_ml_stream *sc = l->s[(size_t) chosen];
    const uint64_t taken_c = sc->ph_taken;
    for (int j = 0; j < n; j++) if (j != chosen) l->s[(size_t) j]->ph_empty = true;
    if (mlo_phase2(sc, out, records) != 0) { drop_phases(); return -1; }
    if (taken_c > 0) l->lru = chosen;
    return 0;
}

extern "C" int flbgpu_ml_list_append(flbgpu_ml_list *l, const void *text, size_t bytes, uint32_t ts_sec, uint32_t ts_nsec, int skip_empty_lines, int flush,
                                     void **out_buf, size_t *out_size, uint64_t *processed, uint64_t *records) {
    *out_buf = nullptr; *out_size = 0;
    if (!l) { set_err("multiline list: no list"); return -1; }
    // (uploaded with a blocking copy: every parser's stream reads it)
    if (bytes && (!l->d_in.ensure(bytes + 16) || hipMemcpy(l->d_in.p, text, bytes, hipMemcpyHostToDevice) != hipSuccess)) { set_err("multiline: upload failed"); return -1; }
    flbgpu_dev_chunk out;
    if (flbgpu_ml_list_append_dev(l, l->d_in.p, bytes, ts_sec, ts_nsec, skip_empty_lines, flush, &out, processed, records) != 0) return -1;
    if (out.bytes == 0) return 0;
    void *hb = malloc(out.bytes);
    if (!hb) { set_err("out of memory"); return -1; }
    if (hipMemcpy(hb, out.data, out.bytes, hipMemcpyDeviceToHost) != hipSuccess) { free(hb); set_err("multiline: download failed"); return -1; }
    *out_buf = hb; *out_size = out.bytes;
    return 0;
}
