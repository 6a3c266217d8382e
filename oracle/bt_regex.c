/*
 * oracle/bt_regex.c -- TEST INFRASTRUCTURE ONLY.  See bt_regex.h for provenance.
 *
 * A compact Perl-syntax backtracking matcher that restates the published behaviour of Boost.Regex's
 * perl_matcher as used by the reference at core/common/StringTools.cpp:183-211:
 *   - leftmost-first ("first alternative that leads to an overall match wins"), greedy quantifiers
 *     unless suffixed with '?', byte-oriented, classic-locale \w \d \s;
 *   - regex_match: MATCH is only accepted at end of input, otherwise the matcher backtracks;
 *   - '.' matches every byte (mod_s default), '^'/'$' also match at embedded line separators
 *     (mod_m default; separators are \n \r \f and never between \r\n);
 *   - single-set repeats use a counted fast path like perl_matcher::match_set_repeat /
 *     match_char_repeat (one backtrack record per repeat, not per character);
 *   - a repeat whose body last matched the empty string stops iterating (match_rep's null check);
 *   - a state-count budget; exceeding it returns -1 (boost throws std::runtime_error, which the
 *     reference swallows into "parse failed", StringTools.cpp:200-205).
 *   - back-references \1 .. \N (round 6; perl_matcher::match_backref): the bytes group N matched last, again; a group that took
 *     no part fails the reference; under ORX_ICASE the comparison folds ASCII case.  Pinned against CPython's `re` on the vectors of
 *     tests/golden/backref_vectors.json (tools/gen_backref_golden.py); boost itself is on neither box.
 *   - conditionals on a group (?(N)yes|no) (round 6).
 * Unsupported (compile error): look-behind bodies of variable length, recursion, conditions on names / look-arounds, \p{..}, collating elements.
 */
#include "bt_regex.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ char sets */
typedef struct { uint32_t b[8]; } cset;
static void cs_clear(cset* s) { memset(s, 0, sizeof *s); }
static void cs_add(cset* s, int c) { s->b[(c >> 5) & 7] |= 1u << (c & 31); }
static int cs_has(const cset* s, int c) { return (s->b[c >> 5] >> (c & 31)) & 1u; }
static void cs_range(cset* s, int lo, int hi) { for (int c = lo; c <= hi; ++c) cs_add(s, c); }
static void cs_or(cset* d, const cset* s) { for (int i = 0; i < 8; ++i) d->b[i] |= s->b[i]; }
static void cs_not(cset* s) { for (int i = 0; i < 8; ++i) s->b[i] = ~s->b[i]; }
static void cs_fold_case(cset* s) {
    for (int c = 'a'; c <= 'z'; ++c) {
        int u = c - 32;
        if (cs_has(s, c) || cs_has(s, u)) { cs_add(s, c); cs_add(s, u); }
    }
}
static int is_word(int c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; }
static int is_sep(int c) { return c == '\n' || c == '\r' || c == '\f'; }

/* shorthand letters per dialect: regexp2 (RE2 option) knows \d \w \s only, and its \s has no \v */
#define SH_LOWER(P) (((P)->flags & ORX_REGEXP2) ? "dws" : "dwshvlu")
#define SH_UPPER(P) (((P)->flags & ORX_REGEXP2) ? "DWS" : "DWSHVLU")
#define SH_KIND(P, e) ((((P)->flags & ORX_REGEXP2) && (e) == 's') ? 'r' : (e))
static void cs_named(cset* s, int kind) { /* kind: 'd','w','s','h','v','l','u' */
    switch (kind) {
        case 'd': cs_range(s, '0', '9'); break;
        case 'w': cs_range(s, '0', '9'); cs_range(s, 'a', 'z'); cs_range(s, 'A', 'Z'); cs_add(s, '_'); break;
        case 's': cs_add(s, ' '); cs_range(s, 9, 13); break;
        case 'r': cs_add(s, ' '); cs_add(s, '\t'); cs_add(s, '\n'); cs_add(s, '\f'); cs_add(s, '\r'); break; /* RE2's \s */
        case 'h': cs_add(s, ' '); cs_add(s, '\t'); break;
        case 'v': cs_range(s, 10, 13); break;
        case 'l': cs_range(s, 'a', 'z'); break;
        case 'u': cs_range(s, 'A', 'Z'); break;
    }
}
static int cs_posix(cset* s, const char* name, size_t n) {
#define IS(x) (n == strlen(x) && memcmp(name, x, n) == 0)
    if (IS("alpha")) { cs_range(s, 'a', 'z'); cs_range(s, 'A', 'Z'); }
    else if (IS("digit") || IS("d")) cs_range(s, '0', '9');
    else if (IS("alnum")) { cs_range(s, 'a', 'z'); cs_range(s, 'A', 'Z'); cs_range(s, '0', '9'); }
    else if (IS("upper") || IS("u")) cs_range(s, 'A', 'Z');
    else if (IS("lower") || IS("l")) cs_range(s, 'a', 'z');
    else if (IS("space") || IS("s")) { cs_add(s, ' '); cs_range(s, 9, 13); }
    else if (IS("blank")) { cs_add(s, ' '); cs_add(s, '\t'); }
    else if (IS("punct")) { cs_range(s, 33, 47); cs_range(s, 58, 64); cs_range(s, 91, 96); cs_range(s, 123, 126); }
    else if (IS("print")) cs_range(s, 32, 126);
    else if (IS("graph")) cs_range(s, 33, 126);
    else if (IS("cntrl")) { cs_range(s, 0, 31); cs_add(s, 127); }
    else if (IS("xdigit")) { cs_range(s, '0', '9'); cs_range(s, 'a', 'f'); cs_range(s, 'A', 'F'); }
    else if (IS("word") || IS("w")) cs_named(s, 'w');
    else return 0;
#undef IS
    return 1;
}

/* ------------------------------------------------------------------ AST */
enum { N_EMPTY, N_SET, N_CAT, N_ALT, N_REP, N_GROUP, N_ASSERT, N_ATOMIC,
       N_LOOKAHEAD /* general (?=X) (?!X): max = negative; min = k > 0: the look-BEHIND (?<=X) (?<!X) of a body of fixed length k */,
       N_BACKREF /* cap = the group referred to */, N_COND /* (?(cap)l|r) */ };
enum {
    A_BOL_ML, A_BOL_SL, A_EOL_ML, A_EOL_SL, A_BUF_START, A_BUF_END, A_BUF_END_NL,
    A_WORDB, A_NWORDB, A_WORD_START, A_WORD_END,
    A_LOOK /* one-byte look-around: node.set = class, node.min = behind, node.max = negative */
};
typedef struct {
    int kind;
    int l, r;          /* children (CAT/ALT: l,r ; REP/GROUP: l) */
    int min, max;      /* REP; max<0 = infinity */
    int greedy;        /* REP */
    int cap;           /* GROUP: capture index, 0 = non-capturing */
    int akind;         /* ASSERT */
    int set;           /* SET: index into sets */
} node;

#define ORX_MAX_GROUPS 1023
enum { I_SET, I_SPLIT, I_JMP, I_SAVE, I_ASSERT, I_MATCH, I_MARK, I_CHK, I_REPSET, I_ATOM_BEGIN, I_ATOM_END, I_LOOK_BEGIN, I_LOOK_END, I_NLOOK_BEGIN, I_NLOOK_END,
       I_BACK /* x = k: step back k bytes (inside a look-behind, right behind its mark); fewer than k behind: backtrack */,
       I_BACKREF /* x = group */, I_COND /* x = group, y = pc of the "no" branch */ };
typedef struct { int op, x, y, z, w; } inst;
/* I_REPSET: x=set, y=min, z=max(-1 inf), w=greedy */

struct orx_prog {
    inst* code; int ncode, capcode;
    cset* sets; int nsets, capsets;
    node* nodes; int nnodes, capnodes;
    int ngroups;
    int nloopregs;
    int maxbackref;
    char* names[ORX_MAX_GROUPS + 1];
    /* parser state */
    const unsigned char* p; size_t n, i;
    unsigned flags;
    char err[256];
    int failed;
};

static void fail(orx_prog* P, const char* msg) {
    if (!P->failed) { snprintf(P->err, sizeof P->err, "%s at offset %zu", msg, P->i); P->failed = 1; }
}
static int new_node(orx_prog* P, int kind) {
    if (P->nnodes == P->capnodes) {
        P->capnodes = P->capnodes ? P->capnodes * 2 : 64;
        P->nodes = (node*)realloc(P->nodes, sizeof(node) * P->capnodes);
    }
    node* nd = &P->nodes[P->nnodes];
    memset(nd, 0, sizeof *nd);
    nd->kind = kind; nd->l = nd->r = -1;
    return P->nnodes++;
}
static int new_set(orx_prog* P, const cset* s) {
    if (P->nsets == P->capsets) {
        P->capsets = P->capsets ? P->capsets * 2 : 32;
        P->sets = (cset*)realloc(P->sets, sizeof(cset) * P->capsets);
    }
    P->sets[P->nsets] = *s;
    return P->nsets++;
}
static int set_node(orx_prog* P, const cset* s) {
    int n = new_node(P, N_SET);
    P->nodes[n].set = new_set(P, s);
    return n;
}
static int lit_node(orx_prog* P, int c) {
    cset s; cs_clear(&s); cs_add(&s, c);
    if (P->flags & ORX_ICASE) cs_fold_case(&s);
    return set_node(P, &s);
}
static int cat2(orx_prog* P, int a, int b) {
    if (a < 0) return b;
    if (b < 0) return a;
    int n = new_node(P, N_CAT);
    P->nodes[n].l = a; P->nodes[n].r = b;
    return n;
}

static int parse_alt(orx_prog* P, int depth);

static int hexval(int c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}
/* Parses the part of an escape that denotes a single byte; P->i is just after the escape letter `c`.
 * Returns byte value or -1 if `c` is not a single-byte escape. */
static int escape_byte(orx_prog* P, int c) {
    switch (c) {
        case 't': return '\t';
        case 'n': return '\n';
        case 'r': return '\r';
        case 'f': return '\f';
        case 'e': return 27;
        case 'a': return 7;
        case 'x': {
            if (P->i < P->n && P->p[P->i] == '{') {
                size_t j = P->i + 1; int v = 0, nd = 0;
                while (j < P->n && hexval(P->p[j]) >= 0) { v = v * 16 + hexval(P->p[j]); ++j; ++nd; if (v > 255) break; }
                if (nd == 0 || j >= P->n || P->p[j] != '}' || v > 255) { fail(P, "bad \\x{..} escape"); return 0; }
                P->i = j + 1;
                return v;
            }
            int v = 0, nd = 0;
            while (nd < 2 && P->i < P->n && hexval(P->p[P->i]) >= 0) { v = v * 16 + hexval(P->p[P->i]); ++P->i; ++nd; }
            if (nd == 0) { fail(P, "bad \\x escape"); return 0; }
            return v;
        }
        case '0': {
            int v = 0, nd = 0;
            while (nd < 3 && P->i < P->n && P->p[P->i] >= '0' && P->p[P->i] <= '7') { v = v * 8 + (P->p[P->i] - '0'); ++P->i; ++nd; }
            if (v > 255) { fail(P, "octal escape out of range"); return 0; }
            return v;
        }
        case 'c': {
            if (P->i >= P->n) { fail(P, "bad \\c escape"); return 0; }
            int v = P->p[P->i++];
            return v % 32;
        }
        default: return -1;
    }
}

static int parse_class(orx_prog* P) { /* P->i just after '[' */
    cset s; cs_clear(&s);
    int negate = 0, first = 1;
    if (P->i < P->n && P->p[P->i] == '^') { negate = 1; ++P->i; }
    for (;;) {
        if (P->i >= P->n) { fail(P, "unterminated character class"); return -1; }
        int c = P->p[P->i];
        if (c == ']' && !first) { ++P->i; break; }
        first = 0;
        int lo = -1; /* single-char endpoint, or -1 if a class was added */
        if (c == '[' && P->i + 1 < P->n && (P->p[P->i + 1] == ':' || P->p[P->i + 1] == '=' || P->p[P->i + 1] == '.')) {
            int kind = P->p[P->i + 1];
            size_t j = P->i + 2;
            while (j + 1 < P->n && !(P->p[j] == kind && P->p[j + 1] == ']')) ++j;
            if (j + 1 >= P->n) { fail(P, "unterminated [: :] in class"); return -1; }
            if (kind != ':') { fail(P, "collating elements unsupported"); return -1; }
            const char* nm = (const char*)P->p + P->i + 2; size_t nl = j - (P->i + 2);
            int neg = 0;
            if (nl > 0 && nm[0] == '^') { neg = 1; ++nm; --nl; }
            cset t; cs_clear(&t);
            if (!cs_posix(&t, nm, nl)) { fail(P, "unknown POSIX class"); return -1; }
            if (neg) cs_not(&t);
            cs_or(&s, &t);
            P->i = j + 2;
            continue;
        } else if (c == '\\') {
            ++P->i;
            if (P->i >= P->n) { fail(P, "trailing backslash"); return -1; }
            int e = P->p[P->i++];
            if (e && strchr(SH_LOWER(P), e)) { cset t; cs_clear(&t); cs_named(&t, SH_KIND(P, e)); cs_or(&s, &t); continue; }
            if (e && strchr(SH_UPPER(P), e)) { cset t; cs_clear(&t); cs_named(&t, SH_KIND(P, e + 32)); cs_not(&t); cs_or(&s, &t); continue; }
            if (e == 'b') lo = 8;
            else {
                int v = escape_byte(P, e);
                if (P->failed) return -1;
                lo = v >= 0 ? v : e;
            }
        } else {
            lo = c; ++P->i;
        }
        /* range? */
        if (P->i + 1 < P->n && P->p[P->i] == '-' && P->p[P->i + 1] != ']') {
            size_t save = P->i;
            ++P->i;
            int hi;
            int c2 = P->p[P->i];
            if (c2 == '[' && P->i + 1 < P->n && P->p[P->i + 1] == ':') { P->i = save; cs_add(&s, lo); continue; }
            if (c2 == '\\') {
                ++P->i;
                if (P->i >= P->n) { fail(P, "trailing backslash"); return -1; }
                int e = P->p[P->i++];
                if (e && strchr("dwshvluDWSHVLU", e)) { fail(P, "class escape as range endpoint"); return -1; }
                if (e == 'b') hi = 8;
                else { int v = escape_byte(P, e); if (P->failed) return -1; hi = v >= 0 ? v : e; }
            } else { hi = c2; ++P->i; }
            if (hi < lo) { fail(P, "invalid range in character class"); return -1; }
            cs_range(&s, lo, hi);
        } else {
            cs_add(&s, lo);
        }
    }
    if (P->flags & ORX_ICASE) cs_fold_case(&s);
    if (negate) cs_not(&s);
    return set_node(P, &s);
}

static void skip_extended(orx_prog* P) {
    if (!(P->flags & ORX_EXTENDED)) return;
    for (;;) {
        while (P->i < P->n && strchr(" \t\n\r\f\v", P->p[P->i]) && P->p[P->i] != 0) ++P->i;
        if (P->i < P->n && P->p[P->i] == '#') { while (P->i < P->n && P->p[P->i] != '\n') ++P->i; continue; }
        break;
    }
}

static int assert_node(orx_prog* P, int kind) {
    int n = new_node(P, N_ASSERT);
    P->nodes[n].akind = kind;
    return n;
}

/* length of every match of node n if that is one number (a look-behind body must have one: Perl, PCRE, regexp2 and boost agree on
 * "fixed length" for the bodies the log patterns use), else -1 */
static int fixed_len(const orx_prog* P, int n) {
    const node* nd = &P->nodes[n];
    switch (nd->kind) {
        case N_EMPTY: case N_ASSERT: case N_LOOKAHEAD: return 0;
        case N_SET: return 1;
        case N_CAT: { int a = fixed_len(P, nd->l), b = fixed_len(P, nd->r); return (a < 0 || b < 0) ? -1 : a + b; }
        case N_ALT: { int a = fixed_len(P, nd->l), b = fixed_len(P, nd->r); return (a < 0 || a != b) ? -1 : a; }
        case N_REP: { int a = fixed_len(P, nd->l); return (a < 0 || nd->min != nd->max) ? -1 : a * nd->min; }
        case N_GROUP: case N_ATOMIC: return fixed_len(P, nd->l);
        case N_COND: { int a = fixed_len(P, nd->l), b = fixed_len(P, nd->r); return (a < 0 || a != b) ? -1 : a; }
    }
    return -1;
}

/* (?=X) (?!X) (?<=X) (?<!X) with X a single character class (one-byte look-around); P->i is past the introducer */
static int parse_lookaround(orx_prog* P, int depth, int behind, int negative) {
    unsigned saved = P->flags;
    int inner = parse_alt(P, depth + 1);
    if (P->failed) return -1;
    if (P->i >= P->n || P->p[P->i] != ')') { fail(P, "missing )"); return -1; }
    ++P->i;
    P->flags = saved;
    while (P->nodes[inner].kind == N_GROUP && P->nodes[inner].cap == 0) inner = P->nodes[inner].l;
    if (P->nodes[inner].kind != N_SET) {
        /* a general look-AHEAD is just a sub-match the backtracker runs in place (atomic, zero-width) */
        int back = 0;
        if (behind) { /* a body of fixed length k: the same sub-match, started k bytes back (it then ends where the assertion stands) */
            back = fixed_len(P, inner);
            if (back <= 0) { fail(P, "unsupported: look-behind body must have a fixed, non-zero length"); return -1; }
        }
        int la = new_node(P, N_LOOKAHEAD);
        P->nodes[la].l = inner;
        P->nodes[la].max = negative;
        P->nodes[la].min = back;
        return la;
    }
    int n = new_node(P, N_ASSERT);
    P->nodes[n].akind = A_LOOK;
    P->nodes[n].set = P->nodes[inner].set;
    P->nodes[n].min = behind;
    P->nodes[n].max = negative;
    return n;
}

/* returns node index, -1 for "no atom" (caller decides), sets *is_assert */
static int parse_atom(orx_prog* P, int depth, int* is_assert) {
    *is_assert = 0;
    int c = P->p[P->i];
    switch (c) {
        case '(': {
            ++P->i;
            unsigned saved = P->flags;
            int cap = 0; char* name = NULL;
            if (P->i < P->n && P->p[P->i] == '?') {
                ++P->i;
                if (P->i >= P->n) { fail(P, "unterminated group"); return -1; }
                int d = P->p[P->i];
                if (d == '#') {
                    while (P->i < P->n && P->p[P->i] != ')') ++P->i;
                    if (P->i >= P->n) { fail(P, "unterminated comment"); return -1; }
                    ++P->i;
                    return new_node(P, N_EMPTY);
                } else if (d == ':') {
                    ++P->i;
                } else if (d == '=' || d == '!') {
                    ++P->i;
                    return parse_lookaround(P, depth, 0, d == '!');
                } else if (d == '>') { /* atomic group */
                    ++P->i;
                    int inner = parse_alt(P, depth + 1);
                    if (P->failed) return -1;
                    if (P->i >= P->n || P->p[P->i] != ')') { fail(P, "missing )"); return -1; }
                    ++P->i;
                    P->flags = saved;
                    int a = new_node(P, N_ATOMIC);
                    P->nodes[a].l = inner;
                    return a;
                } else if (d == '(' && P->i + 1 < P->n && P->p[P->i + 1] >= '1' && P->p[P->i + 1] <= '9' && !(P->flags & ORX_REGEXP2)) {
                    /* (?(N)yes|no): yes when group N has taken part (perl_matcher::match_assert_backref's sub-expression test) */
                    ++P->i;
                    int v = 0;
                    while (P->i < P->n && P->p[P->i] >= '0' && P->p[P->i] <= '9' && v < 1000) v = v * 10 + (P->p[P->i++] - '0');
                    if (P->i >= P->n || P->p[P->i] != ')') { fail(P, "unsupported group construct (recursion/conditional)"); return -1; }
                    ++P->i;
                    int inner = parse_alt(P, depth + 1);
                    if (P->failed) return -1;
                    if (P->i >= P->n || P->p[P->i] != ')') { fail(P, "missing )"); return -1; }
                    ++P->i;
                    P->flags = saved;
                    int c = new_node(P, N_COND);
                    P->nodes[c].cap = v;
                    if (P->nodes[inner].kind == N_ALT) {
                        int yes = P->nodes[inner].l, no = P->nodes[inner].r;
                        if (P->nodes[yes].kind == N_ALT || P->nodes[no].kind == N_ALT) { fail(P, "a conditional has at most two alternatives"); return -1; }
                        P->nodes[c].l = yes; P->nodes[c].r = no;
                    } else {
                        P->nodes[c].l = inner; P->nodes[c].r = new_node(P, N_EMPTY);
                    }
                    if (v > P->maxbackref) P->maxbackref = v;
                    return c;
                } else if (d == '|' || d == '(' || d == 'R' || d == '&' || (d >= '0' && d <= '9') || d == '+') {
                    fail(P, "unsupported group construct (recursion/conditional)");
                    return -1;
                } else if (d == '<' || d == 'P' || d == '\'') {
                    int close = '>';
                    if (d == 'P') {
                        ++P->i;
                        if (P->i >= P->n || P->p[P->i] != '<') { fail(P, "unsupported (?P construct"); return -1; }
                    } else if (d == '\'') close = '\'';
                    else if (P->i + 1 < P->n && (P->p[P->i + 1] == '=' || P->p[P->i + 1] == '!')) {
                        int negative = P->p[P->i + 1] == '!';
                        P->i += 2;
                        return parse_lookaround(P, depth, 1, negative);
                    }
                    ++P->i;
                    size_t j = P->i;
                    while (j < P->n && P->p[j] != (unsigned char)close) ++j;
                    if (j >= P->n || j == P->i) { fail(P, "bad group name"); return -1; }
                    name = (char*)malloc(j - P->i + 1);
                    memcpy(name, P->p + P->i, j - P->i); name[j - P->i] = 0;
                    P->i = j + 1;
                    cap = -1; /* allocate below */
                } else {
                    /* inline flags */
                    int on = 1; unsigned f = P->flags;
                    for (;; ++P->i) {
                        if (P->i >= P->n) { fail(P, "unterminated flag group"); return -1; }
                        d = P->p[P->i];
                        if (d == '-') { on = 0; continue; }
                        unsigned bit = 0; int invert = 0;
                        if (d == 'i') bit = ORX_ICASE;
                        else if (d == 's') { bit = ORX_NO_MOD_S; invert = 1; }
                        else if (d == 'm') { bit = ORX_NO_MOD_M; invert = 1; }
                        else if (d == 'x') bit = ORX_EXTENDED;
                        else break;
                        if (on ^ invert) f |= bit; else f &= ~bit;
                    }
                    if (d == ')') { ++P->i; P->flags = f; return new_node(P, N_EMPTY); }
                    if (d != ':') { fail(P, "unknown inline flag"); return -1; }
                    ++P->i;
                    P->flags = f;
                    int inner = parse_alt(P, depth + 1);
                    if (P->failed) return -1;
                    if (P->i >= P->n || P->p[P->i] != ')') { fail(P, "missing )"); return -1; }
                    ++P->i;
                    P->flags = saved;
                    int g = new_node(P, N_GROUP);
                    P->nodes[g].l = inner; P->nodes[g].cap = 0;
                    return g;
                }
            } else {
                cap = -1;
            }
            if (cap == -1) {
                if (P->ngroups >= ORX_MAX_GROUPS) { fail(P, "too many capture groups"); free(name); return -1; }
                cap = ++P->ngroups;
                P->names[cap] = name;
            }
            int inner = parse_alt(P, depth + 1);
            if (P->failed) return -1;
            if (P->i >= P->n || P->p[P->i] != ')') { fail(P, "missing )"); return -1; }
            ++P->i;
            P->flags = saved;
            int g = new_node(P, N_GROUP);
            P->nodes[g].l = inner; P->nodes[g].cap = cap;
            return g;
        }
        case '[': ++P->i; return parse_class(P);
        case '.': {
            ++P->i;
            cset s; cs_clear(&s); cs_not(&s);
            if (P->flags & ORX_NO_MOD_S) { s.b['\n' >> 5] &= ~(1u << ('\n' & 31)); }
            return set_node(P, &s);
        }
        case '^': ++P->i; *is_assert = 1; return assert_node(P, (P->flags & ORX_NO_MOD_M) ? A_BOL_SL : A_BOL_ML);
        case '$': ++P->i; *is_assert = 1; return assert_node(P, (P->flags & ORX_NO_MOD_M) ? A_EOL_SL : A_EOL_ML);
        case '*': case '+': case '?': fail(P, "nothing to repeat"); return -1;
        case '\\': {
            ++P->i;
            if (P->i >= P->n) { fail(P, "trailing backslash"); return -1; }
            int e = P->p[P->i++];
            if (e && strchr(SH_LOWER(P), e)) { cset s; cs_clear(&s); cs_named(&s, SH_KIND(P, e)); if (P->flags & ORX_ICASE) cs_fold_case(&s); return set_node(P, &s); }
            if (e && strchr(SH_UPPER(P), e)) { cset s; cs_clear(&s); cs_named(&s, SH_KIND(P, e + 32)); if (P->flags & ORX_ICASE) cs_fold_case(&s); cs_not(&s); return set_node(P, &s); }
            if ((P->flags & ORX_REGEXP2) && e && strchr("<>`'", e)) return lit_node(P, e);   /* unknown escapes are the literal */
            if ((P->flags & ORX_REGEXP2) && e == 'u') { fail(P, "unsupported escape"); return -1; }
            switch (e) {
                case 'b': *is_assert = 1; return assert_node(P, A_WORDB);
                case 'B': *is_assert = 1; return assert_node(P, A_NWORDB);
                case '<': *is_assert = 1; return assert_node(P, A_WORD_START);
                case '>': *is_assert = 1; return assert_node(P, A_WORD_END);
                case 'A': case '`': *is_assert = 1; return assert_node(P, A_BUF_START);
                case 'z': case '\'': *is_assert = 1; return assert_node(P, A_BUF_END);
                case 'Z': *is_assert = 1; return assert_node(P, A_BUF_END_NL);
                case 'Q': {
                    int acc = new_node(P, N_EMPTY);
                    while (P->i < P->n) {
                        if (P->p[P->i] == '\\' && P->i + 1 < P->n && P->p[P->i + 1] == 'E') { P->i += 2; break; }
                        acc = cat2(P, acc, lit_node(P, P->p[P->i++]));
                    }
                    int g = new_node(P, N_GROUP); /* keep \Q..\E as one unit for a following quantifier? Perl applies it to last char; keep simple: group */
                    P->nodes[g].l = acc; P->nodes[g].cap = 0;
                    return g;
                }
                case 'E': return new_node(P, N_EMPTY);
                case 'k': case 'g': {
                    /* \k<name> \k{name} \k'name' \g{name}; \gN \g{N} \g{-N} (boost Perl syntax): groups opened in front of the reference */
                    if (((P->flags & ORX_REGEXP2) && !((P->flags & ORX_NAMED_BACKREFS) && e == 'k')) || P->i >= P->n) { fail(P, "unsupported escape"); return -1; }
                    int open = P->p[P->i], close = open == '<' ? '>' : open == '{' ? '}' : open == '\'' ? '\'' : 0;
                    char word[128]; size_t wl = 0;
                    if (close) {
                        ++P->i;
                        while (P->i < P->n && P->p[P->i] != close && wl + 1 < sizeof word) word[wl++] = (char)P->p[P->i++];
                        if (P->i >= P->n || P->p[P->i] != close || wl == 0) { fail(P, "unterminated back-reference name"); return -1; }
                        ++P->i;
                    } else if (e == 'g') {
                        if (P->p[P->i] == '-') word[wl++] = (char)P->p[P->i++];
                        while (P->i < P->n && P->p[P->i] >= '0' && P->p[P->i] <= '9' && wl + 1 < sizeof word) word[wl++] = (char)P->p[P->i++];
                        if (wl == 0 || (wl == 1 && word[0] == '-')) { fail(P, "unsupported escape"); return -1; }
                    } else { fail(P, "unsupported escape"); return -1; }
                    word[wl] = 0;
                    int v = 0, numeric = strspn(word, "-0123456789") == wl;
                    if (numeric && e == 'g') {
                        int k = atoi(word);
                        v = k < 0 ? P->ngroups + 1 + k : k;
                        if (k == 0 || v < 1) { fail(P, "invalid back reference"); return -1; }
                    } else {
                        for (int g = 1; g <= P->ngroups && !v; ++g) if (P->names[g] && strcmp(P->names[g], word) == 0) v = g;
                        if (!v) { fail(P, "back-reference to a group name that is not defined in front of it: unsupported"); return -1; }
                    }
                    int b = new_node(P, N_BACKREF);
                    P->nodes[b].cap = v;
                    if (v > P->maxbackref) P->maxbackref = v;
                    return b;
                }
                case 'p': case 'P': case 'X': case 'C': case 'R': case 'K': case 'G': case 'N':
                    fail(P, "unsupported escape"); return -1;
                default: break;
            }
            if (e >= '1' && e <= '9') {
                int v = e - '0';
                while (P->i < P->n && P->p[P->i] >= '0' && P->p[P->i] <= '9' && v * 10 + (P->p[P->i] - '0') <= 999) v = v * 10 + (P->p[P->i++] - '0');
                int b = new_node(P, N_BACKREF);
                P->nodes[b].cap = v;
                if (v > P->maxbackref) P->maxbackref = v;
                return b;
            }
            int v = escape_byte(P, e);
            if (P->failed) return -1;
            return lit_node(P, v >= 0 ? v : e);
        }
        default:
            ++P->i;
            return lit_node(P, c);
    }
}

/* try to parse {m}, {m,}, {m,n}; returns 1 and advances if well-formed */
static int parse_braces(orx_prog* P, int* mn, int* mx) {
    size_t j = P->i + 1;
    long a = 0, b = -1; int nd = 0;
    while (j < P->n && P->p[j] >= '0' && P->p[j] <= '9') { a = a * 10 + (P->p[j] - '0'); if (a > 100000) return 0; ++j; ++nd; }
    if (nd == 0) return 0;
    if (j < P->n && P->p[j] == '}') { b = a; ++j; }
    else if (j < P->n && P->p[j] == ',') {
        ++j; nd = 0; long t = 0;
        while (j < P->n && P->p[j] >= '0' && P->p[j] <= '9') { t = t * 10 + (P->p[j] - '0'); if (t > 100000) return 0; ++j; ++nd; }
        if (j >= P->n || P->p[j] != '}') return 0;
        ++j;
        b = nd ? t : -1;
    } else return 0;
    *mn = (int)a; *mx = (int)b;
    P->i = j;
    return 1;
}

static int parse_cat(orx_prog* P, int depth) {
    int acc = -1;
    for (;;) {
        skip_extended(P);
        if (P->i >= P->n) break;
        int c = P->p[P->i];
        if (c == '|' || c == ')') break;
        int is_assert = 0;
        int a = parse_atom(P, depth, &is_assert);
        if (P->failed) return -1;
        /* quantifiers */
        int quantified = 0;
        for (;;) {
            skip_extended(P);
            if (P->i >= P->n) break;
            int q = P->p[P->i];
            int mn, mx;
            if (q == '*') { mn = 0; mx = -1; ++P->i; }
            else if (q == '+') { mn = 1; mx = -1; ++P->i; }
            else if (q == '?') { mn = 0; mx = 1; ++P->i; }
            else if (q == '{') { if (!parse_braces(P, &mn, &mx)) break; }
            else break;
            if (quantified) { fail(P, "nested quantifier"); return -1; }
            if (mx >= 0 && mx < mn) { fail(P, "bad repeat range"); return -1; }
            int greedy = 1;
            if (P->i < P->n && P->p[P->i] == '?') { greedy = 0; ++P->i; }
            int possessive = 0;
            if (greedy && P->i < P->n && P->p[P->i] == '+') { possessive = 1; ++P->i; }
            int r = new_node(P, N_REP);
            P->nodes[r].l = a; P->nodes[r].min = mn; P->nodes[r].max = mx; P->nodes[r].greedy = greedy;
            a = r;
            if (possessive) { int at = new_node(P, N_ATOMIC); P->nodes[at].l = a; a = at; } /* X*+ == (?>X*) */
            quantified = 1;
        }
        acc = cat2(P, acc, a);
    }
    if (acc < 0) acc = new_node(P, N_EMPTY);
    return acc;
}

static int parse_alt(orx_prog* P, int depth) {
    if (depth > 200) { fail(P, "nesting too deep"); return -1; }
    unsigned entry_flags = P->flags;
    int left = parse_cat(P, depth);
    if (P->failed) return -1;
    while (P->i < P->n && P->p[P->i] == '|') {
        ++P->i;
        (void)entry_flags; /* Perl: flags set inside an alternative persist to later alternatives of the same group */
        int right = parse_cat(P, depth);
        if (P->failed) return -1;
        int n = new_node(P, N_ALT);
        P->nodes[n].l = left; P->nodes[n].r = right;
        left = n;
    }
    return left;
}

/* ------------------------------------------------------------------ codegen */
static int emit(orx_prog* P, int op, int x, int y) {
    if (P->ncode >= 2000000) { fail(P, "program too large"); return 0; }
    if (P->ncode == P->capcode) {
        P->capcode = P->capcode ? P->capcode * 2 : 128;
        P->code = (inst*)realloc(P->code, sizeof(inst) * P->capcode);
    }
    inst* in = &P->code[P->ncode];
    in->op = op; in->x = x; in->y = y; in->z = 0; in->w = 0;
    return P->ncode++;
}
static int nullable(const orx_prog* P, int n) {
    const node* nd = &P->nodes[n];
    switch (nd->kind) {
        case N_EMPTY: case N_ASSERT: return 1;
        case N_SET: return 0;
        case N_CAT: return nullable(P, nd->l) && nullable(P, nd->r);
        case N_ALT: return nullable(P, nd->l) || nullable(P, nd->r);
        case N_REP: return nd->min == 0 || nullable(P, nd->l);
        case N_GROUP: return nullable(P, nd->l);
        case N_ATOMIC: return nullable(P, nd->l);
        case N_LOOKAHEAD: return 1;
        case N_BACKREF: return 1;
        case N_COND: return nullable(P, nd->l) || nullable(P, nd->r);
    }
    return 1;
}
static void gen(orx_prog* P, int n) {
    if (P->failed) return;
    const node nd = P->nodes[n];
    switch (nd.kind) {
        case N_EMPTY: break;
        case N_SET: emit(P, I_SET, nd.set, 0); break;
        case N_CAT: gen(P, nd.l); gen(P, nd.r); break;
        case N_ALT: {
            int sp = emit(P, I_SPLIT, 0, 0);
            P->code[sp].x = P->ncode;
            gen(P, nd.l);
            int jm = emit(P, I_JMP, 0, 0);
            P->code[sp].y = P->ncode;
            gen(P, nd.r);
            P->code[jm].x = P->ncode;
            break;
        }
        case N_GROUP:
            if (nd.cap) emit(P, I_SAVE, 2 * nd.cap, 0);
            gen(P, nd.l);
            if (nd.cap) emit(P, I_SAVE, 2 * nd.cap + 1, 0);
            break;
        case N_BACKREF: emit(P, I_BACKREF, nd.cap, 0); break;
        case N_COND: {
            int c = emit(P, I_COND, nd.cap, 0);
            gen(P, nd.l);
            int j = emit(P, I_JMP, 0, 0);
            P->code[c].y = P->ncode;
            gen(P, nd.r);
            P->code[j].x = P->ncode;
            break;
        }
        case N_ATOMIC:
            emit(P, I_ATOM_BEGIN, 0, 0);
            gen(P, nd.l);
            emit(P, I_ATOM_END, 0, 0);
            break;
        case N_LOOKAHEAD: {
            int b = emit(P, nd.max ? I_NLOOK_BEGIN : I_LOOK_BEGIN, 0, 0);
            if (nd.min > 0) emit(P, I_BACK, nd.min, 0);
            gen(P, nd.l);
            emit(P, nd.max ? I_NLOOK_END : I_LOOK_END, 0, 0);
            P->code[b].x = P->ncode; /* where a satisfied negative look-ahead continues */
            break;
        }
        case N_ASSERT: {
            int k = emit(P, I_ASSERT, nd.akind, nd.set);
            P->code[k].z = nd.min; P->code[k].w = nd.max;
            break;
        }
        case N_REP: {
            if (P->nodes[nd.l].kind == N_SET) { /* perl_matcher::match_set_repeat style fast path */
                int k = emit(P, I_REPSET, P->nodes[nd.l].set, nd.min);
                P->code[k].z = nd.max; P->code[k].w = nd.greedy;
                break;
            }
            for (int i = 0; i < nd.min; ++i) gen(P, nd.l);
            int body_nullable = nullable(P, nd.l);
            if (nd.max < 0) {
                int reg = body_nullable ? P->nloopregs++ : -1;
                int sp = emit(P, I_SPLIT, 0, 0);
                int body = P->ncode;
                if (reg >= 0) emit(P, I_MARK, reg, 0);
                gen(P, nd.l);
                int chk = -1;
                if (reg >= 0) chk = emit(P, I_CHK, reg, 0);
                emit(P, I_JMP, sp, 0);
                int out = P->ncode;
                if (chk >= 0) P->code[chk].y = out;
                if (nd.greedy) { P->code[sp].x = body; P->code[sp].y = out; }
                else { P->code[sp].x = out; P->code[sp].y = body; }
            } else {
                int cnt = nd.max - nd.min;
                int* sps = cnt ? (int*)malloc(sizeof(int) * cnt) : NULL;
                for (int i = 0; i < cnt; ++i) {
                    sps[i] = emit(P, I_SPLIT, 0, 0);
                    int body = P->ncode;
                    gen(P, nd.l);
                    if (nd.greedy) P->code[sps[i]].x = body; else P->code[sps[i]].y = body;
                }
                int out = P->ncode;
                for (int i = 0; i < cnt; ++i) { if (nd.greedy) P->code[sps[i]].y = out; else P->code[sps[i]].x = out; }
                free(sps);
            }
            break;
        }
    }
}

orx_prog* orx_compile(const char* pattern, size_t len, unsigned flags, char* err, size_t errcap) {
    orx_prog* P = (orx_prog*)calloc(1, sizeof *P);
    P->p = (const unsigned char*)pattern; P->n = len; P->i = 0; P->flags = flags;
    int root = parse_alt(P, 0);
    if (!P->failed && P->i < P->n) fail(P, P->p[P->i] == ')' ? "unmatched )" : "unexpected character");
    if (!P->failed && P->maxbackref > P->ngroups) fail(P, "invalid back reference");
    if (!P->failed) {
        emit(P, I_SAVE, 0, 0);
        gen(P, root);
        emit(P, I_SAVE, 1, 0);
        emit(P, I_MATCH, 0, 0);
    }
    if (P->failed) {
        if (err && errcap) { snprintf(err, errcap, "%s", P->err); }
        orx_free(P);
        return NULL;
    }
    if (err && errcap) err[0] = 0;
    P->p = NULL;
    return P;
}
void orx_free(orx_prog* P) {
    if (!P) return;
    for (int i = 0; i <= ORX_MAX_GROUPS; ++i) free(P->names[i]);
    free(P->code); free(P->sets); free(P->nodes); free(P);
}
int orx_mark_count(const orx_prog* P) { return P->ngroups; }
const char* orx_group_name(const orx_prog* P, int g) { return (g >= 1 && g <= P->ngroups) ? P->names[g] : NULL; }

/* ------------------------------------------------------------------ matcher */
enum { F_ALT, F_UNDO_CAP, F_UNDO_LOOP, F_REP_GREEDY, F_REP_LAZY, F_ATOM_MARK, F_LOOK_MARK /* b=pos */, F_NLOOK_MARK /* a=pc after, b=pos */ };
typedef struct { int kind; int a; long b; long c; } frame;
/* F_ALT: a=pc, b=pos | F_UNDO_*: a=slot, b=old | F_REP_GREEDY: a=pc_next, b=low, c=cur | F_REP_LAZY: a=pc (of REPSET), b=cur pos, c=count */

typedef struct { frame* v; size_t n, cap; } fstack;
static void push(fstack* st, int kind, int a, long b, long c) {
    if (st->n == st->cap) { st->cap = st->cap ? st->cap * 2 : 256; st->v = (frame*)realloc(st->v, st->cap * sizeof(frame)); }
    frame* f = &st->v[st->n++];
    f->kind = kind; f->a = a; f->b = b; f->c = c;
}

static int check_assert(int kind, const uint8_t* s, long n, long pos) {
    switch (kind) {
        case A_BOL_SL: return pos == 0;
        case A_BOL_ML: {
            if (pos == 0) return 1;
            int t = s[pos - 1];
            if (pos != n) return is_sep(t) && !(t == '\r' && s[pos] == '\n');
            return is_sep(t);
        }
        case A_EOL_SL: return pos == n;
        case A_EOL_ML: {
            if (pos == n) return 1;
            if (is_sep(s[pos])) {
                if (pos > 0 && s[pos - 1] == '\r' && s[pos] == '\n') return 0;
                return 1;
            }
            return 0;
        }
        case A_BUF_START: return pos == 0;
        case A_BUF_END: return pos == n;
        case A_BUF_END_NL: { long p = pos; while (p < n && (s[p] >= 10 && s[p] <= 13)) ++p; return p == n; }
        case A_WORDB: case A_NWORDB: {
            int a = pos > 0 && is_word(s[pos - 1]);
            int b = pos < n && is_word(s[pos]);
            return kind == A_WORDB ? (a != b) : (a == b);
        }
        case A_WORD_START: return !(pos > 0 && is_word(s[pos - 1])) && (pos < n && is_word(s[pos]));
        case A_WORD_END: return (pos > 0 && is_word(s[pos - 1])) && !(pos < n && is_word(s[pos]));
    }
    return 0;
}

#define ORX_STEP_BUDGET 100000000L /* BOOST_REGEX_MAX_STATE_COUNT */

static int run(const orx_prog* P, const uint8_t* s, long n, long start, int full, int32_t* caps, fstack* st, long* loopregs,
               long* budget) {
    const inst* code = P->code;
    int pc = 0; long pos = start;
    st->n = 0;
    int ncap = 2 * (P->ngroups + 1);
    for (int i = 0; i < ncap; ++i) caps[i] = -1;
    for (;;) {
        if (--*budget < 0) return -1;
        const inst* in = &code[pc];
        switch (in->op) {
            case I_SET:
                if (pos < n && cs_has(&P->sets[in->x], s[pos])) { ++pos; ++pc; continue; }
                goto backtrack;
            case I_REPSET: {
                const cset* cs = &P->sets[in->x];
                long mn = in->y, mx = in->z < 0 ? (n - pos) : in->z;
                if (mx > n - pos) mx = n - pos;
                if (in->w) { /* greedy */
                    long k = 0;
                    while (k < mx && cs_has(cs, s[pos + k])) ++k;
                    if (k < mn) goto backtrack;
                    if (k > mn) push(st, F_REP_GREEDY, pc + 1, pos + mn, pos + k);
                    pos += k; ++pc; continue;
                } else {
                    long k = 0;
                    while (k < mn && k < mx && cs_has(cs, s[pos + k])) ++k;
                    if (k < mn) goto backtrack;
                    if (in->z < 0 || in->z > mn) push(st, F_REP_LAZY, pc, pos + k, k);
                    pos += k; ++pc; continue;
                }
            }
            case I_SPLIT: push(st, F_ALT, in->y, pos, 0); pc = in->x; continue;
            case I_JMP: pc = in->x; continue;
            case I_SAVE: push(st, F_UNDO_CAP, in->x, caps[in->x], 0); caps[in->x] = (int32_t)pos; ++pc; continue;
            case I_MARK: push(st, F_UNDO_LOOP, in->x, loopregs[in->x], 0); loopregs[in->x] = pos; ++pc; continue;
            case I_CHK: if (loopregs[in->x] == pos) pc = in->y; else ++pc; continue;
            case I_ATOM_BEGIN: push(st, F_ATOM_MARK, 0, 0, 0); ++pc; continue;
            case I_ATOM_END: {
                /* commit: drop every alternative created since the matching mark, keep the undo records (they must
                 * still be replayed if the matcher later backtracks past the whole group) */
                size_t m = st->n;
                int depth = 0;
                while (m > 0) {
                    --m;
                    if (st->v[m].kind == F_ATOM_MARK) { if (depth == 0) break; --depth; }
                }
                size_t w = m; /* overwrite the mark itself */
                for (size_t r = m + 1; r < st->n; ++r)
                    if (st->v[r].kind == F_UNDO_CAP || st->v[r].kind == F_UNDO_LOOP) st->v[w++] = st->v[r];
                st->n = w;
                ++pc; continue;
            }
            case I_LOOK_BEGIN: push(st, F_LOOK_MARK, 0, pos, 0); ++pc; continue;
            case I_LOOK_END: {
                /* body matched: commit like an atomic group (captures made inside stay), but give the input back */
                size_t m = st->n;
                while (m > 0 && st->v[m - 1].kind != F_LOOK_MARK) --m;
                --m;
                pos = st->v[m].b;
                size_t w = m;
                for (size_t r = m + 1; r < st->n; ++r)
                    if (st->v[r].kind == F_UNDO_CAP || st->v[r].kind == F_UNDO_LOOP) st->v[w++] = st->v[r];
                st->n = w;
                ++pc; continue;
            }
            case I_NLOOK_BEGIN: push(st, F_NLOOK_MARK, in->x, pos, 0); ++pc; continue;
            case I_BACK: /* (the mark pushed just before holds the position the assertion stands at; I_*LOOK_END goes back to it) */
                if (pos < (long)in->x) goto backtrack;
                pos -= (long)in->x;
                ++pc; continue;
            case I_NLOOK_END: {
                /* body matched, so the assertion fails: unwind everything the body did, mark included, and backtrack */
                for (;;) {
                    frame* f = &st->v[st->n - 1];
                    if (f->kind == F_UNDO_CAP) caps[f->a] = (int32_t)f->b;
                    else if (f->kind == F_UNDO_LOOP) loopregs[f->a] = f->b;
                    --st->n;
                    if (f->kind == F_NLOOK_MARK) break;
                }
                goto backtrack;
            }
            case I_ASSERT:
                if (in->x == A_LOOK) {
                    int hit = in->z ? (pos > 0 && cs_has(&P->sets[in->y], s[pos - 1]))
                                    : (pos < n && cs_has(&P->sets[in->y], s[pos]));
                    if (hit != in->w) { ++pc; continue; }   /* positive: hit, negative: !hit */
                    goto backtrack;
                }
                if (check_assert(in->x, s, n, pos)) { ++pc; continue; }
                goto backtrack;
            case I_COND: if (caps[2 * in->x + 1] >= 0) ++pc; else pc = in->y; continue;
            case I_BACKREF: {
                long b = caps[2 * in->x], e = caps[2 * in->x + 1];
                if (b < 0 || e < 0 || e < b || e - b > n - pos) goto backtrack;
                long k = 0, len = e - b;
                if (P->flags & ORX_ICASE) {
                    while (k < len) {
                        int c1 = s[b + k], c2 = s[pos + k];
                        if (c1 >= 'A' && c1 <= 'Z') c1 += 32;
                        if (c2 >= 'A' && c2 <= 'Z') c2 += 32;
                        if (c1 != c2) break;
                        ++k;
                    }
                } else {
                    while (k < len && s[b + k] == s[pos + k]) ++k;
                }
                if (k < len) goto backtrack;
                pos += len; ++pc; continue;
            }
            case I_MATCH:
                if (full && pos != n) goto backtrack;
                return 1;
        }
    backtrack:
        for (;;) {
            if (st->n == 0) return 0;
            frame* f = &st->v[st->n - 1];
            if (f->kind == F_UNDO_CAP) { caps[f->a] = (int32_t)f->b; --st->n; continue; }
            if (f->kind == F_UNDO_LOOP) { loopregs[f->a] = f->b; --st->n; continue; }
            if (f->kind == F_ATOM_MARK || f->kind == F_LOOK_MARK) { --st->n; continue; }
            if (f->kind == F_NLOOK_MARK) { pc = f->a; pos = f->b; --st->n; break; } /* body cannot match: assertion holds */
            if (f->kind == F_ALT) { pc = f->a; pos = f->b; --st->n; break; }
            if (f->kind == F_REP_GREEDY) {
                --f->c;
                pc = f->a; pos = f->c;
                if (f->c <= f->b) --st->n;
                break;
            }
            /* F_REP_LAZY: take one more */
            {
                const inst* in2 = &code[f->a];
                long cur = f->b, cnt = f->c;
                if ((in2->z >= 0 && cnt >= in2->z) || cur >= n || !cs_has(&P->sets[in2->x], s[cur])) { --st->n; continue; }
                f->b = cur + 1; f->c = cnt + 1;
                pc = f->a + 1; pos = cur + 1;
                break;
            }
        }
    }
}

typedef struct { fstack st; long* loopregs; } scratch;
static void scratch_init(scratch* sc, const orx_prog* P) {
    memset(sc, 0, sizeof *sc);
    sc->loopregs = (long*)calloc(P->nloopregs + 1, sizeof(long));
}
static void scratch_free(scratch* sc) { free(sc->st.v); free(sc->loopregs); }

int orx_fullmatch(const orx_prog* P, const uint8_t* s, size_t n, int32_t* caps) {
    scratch sc; scratch_init(&sc, P);
    long budget = ORX_STEP_BUDGET;
    int r = run(P, s, (long)n, 0, 1, caps, &sc.st, sc.loopregs, &budget);
    scratch_free(&sc);
    return r;
}

int orx_search(const orx_prog* P, const uint8_t* s, size_t n, size_t start, int32_t* caps) {
    scratch sc; scratch_init(&sc, P);
    long budget = ORX_STEP_BUDGET;
    int r = 0;
    for (long at = (long)start; at <= (long)n; ++at) {
        r = run(P, s, (long)n, at, 0, caps, &sc.st, sc.loopregs, &budget);
        if (r != 0) break;
    }
    scratch_free(&sc);
    return r;
}

/* boost::regex_search(first, last, what, re, match_continuous): the match has to start at `first` (StringTools.cpp:263-289) */
int orx_prefixmatch(const orx_prog* P, const uint8_t* s, size_t n, int32_t* caps) {
    scratch sc; scratch_init(&sc, P);
    long budget = ORX_STEP_BUDGET;
    int r = run(P, s, (long)n, 0, 0, caps, &sc.st, sc.loopregs, &budget);
    scratch_free(&sc);
    return r;
}

long orx_fullmatch_batch(const orx_prog* P, const uint8_t* data, const uint32_t* off, const uint32_t* len, size_t nlines,
                         int ngroups, int32_t* caps, uint8_t* status) {
    scratch sc; scratch_init(&sc, P);
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * 2 * (P->ngroups + 1));
    long matched = 0;
    for (size_t i = 0; i < nlines; ++i) {
        long budget = ORX_STEP_BUDGET;
        int r = run(P, data + off[i], (long)len[i], 0, 1, tmp, &sc.st, sc.loopregs, &budget);
        int32_t* out = caps + (size_t)i * 2 * ngroups;
        if (r == 1) {
            for (int g = 1; g <= ngroups; ++g) {
                if (g <= P->ngroups) { out[2 * (g - 1)] = tmp[2 * g]; out[2 * (g - 1) + 1] = tmp[2 * g + 1]; }
                else { out[2 * (g - 1)] = -1; out[2 * (g - 1) + 1] = -1; }
            }
            status[i] = 1; ++matched;
        } else {
            for (int k = 0; k < 2 * ngroups; ++k) out[k] = -1;
            status[i] = r == 0 ? 0 : 2;
        }
    }
    free(tmp);
    scratch_free(&sc);
    return matched;
}
