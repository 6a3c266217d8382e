// bt_vm.hpp -- the device BACKTRACKING engine (LC_ENGINE_BT, round 6): patterns that are not regular -- back-references \1 .. \N --
// have no automaton; boost::regex_match (core/common/StringTools.cpp:183-211) and regexp2 (plugins/processor/grok/processor_grok.go:156)
// run them the way they run everything, by backtracking.  This is that walk for one value per LANE: a small instruction program
// (bt_program.cpp compiles it from the parser's tree) and an explicit stack in the lane's slice of a scratch pool in HBM.  No CPU
// fallback: a handle whose pattern needs this engine runs here or fails loudly (gpu_runtime.hip launchBt).
//
// Semantics (the ones boost's perl_matcher publishes, as the oracle restates them in oracle/bt_regex.c):
//   * leftmost-first: of two ways on, the preferred one is tried first and the other is pushed; the first way through the whole
//     program that ends at the END of the value is the match (regex_match; searches and prefix matches arrive wrapped into whole-value
//     patterns by regex_handle.cpp wrapForSearch / wrapForPrefix, as for the automata);
//   * a repeat of ONE byte class is counted (perl_matcher::match_set_repeat): one stack record per repeat, not per byte;
//   * a repeat whose body last matched the empty string stops iterating (match_rep's null check: MARK / CHK);
//   * a back-reference to a group that took no part fails; under LC_SYNTAX_ICASE it compares ASCII-folded bytes;
//   * an atomic group drops the alternatives opened inside it when it is left, and keeps the undo records;
//   * a look-around's body runs in place (a look-behind's of fixed length k from k bytes back): a positive one is committed like an
//     atomic group and gives the input back, its captures stay; a negative one holds when its body cannot match;
//   * a step budget and the stack's capacity: a value that exhausts either is reported LC_GAVE_UP -- boost's complexity exception,
//     which the reference counts as a parse failure (StringTools.cpp:200-205) -- never guessed.
//
// The routine is __host__ __device__ so that tests/native/bt_host_check.cpp can walk the same code over the golden vectors on a box
// without a GPU; the product calls it from bt_match_kernel only.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define LC_BT_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define LC_BT_HD inline
#endif

// ---- program blob (u32 words): header, byte classes (8 words each), instructions (4 words each)
enum { BT_NINST = 0, BT_NSETS = 1, BT_NCAPS = 2 /* 2 x (groups + 1) */, BT_NLOOP = 3, BT_OFF_SETS = 4, BT_OFF_CODE = 5, BT_FLAGS = 6 /* bit 0: back-references fold case */,
       BT_HEADER_WORDS = 8 };
// instruction word 0: op | flags << 8; words 1..3: x, y, z
enum {
    BT_SET = 0,      // x = class: the next byte is in it
    BT_REPSET = 1,   // x = class, y = min, z = max (BT_INF: unbounded), flag 1: greedy
    BT_SPLIT = 2,    // go on at x, keep y as the alternative
    BT_JMP = 3,      // x
    BT_SAVE = 4,     // x = capture slot <- position
    BT_MARK = 5,     // x = loop register <- position (start of an iteration of a repeat whose body may be empty)
    BT_CHK = 6,      // x = loop register, y = where the repeat is left: the iteration matched nothing -> leave
    BT_ASSERT = 7,   // x = class, flag 1: the byte BEHIND (else ahead), flag 2: holds at the edge of the value
    BT_BACKREF = 8,  // x = group
    BT_ATOM_BEGIN = 9,
    BT_ATOM_END = 10,
    BT_MATCH = 11,
    BT_LOOK_BEGIN = 12,   // a positive look-around: the body runs in place, then the position comes back (its captures stay)
    BT_LOOK_END = 13,
    BT_NLOOK_BEGIN = 14,  // a negative one; x = where the walk goes on when the body cannot match
    BT_NLOOK_END = 15,    // the body matched: the assertion fails
    BT_BACK = 16,         // x = k: k bytes back (a look-behind's body of fixed length k runs forward from there); fewer behind: fail
    BT_COND = 17          // x = group, y = where the "no" branch starts: group x has taken part -> go on (the "yes" branch), else y
};
constexpr uint32_t BT_INF = 0xFFFFFFFFu;
constexpr uint32_t BT_NONE = 0xFFFFFFFFu;  // unset capture slot (-1 in the result row)

// ---- stack entries: two words, kind | a << 4, b
enum { BT_K_ALT = 0 /* a = pc, b = position */, BT_K_UNDO_CAP = 1 /* a = slot, b = old */, BT_K_UNDO_LOOP = 2, BT_K_ATOM_MARK = 3,
       BT_K_REP_GREEDY = 4 /* a = pc of the repeat, b = current end; the entry below (AUX) holds the lowest end */,
       BT_K_REP_LAZY = 5 /* a = pc of the repeat, b = next byte to take; AUX below: bytes taken */, BT_K_AUX = 6,
       BT_K_LOOK_MARK = 7 /* b = position the assertion stands at */, BT_K_NLOOK_MARK = 8 /* a = pc behind the assertion, b = position */ };

LC_BT_HD bool btHas(const uint32_t* sets, uint32_t set, uint32_t c) { return (sets[set * 8u + (c >> 5)] >> (c & 31u)) & 1u; }
LC_BT_HD uint32_t btFold(uint32_t c) { return (c >= 'A' && c <= 'Z') ? c + 32u : c; }
// how many of the bytes s[0, limit) are in `set`, from the front: eight bytes per load (gfx950 takes the unaligned load), their eight
// class tests issued side by side -- a counted repeat is where a log line's bytes go, and byte loads in a dependent chain were its cost
LC_BT_HD uint32_t btRun8(const uint32_t* sets, uint32_t set, const uint8_t* s, uint32_t limit) {
    uint32_t k = 0;
    while (k + 8u <= limit) {
        uint64_t w;
        __builtin_memcpy(&w, s + k, 8);
        uint32_t ok = 0;
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) ok |= uint32_t(btHas(sets, set, uint32_t(w >> (8u * j)) & 0xFFu)) << j;
        if (ok != 0xFFu) return k + uint32_t(__builtin_ctz(~ok));
        k += 8u;
    }
    while (k < limit && btHas(sets, set, s[k])) ++k;
    return k;
}

// Walks value s[0, n) from offset `from`.  scratch: [captures][loop registers][stack].  Returns 1: match (capture slots in scratch[0 ..
// BT_NCAPS), slot 2g / 2g + 1 = group g, group 0 = the whole match); 0: no match; -1: out of steps; -2: out of stack (the launch
// gives such a value a second pass with a larger slice before it reports LC_GAVE_UP).
LC_BT_HD int btRun(const uint32_t* blob, const uint8_t* s, uint32_t n, uint32_t from, uint32_t* scratch, uint32_t scratchWords, uint32_t budget) {
    const uint32_t nCaps = blob[BT_NCAPS], nLoop = blob[BT_NLOOP];
    const uint32_t* sets = blob + blob[BT_OFF_SETS];
    const uint32_t* code = blob + blob[BT_OFF_CODE];
    const bool fold = (blob[BT_FLAGS] & 1u) != 0;
    if (scratchWords < nCaps + nLoop + 32u) return -1;
    uint32_t* caps = scratch;
    uint32_t* loop = scratch + nCaps;
    uint32_t* st = loop + nLoop;
    const uint32_t cap = (scratchWords - nCaps - nLoop) / 2u;  // stack entries
    for (uint32_t i = 0; i < nCaps + nLoop; ++i) scratch[i] = BT_NONE;
    uint32_t sp = 0, pc = 0, pos = from;
#define BT_PUSH(kind, a, b)                                 \
    do {                                                    \
        if (sp >= cap) return -2;                           \
        st[2u * sp] = uint32_t(kind) | (uint32_t(a) << 4);  \
        st[2u * sp + 1u] = uint32_t(b);                     \
        ++sp;                                               \
    } while (0)
    for (;;) {
        if (budget-- == 0u) return -1;
        const uint32_t* in = code + pc * 4u;
        const uint32_t op = in[0] & 0xFFu, fl = in[0] >> 8;
        bool fail = false;
        switch (op) {
            case BT_SET:
                if (pos < n && btHas(sets, in[1], s[pos])) {
                    ++pos;
                    ++pc;
                } else {
                    fail = true;
                }
                break;
            case BT_REPSET: {
                const uint32_t mn = in[2];
                uint32_t mx = in[3];
                if (mx > n - pos) mx = n - pos;
                uint32_t k = 0;
                if (fl & 1u) {
                    k = btRun8(sets, in[1], s + pos, mx);
                    if (k < mn) {
                        fail = true;
                        break;
                    }
                    if (k > mn) {
                        BT_PUSH(BT_K_AUX, 0, pos + mn);
                        BT_PUSH(BT_K_REP_GREEDY, pc, pos + k);
                    }
                } else {
                    while (k < mn && k < mx && btHas(sets, in[1], s[pos + k])) ++k;
                    if (k < mn) {
                        fail = true;
                        break;
                    }
                    if (in[3] == BT_INF || in[3] > mn) {
                        BT_PUSH(BT_K_AUX, 0, k);
                        BT_PUSH(BT_K_REP_LAZY, pc, pos + k);
                    }
                }
                pos += k;
                ++pc;
                break;
            }
            case BT_SPLIT:
                BT_PUSH(BT_K_ALT, in[2], pos);
                pc = in[1];
                break;
            case BT_JMP: pc = in[1]; break;
            case BT_SAVE:
                BT_PUSH(BT_K_UNDO_CAP, in[1], caps[in[1]]);
                caps[in[1]] = pos;
                ++pc;
                break;
            case BT_MARK:
                BT_PUSH(BT_K_UNDO_LOOP, in[1], loop[in[1]]);
                loop[in[1]] = pos;
                ++pc;
                break;
            case BT_CHK:
                if (loop[in[1]] == pos) pc = in[2];
                else ++pc;
                break;
            case BT_ASSERT: {
                bool ok;
                if (fl & 1u) ok = pos == 0u ? (fl & 2u) != 0u : btHas(sets, in[1], s[pos - 1u]);
                else ok = pos == n ? (fl & 2u) != 0u : btHas(sets, in[1], s[pos]);
                if (ok) ++pc;
                else fail = true;
                break;
            }
            case BT_BACKREF: {
                const uint32_t b = caps[2u * in[1]], e = caps[2u * in[1] + 1u];
                if (b == BT_NONE || e == BT_NONE || e < b || e - b > n - pos) {
                    fail = true;
                    break;
                }
                const uint32_t len = e - b;
                uint32_t k = 0;
                if (fold) {
                    while (k < len && btFold(s[b + k]) == btFold(s[pos + k])) ++k;
                } else {
                    while (k < len && s[b + k] == s[pos + k]) ++k;
                }
                if (k < len) {
                    fail = true;
                    break;
                }
                pos += len;
                ++pc;
                break;
            }
            case BT_ATOM_BEGIN:
                BT_PUSH(BT_K_ATOM_MARK, 0, 0);
                ++pc;
                break;
            case BT_ATOM_END: {
                // commit: every alternative opened since the group's mark goes, the undo records stay (they are replayed if the
                // walk later backtracks past the whole group)
                uint32_t m = sp, depth = 0;
                while (m > 0u) {
                    --m;
                    const uint32_t kind = st[2u * m] & 15u;
                    if (kind == BT_K_ATOM_MARK) {
                        if (depth == 0u) break;
                        --depth;
                    }
                }
                uint32_t w = m;  // (the mark itself is overwritten)
                for (uint32_t r = m + 1u; r < sp; ++r) {
                    const uint32_t kind = st[2u * r] & 15u;
                    if (kind == BT_K_UNDO_CAP || kind == BT_K_UNDO_LOOP) {
                        st[2u * w] = st[2u * r];
                        st[2u * w + 1u] = st[2u * r + 1u];
                        ++w;
                    }
                }
                sp = w;
                ++pc;
                break;
            }
            case BT_LOOK_BEGIN:
                BT_PUSH(BT_K_LOOK_MARK, 0, pos);
                ++pc;
                break;
            case BT_LOOK_END: {
                // the body matched: committed like an atomic group (no way back into it, its captures stay), the input given back
                uint32_t m = sp;
                while (m > 0u && (st[2u * (m - 1u)] & 15u) != BT_K_LOOK_MARK) --m;
                if (m == 0u) return -1;
                --m;
                pos = st[2u * m + 1u];
                uint32_t w = m;
                for (uint32_t r = m + 1u; r < sp; ++r) {
                    const uint32_t kind = st[2u * r] & 15u;
                    if (kind == BT_K_UNDO_CAP || kind == BT_K_UNDO_LOOP) {
                        st[2u * w] = st[2u * r];
                        st[2u * w + 1u] = st[2u * r + 1u];
                        ++w;
                    }
                }
                sp = w;
                ++pc;
                break;
            }
            case BT_NLOOK_BEGIN:
                BT_PUSH(BT_K_NLOOK_MARK, in[1], pos);
                ++pc;
                break;
            case BT_NLOOK_END:
                // the body matched, so the assertion fails: everything the body did is undone, its mark goes, and the walk backtracks
                for (;;) {
                    if (sp == 0u) return -1;
                    const uint32_t top = sp - 1u;
                    const uint32_t kind = st[2u * top] & 15u, a = st[2u * top] >> 4, b = st[2u * top + 1u];
                    if (kind == BT_K_UNDO_CAP) caps[a] = b;
                    else if (kind == BT_K_UNDO_LOOP) loop[a] = b;
                    sp = top;
                    if (kind == BT_K_NLOOK_MARK) break;
                }
                fail = true;
                break;
            case BT_COND:
                if (caps[2u * in[1] + 1u] != BT_NONE) ++pc;
                else pc = in[2];
                break;
            case BT_BACK:
                if (pos < in[1]) {
                    fail = true;
                } else {
                    pos -= in[1];
                    ++pc;
                }
                break;
            case BT_MATCH:
                if (pos == n) return 1;
                fail = true;
                break;
            default: return -1;
        }
        if (!fail) continue;
        // ---- backtrack: undo records are replayed on the way down to the next alternative
        for (;;) {
            if (sp == 0u) return 0;
            const uint32_t top = sp - 1u;
            const uint32_t kind = st[2u * top] & 15u, a = st[2u * top] >> 4, b = st[2u * top + 1u];
            if (kind == BT_K_UNDO_CAP) {
                caps[a] = b;
                sp = top;
                continue;
            }
            if (kind == BT_K_UNDO_LOOP) {
                loop[a] = b;
                sp = top;
                continue;
            }
            if (kind == BT_K_ATOM_MARK || kind == BT_K_AUX || kind == BT_K_LOOK_MARK) {
                sp = top;
                continue;
            }
            if (kind == BT_K_NLOOK_MARK) {  // the body cannot match: the assertion holds, the walk goes on behind it
                pc = a;
                pos = b;
                sp = top;
                break;
            }
            if (kind == BT_K_ALT) {
                pc = a;
                pos = b;
                sp = top;
                break;
            }
            if (kind == BT_K_REP_GREEDY) {  // one byte less ...
                uint32_t cur = b - 1u;
                const uint32_t low = st[2u * (top - 1u) + 1u];
                // ... and on, past the ends behind which what follows cannot start (perl_matcher::unwind_greedy_single_repeat's
                // can_start test): a byte class behind the repeat would fail there at once and come straight back -- same result,
                // without a step per byte.  A line that does NOT match its pattern is mostly this.
                const uint32_t* nx = code + (a + 1u) * 4u;
                while ((nx[0] & 0xFFu) == BT_SAVE) nx += 4;  // (a group that closes behind the repeat: zero-width, never fails)
                if ((nx[0] & 0xFFu) == BT_SET)
                    while (cur > low && !btHas(sets, nx[1], s[cur])) --cur;
                st[2u * top + 1u] = cur;
                pc = a + 1u;
                pos = cur;
                if (cur <= low) sp = top - 1u;
                break;
            }
            {  // BT_K_REP_LAZY: one byte more -- and on, while what follows cannot start there (as above)
                const uint32_t* rep = code + a * 4u;
                const uint32_t* nx = rep + 4u;
                while ((nx[0] & 0xFFu) == BT_SAVE) nx += 4;
                const bool skip = (nx[0] & 0xFFu) == BT_SET;
                uint32_t cur = b, count = st[2u * (top - 1u) + 1u];
                bool dead = false;
                for (;;) {
                    if ((rep[3] != BT_INF && count >= rep[3]) || cur >= n || !btHas(sets, rep[1], s[cur])) {
                        dead = true;
                        break;
                    }
                    ++cur;
                    ++count;
                    if (!skip || cur >= n || btHas(sets, nx[1], s[cur])) break;
                }
                if (dead) {
                    sp = top - 1u;
                    continue;
                }
                st[2u * top + 1u] = cur;
                st[2u * (top - 1u) + 1u] = count;
                pc = a + 1u;
                pos = cur;
                break;
            }
        }
    }
#undef BT_PUSH
}
