// regex_handle.cpp -- host half of the C ABI: compile a pattern into device table blobs (no HIP calls here).
#include "regex_handle.hpp"
#include "bt_program.hpp"

#include <cstdlib>
#include <functional>
#include <tuple>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>

#include "device_tables.h"
#include "screen_kernel_layout.h"
#include "tdfa_l2_layout.h"

namespace lcregex {

namespace {
struct BlobWriter {
    std::vector<uint8_t> bytes;
    uint32_t reserve(size_t n) {
        size_t at = (bytes.size() + 15) & ~size_t(15);
        bytes.resize(at + n, 0);
        return uint32_t(at);
    }
    template <class T>
    uint32_t put(const std::vector<T>& v) {
        uint32_t at = reserve(v.size() * sizeof(T));
        if (!v.empty()) std::memcpy(bytes.data() + at, v.data(), v.size() * sizeof(T));
        return at;
    }
    std::vector<uint32_t> finish(uint32_t totalIndex) {
        size_t total = (bytes.size() + 15) & ~size_t(15);
        bytes.resize(total, 0);
        std::vector<uint32_t> out(total / 4);
        std::memcpy(out.data(), bytes.data(), total);
        out[totalIndex] = uint32_t(total);
        return out;
    }
};
}  // namespace

// ---- folding of multi-stamp register programs (packing only; the logical tables stay as the builder made them).
// A capture group that matches the EMPTY string ("" in a quoted field) stamps its begin and its end on one transition:
// {r = pos, r' = pos}, a general register program.  The kernels replay every chunk in which some lane met one, and real logs
// are full of them (measured: 10 % of the lines with an empty field = +22 % kernel time).  When EVERY general program of a
// table is such a set of position stamps -- no copies anywhere -- each distinct set S gets one extra register both(S) and
// the transition stamps that alone (a plain stamp again).  Positions only grow along a line, so "last stamp wins" is "largest
// position wins": at the end register r reads as max(r, both(S) for the sets S that contain r), with registers cleared to 0
// at the start of a line (an unstamped register can only lose the max).  Tables with any copy keep their general programs.
struct TdfaFold {
    bool ok = false;
    std::vector<std::vector<uint8_t>> sets;  // distinct sorted destination sets
    std::vector<int> listSet;                // per op list: index into sets, or -1
};
static TdfaFold planTdfaFold(const TdfaTables& t) {
    TdfaFold f;
    static const bool off = getenv("LC_TDFA_NO_FOLD") != nullptr;
    const size_t nLists = t.opsStart.size() - 1;
    f.listSet.assign(nLists, -1);
    if (off) return f;
    for (size_t id = 1; id < nLists; ++id) {
        const uint32_t at = t.opsStart[id];
        const uint32_t n = t.ops[at];
        std::vector<uint8_t> dsts;
        for (uint32_t k = 0; k < n; ++k) {
            const uint16_t w = t.ops[at + 1 + k];
            if ((w >> 8) != kRegPos) return f;  // a copy: the table keeps its general programs
            dsts.push_back(uint8_t(w & 0xFF));
        }
        if (n < 2) continue;
        std::sort(dsts.begin(), dsts.end());
        dsts.erase(std::unique(dsts.begin(), dsts.end()), dsts.end());
        auto it = std::find(f.sets.begin(), f.sets.end(), dsts);
        if (it == f.sets.end()) {
            f.sets.push_back(dsts);
            it = f.sets.end() - 1;
        }
        f.listSet[id] = int(it - f.sets.begin());
    }
    if (f.sets.empty() || t.nRegs + f.sets.size() > size_t(kMaxTdfaRegs)) return f;
    f.ok = true;
    return f;
}
// registers the fold adds per line: one per distinct set (0: nothing to fold, or the table keeps its general programs)
uint32_t tdfaFoldRegs(const TdfaTables& t) {
    const TdfaFold f = planTdfaFold(t);
    return f.ok ? uint32_t(f.sets.size()) : 0u;
}

size_t tdfaBlobBytesEstimate(const TdfaTables& t);

// ---- fixed-distance registers (one-stamp pair tables, LC_TDFA_PAIR=2).  In a log-format regex the end of field k is stamped on
// the separator byte and the start of field k+1 on the byte behind it -- ALWAYS: two real stamps in two consecutive bytes, on every
// line, four or five times per line.  A table that stamps once per byte PAIR cannot carry both when they fall into one pair.  Such
// a register b is not stamped at all: it reads as a + 1 at the end of the line (device_tables.h TP_OFF_DERIVE).  Proved on the
// automaton, for a table without general register programs (stamps(t) = the registers transition t sets to its position: one
// register, the members of a folded set, or none):
//   P1  every live transition t2 with b in stamps(t2) leaves a state that is no entry state and ALL of whose live incoming
//       transitions t1 have a in stamps(t1); and a is not in stamps(t2)                 (b is only ever stamped one byte behind a)
//   P2  every live transition t1 with a in stamps(t1) leads to a state s whose final row -- if the line may end there -- does not
//       read b, and every live transition t2 out of s has b in stamps(t2), or a in stamps(t2), or leads to a state from which no line
//       can end with b in the final row unless a is stamped again on the way      (behind the LAST stamp of a, b follows)
// Then, wherever a line ends with b in the final map, b == a + 1: take the LAST stamp of a, at p (there is one: b was stamped, P1).
// The line does not end there; the transition behind it cannot stamp a; if it stamped neither, the line could not end with b in the
// final row without another stamp of a -- so it stamps b = p + 1; and a later stamp of b would have a stamp of a in front of it
// (P1), later than p.  (A greedy field that may contain its own separator -- "([^\"]*) (\S*)\"" -- restamps a at every separator
// and b behind it, except behind a second separator in a row or a byte the next field cannot start with: stale values that the
// "or a" and the "unless a is stamped again" clauses cover.)  (live: between states the start state reaches, not into the
// dead state.)  -> derive[k] = (b, a), in an order in which every a is settled before it is used.
struct TdfaDerive {
    std::vector<std::pair<uint8_t, uint8_t>> pairs;  // (b, a): b = a + 1
    std::vector<bool> derived;                       // per register
};
static TdfaDerive planTdfaDerive(const TdfaTables& t, const TdfaFold& fold) {
    TdfaDerive d;
    d.derived.assign(t.nRegs, false);
    const size_t nLists = t.opsStart.size() - 1;
    // stamps of a list: bit set of registers (nRegs <= 250: four words)
    struct RegSet {
        uint64_t w[4] = {0, 0, 0, 0};
        void add(uint32_t r) { w[r >> 6] |= 1ull << (r & 63); }
        bool has(uint32_t r) const { return (w[r >> 6] >> (r & 63)) & 1ull; }
        bool empty() const { return !(w[0] | w[1] | w[2] | w[3]); }
    };
    std::vector<RegSet> stamps(nLists);
    for (size_t id = 1; id < nLists; ++id) {
        const uint32_t at = t.opsStart[id];
        const uint32_t n = t.ops[at];
        for (uint32_t k = 0; k < n; ++k) {
            const uint16_t w = t.ops[at + 1 + k];
            if ((w >> 8) != kRegPos) return d;  // a copy: no derivation on tables with general programs
            stamps[id].add(w & 0xFFu);
        }
        if (n >= 2 && !(fold.ok && fold.listSet[id] >= 0)) return d;  // a multi-stamp list that is not folded stays general
    }
    const uint32_t nS = t.nStates, nC = t.nClasses;
    // reachable states and entry states
    std::vector<char> reach(nS, 0), entry(nS, 0);
    std::vector<uint32_t> stack;
    auto seed = [&](uint32_t s0) {
        if (s0 == 0 || s0 >= nS) return;
        entry[s0] = 1;
        if (!reach[s0]) {
            reach[s0] = 1;
            stack.push_back(s0);
        }
    };
    seed(t.startState);
    for (uint32_t s0 : t.startAfter) seed(s0);
    while (!stack.empty()) {
        const uint32_t s = stack.back();
        stack.pop_back();
        for (uint32_t c = 0; c < nC; ++c) {
            const uint32_t nx = t.trans[size_t(s) * nC + c] & 0xFFFFu;
            if (nx && !reach[nx]) {
                reach[nx] = 1;
                stack.push_back(nx);
            }
        }
    }
    // reverse adjacency between reachable states: (source, op list) per live transition
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> incoming(nS);
    for (uint32_t s = 1; s < nS; ++s) {
        if (!reach[s]) continue;
        for (uint32_t c = 0; c < nC; ++c) {
            const uint32_t e = t.trans[size_t(s) * nC + c];
            if (e & 0xFFFFu) incoming[e & 0xFFFFu].push_back({s, e >> 16});
        }
    }
    // per state: the intersection of the stamps of its live incoming transitions (entry states: empty), and whether it has any
    std::vector<RegSet> inAll(nS);
    std::vector<char> hasIn(nS, 0);
    for (uint32_t s = 1; s < nS; ++s) {
        if (!reach[s]) continue;
        for (uint32_t c = 0; c < nC; ++c) {
            const uint32_t e = t.trans[size_t(s) * nC + c];
            const uint32_t nx = e & 0xFFFFu;
            if (!nx) continue;
            const RegSet& st = stamps[e >> 16];
            if (!hasIn[nx]) {
                inAll[nx] = st;
                hasIn[nx] = 1;
            } else {
                for (int k = 0; k < 4; ++k) inAll[nx].w[k] &= st.w[k];
            }
        }
    }
    for (uint32_t s = 1; s < nS; ++s)
        if (entry[s]) inAll[s] = RegSet();
    // per state: the intersection of the stamps of its live outgoing transitions (diagnostics: LC_TDFA_PROBE)
    std::vector<RegSet> outAll(nS);
    std::vector<char> hasOut(nS, 0);
    for (uint32_t s = 1; s < nS; ++s) {
        if (!reach[s]) continue;
        for (uint32_t c = 0; c < nC; ++c) {
            const uint32_t e = t.trans[size_t(s) * nC + c];
            if (!(e & 0xFFFFu)) continue;
            const RegSet& st = stamps[e >> 16];
            if (!hasOut[s]) {
                outAll[s] = st;
                hasOut[s] = 1;
            } else {
                for (int k = 0; k < 4; ++k) outAll[s].w[k] &= st.w[k];
            }
        }
    }
    auto finalReads = [&](uint32_t s, uint32_t r) {
        const uint32_t fid = t.finalId[s];
        if (fid == 0xFFFFu) return false;
        for (uint32_t sl = 0; sl < t.nSlots; ++sl)
            if (t.finalMap[size_t(fid) * t.nSlots + sl] == r) return true;
        return false;
    };
    std::vector<int> base(t.nRegs, -1);  // derived b -> its a
    for (uint32_t b = 0; b < t.nRegs; ++b) {
        // P1: candidates = intersection over every transition that stamps b of inAll[source]; none of them may stamp the candidate
        RegSet cand;
        bool any = false, ok = true;
        RegSet alsoStamped;  // registers stamped together with b somewhere: not candidates
        for (uint32_t s = 1; s < nS && ok; ++s) {
            if (!reach[s]) continue;
            for (uint32_t c = 0; c < nC; ++c) {
                const uint32_t e = t.trans[size_t(s) * nC + c];
                if (!(e & 0xFFFFu) || !stamps[e >> 16].has(b)) continue;
                if (entry[s] || !hasIn[s]) {
                    ok = false;
                    break;
                }
                if (!any) {
                    cand = inAll[s];
                    any = true;
                } else {
                    for (int k = 0; k < 4; ++k) cand.w[k] &= inAll[s].w[k];
                }
                for (int k = 0; k < 4; ++k) alsoStamped.w[k] |= stamps[e >> 16].w[k];
            }
        }
        if (!ok || !any) continue;
        for (int k = 0; k < 4; ++k) cand.w[k] &= ~alsoStamped.w[k];
        for (uint32_t a = 0; a < t.nRegs && base[b] < 0; ++a) {
            if (a == b || !cand.has(a)) continue;
            // no cycle: a must not (transitively) read b
            bool cyc = false;
            for (int x = int(a), hops = 0; x >= 0 && hops <= int(t.nRegs); x = base[size_t(x)], ++hops)
                if (uint32_t(x) == b) cyc = true;
            if (cyc) continue;
            // P2.  stale[s]: from s a line can END in a state whose final row reads b WITHOUT a being stamped on the way (backward
            // reachability over the transitions that do not stamp a)
            std::vector<char> stale(nS, 0);
            {
                std::vector<uint32_t> work;
                for (uint32_t s = 1; s < nS; ++s)
                    if (reach[s] && finalReads(s, b)) {
                        stale[s] = 1;
                        work.push_back(s);
                    }
                while (!work.empty()) {
                    const uint32_t s2 = work.back();
                    work.pop_back();
                    for (const auto& in : incoming[s2]) {
                        if (stale[in.first] || stamps[in.second].has(a)) continue;
                        stale[in.first] = 1;
                        work.push_back(in.first);
                    }
                }
            }
            bool p2 = true;
            for (uint32_t s = 1; s < nS && p2; ++s) {
                if (!reach[s]) continue;
                for (uint32_t c = 0; c < nC && p2; ++c) {
                    const uint32_t e = t.trans[size_t(s) * nC + c];
                    const uint32_t nx = e & 0xFFFFu;
                    if (!nx || !stamps[e >> 16].has(a)) continue;
                    if (finalReads(nx, b)) p2 = false;  // (the line may end right behind the stamp of a)
                    for (uint32_t c2 = 0; c2 < nC && p2; ++c2) {
                        const uint32_t e2 = t.trans[size_t(nx) * nC + c2];
                        const uint32_t nx2 = e2 & 0xFFFFu;
                        if (!nx2) continue;
                        const RegSet& st2 = stamps[e2 >> 16];
                        if (!st2.has(b) && !st2.has(a) && stale[nx2]) p2 = false;
                    }
                }
            }
            if (p2) base[b] = int(a);
        }
    }
    // order: every a settled before it is used
    std::vector<char> done(t.nRegs, 0);
    for (uint32_t r = 0; r < t.nRegs; ++r) done[r] = base[r] < 0;
    for (bool progress = true; progress;) {
        progress = false;
        for (uint32_t b = 0; b < t.nRegs; ++b)
            if (!done[b] && done[size_t(base[b])]) {
                d.pairs.push_back({uint8_t(b), uint8_t(base[b])});
                d.derived[b] = true;
                done[b] = 1;
                progress = true;
            }
    }
    return d;
}
}  // namespace lcregex

// Workgroup size of the standard tables, and whether they fold their multi-stamp programs: the fold's extra registers must not
// cost a smaller workgroup (or the fit).  Sets re->tdfaBlock (0 = does not fit) and re->tdfaPackedRegs.
static bool lcPickTdfaBlockAndFold(lc_regex* re) {
    const uint32_t est = uint32_t(lcregex::tdfaBlobBytesEstimate(re->tdfa));
    const uint32_t foldRegs = lcregex::tdfaFoldRegs(re->tdfa);
    re->tdfaBlock = lcTdfaPickBlock(est, re->tdfa.nRegs);
    const bool fold = foldRegs && re->tdfaBlock && lcTdfaPickBlock(est, re->tdfa.nRegs + foldRegs) == re->tdfaBlock;
    re->tdfaPackedRegs = re->tdfa.nRegs + (fold ? foldRegs : 0u);
    return fold;
}

// Round 4: the STANDARD tables (small batches: event groups, the multi-tenant launch, the fused pipeline) take the one-stamp byte-pair
// table by default under the conditions the compact tables take it under -- a full-match pattern without general register programs,
// at most 2 % DOUBLE entries -- and only if the workgroup size chosen without it still leaves two workgroups per CU with it.  A small
// launch is a dependent chain of one LDS lookup per byte per line; the pair table halves the chain.  LC_TDFA_PAIR (any value) keeps
// the round-3 behaviour (the environment decides), LC_TDFA_STD_PAIR=0 switches this default off.
static void lcTryStandardPairTable(lc_regex* re, bool fold) {
    static const bool off = [] {
        const char* e = getenv("LC_TDFA_STD_PAIR");
        return e && e[0] == '0';
    }();
    // (LC_TDFA_STREAM=0, the A/B knob that sends every launch to the phase-separated kernel, cannot run one-stamp tables: the
    // standard blob then keeps its single-byte tables instead of making default patterns fail)
    static const bool streamOff = [] {
        const char* e = getenv("LC_TDFA_STREAM");
        return e && e[0] == '0';
    }();
    if (off || streamOff || getenv("LC_TDFA_PAIR") || !re->tdfa.startAfter.empty() || !re->tdfaBlock) return;
    try {
        std::vector<uint32_t> blob = lcregex::packTdfaBlob(re->tdfa, re->tdfaBlock, false, false, fold, 2);
        const uint32_t po = blob[TD_OFF_PAIR];
        if (!po || blob[po / 4 + TP_FORMAT] != 1) return;  // (general programs left, or the pair table is too large)
        if (lcTdfaLdsBytes(uint32_t(blob.size() * 4), re->tdfaPackedRegs, re->tdfaBlock) > kLcLdsPerCu / 2) return;
        const uint32_t cols = re->tdfa.nClasses + 1;
        const uint32_t* pair = blob.data() + blob[po / 4 + TP_BASE] / 4;
        const size_t entries = size_t(re->tdfa.nStates) * cols * cols;
        size_t doubles = 0;
        for (size_t i = 0; i < entries; ++i) doubles += pair[i] >> 31;
        if (doubles * 100 > entries * 2) return;
        re->tdfaBlob.swap(blob);
    } catch (const lcregex::RegexError&) {
    }
}

namespace lcregex {
size_t tdfaBlobBytesEstimate(const TdfaTables& t) {
    auto pad = [](size_t n) { return (n + 15) & ~size_t(15); };
    size_t n = TD_TRANS_OFFSET + pad(size_t(t.nStates) * (t.nClasses + 1) * 4) + pad(t.finalId.size() * 2) +
               pad(t.finalMap.size()) + pad(t.opsStart.size() * 4) + pad(t.ops.size() * 2) + pad(t.startAfter.size() * 4) +
               pad(size_t(kMaxTdfaRegs) * 4 + 4);  // (+ the fold words of folded register programs)
    const size_t pairBytes = size_t(t.nStates) * (t.nClasses + 1) * (t.nClasses + 1) * 4;
    const char* pairEnv = getenv("LC_TDFA_PAIR");
    if (pairEnv && (pairEnv[0] == '1' || pairEnv[0] == '2') && pairBytes <= TP_MAX_TABLE_BYTES)
        n += 512 + pad(pairBytes) + 32 + pad(size_t(kMaxTdfaRegs) * 4 + 4);  // packTdfaBlob's byte-pair extension (+ derive words)
    return n;
}

// Tables for a COMPACT kernel variant (tdfa_kernel.hpp), chosen when the pattern is compiled.  By default the 256-lane
// variant (class-indexed rows, 16-bit registers: 16 instead of 12 waves per CU on the headline regex, +4 % measured) is
// packed and the launcher uses it for large batches.  LC_TDFA_COMPACT=0: none.  LC_TDFA_COMPACT=256 / 512: that workgroup
// size, used for every batch.  LC_TDFA_COMPACT=1024: byte-indexed rows shared by one 1024-lane workgroup per CU, for
// automata small enough to keep that table, the registers and the staging tiles in the CU's LDS (measured slower: the
// wide rows quadruple the LDS bank conflicts).
std::vector<uint32_t> packTdfaWideBlob(const TdfaTables& t, int* blockOut, bool* forcedOut, uint32_t* packedRegsOut) {
    const char* env = getenv("LC_TDFA_COMPACT");
    const int want = env ? atoi(env) : 256;
    *blockOut = 0;
    *forcedOut = env != nullptr;
    *packedRegsOut = t.nRegs;
    const uint32_t foldRegs = tdfaFoldRegs(t);
    // Default (neither LC_TDFA_COMPACT nor LC_TDFA_PAIR set), round 3: small automata whose byte-pair table is a ONE-STAMP table with
    // few DOUBLE entries take 512 lanes and that table -- the capture stamp is issued once per two bytes (headline batch: 0.2025 ->
    // 0.1915 ms, profiles/round3_lab_pair1.txt).  "Few": a DOUBLE is settled by a read-modify-write behind its chunk, for the whole
    // wavefront; regex A keeps 0.7 % of its entries as DOUBLEs after the fixed-distance registers are derived (one-byte fields),
    // regex B 1.8 % (fields that may be empty on both sides of a separator).  Round 4: B's tables fit two workgroups per CU once the
    // unused op lists were dropped, and measured on the GPU the pair table pays for B too (0.2060 -> 0.1944 ms, frac 0.385 -> 0.408):
    // the threshold went from 1 % to 2 %.
    // (full-match patterns only -- the parse processor's regex_match: that is what the GPU validation of the round covered; search
    // patterns, i.e. Grok entries and the Go regex plugin, keep the single-byte tables unless LC_TDFA_PAIR=2 asks)
    if (!env && !getenv("LC_TDFA_PAIR") && t.startAfter.empty()) {
        for (int fold = foldRegs ? 1 : 0; fold >= 0; --fold) {
            const uint32_t packedRegs = t.nRegs + (fold ? foldRegs : 0u);
            try {
                if (size_t(packedRegs + 1) * 512 * 2 > TD_MAX_REG_AREA) continue;
                std::vector<uint32_t> blob = packTdfaBlob(t, 512, false, true, fold != 0, 2);
                const uint32_t po = blob[TD_OFF_PAIR];
                if (!po || blob[po / 4 + TP_FORMAT] != 1) break;  // (general programs, or the pair table is too large)
                if (lcTdfaCompactLdsBytes(uint32_t(blob.size() * 4), packedRegs, 512) * 2 > kLcLdsPerCu) break;  // two workgroups per CU
                const uint32_t cols = t.nClasses + 1;
                const uint32_t* pair = blob.data() + blob[po / 4 + TP_BASE] / 4;
                size_t doubles = 0;
                const size_t entries = size_t(t.nStates) * cols * cols;
                for (size_t i = 0; i < entries; ++i) doubles += pair[i] >> 31;
                // more than 2 % DOUBLE entries (LC_TDFA_PAIR_DOUBLE_PCT: the threshold, for A/B runs -- regex B sits at 1.8 %)
                static const size_t pct = [] {
                    const char* e = getenv("LC_TDFA_PAIR_DOUBLE_PCT");
                    return e ? size_t(atol(e)) : size_t(2);
                }();
                if (doubles * 100 > entries * pct) break;
                *blockOut = 512;
                *packedRegsOut = packedRegs;
                return blob;
            } catch (const RegexError&) {
                break;
            }
        }
    }
    // (the fold costs registers: tables that only fit without it are packed without it)
    for (int fold = foldRegs ? 1 : 0; fold >= 0; --fold) {
        const uint32_t packedRegs = t.nRegs + (fold ? foldRegs : 0u);
        try {
            if (want == 256 || want == 512) {
                if (size_t(packedRegs + 1) * size_t(want) * 2 > TD_MAX_REG_AREA) continue;
                std::vector<uint32_t> blob = packTdfaBlob(t, want, false, true, fold != 0);
                if (lcTdfaCompactLdsBytes(uint32_t(blob.size() * 4), packedRegs, want) > kLcLdsPerCu) continue;
                *blockOut = want;
                *packedRegsOut = packedRegs;
                return blob;
            }
            if (want != kLcTdfaWideBlock) return {};
            const uint64_t tableEnd = TD_TRANS_OFFSET + uint64_t(t.nStates) * 257 * 4;
            if (tableEnd > TD_MAX_TABLE_END) return {};
            if (lcTdfaWideRegBytes(packedRegs) > TD_MAX_REG_AREA) continue;
            const size_t rest = tdfaBlobBytesEstimate(t) - size_t(t.nStates) * (t.nClasses + 1) * 4;  // everything but the rows
            if (lcTdfaWideLdsBytes(uint32_t(size_t(t.nStates) * 257 * 4 + rest + 64), packedRegs) > kLcLdsPerCu) continue;
            std::vector<uint32_t> blob = packTdfaBlob(t, kLcTdfaWideBlock, true, true, fold != 0);
            if (lcTdfaWideLdsBytes(uint32_t(blob.size() * 4), packedRegs) > kLcLdsPerCu) continue;
            *blockOut = kLcTdfaWideBlock;
            *packedRegsOut = packedRegs;
            return blob;
        } catch (const RegexError&) {
            return {};
        }
    }
    return {};
}

std::vector<uint32_t> packTdfaBlob(const TdfaTables& t, int block, bool wide, bool compact, bool foldPrograms, int pairMode) {
    // wide: rows indexed by the byte itself (256 columns + identity); compact: 16-bit offset registers (tdfa_kernel.hpp)
    const uint32_t cols = (wide ? 256 : t.nClasses) + 1;  // + identity column
    const uint32_t rowBytes = cols * 4;
    if (TD_TRANS_OFFSET + uint64_t(t.nStates) * rowBytes > TD_MAX_TABLE_END)
        throw RegexError("tdfa: transition table exceeds the 64 KiB LDS window");
    if (t.nClasses > 63) throw RegexError("tdfa: more than 63 byte classes");
    const TdfaFold fold = foldPrograms ? planTdfaFold(t) : TdfaFold{false, {}, std::vector<int>(t.opsStart.size() - 1, -1)};
    const uint32_t nRegsPacked = t.nRegs + (fold.ok ? uint32_t(fold.sets.size()) : 0u);
    const uint32_t dummyReg = nRegsPacked;  // one past the real (and the folded-set) registers
    const uint32_t regStride = uint32_t(block) * (compact ? 2 : 4);
    if (uint64_t(dummyReg + 1) * regStride > TD_MAX_REG_AREA) throw RegexError("tdfa: register file exceeds 64 KiB");
    const size_t nLists = t.opsStart.size() - 1;
    // "one register = pos" is folded into the transition word; everything else stays a list
    std::vector<uint32_t> field(nLists, 0);
    field[0] = dummyReg * regStride;
    for (size_t id = 1; id < nLists; ++id) {
        const uint32_t at = t.opsStart[id];
        const uint32_t n = t.ops[at];
        const uint16_t w0 = t.ops[at + 1];
        if (n == 1 && (w0 >> 8) == kRegPos) {
            field[id] = uint32_t(w0 & 0xFF) * regStride;
        } else if (fold.ok && fold.listSet[id] >= 0) {
            field[id] = (t.nRegs + uint32_t(fold.listSet[id])) * regStride;  // the set's own register
        } else {
            if (id > TD_MAX_LISTS) throw RegexError("tdfa: too many register programs");
            field[id] = (uint32_t(id) << 1) | TD_OP_GENERAL;
        }
    }
    BlobWriter w;
    w.reserve(TD_HEADER_WORDS * 4);
    std::vector<uint8_t> cmap(256);
    for (int b = 0; b < 256; ++b) cmap[size_t(b)] = uint8_t(t.classMap[size_t(b)] * 4);
    std::vector<uint32_t> trans(size_t(t.nStates) * cols);
    auto rowAddr = [&](uint32_t state) { return TD_TRANS_OFFSET + state * rowBytes; };
    for (uint32_t s = 0; s < t.nStates; ++s) {
        for (uint32_t c = 0; c + 1 < cols; ++c) {
            const uint32_t e = t.trans[size_t(s) * t.nClasses + (wide ? t.classMap[c] : c)];
            trans[size_t(s) * cols + c] = rowAddr(e & 0xFFFF) | (field[e >> 16] << 16);
        }
        trans[size_t(s) * cols + cols - 1] = rowAddr(s) | (field[0] << 16);  // identity column
    }
    uint32_t hdr[TD_HEADER_WORDS] = {};
    hdr[TD_MAGIC] = TD_MAGIC_VALUE;
    hdr[TD_NSTATES] = t.nStates;
    hdr[TD_NCLASSES] = t.nClasses;
    hdr[TD_NREGS] = nRegsPacked + 1;  // low half; the high half (fold words, "no general program" flag) is set below
    hdr[TD_NSLOTS] = t.nSlots;
    hdr[TD_START_ROW] = rowAddr(t.startState);
    hdr[TD_ROW_BYTES] = rowBytes;
    hdr[TD_ID_COL] = (cols - 1) * 4;
    hdr[TD_BLOCK] = uint32_t(block);
    const uint32_t cmapAt = w.put(cmap);
    const uint32_t transAt = w.put(trans);
    if (cmapAt != TD_CMAP_OFFSET || transAt != TD_TRANS_OFFSET) throw RegexError("tdfa: internal layout error");
    if (!t.startAfter.empty()) {
        std::vector<uint32_t> rows;
        for (uint32_t st : t.startAfter) rows.push_back(rowAddr(st));
        hdr[TD_OFF_STARTAFTER] = w.put(rows);
    }
    hdr[TD_OFF_FINALID] = w.put(t.finalId);
    hdr[TD_OFF_FINALMAP] = w.put(t.finalMap);
    {
        // (no transition of this blob names a general program -- every list is one stamp, or folded into its set's register: the
        // lists themselves are never read; 200-300 bytes of LDS that decide whether regex B's pair tables fit two workgroups per CU)
        bool anyList = false;
        for (size_t id = 1; id < nLists; ++id) anyList = anyList || (field[id] & TD_OP_GENERAL);
        if (anyList) {
            hdr[TD_OFF_OPSSTART] = w.put(t.opsStart);
            hdr[TD_OFF_OPS] = w.put(t.ops);
        } else {
            hdr[TD_OFF_OPSSTART] = w.put(std::vector<uint32_t>{0u, 0u});
            hdr[TD_OFF_OPS] = w.put(std::vector<uint16_t>{0});
        }
    }
    // ---- byte-pair extension: one dependent LDS lookup per TWO bytes.  Only for small automata (log-format regexes
    // have 10-20 byte classes and a few dozen states): the table grows with (classes+1)^2.
    const uint64_t pairRowBytes = uint64_t(cols) * cols * 4;
    const uint64_t pairBytes = pairRowBytes * t.nStates;
    // Opt-in (LC_TDFA_PAIR=1 when the pattern is compiled): measured on the headline corpus it buys ~10 % at equal
    // occupancy, but the 20 KiB table costs one of the three resident workgroups per CU.
    if (pairMode < 0) {
        const char* pairEnv = getenv("LC_TDFA_PAIR");
        pairMode = (pairEnv && (pairEnv[0] == '1' || pairEnv[0] == '2')) ? pairEnv[0] - '0' : 0;
    }
    const bool pairOne = pairMode == 2;  // ONE stamp per pair entry (device_tables.h TP1_*)
    bool anyGeneralList = false;
    for (size_t id = 1; id < nLists; ++id) anyGeneralList = anyGeneralList || (field[id] & TD_OP_GENERAL);
    if (!wide && (pairMode == 1 || (pairOne && !anyGeneralList)) && pairBytes <= TP_MAX_TABLE_BYTES && dummyReg < TP_GENERAL) {
        std::vector<uint16_t> cmapA(256);
        for (int b = 0; b < 256; ++b) cmapA[size_t(b)] = uint16_t(t.classMap[size_t(b)] * cols * 4);
        const uint32_t cmapAOff = w.put(cmapA);
        const uint32_t pairBase = w.reserve(size_t(pairBytes));
        if (uint64_t(pairBase) + pairBytes <= TD_MAX_TABLE_END) {
            // registers that read as another one + 1 at the end of the line: their own stamps are dropped from the pair table
            const TdfaDerive derive = pairOne ? planTdfaDerive(t, fold) : TdfaDerive();
            std::vector<char> dropList(nLists, 0);
            if (pairOne) {
                for (size_t id = 1; id < nLists; ++id) {
                    const uint32_t at = t.opsStart[id];
                    if (t.ops[at] == 1 && (t.ops[at + 1] >> 8) == kRegPos && derive.derived[t.ops[at + 1] & 0xFFu]) dropList[id] = 1;
                }
            }
            // one single-byte step: next state + what it stamps (register index, dummy, or "general")
            auto step = [&](uint32_t s, uint32_t c, uint32_t& next, uint32_t& stamp) {
                if (c == t.nClasses) {  // identity class: bytes outside the line
                    next = s;
                    stamp = dummyReg;
                    return;
                }
                const uint32_t e = t.trans[size_t(s) * t.nClasses + c];
                next = e & 0xFFFF;
                const uint32_t f = field[e >> 16];
                stamp = (f & TD_OP_GENERAL) ? TP_GENERAL : dropList[e >> 16] ? dummyReg : f / regStride;
            };
            uint32_t* pair = reinterpret_cast<uint32_t*>(w.bytes.data() + pairBase);
            for (uint32_t s = 0; s < t.nStates; ++s)
                for (uint32_t c1 = 0; c1 < cols; ++c1)
                    for (uint32_t c2 = 0; c2 < cols; ++c2) {
                        uint32_t s1, s2, st1, st2;
                        step(s, c1, s1, st1);
                        step(s1, c2, s2, st2);
                        uint32_t hi;
                        if (!pairOne) {
                            hi = st1 | (st2 << 8);
                        } else if (st2 == dummyReg) {
                            hi = st1;  // (the first byte's register, or nothing)
                        } else if (st1 == dummyReg || st1 == st2) {
                            hi = st2 | (TP1_DELTA >> 16);
                        } else {
                            hi = st1 | (st2 << 8) | (TP1_DOUBLE >> 16);
                        }
                        pair[(size_t(s) * cols + c1) * cols + c2] = (pairBase + s2 * uint32_t(pairRowBytes)) | (hi << 16);
                    }
            std::vector<uint32_t> ph(TP_HEADER_WORDS, 0u);
            ph[TP_BASE] = pairBase;
            ph[TP_ROW_BYTES] = uint32_t(pairRowBytes);
            ph[TP_OFF_CMAPA] = cmapAOff;
            ph[TP_ID_A] = t.nClasses * cols * 4;
            if (pairOne) {
                ph[TP_FORMAT] = 1;
                if (!derive.pairs.empty()) {
                    std::vector<uint32_t> words{uint32_t(derive.pairs.size())};
                    for (const auto& ba : derive.pairs) words.push_back(uint32_t(ba.first) | (uint32_t(ba.second) << 8) | (1u << 16));
                    ph[TP_OFF_DERIVE] = w.put(words);
                }
            }
            hdr[TD_OFF_PAIR] = w.put(ph);
        }
    }
    if (fold.ok) {  // fold words (device_tables.h TD_NREGS): set register + up to three members per word
        std::vector<uint32_t> words{0};
        for (size_t si = 0; si < fold.sets.size(); ++si) {
            const auto& set = fold.sets[si];
            for (size_t k = 0; k < set.size(); k += 3) {
                uint32_t wd = (t.nRegs + uint32_t(si)) | 0xFFFFFF00u;
                for (size_t j = 0; j < 3 && k + j < set.size(); ++j)
                    wd = (wd & ~(0xFFu << (8 * (j + 1)))) | (uint32_t(set[k + j]) << (8 * (j + 1)));
                words.push_back(wd);
            }
        }
        words[0] = uint32_t(words.size() - 1);
        const uint32_t at = w.put(words);
        if (at / 16 > 0x1FFFu) throw RegexError("tdfa: internal layout error");
        hdr[TD_NREGS] |= (at / 16) << 16;
    }
    bool anyGeneral = false;
    for (size_t id = 1; id < nLists; ++id) anyGeneral = anyGeneral || (field[id] & TD_OP_GENERAL);
    if (!anyGeneral) hdr[TD_NREGS] |= TD_NREGS_NO_GENERAL;
    std::memcpy(w.bytes.data(), hdr, sizeof hdr);
    return w.finish(TD_TOTAL_BYTES);
}

// tdfa_l2_layout.h: the logical tables as they are, one after the other
std::vector<uint32_t> packTdfaL2Blob(const TdfaTables& t) {
    BlobWriter w;
    w.reserve(TL_HEADER_WORDS * 4);
    w.put(t.classMap);
    uint32_t hdr[TL_HEADER_WORDS] = {};
    hdr[TL_MAGIC] = TL_MAGIC_VALUE;
    hdr[TL_NSTATES] = t.nStates;
    hdr[TL_NCLASSES] = t.nClasses;
    hdr[TL_NREGS] = t.nRegs;
    hdr[TL_NSLOTS] = t.nSlots;
    hdr[TL_START] = t.startState;
    hdr[TL_OFF_TRANS] = w.put(t.trans);
    hdr[TL_OFF_OPSSTART] = w.put(t.opsStart);
    hdr[TL_OFF_OPS] = w.put(t.ops);
    hdr[TL_OFF_FINALID] = w.put(t.finalId);
    hdr[TL_OFF_FINALMAP] = w.put(t.finalMap);
    if (!t.startAfter.empty()) hdr[TL_OFF_STARTAFTER] = w.put(t.startAfter);
    {
        std::vector<uint64_t> quiet(t.nStates, 0);
        for (uint32_t st = 1; st < t.nStates; ++st)
            for (uint32_t c = 0; c < t.nClasses && c < 64; ++c)
                if (t.trans[size_t(st) * t.nClasses + c] == st) quiet[st] |= uint64_t(1) << c;
        hdr[TL_OFF_QUIET] = w.put(quiet);
    }
    hdr[TL_MISS] = t.missState;
    for (uint32_t st = 1; st < t.nStates && !hdr[TL_ABSORB]; ++st) {
        bool self = t.finalId[st] != 0xFFFF && st != t.missState;  // (the MISS sink of a lazy automaton decides nothing)
        for (uint32_t c = 0; c < t.nClasses && self; ++c) self = t.trans[size_t(st) * t.nClasses + c] == st;  // (next = st, program 0)
        if (self) hdr[TL_ABSORB] = st;
    }
    std::memcpy(w.bytes.data(), hdr, sizeof hdr);
    return w.finish(TL_TOTAL_BYTES);
}

std::vector<uint32_t> packNfaBlob(const FollowNfa& nfa, std::vector<uint8_t>& classMapOut) {
    if (nfa.atomicCount > 255) throw RegexError("nfa: more than 255 atomic group instances");
    const int npos = int(nfa.positions.size());
    classMapOut.assign(256, 0);
    std::map<std::vector<bool>, int> sig2cls;
    std::vector<unsigned> rep;
    for (unsigned b = 0; b < 256; ++b) {
        std::vector<bool> sig;
        for (int p = 0; p < npos; ++p) sig.push_back(nfa.positions[size_t(p)].has(b));
        for (const auto& a : nfa.asserts) sig.push_back(a.set.has(b));
        auto it = sig2cls.find(sig);
        if (it == sig2cls.end()) {
            it = sig2cls.emplace(sig, int(rep.size())).first;
            rep.push_back(b);
        }
        classMapOut[b] = uint8_t(it->second);
    }
    if (rep.size() > 128) throw RegexError("nfa: more than 128 byte classes (" + std::to_string(rep.size()) + ")");
    for (const auto& lst : nfa.follow)
        if (lst.size() > 64) throw RegexError("nfa: follow list longer than 64 paths");
    // per-position class masks: 2 words per position, 4 when the pattern tells more than 64 byte classes apart
    typedef unsigned __int128 ClassMask;
    const size_t mw = rep.size() > 64 ? 4 : 2;
    auto putMask = [mw](std::vector<uint32_t>& v, size_t p, ClassMask m) {
        for (size_t k = 0; k < mw; ++k) v[p * mw + k] = uint32_t(m >> (32 * k));
    };
    std::vector<uint32_t> posMask(size_t(npos + 1) * mw, 0);
    for (int p = 0; p < npos; ++p) {
        ClassMask m = 0;
        for (size_t c = 0; c < rep.size(); ++c)
            if (nfa.positions[size_t(p)].has(rep[c])) m |= ClassMask(1) << c;
        putMask(posMask, size_t(p), m);
    }
    std::vector<uint32_t> stableMask(size_t(npos + 1) * mw, 0);
    for (int p = 0; p < npos; ++p) {
        ClassMask m = 0;
        for (size_t c = 0; c < rep.size(); ++c) {
            int passing = 0;
            bool steady = true;
            for (const auto& path : nfa.follow[size_t(p)]) {
                if (path.target < 0 || !nfa.positions[size_t(path.target)].has(rep[c])) continue;
                ++passing;
                const bool clean = path.cond == 0 && path.atoms.empty();
                if (passing == 1) {  // the move of highest priority: the thread's own tag-free self loop
                    if (path.target != p || path.tags.any() || !clean) steady = false;
                } else {
                    // Search patterns: further moves may lead to the wrapper's suffix position (the match could end here:
                    // "%{GREEDYDATA:message}" at the end of a Grok pattern spawns such a thread at EVERY byte of the
                    // message).  That thread ranks right below this one, replaces the one spawned a byte earlier (same
                    // position, higher rank) and is consulted only once this thread is gone -- and the step that ends
                    // this thread spawns it afresh from the same, unchanged captures (the path is unconditional and the
                    // suffix position takes every byte).  At the end of the line this thread's own MATCH path outranks
                    // it.  So the steps in between need not happen.
                    if (path.target != nfa.searchSuffix || !clean) steady = false;
                }
            }
            if (passing >= 1 && steady) m |= ClassMask(1) << c;
        }
        // a path that LEAVES an atomic group acts even when its target cannot take the byte (leaving commits the group):
        // such a position is never in a steady state.  (Entering a group on a path that goes nowhere has no effect.)
        for (const auto& path : nfa.follow[size_t(p)])
            for (const auto& ev : path.atoms)
                if (ev.code < 0) m = 0;
        putMask(stableMask, size_t(p), m);
    }
    if (npos >= 0xFFFF) throw RegexError("nfa: more than 65534 positions");
    // ---- doomed spawns (device_tables.h NF_OFF_QUASI).  cont[t]: the byte classes some follow path of t takes (MATCH paths take no
    // byte; assertions are ignored: a superset).  For (p, c) that is not stable: exactly one passing path is p's clean self loop, the
    // others are plain (no atomic events) moves to positions t != p; then the classes d outside every cont[t] are the bytes behind
    // which the spawned threads are gone again.  A position that leaves an atomic group on some path is never steady (as above).
    std::vector<uint16_t> quasiIdx(size_t(npos) + 1, 0);
    std::vector<uint32_t> quasiRows;
    {
        static const bool off = getenv("LC_NFA_NO_QUASI") != nullptr;  // (A/B measurements)
        std::vector<ClassMask> cont(size_t(npos), 0);
        const ClassMask allClasses = rep.size() >= 128 ? ~ClassMask(0) : ((ClassMask(1) << rep.size()) - 1);
        for (int t = 0; t < npos; ++t)
            for (const auto& g : nfa.follow[size_t(t)])
                if (g.target >= 0)
                    for (size_t d = 0; d < rep.size(); ++d)
                        if (nfa.positions[size_t(g.target)].has(rep[d])) cont[size_t(t)] |= ClassMask(1) << d;
        // (a spawned thread that LEAVES an atomic group on its next step commits the group and ends the threads behind it -- p's self
        // loop among them -- whatever byte follows: such a spawn is not "gone again", found by the forced-engine device fuzz of round 6,
        // '(?>a+?.)' on "aa1", profiles/round6_bt_fuzz_gpu.txt D)
        std::vector<char> leavesPos(size_t(npos), 0);
        for (int t = 0; t < npos; ++t)
            for (const auto& path : nfa.follow[size_t(t)])
                for (const auto& ev : path.atoms)
                    if (ev.code < 0) leavesPos[size_t(t)] = 1;
        for (int p = 0; p < npos && !off; ++p) {
            if (leavesPos[size_t(p)]) continue;
            std::vector<ClassMask> row(rep.size(), 0);
            bool any = false;
            for (size_t c = 0; c < rep.size(); ++c) {
                if ((ClassMask(stableMask[size_t(p) * mw]) | (ClassMask(stableMask[size_t(p) * mw + 1]) << 32) |
                     (mw == 4 ? (ClassMask(stableMask[size_t(p) * mw + 2]) << 64) | (ClassMask(stableMask[size_t(p) * mw + 3]) << 96) : 0)) >> c & 1)
                    continue;  // stable already
                int selfLoops = 0;
                bool plain = true;
                ClassMask survive = 0;
                for (const auto& path : nfa.follow[size_t(p)]) {
                    if (path.target < 0 || !nfa.positions[size_t(path.target)].has(rep[c])) continue;
                    if (!path.atoms.empty()) plain = false;
                    if (path.target == p) {
                        if (path.tags.any() || path.cond != 0) plain = false;  // (a tagged or conditional self loop is a real move)
                        ++selfLoops;
                    } else {
                        if (leavesPos[size_t(path.target)]) plain = false;
                        survive |= cont[size_t(path.target)];
                    }
                }
                if (!plain || selfLoops != 1) continue;
                row[c] = allClasses & ~survive;
                any = any || row[c] != 0;
            }
            if (!any) continue;
            quasiIdx[size_t(p)] = uint16_t(quasiRows.size() / (rep.size() * mw) + 1);
            for (size_t c = 0; c < rep.size(); ++c)
                for (size_t k = 0; k < mw; ++k) quasiRows.push_back(uint32_t(row[c] >> (32 * k)));
            if (quasiRows.size() / (rep.size() * mw) >= 0xFFFE) break;
        }
    }
    // aux entries: (cond, tags words...) padded to 4 words -- 8 / 16 when the pattern has more than 64 / 128 capture slots
    const size_t aw = nfa.slotCount() > 128 ? 16 : (nfa.slotCount() > 64 ? 8 : 4);
    std::vector<uint32_t> followStart, paths, events, aux(aw, 0);  // aux entry 0 = (no cond, no tags)
    std::map<std::tuple<uint32_t, TagSet>, uint32_t> auxIndex;
    std::map<std::vector<uint32_t>, uint32_t> eventSeqs;
    std::vector<uint32_t> atomicPos(size_t(npos) / 32 + 2, 0);  // bit p: some path out of position p enters/leaves a group
    for (int p = 0; p <= npos; ++p) {
        followStart.push_back(uint32_t(paths.size() / 2));
        for (const auto& path : nfa.follow[size_t(p)]) {
            uint32_t a = 0;
            if (path.cond || path.tags.any()) {
                auto it = auxIndex.find({path.cond, path.tags});
                if (it == auxIndex.end()) {
                    if (aux.size() / aw >= 0xFFFF) throw RegexError("nfa: too many distinct tag sets");
                    it = auxIndex.emplace(std::make_tuple(path.cond, path.tags), uint32_t(aux.size() / aw)).first;
                    aux.push_back(path.cond);
                    for (size_t k = 0; k + 1 < aw; ++k) aux.push_back(path.tags.word32(k));
                }
                a = it->second;
            }
            paths.push_back((path.target == kMatchTarget ? 0xFFFFu : uint32_t(path.target)) | (a << 16));
            uint32_t evWord = 0;
            if (nfa.atomicCount) {
                if (path.atoms.size() > 255) throw RegexError("nfa: atomic event list too long");
                std::vector<uint32_t> seq;
                for (const auto& ev : path.atoms) {
                    if (ev.visit > 0xFFFF) throw RegexError("nfa: too many atomic exits on one follow list");
                    seq.push_back(uint32_t(uint16_t(int16_t(ev.code))) | (uint32_t(ev.visit) << 16));
                    if (ev.code < kAssertEvent) atomicPos[size_t(p) / 32] |= 1u << (p % 32);
                }
                auto it = eventSeqs.find(seq);  // identical event sequences are stored once
                if (it == eventSeqs.end()) {
                    if (events.size() >= (1u << 24)) throw RegexError("nfa: atomic event list too long");
                    it = eventSeqs.emplace(seq, uint32_t(events.size())).first;
                    events.insert(events.end(), seq.begin(), seq.end());
                }
                evWord = (it->second << 8) | uint32_t(seq.size());
            }
            paths.push_back(evWord);
        }
    }
    followStart.push_back(uint32_t(paths.size() / 2));
    BlobWriter w;
    w.reserve(NF_HEADER_WORDS * 4);
    uint32_t hdr[NF_HEADER_WORDS] = {};
    hdr[NF_MAGIC] = NF_MAGIC_VALUE;
    hdr[NF_NPOS] = uint32_t(npos);
    hdr[NF_NSLOTS] = uint32_t(nfa.slotCount());
    hdr[NF_NCLASSES] = uint32_t(rep.size());
    hdr[NF_NPATHS] = uint32_t(paths.size() / 2);
    hdr[NF_CONDS_USED] = nfa.condsUsed;
    hdr[NF_SEARCH] = nfa.searchPrefix == 0 ? 1u : 0u;
    hdr[NF_SUFFIX] = (nfa.searchSuffix >= 0 && nfa.searchSuffix == npos - 1) ? 1u : 0u;
    hdr[NF_MASK_WORDS] = uint32_t(mw);
    hdr[NF_AUX_WORDS] = uint32_t(aw);
    hdr[NF_OFF_CLASSMAP] = w.put(classMapOut);
    hdr[NF_OFF_POSMASK] = w.put(posMask);
    hdr[NF_OFF_FOLLOWSTART] = w.put(followStart);
    hdr[NF_OFF_PATHS] = w.put(paths);
    hdr[NF_OFF_AUX] = w.put(aux);
    hdr[NF_OFF_STABLE] = w.put(stableMask);
    if (!quasiRows.empty()) {
        const uint32_t rowsAt = w.put(quasiRows);
        std::vector<uint16_t> block(4, 0);  // u32 rows | u32 offset of the rows | idx...
        const uint32_t nRows = uint32_t(quasiRows.size() / (rep.size() * mw));
        block[0] = uint16_t(nRows & 0xFFFF);
        block[1] = uint16_t(nRows >> 16);
        block[2] = uint16_t(rowsAt & 0xFFFF);
        block[3] = uint16_t(rowsAt >> 16);
        block.insert(block.end(), quasiIdx.begin(), quasiIdx.end());
        hdr[NF_OFF_QUASI] = w.put(block);
    }
    // look assertions per byte class: behind[c] / ahead[c] = cond bits that hold when the previous / next byte has
    // class c; entry nClasses = START / END
    std::vector<uint32_t> behind(rep.size() + 1), ahead(rep.size() + 1);
    for (size_t c = 0; c < rep.size(); ++c) {
        behind[c] = nfa.behindBits(int(rep[c]));
        ahead[c] = nfa.aheadBits(int(rep[c]));
    }
    behind[rep.size()] = nfa.behindBits(kEdge);
    ahead[rep.size()] = nfa.aheadBits(kEdge);
    hdr[NF_OFF_BEHIND] = w.put(behind);
    hdr[NF_OFF_AHEAD] = w.put(ahead);
    if (nfa.atomicCount) {
        // touchy[p] bit c: on byte class c a thread on position p needs the ordered commit pass -- some path out of p
        // leaves a group (acts whatever the byte is) or enters one on its way to a position that takes class c
        std::vector<uint32_t> touchy(size_t(npos + 1) * mw, 0);
        for (int p = 0; p <= npos; ++p) {
            ClassMask m = 0;
            for (const auto& path : nfa.follow[size_t(p)]) {
                bool enters = false, leaves = false;
                for (const auto& ev : path.atoms) {
                    enters |= ev.code > 0 && ev.code < kAssertEvent;
                    leaves |= ev.code < 0;
                }
                if (leaves) m = ~ClassMask(0);
                if (enters && path.target >= 0)
                    for (size_t c = 0; c < rep.size(); ++c)
                        if (nfa.positions[size_t(path.target)].has(rep[c])) m |= ClassMask(1) << c;
            }
            putMask(touchy, size_t(p), m);
        }
        hdr[NF_OFF_TOUCHY] = w.put(touchy);
        if (events.empty()) events.push_back(0);
        hdr[NF_ATOMIC] = uint32_t(nfa.atomicCount);
        hdr[NF_OFF_EVENTS] = w.put(events);
        hdr[NF_OFF_ATOMICPOS] = w.put(atomicPos);
    }
    // ---- follow lists by byte class (device_tables.h NF_OFF_CSTART): behind everything the LDS kernels stage
    hdr[NF_STAGE_BYTES] = uint32_t((w.bytes.size() + 15) & ~size_t(15));
    {
        // MEASURED AND LEFT OFF (LC_NFA_CLASS_LISTS=1 packs them; read when a pattern is compiled): on the Grok entries that are the long
        // poles of a batch a step has ~14 candidates in ONE election round either way -- what a step costs there is the latency of its
        // own ~650 instructions, not the number of candidates (profiles/round5_grok_steps.txt); the lists add up to 1.6 MB per pattern.
        const char* on = getenv("LC_NFA_CLASS_LISTS");
        const bool off = !(on && on[0] == '1');
        const size_t nc = rep.size();
        std::vector<uint32_t> cstart, cpaths;
        bool fits = !off;
        if (fits) {
            cstart.reserve(size_t(npos + 1) * nc + 1);
            for (int p = 0; p <= npos && fits; ++p) {
                const uint32_t base = followStart[size_t(p)];
                for (size_t c = 0; c < nc; ++c) {
                    cstart.push_back(uint32_t(cpaths.size()));
                    uint32_t k = 0;
                    for (const auto& path : nfa.follow[size_t(p)]) {
                        if (path.target >= 0 && nfa.positions[size_t(path.target)].has(rep[c])) cpaths.push_back(base + k);
                        ++k;
                    }
                }
                fits = (cstart.size() + cpaths.size()) * 4 <= (size_t(16) << 20);
            }
            cstart.push_back(uint32_t(cpaths.size()));
        }
        if (fits) {
            if (cpaths.empty()) cpaths.push_back(0);
            hdr[NF_OFF_CSTART] = w.put(cstart);
            hdr[NF_OFF_CPATHS] = w.put(cpaths);
        }
    }
    std::memcpy(w.bytes.data(), hdr, sizeof hdr);
    return w.finish(NF_TOTAL_BYTES);
}

}  // namespace lcregex

using namespace lcregex;

// Leftmost-first SEARCH as a whole-line match:  search(re)  ==  regex_match( (?s:.*?)(re)(?s:.*) )
// The lazy prefix makes the earliest start win, backtracking priority inside `re` is untouched, the greedy
// suffix swallows the rest.  Group 1 becomes the whole match, the pattern's own groups shift up by one.
// (Go processor_regex without FullMatch: plugins/processor/regex/regex.go:105-129; regexp2.FindStringMatch in
// plugins/processor/grok/processor_grok.go:156.)
static void shiftCaptures(Node& n) {
    if ((n.kind == Node::Group || n.kind == Node::BackRef || n.kind == Node::Cond) && n.capture) ++n.capture;
    for (auto& k : n.kids) shiftCaptures(*k);
}
// The longest byte string every match of the sub-expression must contain (possibly empty).  A line without it cannot
// match, which lets an ordered-pattern-list caller (Grok) skip the automaton for most (line, pattern) pairs.
// Returns the best literal found inside `n`; `run` is the literal run being built across a concatenation.
static std::string requiredLiteral(const Node& n) {
    auto singleByte = [](const Node& k, unsigned& b) {
        if (k.kind != Node::Set) return false;
        int count = 0;
        for (int w = 0; w < 4; ++w) count += __builtin_popcountll(k.set.w[w]);
        if (count != 1) return false;
        for (unsigned v = 0; v < 256; ++v)
            if (k.set.has(v)) b = v;
        return true;
    };
    auto better = [](const std::string& a, const std::string& b) { return b.size() > a.size() ? b : a; };
    unsigned b = 0;
    switch (n.kind) {
        case Node::Set: return singleByte(n, b) ? std::string(1, char(b)) : std::string();
        case Node::Cat: {
            std::string best, run;
            for (const auto& k : n.kids) {
                if (singleByte(*k, b)) {
                    run.push_back(char(b));
                    best = better(best, run);
                    continue;
                }
                if (k->kind == Node::Assert || k->kind == Node::Empty) continue;  // zero-width: the run goes on
                run.clear();
                best = better(best, requiredLiteral(*k));
            }
            return best;
        }
        case Node::Group:
        case Node::Atomic: return requiredLiteral(*n.kids[0]);
        case Node::Repeat: return n.min >= 1 ? requiredLiteral(*n.kids[0]) : std::string();
        default: return std::string();  // Alt: no branch is certain; Assert / Empty: nothing consumed
    }
}

// anchored: no lazy prefix -- (re)(?s:.*): the search wrapper for a match that must start at the first byte (same group layout:
// group 1 = the whole match)
static void wrapForSearch(ParsedRegex& re, bool anchored = false) {
    shiftCaptures(*re.root);
    auto anyStar = [](bool greedy) {
        auto set = std::make_unique<Node>();
        set->kind = Node::Set;
        set->set = ByteSet::all();
        auto rep = std::make_unique<Node>();
        rep->kind = Node::Repeat;
        rep->min = 0;
        rep->max = -1;
        rep->greedy = greedy;
        rep->kids.push_back(std::move(set));
        return rep;
    };
    auto whole = std::make_unique<Node>();
    whole->kind = Node::Group;
    whole->capture = 1;
    whole->kids.push_back(std::move(re.root));
    auto cat = std::make_unique<Node>();
    cat->kind = Node::Cat;
    if (!anchored) cat->kids.push_back(anyStar(false));
    cat->kids.push_back(std::move(whole));
    cat->kids.push_back(anyStar(true));
    re.root = std::move(cat);
    re.groupCount += 1;
    re.groupNames.insert(re.groupNames.begin() + 1, std::string());
}

// regex_search(match_continuous): the match must start at the first byte and may end anywhere.  As a whole-line match:
// (re)(?s:.*) -- leftmost-first picks the same (preferred) way through `re` that the backtracker finds first.
static void wrapForPrefix(ParsedRegex& re) {
    auto set = std::make_unique<Node>();
    set->kind = Node::Set;
    set->set = ByteSet::all();
    auto rest = std::make_unique<Node>();
    rest->kind = Node::Repeat;
    rest->min = 0;
    rest->max = -1;
    rest->greedy = true;
    rest->kids.push_back(std::move(set));
    auto cat = std::make_unique<Node>();
    cat->kind = Node::Cat;
    cat->kids.push_back(std::move(re.root));
    cat->kids.push_back(std::move(rest));
    re.root = std::move(cat);
}

static void setErr(char* err, size_t cap, const std::string& msg) {
    if (err && cap) std::snprintf(err, cap, "%s", msg.c_str());
}

// ---- prefix screen ---------------------------------------------------------------------------------------------
// A pattern whose tagged DFA is far too large for LDS (Grok log formats: > 65 000 states) still has a PREFIX that is
// small: the first k elements of its top-level concatenation, captures dropped.  Every match of the pattern contains a
// match of that prefix, so "the prefix occurs somewhere in the line" is a necessary condition that the fast TDFA kernel
// can test (status only) before the slow NFA kernel is asked for the captures.
static std::unique_ptr<Node> cloneWithoutCaptures(const Node& n) {
    auto c = std::make_unique<Node>();
    if (n.kind == Node::Assert && n.window) return c;  // (a look-ahead window only narrows the matches: a screen may ignore it)
    c->kind = n.kind;
    c->set = n.set;
    c->min = n.min;
    c->max = n.max;
    c->greedy = n.greedy;
    c->capture = 0;
    c->look = n.look;
    for (const auto& k : n.kids) c->kids.push_back(cloneWithoutCaptures(*k));
    return c;
}

// yes/no DFA blob (screen_kernel_layout.h) from the state graph of `t` (register programs, if any, are ignored: the states of a
// tagged DFA recognise the same language): class map, accept flags, u16 next-state table.  Needs nStates <= 65535.
static std::vector<uint32_t> packScreenBlob(const TdfaTables& t) {
    uint32_t sink = 0xFFFFFFFFu;
    for (uint32_t st = 1; st < t.nStates && sink == 0xFFFFFFFFu; ++st) {
        bool self = t.finalId[st] != 0xFFFF;
        for (uint32_t c = 0; c < t.nClasses && self; ++c) self = (t.trans[size_t(st) * t.nClasses + c] & 0xFFFF) == st;
        if (self) sink = st;
    }
    BlobWriter w;
    w.reserve(SC_HEADER_WORDS * 4);
    w.put(t.classMap);
    std::vector<uint8_t> accept(t.nStates, 0);
    for (uint32_t st = 1; st < t.nStates; ++st) accept[st] = t.finalId[st] != 0xFFFF;
    uint32_t hdr[SC_HEADER_WORDS] = {};
    hdr[SC_MAGIC] = SC_MAGIC_VALUE;
    hdr[SC_NSTATES] = t.nStates;
    hdr[SC_NCLASSES] = t.nClasses;
    hdr[SC_START] = t.startState;
    hdr[SC_SINK] = sink;
    hdr[SC_OFF_ACCEPT] = w.put(accept);
    std::vector<uint16_t> table(size_t(t.nStates) * t.nClasses);
    for (size_t i = 0; i < table.size(); ++i) table[i] = uint16_t(t.trans[i] & 0xFFFF);
    hdr[SC_OFF_TABLE] = w.put(table);
    std::memcpy(w.bytes.data(), hdr, sizeof hdr);
    return w.finish(SC_TOTAL_BYTES);
}

// compile one candidate screen (a sub-expression every match of the pattern must contain) as a status-only search; nullptr if
// it is nullable (screens nothing) or its automaton is too large
static lc_regex* compileScreenNode(std::unique_ptr<Node> node, uint32_t syntax_flags, uint32_t maxStates, size_t maxBlobBytes,
                                   const std::string& description, uint32_t k, uint32_t n, bool plainDfa = false) {
    auto re = std::make_unique<lc_regex>();
    try {
        ParsedRegex sub;
        sub.root = std::move(node);
        sub.groupCount = 0;
        sub.groupNames.emplace_back();
        {
            FollowNfa probe = buildFollowNfa(sub);
            for (const auto& path : probe.follow[size_t(probe.startIndex())])
                if (path.target == kMatchTarget) return nullptr;
        }
        wrapForSearch(sub);
        re->nfa = buildFollowNfa(sub);
        re->nfa.searchPrefix = 0;
        re->nfa.searchSuffix = int(re->nfa.positions.size()) - 1;
        TdfaLimits lim;
        lim.maxStates = maxStates;
        if (plainDfa) {  // yes/no DFA, table in global memory (screen_kernel.hpp): no LDS window to respect
            lim.ldsWindow = false;
            re->tdfa = buildScreenDfa(re->nfa, lim);
            const TdfaTables& t = re->tdfa;
            const size_t tableBytes = size_t(t.nStates) * t.nClasses * 2;
            if (getenv("LC_RELAX_DEBUG"))
                fprintf(stderr, "  screen dfa: %u states, %u classes, %zu table bytes\n", t.nStates, t.nClasses, tableBytes);
            if (tableBytes > maxBlobBytes || t.nStates > 0xFFFF) return nullptr;
            re->screenBlob = packScreenBlob(t);
            re->engine = LC_ENGINE_TDFA;
            re->pattern = description;
            re->syntaxFlags = syntax_flags | LC_SYNTAX_SEARCH;
            re->tdfaHeader = {t.nStates, t.nClasses, t.nRegs, t.nSlots, t.startState, k, n, 0};
            return re.release();
        }
        re->tdfa = buildTdfa(re->nfa, lim);
        if (getenv("LC_RELAX_DEBUG"))
            fprintf(stderr, "  screen automaton: %u states, %u classes, %zu table bytes\n", re->tdfa.nStates, re->tdfa.nClasses,
                    tdfaBlobBytesEstimate(re->tdfa));
        if (tdfaBlobBytesEstimate(re->tdfa) > maxBlobBytes) return nullptr;
        const bool fold = lcPickTdfaBlockAndFold(&*re);
        if (!re->tdfaBlock) return nullptr;
        re->tdfaBlob = packTdfaBlob(re->tdfa, re->tdfaBlock, false, false, fold);
        re->tdfaWideBlob = packTdfaWideBlob(re->tdfa, &re->tdfaWideBlock, &re->tdfaWideForced, &re->tdfaWidePackedRegs);
        re->hasTdfa = true;
        // the same automaton as a plain yes/no DFA: the merged screen launch of the Grok matcher (grok_plan_kernel.hpp) walks
        // every entry's screen in this one format
        if (re->tdfa.nStates <= 0xFFFF) re->screenBlob = packScreenBlob(re->tdfa);
        re->engine = LC_ENGINE_TDFA;
        re->pattern = description;
        re->syntaxFlags = syntax_flags | LC_SYNTAX_SEARCH;
        re->tdfaHeader = {re->tdfa.nStates, re->tdfa.nClasses, re->tdfa.nRegs, re->tdfa.nSlots, re->tdfa.startState, k, n, 0};
        return re.release();
    } catch (const RegexError& e) {
        if (getenv("LC_RELAX_DEBUG")) fprintf(stderr, "  screen refused: %s\n", e.what());
        return nullptr;
    }
}

static ParsedRegex parseForScreen(const char* pattern, size_t len, uint32_t syntax_flags) {
    Syntax syn;
    syn.icase = syntax_flags & LC_SYNTAX_ICASE;
    syn.dotAll = !(syntax_flags & LC_SYNTAX_NO_DOTALL);
    syn.multiLine = !(syntax_flags & LC_SYNTAX_NO_MULTILINE);
    syn.extended = syntax_flags & LC_SYNTAX_EXTENDED;
    syn.namedOnly = true;
    syn.regexp2 = syntax_flags & LC_SYNTAX_REGEXP2;
    return parseRegex(std::string_view(pattern, len), syn);
}

// ---- relaxed screen -----------------------------------------------------------------------------------------------
// The WHOLE pattern, made small by accepting more: captures, assertions and atomic brackets dropped, long counters opened
// ({0,62} -> *), and every sub-expression that still needs more than `budget` positions -- the IPv6|IPv4|hostname
// alternations, lists of month names or firewall actions -- replaced by  F C*  (F = the bytes it can start with, C = all
// bytes it can contain; C* when it can be empty).  Every match of the pattern is a match of the relaxed pattern, so a line
// without one cannot match: a necessary condition over the whole line, where the prefix screen only sees the first elements
// ("... connection denied from ..." passes a prefix that ends at "connection" and fails here, where a number must follow).
namespace {
struct Relaxer {
    size_t budget;
    static size_t positions(const Node& n) {
        switch (n.kind) {
            case Node::Set: return 1;
            case Node::Empty:
            case Node::Assert: return 0;
            case Node::Repeat: {
                const size_t k = n.kids.empty() ? 0 : positions(*n.kids[0]);
                const size_t copies = size_t(n.max < 0 ? std::max(n.min, 1) : std::max(n.max, 1));
                return std::min<size_t>(k * copies, 1u << 20);
            }
            default: {
                size_t t = 0;
                for (const auto& k : n.kids) t += positions(*k);
                return std::min<size_t>(t, 1u << 20);
            }
        }
    }
    static bool nullable(const Node& n) {
        switch (n.kind) {
            case Node::Set: return false;
            case Node::Empty:
            case Node::Assert: return true;
            case Node::Repeat: return n.min == 0 || n.kids.empty() || nullable(*n.kids[0]);
            case Node::Alt:
                for (const auto& k : n.kids)
                    if (nullable(*k)) return true;
                return n.kids.empty();
            default:  // Cat, Group, Atomic
                if (n.kind == Node::Group && n.runCapture) return true;
                for (const auto& k : n.kids)
                    if (!nullable(*k)) return false;
                return true;
        }
    }
    static void bytes(const Node& n, ByteSet& acc) {
        if (n.kind == Node::Set) acc.unite(n.set);
        if (n.kind == Node::Group && n.runCapture) return;
        for (const auto& k : n.kids) bytes(*k, acc);
    }
    static void first(const Node& n, ByteSet& acc) {  // bytes a non-empty match of n can start with
        switch (n.kind) {
            case Node::Set: acc.unite(n.set); break;
            case Node::Empty:
            case Node::Assert: break;
            case Node::Alt:
                for (const auto& k : n.kids) first(*k, acc);
                break;
            case Node::Repeat:
                if (!n.kids.empty()) first(*n.kids[0], acc);
                break;
            default:
                if (n.kind == Node::Group && n.runCapture) break;
                for (const auto& k : n.kids) {
                    first(*k, acc);
                    if (!nullable(*k)) break;
                }
        }
    }
    static std::unique_ptr<Node> make(Node::Kind kind) {
        auto n = std::make_unique<Node>();
        n->kind = kind;
        return n;
    }
    static std::unique_ptr<Node> collapse(const Node& n) {
        ByteSet all;
        bytes(n, all);
        auto star = make(Node::Repeat);
        star->min = 0;
        star->max = -1;
        auto body = make(Node::Set);
        body->set = all;
        star->kids.push_back(std::move(body));
        if (nullable(n)) return star;
        auto head = make(Node::Set);
        first(n, head->set);
        auto cat = make(Node::Cat);
        cat->kids.push_back(std::move(head));
        cat->kids.push_back(std::move(star));
        return cat;
    }
    std::unique_ptr<Node> relax(const Node& n) const { return relax(n, budget); }
    std::unique_ptr<Node> relax(const Node& n, size_t budget) const {
        switch (n.kind) {
            case Node::Empty:
            case Node::Assert: return make(Node::Empty);
            case Node::Set: {
                auto c = make(Node::Set);
                c->set = n.set;
                return c;
            }
            case Node::Group:
                if (n.runCapture) return make(Node::Empty);
                [[fallthrough]];
            case Node::Atomic: return n.kids.empty() ? make(Node::Empty) : relax(*n.kids[0], budget);
            case Node::Cat: {
                auto c = make(Node::Cat);
                for (const auto& k : n.kids) c->kids.push_back(relax(*k, budget));
                return c;
            }
            case Node::Alt: {
                // too large as a whole: first coarsen the alternatives (two timestamp formats keep their shape, the month
                // names inside them go), only then give up the alternation itself
                for (size_t b = budget; b >= 3; b /= 2) {
                    auto c = make(Node::Alt);
                    for (const auto& k : n.kids) c->kids.push_back(relax(*k, b));
                    if (positions(*c) <= 4 * budget) return c;
                }
                return collapse(n);
            }
            case Node::Repeat: {
                if (n.kids.empty()) return make(Node::Empty);
                auto c = make(Node::Repeat);
                c->min = n.min;
                c->max = n.max;
                c->kids.push_back(relax(*n.kids[0], budget));
                // "[^\n]*" (Grok's DATA / GREEDYDATA) -> "(?s:.)*": a universal loop, which the screen's determinisation
                // (screen_dfa.cpp) uses to forget everything that lies before it
                if (c->max < 0 && c->min == 0 && c->kids[0]->kind == Node::Set) {
                    int cnt = 0;
                    for (unsigned b = 0; b < 256; ++b) cnt += c->kids[0]->set.has(b);
                    if (cnt >= 250) c->kids[0]->set = ByteSet::all();
                }
                if (positions(*c) > budget && (c->max < 0 || c->max > 2)) {  // open the counter
                    c->min = std::min(c->min, 1);
                    c->max = -1;
                }
                if (positions(*c) > budget) return collapse(n);
                return c;
            }
        }
        return make(Node::Empty);
    }
};
}  // namespace

// (debugging aid, LC_RELAX_DEBUG: the relaxed pattern in a readable form)
static std::string dumpNode(const Node& n) {
    auto setStr = [](const ByteSet& s) {
        int cnt = 0;
        for (unsigned c = 0; c < 256; ++c) cnt += s.has(c);
        if (cnt == 256) return std::string(".");
        std::string o;
        if (cnt == 1) {
            for (unsigned c = 0; c < 256; ++c)
                if (s.has(c)) {
                    if (c >= 33 && c < 127) o += char(c);
                    else if (c == 32) o += "\\s";
                    else o += "\\x" + std::to_string(c);
                }
            return o;
        }
        o = "[";
        for (unsigned c = 0; c < 256;) {
            if (!s.has(c)) {
                ++c;
                continue;
            }
            unsigned e = c;
            while (e + 1 < 256 && s.has(e + 1)) ++e;
            auto ch = [](unsigned x) { return (x >= 33 && x < 127) ? std::string(1, char(x)) : "\\x" + std::to_string(x); };
            o += ch(c);
            if (e > c) o += "-" + ch(e);
            c = e + 1;
        }
        return o + "]";
    };
    switch (n.kind) {
        case Node::Empty: return "";
        case Node::Set: return setStr(n.set);
        case Node::Assert: return "(?A)";
        case Node::Repeat: {
            std::string k = n.kids.empty() ? "" : dumpNode(*n.kids[0]);
            return "(" + k + "){" + std::to_string(n.min) + "," + (n.max < 0 ? "" : std::to_string(n.max)) + "}";
        }
        case Node::Alt: {
            std::string o = "(";
            for (size_t i = 0; i < n.kids.size(); ++i) o += (i ? "|" : "") + dumpNode(*n.kids[i]);
            return o + ")";
        }
        default: {
            std::string o;
            for (const auto& k : n.kids) o += dumpNode(*k);
            return o;
        }
    }
}

lc_regex* lcCompileRelaxedScreen(const char* pattern, size_t len, uint32_t syntax_flags, uint32_t maxStates, size_t maxBlobBytes) {
    return lcCompileRelaxedScreenPreferring(pattern, len, syntax_flags, maxStates, maxBlobBytes, 0);
}

// preferStageBytes != 0 (round 6): the automaton whose [accept flags | table] block is at most this large is preferred -- the Grok plan
// stages such a screen into LDS (45 ns a byte) and walks a larger one through L2 (100 ns a byte: grok_device.hip kGrokScreenStageMax).
// The budgets go on shrinking behind the first automaton that fits at all; the first one that fits the preference wins, and if none
// does the first one is kept.  A smaller budget forgets more of the pattern: the screen lets more values through.
lc_regex* lcCompileRelaxedScreenPreferring(const char* pattern, size_t len, uint32_t syntax_flags, uint32_t maxStates, size_t maxBlobBytes,
                                           size_t preferStageBytes) {
    ParsedRegex parsed;
    try {
        parsed = parseForScreen(pattern, len, syntax_flags);
    } catch (const RegexError&) {
        return nullptr;
    }
    // LC_RELAX_BUDGET (tests): start at this budget instead of 96 -- small patterns get relaxed too
    size_t firstBudget = 96;
    if (const char* e = getenv("LC_RELAX_BUDGET")) firstBudget = size_t(std::max(1, atoi(e)));
    std::vector<size_t> budgets;
    for (size_t b = firstBudget; b >= 1 && budgets.size() < 5; b = b > 6 ? b / 2 : (b > 1 ? b - 1 : 0)) budgets.push_back(b);
    lc_regex* firstFit = nullptr;  // (preferStageBytes: the automaton of the largest budget that fits at all)
    for (size_t budget : budgets) {
        Relaxer rx{budget};
        std::unique_ptr<Node> node = rx.relax(*parsed.root);
        const size_t npos = Relaxer::positions(*node);
        if (getenv("LC_RELAX_DEBUG")) fprintf(stderr, "relaxed(budget %zu, %zu positions): %s\n", budget, npos, dumpNode(*node).c_str());
        if (npos > 3000) continue;
        if (lc_regex* re = compileScreenNode(std::move(node), syntax_flags, maxStates, maxBlobBytes,
                                             "<relaxed screen: budget " + std::to_string(budget) + ", " + std::to_string(npos) +
                                                 " positions>",
                                             uint32_t(budget), uint32_t(npos), true)) {
            const size_t stage = re->screenBlob.empty() ? 0 : size_t(re->screenBlob[SC_TOTAL_BYTES] - re->screenBlob[SC_OFF_ACCEPT]);
            if (!preferStageBytes || (stage && stage <= preferStageBytes)) {
                if (firstFit) lc_regex_free(firstFit);
                return re;
            }
            if (!firstFit) firstFit = re;
            else lc_regex_free(re);
        }
    }
    return firstFit;
}

lc_regex* lcCompilePrefixScreen(const char* pattern, size_t len, uint32_t syntax_flags, uint32_t maxStates,
                                size_t maxBlobBytes) {
    ParsedRegex parsed;
    try {
        parsed = parseForScreen(pattern, len, syntax_flags);
    } catch (const RegexError&) {
        return nullptr;
    }
    // the pattern as one flat concatenation: groups (captures are dropped anyway) and nested concatenations are spliced
    std::vector<const Node*> flat;
    std::function<void(const Node&)> flatten = [&](const Node& n) {
        if (n.kind == Node::Group && n.kids.size() == 1) flatten(*n.kids[0]);
        else if (n.kind == Node::Cat)
            for (const auto& k : n.kids) flatten(*k);
        else if (n.kind != Node::Empty) flat.push_back(&n);
    };
    flatten(*parsed.root);
    if (flat.size() < 2) return nullptr;  // nothing to cut
    auto tryScreen = [&](std::unique_ptr<Node> node, uint32_t k, const char* how) -> lc_regex* {
        return compileScreenNode(std::move(node), syntax_flags, maxStates, maxBlobBytes,
                                 std::string("<prefix screen: ") + how + ", " + std::to_string(k) + " of " +
                                     std::to_string(flat.size()) + " elements>",
                                 k, uint32_t(flat.size()));
    };
    auto prefixOf = [&](const std::vector<const Node*>& list, size_t k) {
        auto cat = std::make_unique<Node>();
        cat->kind = Node::Cat;
        for (size_t i = 0; i < k && i < list.size(); ++i) cat->kids.push_back(cloneWithoutCaptures(*list[i]));
        return cat;
    };
    // 1. a long prefix of the concatenation; halve until the automaton is small enough
    for (size_t k = std::min<size_t>(flat.size() - 1, 48); k >= 1; k = k / 2)
        if (lc_regex* re = tryScreen(prefixOf(flat, k), uint32_t(k), "concatenation")) return re;
    // 2. the pattern starts with an alternation that is too large as a whole (a syslog line starts with one of two
    //    timestamp formats): a match starts with a match of SOME alternative, hence with a prefix of that alternative
    if (flat[0]->kind == Node::Alt) {
        std::vector<std::vector<const Node*>> alts;
        for (const auto& kid : flat[0]->kids) {
            std::vector<const Node*> saved;
            saved.swap(flat);
            flatten(*kid);
            alts.push_back(flat);
            flat.swap(saved);
        }
        for (size_t budget = 16; budget >= 1; budget /= 2) {
            auto alt = std::make_unique<Node>();
            alt->kind = Node::Alt;
            for (const auto& list : alts) alt->kids.push_back(prefixOf(list, std::max<size_t>(1, std::min(list.size(), budget))));
            if (lc_regex* re = tryScreen(std::move(alt), uint32_t(budget), "leading alternation")) return re;
        }
    }
    return nullptr;
}

extern "C" lc_regex_t* lc_regex_compile_screen(const char* pattern, size_t pattern_len, uint32_t syntax_flags,
                                               uint32_t max_states, size_t max_table_bytes) {
    if (!pattern) return nullptr;
    try {
        return lcCompilePrefixScreen(pattern, pattern_len, syntax_flags, max_states, max_table_bytes);
    } catch (const std::exception&) {
        return nullptr;
    }
}

extern "C" lc_regex_t* lc_regex_compile_relaxed_screen(const char* pattern, size_t pattern_len, uint32_t syntax_flags,
                                                       uint32_t max_states, size_t max_table_bytes) {
    if (!pattern) return nullptr;
    try {
        return lcCompileRelaxedScreen(pattern, pattern_len, syntax_flags, max_states, max_table_bytes);
    } catch (const std::exception&) {
        return nullptr;
    }
}

void lcPreferWaveTdfa(lc_regex* re) {
    if (!re || re->engine != LC_ENGINE_TDFA) return;
    if (re->tdfaL2Blob.empty() && re->hasTdfa) {
        const size_t tableBytes = size_t(re->tdfa.nStates) * re->tdfa.nClasses * 4;
        if (tableBytes > (size_t(16) << 20) || size_t(re->tdfa.nRegs) * 64 * 4 > 64 * 1024) return;
        try {
            re->tdfaL2Blob = lcregex::packTdfaL2Blob(re->tdfa);
        } catch (const lcregex::RegexError&) {
            return;
        }
    }
    re->preferWave = !re->tdfaL2Blob.empty();
}

// ------------------------------------------------------------------------------------------------ lazy automata (round 6)
int lcRegexLazyTrain(lc_regex* re, const uint8_t* data, const uint32_t* off, const uint32_t* len, uint32_t n, uint64_t out[5]) {
    if (out)
        for (int i = 0; i < 5; ++i) out[i] = 0;
    if (!re || (n && (!data || !off || !len))) return LC_ERR_ARG;
    if (re->engine != LC_ENGINE_NFA || re->nfaBlob.empty()) return LC_OK;  // a complete automaton, or nothing to stand in front of
    LcLazyTdfa& Z = re->lazy;
    std::lock_guard<std::mutex> train(Z.trainMutex);
    {
        std::lock_guard<std::mutex> g(Z.m);
        if (Z.disabled) return LC_OK;
    }
    constexpr size_t kMaxSampleValues = 8192, kMaxSampleBytes = size_t(12) << 20;
    constexpr uint32_t kPerRound = 256;  // values added between two rebuilds
    Z.offered.fetch_add(n, std::memory_order_relaxed);
    // where a value ends on the current tables: 0 = dead, missState = undecided, anything else = decided
    auto endsInMiss = [&](const uint8_t* p, uint32_t L) {
        if (!Z.haveTables) return true;
        const lcregex::TdfaTables& t = Z.tables;
        uint32_t st = t.startState;
        for (uint32_t i = 0; i < L && st != 0 && st != t.missState; ++i) st = t.trans[size_t(st) * t.nClasses + t.classMap[p[i]]] & 0xFFFFu;
        return st == t.missState;
    };
    auto sampleFull = [&](uint32_t L) { return Z.sampleLen.size() >= kMaxSampleValues || Z.sampleData.size() + L > kMaxSampleBytes; };
    uint32_t stillMissing = 0;
    bool built = false;
    for (uint32_t next = 0, rounds = 0; rounds < 64; ++rounds) {
        // the offered values the tables do not decide join the sample, a few hundred at a time (most of a round's misses are decided
        // by what the round's first values add)
        uint32_t added = 0;
        stillMissing = 0;
        if (!Z.frozen)
            for (; next < n && added < kPerRound; ++next) {
                if (!endsInMiss(data + off[next], len[next])) continue;
                if (sampleFull(len[next])) {
                    ++stillMissing;
                    continue;
                }
                Z.sampleOff.push_back(uint32_t(Z.sampleData.size()));
                Z.sampleLen.push_back(len[next]);
                Z.sampleData.insert(Z.sampleData.end(), data + off[next], data + off[next] + len[next]);
                ++added;
            }
        if (!added) break;
        Z.kept.fetch_add(added, std::memory_order_relaxed);
        lcregex::TdfaLimits lim;
        lim.maxStates = 60000;
        lim.ldsWindow = false;
        lim.maxPathWork = uint64_t(1) << 31;   // (a step of a 3 000-position format looks at a few hundred paths: room for millions of steps)
        lim.maxCommitWork = uint64_t(1) << 30;
        lcregex::TdfaLazyGuide guide;
        guide.data = Z.sampleData.data();
        guide.off = Z.sampleOff.data();
        guide.len = Z.sampleLen.data();
        guide.n = uint32_t(Z.sampleLen.size());
        guide.maxTableBytes = size_t(8) << 20;
        lcregex::TdfaTables t;
        lcregex::TdfaLazyReport rep;
        bool failed = false;
        try {
            t = lcregex::buildTdfaLazy(re->nfa, lim, guide, &rep);
        } catch (const lcregex::RegexError&) {  // (register limit, layout limits: this pattern stays on the thread-list engine alone)
            failed = true;
        }
        if (failed || t.missState == 0 || t.nRegs > 250 || (rep.stopped && rep.transitionsComputed < 64)) {
            std::lock_guard<std::mutex> g(Z.m);
            Z.disabled = true;
            Z.blob.clear();
            re->lazyReady.store(false, std::memory_order_release);
            return LC_OK;
        }
        Z.tables = std::move(t);
        Z.haveTables = true;
        Z.report = rep;
        Z.frozen = rep.stopped;  // a limit: what is there stays in use, nothing is added any more
        built = true;  // (the values in front of `next` are decided by the new tables: they were decided before, or have just joined the sample)
    }
    if (built) {
        std::vector<uint32_t> blob = lcregex::packTdfaL2Blob(Z.tables);
        std::lock_guard<std::mutex> g(Z.m);
        Z.blob.swap(blob);
        ++Z.version;
        Z.builds.fetch_add(1, std::memory_order_relaxed);
        re->lazyReady.store(true, std::memory_order_release);
    }
    if (out && Z.haveTables) {
        out[0] = Z.tables.nStates;
        out[1] = Z.report.transitionsComputed;
        out[2] = Z.sampleLen.size();
        out[3] = stillMissing;
        out[4] = 1;
    }
    return LC_OK;
}

extern "C" int lc_regex_lazy_train(lc_regex_t* re, const uint8_t* data, const uint32_t* off, const uint32_t* len, uint32_t n, uint64_t out[5]) {
    return lcRegexLazyTrain(re, data, off, len, n, out);
}

extern "C" int lc_runtime_set_table_cache_dir(const char* dir) {
    lcregex::lcSetTableCacheDir(dir);
    return LC_OK;
}
extern "C" const char* lc_runtime_table_cache_stamp(void) { return lcregex::lcTableCacheStamp(); }
extern "C" void lc_runtime_table_cache_stats(uint64_t out[4]) {
    if (!out) return;
    const lcregex::TableCacheStats s = lcregex::lcTableCacheStats();
    out[0] = s.hits;
    out[1] = s.misses;
    out[2] = s.stored;
    out[3] = s.failuresRecalled;
}

extern "C" int lc_regex_prefer_wave_tdfa(lc_regex_t* re) {
    if (!re) return 0;
    lcPreferWaveTdfa(re);
    return re->preferWave ? 1 : 0;
}

extern "C" void lc_regex_atomic_groups(const lc_regex_t* re, uint32_t* kept, uint32_t* elided) {
    if (kept) *kept = re ? uint32_t(re->nfa.atomicCount) : 0u;
    if (elided) *elided = re ? re->atomicsElided : 0u;
}

// verdicts of global-memory constructions that failed on their limits (see lc_regex_compile); LC_TDFA_NO_FAILURE_MEMO: off
namespace {
std::mutex gTdfaFailureMutex;
std::unordered_map<std::string, std::string> gTdfaFailures;
bool tdfaFailureMemoOn() {
    static const bool on = getenv("LC_TDFA_NO_FAILURE_MEMO") == nullptr;
    return on;
}
bool lcRecallTdfaFailure(const std::string& key, std::string& suffix) {
    if (!tdfaFailureMemoOn()) return false;
    std::lock_guard<std::mutex> g(gTdfaFailureMutex);
    auto it = gTdfaFailures.find(key);
    if (it == gTdfaFailures.end()) return false;
    suffix = it->second;
    return true;
}
void lcRememberTdfaFailure(const std::string& key, const std::string& suffix) {
    if (!tdfaFailureMemoOn()) return;
    std::lock_guard<std::mutex> g(gTdfaFailureMutex);
    if (gTdfaFailures.size() < 1024) gTdfaFailures.emplace(key, suffix);  // (bounded: patterns come from configuration files)
}
}  // namespace

extern "C" int lc_regex_compile(const char* pattern, size_t pattern_len, uint32_t syntax_flags, int engine,
                                lc_regex_t** out, char* err, size_t errcap) {
    if (!pattern || !out || engine < LC_ENGINE_AUTO || (engine > LC_ENGINE_NFA && engine != LC_ENGINE_BT)) {
        setErr(err, errcap, "bad argument");
        return LC_ERR_ARG;
    }
    *out = nullptr;
    Syntax syn;
    syn.icase = syntax_flags & LC_SYNTAX_ICASE;
    syn.dotAll = !(syntax_flags & LC_SYNTAX_NO_DOTALL);
    syn.multiLine = !(syntax_flags & LC_SYNTAX_NO_MULTILINE);
    syn.extended = syntax_flags & LC_SYNTAX_EXTENDED;
    syn.namedOnly = syntax_flags & LC_SYNTAX_NAMED_ONLY;
    syn.regexp2 = syntax_flags & LC_SYNTAX_REGEXP2;
    auto re = new lc_regex();
    re->pattern.assign(pattern, pattern_len);
    re->syntaxFlags = syntax_flags;
    try {
        ParsedRegex parsed;
        try {
            parsed = parseRegex(std::string_view(pattern, pattern_len), syn);
        } catch (const RegexError& e) {
            const bool unsupported = std::strstr(e.what(), "unsupported") != nullptr;
            setErr(err, errcap, e.what());
            delete re;
            return unsupported ? LC_ERR_UNSUPPORTED : LC_ERR_SYNTAX;
        }
        // Not regular (back-references), or the caller asked for it: the device backtracking engine (bt_vm.hpp).  No automaton is
        // built; a search / prefix match is the same wrapped whole-value pattern the automata get.
        if (parsed.hasBackRef || parsed.hasGeneralLook || engine == LC_ENGINE_BT) {
            if (engine == LC_ENGINE_TDFA || engine == LC_ENGINE_NFA)
                throw RegexError("back-references and general look-arounds need the backtracking engine (LC_ENGINE_AUTO / LC_ENGINE_BT): unsupported on this engine");
            const bool anchored = (syntax_flags & LC_SYNTAX_SEARCH) && (syntax_flags & LC_SYNTAX_PREFIX);
            if (syntax_flags & LC_SYNTAX_SEARCH) wrapForSearch(parsed, anchored);
            else if (syntax_flags & LC_SYNTAX_PREFIX) wrapForPrefix(parsed);
            re->btBlob = lcregex::buildBtProgram(parsed, syn.icase);
            re->engine = LC_ENGINE_BT;
            re->nfa.groupCount = parsed.groupCount;  // (mark count and names are read off the handle's nfa member by every consumer)
            re->nfa.groupNames = parsed.groupNames;
            setErr(err, errcap, "");
            *out = re;
            return LC_OK;
        }
        // atomic groups that provably change nothing become plain groups (atomic_elide.cpp); LC_NO_ATOMIC_ELIDE: A/B measurements
        {
            static const bool off = getenv("LC_NO_ATOMIC_ELIDE") != nullptr;
            if (!off) re->atomicsElided = uint32_t(lcregex::elideRedundantAtomics(parsed));
        }
        re->requiredLiteral = requiredLiteral(*parsed.root);
        // LC_SYNTAX_SEARCH | LC_SYNTAX_PREFIX: the search whose match must START at the first byte -- what a search finds whenever
        // its leftmost match starts there, with the search's group layout (lc_regex_gpu.h)
        const bool anchoredSearch = (syntax_flags & LC_SYNTAX_SEARCH) && (syntax_flags & LC_SYNTAX_PREFIX);
        if (syntax_flags & LC_SYNTAX_SEARCH) wrapForSearch(parsed, anchoredSearch);
        else if (syntax_flags & LC_SYNTAX_PREFIX) wrapForPrefix(parsed);
        try {
            re->nfa = buildFollowNfa(parsed);
        } catch (const RegexError& nfaError) {
            // A tree the parser accepts and the position automaton cannot express ("(a*)*": an unbounded repeat of a body that may match
            // nothing): boost backtracks through it like through anything else, and so does the device backtracking engine -- unless the
            // caller asked for an automaton, or the pattern comes in the Go regex plugin's dialect (RE2 backtracks nowhere).  A Grok
            // handle that holds such an entry walks its list entry by entry (processor_grok_gpu.cpp: the speculative plan is automata).
            if (engine != LC_ENGINE_AUTO || (syn.regexp2 && !syn.namedOnly)) throw;
            try {
                re->btBlob = lcregex::buildBtProgram(parsed, syn.icase);
            } catch (const RegexError&) {
                throw nfaError;
            }
            re->engine = LC_ENGINE_BT;
            re->nfa = lcregex::FollowNfa();
            re->nfa.groupCount = parsed.groupCount;
            re->nfa.groupNames = parsed.groupNames;
            setErr(err, errcap, "");
            *out = re;
            return LC_OK;
        }
        if (syntax_flags & LC_SYNTAX_SEARCH) {  // wrapForSearch generates its prefix '.' first and its suffix '.' last
            re->nfa.searchPrefix = anchoredSearch ? -1 : 0;
            re->nfa.searchSuffix = int(re->nfa.positions.size()) - 1;
        }
        if (engine != LC_ENGINE_NFA) {
            try {
                TdfaLimits lim;
                if (const char* e = getenv("LC_TDFA_MAX_STATES")) lim.maxStates = uint32_t(atoi(e));
                if (const char* e = getenv("LC_TDFA_MAX_WORK")) lim.maxPathWork = uint64_t(atoll(e));
                static const bool probe = getenv("LC_TDFA_PROBE") != nullptr;  // size of the automaton, whatever becomes of it
                if (probe) {
                    const auto t0 = std::chrono::steady_clock::now();
                    std::string what = "ok";
                    TdfaTables t;
                    try {
                        TdfaLimits big = lim;
                        big.ldsWindow = false;
                        t = buildTdfa(re->nfa, big);
                    } catch (const RegexError& e) {
                        what = e.what();
                    }
                    fprintf(stderr, "tdfa-probe: positions %zu states %u classes %u regs %u lists %zu  %.2f s  %s\n",
                            re->nfa.positions.size(), t.nStates, t.nClasses, t.nRegs, t.opsStart.size(),
                            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), what.c_str());
                }
                re->tdfa = buildTdfa(re->nfa, lim);
                const bool fold = lcPickTdfaBlockAndFold(&*re);
                if (!re->tdfaBlock) throw RegexError("tdfa: tables + registers exceed the 160 KiB LDS of a CU");
                re->tdfaBlob = packTdfaBlob(re->tdfa, re->tdfaBlock, false, false, fold);
                lcTryStandardPairTable(&*re, fold);
                re->tdfaWideBlob = packTdfaWideBlob(re->tdfa, &re->tdfaWideBlock, &re->tdfaWideForced, &re->tdfaWidePackedRegs);
                re->hasTdfa = true;
                re->tdfaHeader = {re->tdfa.nStates, re->tdfa.nClasses, re->tdfa.nRegs, re->tdfa.nSlots,
                                  re->tdfa.startState, 0, 0, 0};
            } catch (const RegexError& e) {
                re->tdfaError = e.what();
                // Too large for the LDS kernels: the same automaton with its tables in global memory (tdfa_l2_kernel.hpp), as
                // long as it stays a few MB -- still one line per lane, two orders of magnitude faster than the NFA kernel.
                static const uint32_t l2States = [] {
                    const char* v = getenv("LC_TDFA_L2_MAX_STATES");
                    return uint32_t(v ? atoi(v) : 32768);
                }();
                bool l2 = false;
                // (a construction that ran out of WORK would run out of it again; one that ran out of states is redone with
                // the larger bound; a table that was built and only failed to pack -- byte classes, registers -- is kept)
                const bool outOfWork = re->tdfaError.find("work limit") != std::string::npos;
                const bool outOfStates = re->tdfaError.find("state limit") != std::string::npos;
                // A construction that failed on its limits fails the same way every time, and finding that out is the expensive part
                // of an anchored Grok format that does not determinise (5-50 s of a core each, 16 of the 50 entries of configs[2]):
                // the verdict is remembered for the life of the process, so that reloading a pipeline -- the agent does that in
                // place -- does not pay for it again.  Same return code, same message; nothing is remembered about successes.
                // (the limits that produce the verdict are part of the key: a process that changes LC_TDFA_L2_MAX_* between two
                // compiles must not get the old verdict back)
                const char* envWork = getenv("LC_TDFA_L2_MAX_WORK");
                const char* envCommit = getenv("LC_TDFA_L2_MAX_COMMIT");
                const std::string memoKey = re->pattern + '\0' + std::to_string(syntax_flags) + '/' + std::to_string(engine) + '/' +
                                            std::to_string(l2States) + '/' + (envWork ? envWork : "") + '/' + (envCommit ? envCommit : "");
                std::string remembered;
                if (l2States && !outOfWork && lcRecallTdfaFailure(memoKey, remembered)) {
                    re->tdfaError += remembered;
                } else if (l2States && !outOfWork) {
                    try {
                        if (outOfStates || re->tdfa.nStates == 0) {
                            TdfaLimits lim;
                            lim.maxStates = l2States;
                            lim.ldsWindow = false;
                            // (an anchored search is compiled because its automaton is wanted: it may cost seconds)
                            // (80 M: the anchored Grok formats of configs[2] that build at all need 24-69 M path steps and 2-5 s)
                            lim.maxPathWork = anchoredSearch ? (80u << 20) : (4u << 20);
                            // (the table format's state ids are 16 bits: CISCOFW713172 anchored needs 61 034 of them before
                            // minimisation; the formats that do not build even so pass 170 000-600 000)
                            if (anchoredSearch) lim.maxStates = std::max<uint32_t>(lim.maxStates, 65535);
                            if (const char* v = getenv("LC_TDFA_L2_MAX_WORK")) lim.maxPathWork = uint64_t(atoll(v));
                            if (const char* v = getenv("LC_TDFA_L2_MAX_COMMIT")) lim.maxCommitWork = uint64_t(atoll(v));
                            re->tdfa = buildTdfa(re->nfa, lim);
                        }
                        // (built without the window, MINIMISED, the automaton may fit the LDS kernels after all: dead stores gone,
                        // the states of a search pattern's tail collapse -- CATALINALOG 1 410 -> 136 states)
                        bool lds = false;
                        if (outOfStates) {
                            try {
                                const bool fold = lcPickTdfaBlockAndFold(&*re);
                                if (re->tdfaBlock) {
                                    re->tdfaBlob = packTdfaBlob(re->tdfa, re->tdfaBlock, false, false, fold);
                                    re->tdfaWideBlob =
                                        packTdfaWideBlob(re->tdfa, &re->tdfaWideBlock, &re->tdfaWideForced, &re->tdfaWidePackedRegs);
                                    re->hasTdfa = true;
                                    re->tdfaHeader = {re->tdfa.nStates, re->tdfa.nClasses, re->tdfa.nRegs, re->tdfa.nSlots,
                                                      re->tdfa.startState, 0, 0, 0};
                                    lds = l2 = true;
                                }
                            } catch (const RegexError&) {
                                re->tdfaBlob.clear();
                                re->tdfaWideBlob.clear();
                                re->hasTdfa = false;
                            }
                        }
                        const size_t tableBytes = size_t(re->tdfa.nStates) * re->tdfa.nClasses * 4;
                        if (!lds && tableBytes <= (size_t(16) << 20) && size_t(re->tdfa.nRegs) * 64 * 4 <= 64 * 1024) {
                            re->tdfaL2Blob = packTdfaL2Blob(re->tdfa);
                            re->tdfaHeader = {re->tdfa.nStates, re->tdfa.nClasses, re->tdfa.nRegs, re->tdfa.nSlots,
                                              re->tdfa.startState, 0, 0, 0};
                            l2 = true;
                        }
                    } catch (const RegexError& e2) {
                        const std::string suffix = std::string("; with its tables in global memory: ") + e2.what();
                        re->tdfaError += suffix;
                        // only what is expensive to find out again: a construction that ran into its state or work limit
                        if (suffix.find("limit") != std::string::npos) lcRememberTdfaFailure(memoKey, suffix);
                    }
                }
                if (!l2 && engine == LC_ENGINE_TDFA) throw RegexError(re->tdfaError);
            }
        }
        if (re->hasTdfa || !re->tdfaL2Blob.empty()) {
            re->engine = LC_ENGINE_TDFA;
        } else {
            re->engine = LC_ENGINE_NFA;
        }
        // the NFA blob is always packed when it fits: tests cross-check both engines on one handle
        try {
            re->nfaBlob = packNfaBlob(re->nfa, re->nfaClassMap);
            for (const auto& lst : re->nfa.follow)
                for (const auto& path : lst) {
                    uint32_t enters = 0, exits = 0;
                    for (const auto& ev : path.atoms) {
                        enters += ev.code > 0 && ev.code < kAssertEvent;
                        exits += ev.code < 0;
                    }
                    re->decideMaxEnter = std::max(re->decideMaxEnter, enters);
                    re->decideClosedCap = std::max(re->decideClosedCap, enters + exits);
                }
            // (a program that does not fit into LDS next to its scratch is read in place from HBM by the kernel)
            if (lcNfaLdsBytes(0, uint32_t(re->nfa.positions.size()), re->nfa.atomicCount > 0) > kLcLdsPerCu)
                throw RegexError("nfa: per-wave scratch exceeds the 160 KiB LDS of a CU");
        } catch (const RegexError&) {
            re->nfaBlob.clear();
            if (re->engine == LC_ENGINE_NFA) throw;
        }
    } catch (const RegexError& e) {
        setErr(err, errcap, e.what());
        delete re;
        return LC_ERR_UNSUPPORTED;
    } catch (const std::exception& e) {
        setErr(err, errcap, e.what());
        delete re;
        return LC_ERR_UNSUPPORTED;
    }
    setErr(err, errcap, "");
    *out = re;
    return LC_OK;
}

extern "C" void lc_regex_free(lc_regex_t* re) {
    if (!re) return;
    lcReleaseDeviceTables(re);
    delete re;
}

extern "C" int lc_regex_mark_count(const lc_regex_t* re) { return re ? re->nfa.groupCount : -1; }

extern "C" const char* lc_regex_group_name(const lc_regex_t* re, int g) {
    if (!re || g < 1 || g > re->nfa.groupCount) return nullptr;
    const std::string& s = re->nfa.groupNames[size_t(g)];
    return s.empty() ? nullptr : s.c_str();
}

static std::atomic<uint64_t> gGaveUpValues{0};
void lcNoteGaveUp(uint64_t n) { gGaveUpValues += n; }
extern "C" uint64_t lc_gave_up_values_total(void) { return gGaveUpValues.load(); }

extern "C" int lc_regex_prepare_span_filter(lc_regex_t* re) {
    if (!re) return LC_ERR_ARG;
    std::lock_guard<std::mutex> g(re->deviceMutex);
    if (!re->screenBlob.empty()) return LC_OK;
    // the state graph of the tagged DFA (LDS-sized or the large one kept for the L2 kernel) recognises the same language
    if ((!re->hasTdfa && re->tdfaL2Blob.empty()) || re->tdfa.nStates == 0 || re->tdfa.nStates > 0xFFFF) return LC_ERR_UNSUPPORTED;
    if ((re->syntaxFlags & (LC_SYNTAX_SEARCH | LC_SYNTAX_PREFIX)) != 0) return LC_ERR_UNSUPPORTED;  // regex_match semantics only
    re->screenBlob = packScreenBlob(re->tdfa);
    return LC_OK;
}

extern "C" int lc_regex_info(const lc_regex_t* re, lc_regex_info_t* out) {
    if (!re || !out) return LC_ERR_ARG;
    out->engine = re->engine;
    out->mark_count = re->nfa.groupCount;
    out->positions = uint32_t(re->nfa.positions.size());
    const bool tables = re->hasTdfa || !re->screenBlob.empty() || !re->tdfaL2Blob.empty();
    out->states = tables ? re->tdfa.nStates : 0;
    out->classes = tables ? re->tdfa.nClasses : (re->nfaBlob.empty() ? 0 : re->nfaBlob[NF_NCLASSES]);
    out->registers = tables ? re->tdfa.nRegs : 0;
    out->table_bytes = uint32_t((!re->hasTdfa && !re->screenBlob.empty() ? re->screenBlob.size()
                                 : !re->tdfaL2Blob.empty() ? re->tdfaL2Blob.size()
                                 : re->engine == LC_ENGINE_BT ? re->btBlob.size()
                                 : re->engine == LC_ENGINE_TDFA ? re->tdfaBlob.size() : re->nfaBlob.size()) * 4);
    return LC_OK;
}

extern "C" const uint8_t* lc_regex_required_literal(const lc_regex_t* re, size_t* len) {
    if (!re || !len) return nullptr;
    *len = re->requiredLiteral.size();
    return reinterpret_cast<const uint8_t*>(re->requiredLiteral.data());
}

extern "C" int lc_regex_run_captures(const lc_regex_t* re, int32_t* groups, uint8_t* sets, int cap) {
    if (!re) return 0;
    int n = 0;
    for (const auto& rg : re->nfa.runGroups) {
        if (n < cap && groups && sets) {
            groups[n] = rg.first;
            for (int b = 0; b < 32; ++b) sets[n * 32 + b] = uint8_t(rg.second.w[size_t(b) / 8] >> (8 * (b & 7)));
        }
        ++n;
    }
    return n;
}

extern "C" int lc_regex_table(const lc_regex_t* re, int which, const void** data, size_t* bytes) {
    if (!re || !data || !bytes) return LC_ERR_ARG;
    auto view = [&](const void* p, size_t n) {
        *data = p;
        *bytes = n;
        return LC_OK;
    };
    if (which == LC_TABLE_NFA_BLOB) {
        if (re->nfaBlob.empty()) return LC_ERR_ARG;
        return view(re->nfaBlob.data(), re->nfaBlob.size() * 4);
    }
    if (which == LC_TABLE_BT_BLOB) {
        if (re->btBlob.empty()) return LC_ERR_ARG;
        return view(re->btBlob.data(), re->btBlob.size() * 4);
    }
    if (which == LC_TABLE_LAZY_TDFA_BLOB) {
        lc_regex* mre = const_cast<lc_regex*>(re);
        std::lock_guard<std::mutex> g(mre->lazy.m);
        if (mre->lazy.blob.empty()) return LC_ERR_ARG;
        return view(mre->lazy.blob.data(), mre->lazy.blob.size() * 4);
    }
    if (!re->hasTdfa && re->screenBlob.empty() && re->tdfaL2Blob.empty()) return LC_ERR_ARG;  // (logical tables: all three)
    const TdfaTables& t = re->tdfa;
    if ((which == LC_TABLE_TDFA_BLOB || which == LC_TABLE_TDFA_WIDE_BLOB) && !re->hasTdfa) return LC_ERR_ARG;
    if (which == LC_TABLE_TDFA_L2_BLOB) {
        if (re->tdfaL2Blob.empty()) return LC_ERR_ARG;
        return view(re->tdfaL2Blob.data(), re->tdfaL2Blob.size() * 4);
    }
    switch (which) {
        case LC_TABLE_CLASSMAP: return view(t.classMap.data(), t.classMap.size());
        case LC_TABLE_TDFA_TRANS: return view(t.trans.data(), t.trans.size() * 4);
        case LC_TABLE_TDFA_OPSSTART: return view(t.opsStart.data(), t.opsStart.size() * 4);
        case LC_TABLE_TDFA_OPS: return view(t.ops.data(), t.ops.size() * 2);
        case LC_TABLE_TDFA_FINALID: return view(t.finalId.data(), t.finalId.size() * 2);
        case LC_TABLE_TDFA_FINALMAP: return view(t.finalMap.data(), t.finalMap.size());
        case LC_TABLE_TDFA_HEADER: return view(re->tdfaHeader.data(), re->tdfaHeader.size() * 4);
        case LC_TABLE_TDFA_STARTAFTER:
            if (t.startAfter.empty()) return LC_ERR_ARG;
            return view(t.startAfter.data(), t.startAfter.size() * 4);
        case LC_TABLE_TDFA_BLOB: return view(re->tdfaBlob.data(), re->tdfaBlob.size() * 4);
        case LC_TABLE_TDFA_WIDE_BLOB:
            if (re->tdfaWideBlob.empty()) return LC_ERR_ARG;
            return view(re->tdfaWideBlob.data(), re->tdfaWideBlob.size() * 4);
        default: return LC_ERR_ARG;
    }
}
