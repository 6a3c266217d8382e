// tdfa.hpp -- determinisation of the follow NFA into a tagged DFA (Laurikari-style TDFA, leftmost-first).
//
// One DFA state = ordered list of (NFA position, register map) items; order = backtracking priority.  Registers
// hold byte offsets of capture-group boundaries.  Register names are canonicalised per state (first appearance),
// so state identity is (positions, prev-byte context, register-sharing pattern) and the construction is finite.
// The device kernel (kernels.hip: tdfa_match_kernel) steps one log line per lane through `trans`, running the
// few register moves attached to a transition, and finishes with the per-state final map.
#pragma once

#include <cstdint>
#include <vector>

#include "follow_nfa.hpp"

namespace lcregex {

constexpr uint8_t kRegPos = 0xFF;   // "current offset" pseudo-register in op / final lists
constexpr uint8_t kRegNone = 0xFE;  // capture slot never written -> -1
constexpr int kMaxTdfaRegs = 250;

struct TdfaTables {
    uint32_t nStates = 0, nClasses = 0, nRegs = 0, nSlots = 0, startState = 0;
    std::vector<uint8_t> classMap;    // [256] byte -> class
    std::vector<uint32_t> trans;      // [nStates*nClasses]  low16 = next state (0 = dead), high16 = op-list id (0 = none)
    std::vector<uint32_t> opsStart;   // [nOpLists+1] offset of each op list in `ops` (list id 0 is the empty list)
    std::vector<uint16_t> ops;        // per list: n, then n words (dst | src<<8); src may be kRegPos
    std::vector<uint16_t> finalId;    // [nStates] 0xFFFF = not accepting, else row in finalMap
    std::vector<uint8_t> finalMap;    // [nFinal*nSlots] register id | kRegPos | kRegNone
    // search patterns only: [nClasses] state to RESUME a search in when the byte before the resume point has class c
    // (only the wrapper's prefix thread is alive, no register is set); empty otherwise
    std::vector<uint32_t> startAfter;
    // lazy form only (buildTdfaLazy): the state that stands for "a transition nobody has computed"; 0 = a complete automaton
    uint32_t missState = 0;
};

// buildTdfaLazy: the values the construction follows (host memory) and what it may grow to
struct TdfaLazyGuide {
    const uint8_t* data = nullptr;
    const uint32_t* off = nullptr;
    const uint32_t* len = nullptr;
    uint32_t n = 0;
    size_t maxTableBytes = 0;  // states x classes x 4 (0: 8 MiB)
    bool completeRows = true;  // every state a sample value has been in gets its whole row (tdfa.cpp)
};
struct TdfaLazyReport {
    uint32_t statesBuilt = 0, statesKept = 0;
    uint64_t transitionsComputed = 0, transitionsUnknown = 0, stepsWalked = 0;
    bool stopped = false;  // a limit ended the exploration: the table is as far as it got
};

struct TdfaLimits {
    uint32_t maxStates = 4096;
    // epsilon paths the construction may look at: the densest automata that still fit a table (and the Grok monsters
    // that fail on maxStates) stay under 0.7 M; "(a?){200}a{200}" would spend minutes before failing on the table size
    uint64_t maxPathWork = 8u << 20;
    // steps the ordered commit of atomic groups may take over the whole construction (4x what the densest pattern that still fits
    // an LDS table needs)
    uint64_t maxCommitWork = 20u << 20;
    // true: the transition table must also fit the 64 KiB LDS window of the lane-per-line kernels (rows of (classes+1) words)
    bool ldsWindow = true;
};

// Throws RegexError("tdfa: ...") when the automaton exceeds the limits (caller falls back to the NFA engine).
TdfaTables buildTdfa(const FollowNfa& nfa, const TdfaLimits& limits = TdfaLimits());

// The constructions themselves (buildTdfa / buildScreenDfa go through the table cache, table_cache.cpp, when a cache directory is set)
TdfaTables buildTdfaUncached(const FollowNfa& nfa, const TdfaLimits& limits);
// tdfa.cpp: the data-guided partial automaton (never cached: it depends on the sample)
TdfaTables buildTdfaLazy(const FollowNfa& nfa, const TdfaLimits& limits, const TdfaLazyGuide& guide, TdfaLazyReport* report = nullptr);
TdfaTables buildScreenDfaUncached(const FollowNfa& nfa, const TdfaLimits& limits);

// table_cache.cpp -- compiled automata across process restarts (round 5).  Determinising an anchored Grok format costs seconds (and
// finding out that one does NOT determinise within the limits costs as much: 16 of the 50 entries of BASELINE configs[2]); an agent
// that restarts, or reloads a pipeline in a new process, pays that again -- 8-10 s on the MI355X box's host.  With a cache directory
// set, every construction is looked up first under a hash of its INPUT -- the follow NFA itself (positions, paths, tags, conditions,
// atomic events, search wrapper), the limits and the stamp of this build of the library -- and stored afterwards: the tables, or the
// verdict of a construction that ran into its limits.  Same tables bit for bit (tools/table_snapshot.py); a file that is missing,
// short or from another build is ignored.  Process-wide; empty = off (default).  Also: environment LC_TABLE_CACHE_DIR.
void lcSetTableCacheDir(const char* dir);
const char* lcTableCacheDir();
struct TableCacheStats {
    uint64_t hits = 0, misses = 0, stored = 0, failuresRecalled = 0;
};
TableCacheStats lcTableCacheStats();
const char* lcTableCacheStamp();  // what stands for "this build's constructions" in every key (build.py: a hash of the shaping sources)

// tdfa.cpp: merges states that behave alike (same final row, same register programs into equivalent states); buildTdfa and
// buildScreenDfa end with it
void minimizeTdfaStates(TdfaTables& tables);

// screen_dfa.cpp: plain yes/no DFA (no registers, no thread order) for a pattern without assertions and atomic groups; positions
// that can only reach MATCH through a universal loop (?s:.)* that is already alive are forgotten, so "X.*Y.*Z" costs the sum, not
// the product, of its parts.  Same table format (one empty register program); throws RegexError on the limits.
TdfaTables buildScreenDfa(const FollowNfa& nfa, const TdfaLimits& limits = TdfaLimits());

}  // namespace lcregex
