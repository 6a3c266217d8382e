// regex_handle.hpp -- the object behind lc_regex_t: compiled tables (host) + per-device copies.
#pragma once

#include <cstdint>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lc_regex_gpu.h"
#include "follow_nfa.hpp"
#include "tdfa.hpp"

constexpr int kLcMaxDevices = 16;

// A LAZY / partial tagged DFA in front of the thread-list engine (round 6; tdfa.hpp buildTdfaLazy).  Handles whose automaton does not
// determinise within the limits (170 000 to 600 000 states for the log formats of BASELINE configs[2]) run the thread-list kernels at
// ~12 000 cycles per byte step; the values they see visit a few thousand of those states.  lcRegexLazyTrain builds the automaton along
// SAMPLE values (host memory); the launchers then walk every value through it first (tdfa_wave_kernel / tdfa_l2_kernel) and leave
// only the values that step on an uncomputed transition (LC_PENDING) to the thread-list kernels of the same launch.  Results do not
// depend on the sample: what the partial automaton decides it decides as the complete one would.
struct LcLazyTdfa {
    // ---- the trainer's side: one training call at a time (a call may take tens of milliseconds; launches never wait for it)
    std::mutex trainMutex;
    std::vector<uint8_t> sampleData;   // the values the tables are built from: only values that MISSED the tables of their time are
    std::vector<uint32_t> sampleOff, sampleLen;  // kept, so the sample stays small and still covers everything ever offered
    lcregex::TdfaTables tables;        // the logical tables of the current build (the trainer walks offered values through them)
    bool haveTables = false;
    bool frozen = false;               // a limit ended a construction: the tables stay as they are
    lcregex::TdfaLazyReport report;
    // ---- published: read by the launch path (short critical sections)
    std::mutex m;
    std::vector<uint32_t> blob;        // tdfa_l2_layout.h with TL_MISS != 0; empty: none
    uint32_t version = 0;              // bumped by every (re)build: device copies of older versions retire
    bool disabled = false;             // the construction does not get anywhere (register limit, a limit hit at once): never launched
    void* dBlob[kLcMaxDevices] = {};
    uint32_t dVersion[kLcMaxDevices] = {};
    std::vector<std::pair<int, void*>> retired;  // (device, pointer): copies of older versions, freed with the handle
    std::atomic<uint64_t> launches{0}, builds{0}, offered{0}, kept{0};
};

struct lc_regex {
    std::string pattern;
    uint32_t syntaxFlags = 0;
    int engine = LC_ENGINE_TDFA;
    lcregex::FollowNfa nfa;
    bool hasTdfa = false;
    lcregex::TdfaTables tdfa;
    std::vector<uint32_t> tdfaHeader;   // LC_TABLE_TDFA_HEADER view
    std::vector<uint32_t> tdfaBlob;     // device_tables.h TDFA layout
    int tdfaBlock = 0;                  // workgroup size tdfaBlob was packed for
    uint32_t tdfaPackedRegs = 0;        // registers per line tdfaBlob uses: tdfa.nRegs + one per folded multi-stamp set
    uint32_t tdfaWidePackedRegs = 0;    // the same for tdfaWideBlob (each blob folds only if its register area allows)
    std::vector<uint32_t> tdfaWideBlob; // tables of the COMPACT kernel variant (16-bit offset registers), or empty
    int tdfaWideBlock = 0;              // its workgroup size: 256 / 512 (class-indexed rows) or 1024 (byte-indexed rows)
    bool tdfaWideForced = false;        // LC_TDFA_COMPACT was set: use it for every batch, not only for large ones
    std::vector<uint32_t> tdfaL2Blob;   // tdfa_l2_layout.h: the tagged DFA with its tables in global memory, when `tdfa` is too large
                                        // for the LDS kernels (then hasTdfa is false and tdfaBlob empty) and small enough for L2
    std::vector<uint32_t> screenBlob;   // screen_kernel.hpp layout: a yes/no DFA too large for LDS (relaxed screens), or empty
    std::vector<uint32_t> nfaBlob;      // device_tables.h NFA layout
    std::vector<uint32_t> btBlob;       // bt_vm.hpp: the backtracking program (engine LC_ENGINE_BT: back-references), or empty
    std::vector<uint8_t> nfaClassMap;
    std::string tdfaError;              // why the TDFA was not built (AUTO fell back to NFA)
    std::string requiredLiteral;        // longest byte string every match must contain ("" if none is certain)
    bool preferWave = false;            // lcPreferWaveTdfa: small batches of this handle run one value per wavefront (tdfa_wave_kernel)
                                        // even when the automaton fits the LDS kernels -- callers whose batches wait for their longest
                                        // value (the Grok matcher's per-entry batches of long values)
    uint32_t atomicsElided = 0;         // atomic groups turned into plain groups because they provably change nothing (atomic_elide.cpp)
    // the decide kernel's per-frame capacities (nfa_decide_kernel.hpp DecideShape): enter / enter+exit events of the longest path
    uint32_t decideMaxEnter = 0, decideClosedCap = 0;

    // device residency, managed by gpu_runtime.hip
    std::mutex deviceMutex;
    void* dTdfaBlob[kLcMaxDevices] = {};
    void* dTdfaWideBlob[kLcMaxDevices] = {};
    std::atomic<uint32_t> nfaSeq[kLcMaxDevices] = {};      // launch sequence numbers of the NFA kernel (its overflow flag)
    std::atomic<uint32_t> tdfaWideSeq[kLcMaxDevices] = {};  // launch sequence numbers of the compact kernel (its long-line flag)
    void* dNfaBlob[kLcMaxDevices] = {};
    void* dScreenBlob[kLcMaxDevices] = {};
    void* dTdfaL2Blob[kLcMaxDevices] = {};
    void* dBtBlob[kLcMaxDevices] = {};
    LcLazyTdfa lazy;
    std::atomic<bool> lazyReady{false};  // lazy.blob is there (checked without the lock on the launch path)
    // Grok (grok_device.hip): search rounds this Match entry queues ahead per batch (FindStringMatch + FindNextMatch ...); follows
    // what the batches turn out to need
    std::atomic<uint32_t> grokRounds{2}, grokRoundsSlack{0};
    // ... and what its round 0 (first-chance launch) and its leftovers (second chance, search proper) cost on the device, in ns per
    // candidate, as measured with events on calibration batches (0 = not measured yet): the matcher deals the entries to its worker
    // streams longest-first by these
    std::atomic<uint32_t> grokCost0Ns{0}, grokCost1Ns{0}, grokBatches{0};
    // what recent batches of this entry looked like (grok_device.hip): values that needed more than 64 threads on the thread-list
    // engine (8 = seen in the last batch, forgotten a batch at a time: such an entry's first chance is the wide kernel), and whether
    // most of the slots in play behind a first match pass the entry's remainder screen (then its search rounds are queued ahead,
    // unscreened)
    std::atomic<uint32_t> grokOverflowSeen{0}, grokRemainderSeen{0};
};

// regex_handle.cpp: walks `n` values (host memory) through the handle's lazy automaton, keeps the ones that miss in its sample and
// rebuilds until none does (or a limit is reached); thread-list handles only (others: LC_OK, nothing done).  Safe beside launches on
// other threads.  out (optional) = {states, transitions computed, sample values kept, offered values the tables still miss, 1 = in use}.
int lcRegexLazyTrain(lc_regex* re, const uint8_t* data, const uint32_t* off, const uint32_t* len, uint32_t n, uint64_t out[5]);

// Values a consumer without a parse-failure notion of its own (filter leaves, multiline flags, the Go regex plugin) had to take as
// "no match" because the decide kernel gave up on them (LC_GAVE_UP: boost's complexity exception -- BoostRegexMatch / BoostRegexSearch
// return false there too, core/common/StringTools.cpp:200-205,277-282).  Never silent: counted here, lc_gave_up_values_total().
void lcNoteGaveUp(uint64_t n);

namespace lcregex {
// atomic_elide.cpp: (?>X) -> (?:X) wherever that provably changes no match and no capture; returns how many groups went plain
int elideRedundantAtomics(ParsedRegex& re);
// `block` = workgroup size the register offsets are encoded for (lcTdfaPickBlock)
// foldPrograms: multi-stamp register programs become stamps of set registers when every program of the table allows it
// pairMode: the byte-pair extension -- -1: what LC_TDFA_PAIR says (unset / 0: none, 1: two stamps per entry, 2: one stamp);
// 0 / 1 / 2: that, whatever the environment says
std::vector<uint32_t> packTdfaBlob(const TdfaTables& t, int block, bool wide = false, bool compact = false,
                                   bool foldPrograms = true, int pairMode = -1);
uint32_t tdfaFoldRegs(const TdfaTables& t);  // registers that fold adds per line (0 = nothing to fold)
// tables of the COMPACT kernel variant (LC_TDFA_COMPACT picks it; empty: switched off, or the automaton is too large)
std::vector<uint32_t> packTdfaWideBlob(const TdfaTables& t, int* blockOut, bool* forcedOut, uint32_t* packedRegsOut);
size_t tdfaBlobBytesEstimate(const TdfaTables& t);
// throws RegexError when the NFA does not fit the device format (more than 128 byte classes, follow lists over 64 paths ...)
std::vector<uint32_t> packNfaBlob(const FollowNfa& nfa, std::vector<uint8_t>& classMapOut);
}  // namespace lcregex

// LDS budgeting shared by the compile-time engine choice (regex_handle.cpp) and the launchers (gpu_runtime.hip).
// TDFA: tables + nRegs x BLOCK x 4 B of offset registers.  Prefer 256-lane workgroups while a workgroup stays under
// 80 KiB (2 workgroups per CU; round 4: was 64 KiB, which cost regex B's byte-pair tables their folded registers); shrink the
// workgroup before giving up.  0 = does not fit in 160 KiB at all.
constexpr size_t kLcLdsPerCu = 160 * 1024;
// nRegs counts the real offset registers; the kernel adds one dummy register (see device_tables.h)
inline size_t lcTdfaRegBytes(uint32_t nRegs, int block) { return size_t(nRegs + 1) * size_t(block) * 4; }
// per wavefront: 64 staging rows of 64+16 bytes (tdfa_kernel.hpp: kTdfaStagePerWave)
#ifndef LC_TDFA_STAGE_BYTES
#define LC_TDFA_STAGE_BYTES 64
#endif
inline size_t lcTdfaStageBytes(int block) { return size_t(block / 64) * 64 * (LC_TDFA_STAGE_BYTES + 16); }
inline size_t lcTdfaLdsBytes(uint32_t blobBytes, uint32_t nRegs, int block) {
    return size_t(blobBytes) + lcTdfaRegBytes(nRegs, block) + lcTdfaStageBytes(block);
}
// COMPACT kernel variants (tdfa_kernel.hpp): 16-bit offset registers, unpadded staging tiles; the byte-indexed one runs as
// one 1024-lane workgroup per CU
constexpr int kLcTdfaWideBlock = 1024;
inline size_t lcTdfaWideRegBytes(uint32_t nRegs) { return size_t(nRegs + 1) * kLcTdfaWideBlock * 2; }
inline size_t lcTdfaWideLdsBytes(uint32_t blobBytes, uint32_t nRegs) {
    return size_t(blobBytes) + lcTdfaWideRegBytes(nRegs) + size_t(kLcTdfaWideBlock / 64) * 64 * LC_TDFA_STAGE_BYTES;
}
inline size_t lcTdfaCompactLdsBytes(uint32_t blobBytes, uint32_t nRegs, int block) {
    return size_t(blobBytes) + size_t(nRegs + 1) * size_t(block) * 2 + size_t(block / 64) * 64 * LC_TDFA_STAGE_BYTES;
}
inline int lcTdfaPickBlock(uint32_t blobBytes, uint32_t nRegs) {
    auto fits = [&](int b, size_t budget) {
        return lcTdfaRegBytes(nRegs, b) <= 0x10000 && lcTdfaLdsBytes(blobBytes, nRegs, b) <= budget;
    };
    for (int b : {256, 128, 64})
        if (fits(b, kLcLdsPerCu / 2)) return b;  // (two workgroups per CU; above 64 KiB the launcher raises the function's dynamic-LDS limit)
    for (int b : {256, 128, 64})
        if (fits(b, kLcLdsPerCu)) return b;
    return 0;
}
// 4 waves x (best[nPos] + 4x64 words) [+ 4 x the atomic path's scratch, nfa_kernel.hpp kNfaAtomicScratchWords = 1344]
// (waves: values per workgroup -- 4, or 2 / 1 for small batches whose program then fits LDS, nfa_kernel.hpp BLOCK)
inline size_t lcNfaLdsBytes(uint32_t blobBytes, uint32_t nPos, bool atomic, uint32_t waves = 4) {
    return size_t(blobBytes) + size_t(waves) * (((nPos + 3) & ~3u) + 256) * 4 + (atomic ? size_t(waves) * 1344 * 4 : 0);
}

// regex_handle.cpp: TDFA-only handle for the longest prefix of the pattern's top-level concatenation whose automaton stays
// within the limits (nullptr if there is none); status-only screening before the NFA engine (Grok)
lc_regex* lcCompilePrefixScreen(const char* pattern, size_t len, uint32_t syntax_flags, uint32_t maxStates,
                                size_t maxBlobBytes);
// regex_handle.cpp: TDFA-only handle for the WHOLE pattern relaxed until its automaton is small (sub-expressions that are too
// large become "any of their bytes, repeated"); a necessary condition over the whole line (nullptr if there is none)
lc_regex* lcCompileRelaxedScreen(const char* pattern, size_t len, uint32_t syntax_flags, uint32_t maxStates, size_t maxBlobBytes);
lc_regex* lcCompileRelaxedScreenPreferring(const char* pattern, size_t len, uint32_t syntax_flags, uint32_t maxStates, size_t maxBlobBytes,
                                           size_t preferStageBytes);
// implemented in gpu_runtime.hip: one pass of a screen handle that carries a screenBlob (dfa_screen_kernel) over the values
// listed in d_in (nullptr: all n); accepted values are appended to d_out, their number added to d_counters[0]
int lcScreenOnStream(lc_regex* re, int dev, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len, uint32_t n,
                     const uint32_t* d_in, uint32_t* d_out, uint32_t* d_counters, void* stream);
// implemented in gpu_runtime.hip; frees device copies
void lcReleaseDeviceTables(lc_regex* re);
// regex_handle.cpp: ask for the wave-per-value kernel on small batches of this handle (packs the global-memory form of its tagged
// DFA beside the LDS form; no effect on handles without a tagged DFA).  Call before the handle's first launch.
void lcPreferWaveTdfa(lc_regex* re);
// implemented in gpu_runtime.hip: one launch of the engine's kernel.  d_n (optional): line count on the device;
// d_order (optional): the lines to process; d_resume (optional, indexed by line): resume offsets of a search pattern.
int lcMatchOnStream(lc_regex* re, int engine, int dev, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len,
                    uint32_t sep, uint32_t n, const uint32_t* d_n, const uint32_t* d_order, const uint32_t* d_resume,
                    uint32_t ngroups, int32_t* d_caps, uint8_t* d_status, void* stream);
// the engine's main kernel only (*seq: see below), and the second chance for the lines it left LC_OVERFLOW (gpu_runtime.hip)
int lcMatchFirstOnStream(lc_regex* re, int engine, int dev, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len,
                         uint32_t sep, uint32_t n, const uint32_t* d_n, const uint32_t* d_order, const uint32_t* d_resume, uint32_t ngroups,
                         int32_t* d_caps, uint8_t* d_status, uint32_t* seq, void* streamPtr);
int lcMatchSecondChanceOnStream(lc_regex* re, int engine, int dev, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len,
                                uint32_t sep, uint32_t n, const uint32_t* d_n, const uint32_t* d_order, const uint32_t* d_resume,
                                uint32_t ngroups, int32_t* d_caps, uint8_t* d_status, uint32_t seq, void* streamPtr);
// Round 5, "wide first" (nfa_wide_kernel.hpp): for a caller that knows the pattern needs more than 64 threads on its data.
//   part 0: the wide kernel over every line as the first chance (*seq as above; lines beyond 128 threads keep LC_OVERFLOW);
//   part 1: what is left behind part 0 -- the decide kernels alone (seq: what part 0 returned);
//   part 2: both in one call.
// Engines and programs the wide kernel does not run (tagged DFAs, atomic groups, more than 64 capture slots) take their usual
// kernels: part 0 = lcMatchFirstOnStream, part 2 = lcMatchOnStream.  wideNote (optional, device word): set to 1 by the launch when
// some line did need more than 64 threads.
int lcMatchWideFirstOnStream(int part, lc_regex* re, int engine, int dev, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len,
                             uint32_t sep, uint32_t n, const uint32_t* d_n, const uint32_t* d_order, const uint32_t* d_resume,
                             uint32_t ngroups, int32_t* d_caps, uint8_t* d_status, uint32_t* seq, uint32_t* wideNote, void* streamPtr);
// does the wide kernel run this handle's thread-list program at all?
bool lcNfaWideApplies(const lc_regex* re);
