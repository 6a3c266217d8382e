// grok_device.hip -- device half of the Grok processor (include/lc_grok.h):
// ProcessorGrok.processGrok (plugins/processor/grok/processor_grok.go:148-194) for a whole batch of values.
//
// Two ways through a Match list, same results:
//   * speculative (default, lists of up to 64 entries): grok_plan_kernel.hpp -- one literal pass, ONE launch for the screens of
//     all entries, then every surviving (entry, value) pair evaluated at the same time on a few streams, the first contributing
//     entry per value taken at the end.  Round control stays on the device; the host synchronises twice per batch.
//   * sequential: grok_kernel.hpp -- the list walked entry by entry over the values still undecided (a host round trip per
//     filter, screen and search round).
// The regex kernels themselves (tdfa_* / nfa_*) live in gpu_runtime.hip and are reached through lcMatchOnStream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "grok_kernel.hpp"
#include "grok_plan_kernel.hpp"
#include "grok_runtime.hpp"
#include "group_combiner.hpp"
#include "regex_handle.hpp"
#include "runtime_internal.hpp"
#include "tdfa_l2_layout.h"

#define HIP_TRY LC_HIP_TRY

namespace {
size_t alignUp(size_t v, size_t a) { return (v + a - 1) / a * a; }

thread_local GrokBatchStats tlsStats;
// lcGrokMatchHost asks the device call to queue the copies of its results (pinned host memory) behind its last kernels, so that the
// call's final synchronisation covers them
struct GrokTailCopy {
    int32_t* hPattern = nullptr;
    int32_t* hFirst = nullptr;
    bool armed = false, done = false;
    uint32_t nextra = 0;
};
thread_local GrokTailCopy tlsTail;
hipError_t syncCounted(hipStream_t st) {
    ++tlsStats.hostSyncs;
    return hipStreamSynchronize(st);
}
}  // namespace

GrokBatchStats lcGrokLastBatchStats() { return tlsStats; }

// Scratch layout.  Sequential path: caps int32[n][row] | status u8[n] (padded) | from, nmatch, tried, next, roundA, roundB,
// unanchored : u32[n] each | counters u32[64] | perPattern u32[64] | masks u64[n] | work u32[1024].  The speculative path only uses the tail
// (counters .. masks) plus winner / undecided u32[n] carved from the head; its per-entry arrays come from the thread's arena.
size_t lcGrokScratchBytes(uint32_t n, uint32_t rowInts) {
    const size_t m = n ? n : 1;
    return alignUp(m * rowInts * 4, 256) + alignUp(m, 256) + 7 * alignUp(m * 4, 256) + 512 + alignUp(m * 8, 256) + 4096;
}

// ------------------------------------------------------------------------------------------------ per-handle device state
struct GrokDeviceState {
    std::mutex m;
    bool literalsBuilt = false;
    std::vector<uint32_t> literalBlob;           // empty: the list is not indexable
    void* dLiteral[kLcMaxDevices] = {};
    // screens of the speculative path: per device, a table of GrokScreenDev (at most one screen per entry)
    bool screensBuilt[kLcMaxDevices] = {};
    void* dScreens[kLcMaxDevices] = {};
    uint32_t nScreens[kLcMaxDevices] = {};
    uint32_t screenLdsBytes[kLcMaxDevices] = {};  // largest staged table
    uint32_t nBigScreens[kLcMaxDevices] = {};     // the table's first entries: screens too large for that, see GrokScreenDev::bigBytes
    uint32_t bigScreenLdsBytes[kLcMaxDevices] = {};
    std::vector<GrokScreenDev> hostScreens[kLcMaxDevices];  // the same table on the host (kernel arguments of the remainder screens)
    // lcGrokMatchHost: the groups of concurrent runner threads travel as ONE batch per device (group_combiner.hpp): a worker thread
    // per (processor, device) owns the plan, the streams and the pinned staging
    struct HostJob;
    struct HostCombiner;
    HostCombiner* combiner[kLcMaxDevices] = {};
};

static void grokFreeCombiners(GrokDeviceState* s);
GrokDeviceState* lcGrokStateCreate() { return new GrokDeviceState(); }
void lcGrokStateFree(GrokDeviceState* s) {
    if (!s) return;
    grokFreeCombiners(s);  // (the worker threads end first: they use the tables freed below)
    if (lcRuntimeUsable()) {
        int cur = 0;
        const bool haveCur = hipGetDevice(&cur) == hipSuccess;
        for (int d = 0; d < kLcMaxDevices; ++d) {
            if (!s->dLiteral[d] && !s->dScreens[d]) continue;
            if (hipSetDevice(d) != hipSuccess) continue;
            if (s->dLiteral[d]) (void)hipFree(s->dLiteral[d]);
            if (s->dScreens[d]) (void)hipFree(s->dScreens[d]);
        }
        if (haveCur) (void)hipSetDevice(cur);
    }
    delete s;
}

namespace {
// device blob of the list's literal index, or nullptr (more than 64 entries, no literal at all, automaton too large).  Built once
// per handle, uploaded once per device; freed with the handle.
int grokLiteralIndex(const std::vector<GrokDevicePattern>& patterns, GrokDeviceState* state, int dev, const uint32_t** out) {
    *out = nullptr;
    static const bool off = getenv("LC_GROK_NO_LITERAL_INDEX") != nullptr;
    if (off || patterns.size() > 64) return LC_OK;
    std::lock_guard<std::mutex> g(state->m);
    if (!state->literalsBuilt) {
        std::vector<std::string> lits;
        size_t withLiteral = 0;
        for (const auto& gp : patterns) {
            lits.push_back(lcGrokLiteralOf(gp.re));
            withLiteral += !lits.back().empty();
        }
        if (withLiteral >= 2) state->literalBlob = lcBuildGrokLiteralBlob(lits);
        state->literalsBuilt = true;
    }
    if (state->literalBlob.empty()) return LC_OK;
    if (!state->dLiteral[dev]) {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, state->literalBlob.size() * 4));
        const hipError_t e = hipMemcpy(p, state->literalBlob.data(), state->literalBlob.size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(p);
            return lcHipFail(e, "hipMemcpy(literal index)");
        }
        state->dLiteral[dev] = p;
    }
    *out = static_cast<const uint32_t*>(state->dLiteral[dev]);
    return LC_OK;
}
}  // namespace

// ---- the sequential path: the Match list walked entry by entry over the values still undecided (the control flow of
// grok_kernel.hpp).  Lists of more than 64 entries, and handles configured with "Speculative": false.
static int grokMatchSequential(const std::vector<GrokDevicePattern>& patterns, GrokDeviceState* state, const GrokOptions& opts,
                               uint32_t row, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len, uint32_t n,
                               int32_t* d_pattern, int32_t* d_first, int32_t* d_extra, uint32_t extraCap, uint32_t* d_nextra,
                               void* d_scratch, hipStream_t st, int dev) {

    uint8_t* base = static_cast<uint8_t*>(d_scratch);
    int32_t* caps = reinterpret_cast<int32_t*>(base);
    base += alignUp(size_t(n) * row * 4, 256);
    uint8_t* status = base;
    base += alignUp(n, 256);
    uint32_t* lists[7];
    for (auto& l : lists) {
        l = reinterpret_cast<uint32_t*>(base);
        base += alignUp(size_t(n) * 4, 256);
    }
    uint32_t *from = lists[0], *nmatch = lists[1], *tried = lists[2], *next = lists[3], *roundIn = lists[4],
             *roundOut = lists[5], *unanchored = lists[6];
    uint32_t* counters = reinterpret_cast<uint32_t*>(base);
    base += 256;
    uint32_t* perPattern = reinterpret_cast<uint32_t*>(base);  // [64]: values that carry each entry's literal (literal index pass)
    base += 256;
    uint64_t* masks = reinterpret_cast<uint64_t*>(base);

    const uint32_t gridAll = (n + kGrokBlock - 1) / kGrokBlock;
    hipLaunchKernelGGL(grok_init_kernel, dim3(gridAll), dim3(kGrokBlock), 0, st, n, d_pattern, tried, from, nmatch);
    // which Match entries' required literals each value contains: one pass for the whole list
    const uint32_t* literalIndex = nullptr;
    {
        int rc = grokLiteralIndex(patterns, state, dev, &literalIndex);
        if (rc != LC_OK) return rc;
    }
    std::vector<uint32_t> carriers;  // per entry: values of the batch that carry its literal (empty: no index)
    if (literalIndex) {
        HIP_TRY(hipMemsetAsync(perPattern, 0, 256, st));
        hipLaunchKernelGGL(grok_literal_index_kernel, dim3(gridAll), dim3(kGrokBlock), 0, st, d_data, d_off, d_len, n, literalIndex,
                           masks, uint32_t(patterns.size()), perPattern, static_cast<const uint32_t*>(nullptr));
        carriers.resize(64);
        HIP_TRY(hipMemcpyAsync(carriers.data(), perPattern, 256, hipMemcpyDeviceToHost, st));
        HIP_TRY(syncCounted(st));
    }
    HIP_TRY(hipMemsetAsync(d_first, 0xFF, size_t(n) * row * 4, st));
    HIP_TRY(hipMemsetAsync(counters, 0, 16, st));

    uint32_t nTried = n;
    uint32_t host[4] = {0, 0, 0, 0};
    // LC_GROK_TRACE=1: one stderr line per Match entry -- values tried / with the literal / past the screen, then (values, ms)
    // per search round (the host waits for a counter after every step anyway, so the clock reads are exact)
    static const bool trace = getenv("LC_GROK_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto msSince = [&](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<double, std::milli>(now() - t0).count();
    };
    for (size_t p = 0; p < patterns.size() && nTried; ++p) {
        const GrokDevicePattern& gp = patterns[p];
        // no value of the batch carries this entry's literal: it cannot match anything, and an entry that matches nothing
        // leaves every list as it is
        if (!carriers.empty() && carriers[p] == 0) continue;
        auto tPattern = now();
        std::string traceLine;
        if (trace) traceLine = "grok[" + std::to_string(p) + "] engine " + std::to_string(gp.re->engine) + " tried " + std::to_string(nTried);
        const uint32_t* in = tried;  // round 0 searches every value still undecided, from its first byte ...
        uint32_t nIn = nTried;
        uint32_t* outs[2] = {roundIn, roundOut};
        int flip = 0;
        if (!gp.re->requiredLiteral.empty()) {  // ... that contains the literal every match of this pattern must contain
            GrokLiteral lit;
            const std::string& s = gp.re->requiredLiteral;
            lit.len = uint32_t(std::min<size_t>(s.size(), sizeof lit.bytes));
            std::memcpy(lit.bytes, s.data() + (s.size() - lit.len), lit.len);
            const uint32_t perBlock = kGrokBlock / 64;
            if (literalIndex)
                hipLaunchKernelGGL(grok_mask_filter_kernel, dim3((nTried + kGrokBlock - 1) / kGrokBlock), dim3(kGrokBlock), 0, st, tried,
                                   nTried, masks, uint32_t(p), outs[1], counters);
            else
                hipLaunchKernelGGL(grok_literal_filter_kernel, dim3((nTried + perBlock - 1) / perBlock), dim3(kGrokBlock), 0, st,
                                   tried, nTried, d_data, d_off, d_len, lit, outs[1], counters);
            HIP_TRY(hipMemcpyAsync(host, counters, 16, hipMemcpyDeviceToHost, st));
            HIP_TRY(syncCounted(st));
            nIn = host[0];
            in = outs[1];
            HIP_TRY(hipMemsetAsync(counters, 0, 4, st));
            if (trace) traceLine += " literal " + std::to_string(nIn) + " (" + std::to_string(msSince(tPattern)).substr(0, 6) + " ms)";
        }
        // ... and a match of the pattern's prefix, then of the relaxed whole pattern (fast TDFA kernel, status only)
        // (the relaxed screen rejects nearly everything the prefix screen rejects, and each pass costs a launch and a counter read,
        // ~0.25 ms: the prefix screen only goes first where it saves the relaxed one a large candidate set -- LC_GROK_PREFIX_ABOVE)
        const uint32_t prefixAbove = opts.prefixScreenAbove;
        for (lc_regex* scr : {gp.screen, gp.relaxed}) {
            if (!scr || !nIn) continue;
            if (scr == gp.screen && gp.relaxed && nIn <= prefixAbove) continue;
            uint32_t* out = in == outs[0] ? outs[1] : outs[0];
            if (!scr->hasTdfa && !scr->screenBlob.empty()) {  // a plain DFA with its table in L2: screens and filters in one kernel
                int rc = lcScreenOnStream(scr, dev, d_data, d_off, d_len, nIn, in, out, counters, st);
                if (rc != LC_OK) return rc;
            } else {
                int rc = lcMatchOnStream(scr, LC_ENGINE_TDFA, dev, d_data, d_off, d_len, 0, nIn, nullptr, in, nullptr, 0, caps,
                                         status, st);
                if (rc != LC_OK) return rc;
                hipLaunchKernelGGL(grok_status_filter_kernel, dim3((nIn + kGrokBlock - 1) / kGrokBlock), dim3(kGrokBlock), 0, st,
                                   in, nIn, status, out, counters);
            }
            HIP_TRY(hipMemcpyAsync(host, counters, 16, hipMemcpyDeviceToHost, st));
            HIP_TRY(syncCounted(st));
            nIn = host[0];
            in = out;
            flip = in == outs[0] ? 1 : 0;
            HIP_TRY(hipMemsetAsync(counters, 0, 4, st));
            if (trace)
                traceLine += std::string(scr == gp.screen ? " screen " : " relaxed ") + std::to_string(nIn) + " (" +
                             std::to_string(msSince(tPattern)).substr(0, 6) + " ms)";
        }
        // the match kernels write this pattern's own groups only (whole match + its columns), not the widest pattern's row
        const uint32_t capsRow = 2 * (gp.columns + 1);
        // one search round over the values listed in `list`: the pattern's kernel, then grok_advance_kernel (matches recorded,
        // values that stay in play appended to `out`, their number added to counters[0])
        auto searchRound = [&](lc_regex* re, const uint32_t* list, uint32_t nList, const uint32_t* resume, uint32_t* out) -> int {
            int rc = lcMatchOnStream(re, re->engine, dev, d_data, d_off, d_len, 0, nList, nullptr, list, resume, capsRow / 2, caps,
                                     status, st);
            if (rc != LC_OK) return rc;
            hipLaunchKernelGGL(grok_advance_kernel, dim3((nList + kGrokBlock - 1) / kGrokBlock), dim3(kGrokBlock), 0, st, list, nList,
                               status, caps, capsRow, row, gp.columns, d_len, from, nmatch, d_pattern, d_first, d_extra, extraCap, out,
                               counters);
            return LC_OK;
        };
        if (gp.anchored && nIn) {
            // Round 0 searches every value from its first byte, and a log format matches FROM the first byte: the anchored search
            // (a tagged DFA, tables in L2, one value per lane) finds exactly what the search would find whenever the search's
            // leftmost match starts at byte 0; the values it does not match go to the search proper (their match, if any, starts
            // later).  Both append to the same next-round list.
            auto tRound = now();
            uint32_t* out = outs[flip];
            int rc = lcMatchOnStream(gp.anchored, gp.anchored->engine, dev, d_data, d_off, d_len, 0, nIn, nullptr, in, nullptr, capsRow / 2,
                                     caps, status, st);
            if (rc != LC_OK) return rc;
            hipLaunchKernelGGL(grok_unmatched_kernel, dim3((nIn + kGrokBlock - 1) / kGrokBlock), dim3(kGrokBlock), 0, st, in, nIn, status,
                               unanchored, counters + 3);
            hipLaunchKernelGGL(grok_advance_kernel, dim3((nIn + kGrokBlock - 1) / kGrokBlock), dim3(kGrokBlock), 0, st, in, nIn, status,
                               caps, capsRow, row, gp.columns, d_len, from, nmatch, d_pattern, d_first, d_extra, extraCap, out, counters);
            HIP_TRY(hipMemcpyAsync(host, counters, 16, hipMemcpyDeviceToHost, st));
            HIP_TRY(syncCounted(st));
            const uint32_t nRest = host[3];
            HIP_TRY(hipMemsetAsync(counters + 3, 0, 4, st));
            if (trace)
                traceLine += " anchored " + std::to_string(nIn) + " -> rest " + std::to_string(nRest) + " (" +
                             std::to_string(msSince(tRound)).substr(0, 7) + " ms)";
            if (nRest) {
                auto tRest = now();
                rc = searchRound(gp.re, unanchored, nRest, from, out);
                if (rc != LC_OK) return rc;
                HIP_TRY(hipMemcpyAsync(host, counters, 16, hipMemcpyDeviceToHost, st));
                HIP_TRY(syncCounted(st));
                if (trace) traceLine += " round " + std::to_string(nRest) + " (" + std::to_string(msSince(tRest)).substr(0, 7) + " ms)";
            }
            nIn = host[0];
            in = out;
            flip ^= 1;
            HIP_TRY(hipMemsetAsync(counters, 0, 4, st));
        }
        while (nIn) {
            auto tRound = now();
            const uint32_t roundValues = nIn;
            uint32_t* out = outs[flip];
            int rc = searchRound(gp.re, in, nIn, from, out);
            if (rc != LC_OK) return rc;
            HIP_TRY(hipMemcpyAsync(host, counters, 16, hipMemcpyDeviceToHost, st));
            HIP_TRY(syncCounted(st));
            nIn = host[0];
            in = out;
            flip ^= 1;
            HIP_TRY(hipMemsetAsync(counters, 0, 4, st));  // counters[0] only; the extra-row count keeps running
            if (trace) traceLine += " round " + std::to_string(roundValues) + " (" + std::to_string(msSince(tRound)).substr(0, 7) + " ms)";
        }
        hipLaunchKernelGGL(grok_finish_kernel, dim3((nTried + kGrokBlock - 1) / kGrokBlock), dim3(kGrokBlock), 0, st, tried,
                           nTried, int32_t(p), nmatch, d_pattern, from, next, counters);
        HIP_TRY(hipMemcpyAsync(host, counters, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(syncCounted(st));
        if (trace) fprintf(stderr, "%s settled %u total %.3f ms\n", traceLine.c_str(), nTried - host[2], msSince(tPattern));
        nTried = host[2];
        std::swap(tried, next);
        HIP_TRY(hipMemsetAsync(counters + 2, 0, 4, st));
    }
    HIP_TRY(hipMemcpyAsync(d_nextra, counters + 1, 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(host, counters, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(syncCounted(st));
    HIP_TRY(hipGetLastError());
    if (host[1] > extraCap) {
        lcSetLastError("grok: " + std::to_string(host[1]) + " extra match rows needed, buffer holds " + std::to_string(extraCap));
        return LC_ERR_OVERFLOW;
    }
    return LC_OK;
}


// ------------------------------------------------------------------------------------------------ speculative path
namespace {
constexpr int kGrokMaxStreams = 16;  // (LC_GROK_STREAMS; the default stays opts.streams = 8)
constexpr uint32_t kGrokScreenStageMax = 44 * 1024;  // a screen's accept flags + table are staged into LDS up to this size
constexpr uint32_t kGrokScreenBigMax = 150 * 1024;   // ... and up to this size by a workgroup that has its CU's LDS to itself (small batches)
constexpr uint32_t kGrokMaxRounds = GC_BOUND - GC_ROUND0 - 1;  // search rounds that can be queued ahead per entry
constexpr uint32_t kGrokSmallBatch = 262144;  // up to here phase 1 takes the chunk-parallel literal pass and LDS-staged screens, one value per
                                              // lane (round 5: measured better than the lane-per-value passes at 64 Ki (9.3 -> 7.3 ms), 128 Ki
                                              // (14.0 -> 12.1) and 256 Ki values (21.6 -> 20.6); round 4 drew the line at 32 Ki)
constexpr uint32_t kGrokWideFirstBatch = 32768;  // ... and up to here a batch waits for its longest value: wide first pays (16 Ki: -0.3 ms; 64 Ki: +0.5 ms)

// Pinned host words of a thread: what the two syncs of a batch read back.
// (HW_CAND | HW_FIRST | HW_SHADOW mirror the device block dPlanWords: one copy brings all three)
enum { HW_CAND = 0, HW_FIRST = 64, HW_SHADOW = 128, HW_TAIL = 128 + 64 * 64, HW_CNT = 192 + 64 * 64,
       HW_WORDS = 192 + 64 * 64 + 64 * GC_WORDS };
constexpr uint32_t kPlanWords = 64 + 64 + 64 * 64;  // device: candidates per entry | first-candidate counts | who shadows whom
// device tail words (scratch `counters`): [0] gate  [1] xcount (extra rows wanted in xtmp)
// [16] a copy of the caller's nextra word (grok_commit_extra_kernel counts into both)
enum { TW_GATE = 0, TW_XCOUNT = 1, TW_WORDS = 16, TW_NEXTRA = 16 };
static_assert(HW_CNT - HW_TAIL == 64, "hostWords[HW_TAIL ..] mirrors the device block dTail | dCnt");

constexpr size_t kEntryBlockBytes = 64 * sizeof(GrokEntryDev) + 64 * sizeof(GrokScreenDev);

struct PlanThread {
    int device = -1;
    hipStream_t workers[kGrokMaxStreams] = {};
    hipEvent_t fork = nullptr, join[kGrokMaxStreams] = {};
    hipEvent_t tick[2 * 64 + 2 * 64] = {};  // calibration batches: [2a begin/end per entry | 2c begin/end per entry], created on first use
    uint32_t* hostWords = nullptr;        // pinned, HW_*
    // (round 6: the entry table and the remainder screens are ONE pinned block and ONE device block -- one copy per batch;
    // the tail words and the entries' counters likewise, mirrored by hostWords[HW_TAIL ..] -- one copy back per host synchronisation)
    GrokEntryDev* hostEntries = nullptr;  // pinned [64], then hostRemScreens
    GrokEntryDev* dEntries = nullptr;     // device [64], then dRemScreens
    uint32_t* dTail = nullptr;            // device [64] tail words (TW_*), then dCnt
    uint32_t* dCnt = nullptr;             // device [64][GC_WORDS] (inside dTail's block)
    GrokScreenDev* hostRemScreens = nullptr;  // pinned [64]: per ACTIVE entry, its screen (blob == nullptr: none)
    GrokScreenDev* dRemScreens = nullptr;     // device [64]
    uint32_t* dPlanWords = nullptr;           // device [kPlanWords]: perEntry[64] | firstOf[64] | shadow[64][64] (candidates of entry p whose
                                              // value's first candidate is entry f)
    void* hostJobs = nullptr;                 // pinned / device: the job table of round 0's fused launch (runtime_internal.hpp lcLaunchWaveJobs)
    void* dJobs = nullptr;
    void* arena = nullptr;                // device, grow-only: the per-entry arrays of the batch in flight
    size_t arenaCap = 0;
    ~PlanThread() { release(); }
    void release() {
        if (device >= 0 && lcRuntimeUsable()) {
            int cur = 0;
            const bool haveCur = hipGetDevice(&cur) == hipSuccess;
            if (hipSetDevice(device) == hipSuccess) {
                for (auto& w : workers)
                    if (w) {
                        (void)hipStreamSynchronize(w);
                        (void)hipStreamDestroy(w);
                    }
                if (fork) (void)hipEventDestroy(fork);
                for (auto& e : tick)
                    if (e) (void)hipEventDestroy(e);
                for (auto& e : join)
                    if (e) (void)hipEventDestroy(e);
                if (hostWords) (void)hipHostFree(hostWords);
                if (hostEntries) (void)hipHostFree(hostEntries);
                if (dEntries) (void)hipFree(dEntries);
                if (dTail) (void)hipFree(dTail);
                if (dPlanWords) (void)hipFree(dPlanWords);
                if (hostJobs) (void)hipHostFree(hostJobs);
                if (dJobs) (void)hipFree(dJobs);
                if (arena) (void)hipFree(arena);
            }
            if (haveCur) (void)hipSetDevice(cur);
        }
        for (auto& w : workers) w = nullptr;
        for (auto& e : join) e = nullptr;
        for (auto& e : tick) e = nullptr;
        fork = nullptr;
        hostWords = nullptr;
        hostEntries = nullptr;
        dEntries = nullptr;
        dTail = nullptr;
        dCnt = nullptr;
        hostRemScreens = nullptr;
        dRemScreens = nullptr;
        dPlanWords = nullptr;
        hostJobs = nullptr;
        dJobs = nullptr;
        arena = nullptr;
        arenaCap = 0;
        device = -1;
    }
    int ensure(int dev, uint32_t nStreams) {
        if (device != dev) {
            release();
            lcRegisterExitHook();
            device = dev;
            HIP_TRY(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&hostWords), HW_WORDS * 4, hipHostMallocDefault));
            static_assert((64 * sizeof(GrokEntryDev)) % alignof(GrokScreenDev) == 0, "the remainder screens sit behind the entry table");
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&hostEntries), kEntryBlockBytes, hipHostMallocDefault));
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dEntries), kEntryBlockBytes));
            hostRemScreens = reinterpret_cast<GrokScreenDev*>(hostEntries + 64);
            dRemScreens = reinterpret_cast<GrokScreenDev*>(dEntries + 64);
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dTail), (64 + 64 * GC_WORDS) * 4));
            dCnt = dTail + 64;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dPlanWords), kPlanWords * 4));
            HIP_TRY(hipHostMalloc(&hostJobs, lcWaveJobTableBytes(), hipHostMallocDefault));
            HIP_TRY(hipMalloc(&dJobs, lcWaveJobTableBytes()));
        }
        for (uint32_t s = 0; s < nStreams; ++s)
            if (!workers[s]) {
                HIP_TRY(hipStreamCreateWithFlags(&workers[s], hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&join[s], hipEventDisableTiming));
            }
        return LC_OK;
    }
    int ensureArena(size_t bytes) {
        if (arenaCap >= bytes) return LC_OK;
        if (arena) {
            HIP_TRY(hipDeviceSynchronize());  // (a larger batch than ever before: rare; nothing of ours is in flight here anyway)
            (void)hipFree(arena);
            arena = nullptr;
            arenaCap = 0;
        }
        const size_t want = bytes + (bytes >> 2) + (1u << 20);
        HIP_TRY(hipMalloc(&arena, want));
        arenaCap = want;
        return LC_OK;
    }
};
thread_local PlanThread tlsPlan;

// the screen the merged launch walks for an entry: the relaxed whole-pattern screen when there is one (it rejects nearly
// everything the prefix screen rejects), else the prefix screen
lc_regex* planScreenOf(const GrokDevicePattern& gp) {
    // (round 6) ... unless the relaxed screen is too big to be staged into LDS and the prefix screen is not: a table walked through L2
    // costs 100 ns a byte -- CISCOFW106006_106007_106010's 2 115 states x 57 classes (243 KB) were the longest walk of the screen
    // launch AND of the remainder screens, 0.42 + 0.44 ms of every batch for 345 candidates (profiles/round6_grok_screen_split.txt) --
    // against 45 ns a byte in LDS.  Either screen is a necessary condition; the prefix screen lets more values through to round 0.
    // LC_GROK_SCREEN_PREFIX: 0 (default) never, 1 whenever the relaxed screen is not LDS-staged, 2 only beyond kGrokScreenBigMax.
    // MEASURED AND LEFT OFF: the prefix screen of that entry lets 5 % more (entry, value) pairs through, and those are long values that
    // walk a whole automaton in round 0 -- 16 Ki values 2.31 -> 2.73 ms (mode 2), 2.60 ms (mode 1); profiles/round6_grok_steps.txt.
    static const int prefixMode = [] {
        const char* v = getenv("LC_GROK_SCREEN_PREFIX");
        return v ? atoi(v) : 0;
    }();
    auto stageBytesOf = [](const lc_regex* r) { return r->screenBlob[SC_TOTAL_BYTES] - r->screenBlob[SC_OFF_ACCEPT]; };
    if (prefixMode && gp.relaxed && !gp.relaxed->screenBlob.empty() && gp.screen && !gp.screen->screenBlob.empty() &&
        stageBytesOf(gp.relaxed) > (prefixMode == 1 ? kGrokScreenStageMax : kGrokScreenBigMax) && stageBytesOf(gp.screen) <= kGrokScreenStageMax)
        return gp.screen;
    if (gp.relaxed && !gp.relaxed->screenBlob.empty()) return gp.relaxed;
    if (gp.screen && !gp.screen->screenBlob.empty()) return gp.screen;
    return nullptr;
}

int grokScreenTable(const std::vector<GrokDevicePattern>& patterns, GrokDeviceState* state, int dev, const GrokScreenDev** table,
                    uint32_t* count, uint32_t* ldsBytes) {
    std::lock_guard<std::mutex> g(state->m);
    if (!state->screensBuilt[dev]) {
        static const bool noStage = getenv("LC_GROK_SCREEN_NO_LDS") != nullptr;
        std::vector<GrokScreenDev> host;
        uint32_t maxStage = 0, maxBig = 0;
        for (size_t p = 0; p < patterns.size(); ++p) {
            lc_regex* scr = planScreenOf(patterns[p]);
            if (!scr) continue;
            GrokScreenDev d{};
            int rc = lcEnsureScreenUploaded(scr, dev, &d.blob);
            if (rc != LC_OK) return rc;
            d.bit = uint32_t(p);
            const uint32_t stage = scr->screenBlob[SC_TOTAL_BYTES] - scr->screenBlob[SC_OFF_ACCEPT];
            d.ldsBytes = (!noStage && stage <= kGrokScreenStageMax) ? (stage + 3u) & ~3u : 0u;
            d.bigBytes = (!noStage && !d.ldsBytes && stage <= kGrokScreenBigMax) ? (stage + 3u) & ~3u : 0u;
            maxStage = std::max(maxStage, d.ldsBytes);
            maxBig = std::max(maxBig, d.bigBytes);
            if (getenv("LC_GROK_TRACE"))
                fprintf(stderr, "grok screen of entry %zu: %u states x %u classes, %u bytes (%s)%s; prefix screen %u states\n", p, scr->screenBlob[SC_NSTATES],
                        scr->screenBlob[SC_NCLASSES], stage, d.ldsBytes ? "staged in LDS" : d.bigBytes ? "big: through L2" : "through L2",
                        scr == patterns[p].relaxed ? " relaxed" : " prefix",
                        (patterns[p].screen && !patterns[p].screen->screenBlob.empty()) ? patterns[p].screen->screenBlob[SC_NSTATES] : 0u);
            host.push_back(d);
        }
        std::stable_sort(host.begin(), host.end(), [](const GrokScreenDev& x, const GrokScreenDev& y) { return (x.bigBytes != 0) > (y.bigBytes != 0); });
        uint32_t nBig = 0;
        for (const GrokScreenDev& d : host) nBig += d.bigBytes != 0;
        state->nBigScreens[dev] = nBig;
        state->bigScreenLdsBytes[dev] = maxBig;
        if (!host.empty()) {
            void* p = nullptr;
            HIP_TRY(hipMalloc(&p, host.size() * sizeof(GrokScreenDev)));
            const hipError_t e = hipMemcpy(p, host.data(), host.size() * sizeof(GrokScreenDev), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                (void)hipFree(p);
                return lcHipFail(e, "hipMemcpy(screen table)");
            }
            state->dScreens[dev] = p;
        }
        state->nScreens[dev] = uint32_t(host.size());
        state->hostScreens[dev] = host;
        state->screenLdsBytes[dev] = maxStage;
        state->screensBuilt[dev] = true;
    }
    *table = static_cast<const GrokScreenDev*>(state->dScreens[dev]);
    *count = state->nScreens[dev];
    *ldsBytes = state->screenLdsBytes[dev];
    return LC_OK;
}

// host view of one active entry of the batch
struct PlanEntry {
    uint32_t p = 0, cand = 0, capsRow = 0, columns = 0, rounds = 0, ran = 0;
    bool second = false;  // level > 0
    uint32_t level = 0;   // 0: evaluated at once; L > 0: most of its candidates have an EARLIER candidate entry (of level < L) -- it waits
                          // for those and only looks at the values none of them has won (a general format behind specific ones)
    bool queued = false;  // rounds behind the first match were queued for this entry
    bool wideFirst = false;  // round 0 ran nfa_wide_kernel over every candidate (the entry's history says its values overflow 64 threads)
    uint32_t seq0 = 0;    // launch sequence of round 0's first-chance kernel (lcMatchSecondChanceOnStream)
    uint32_t seqS = 0;    // ... and of the search proper's (phase 2c)
    const GrokScreenDev* remainderScreen = nullptr;  // the entry's screen (host copy), walked over what is left behind a first match
    int stream = 0;
    double cost = 0, cost0 = 0, cost1 = 0;  // heuristic; measured round 0 / leftovers (ns, 0 = unknown)
    GrokEntryDev dev{};
    uint32_t *listA = nullptr, *listB = nullptr, *unanchored = nullptr;
    int32_t* caps = nullptr;
    uint8_t* status = nullptr;
};

int grokMatchSpeculative(const std::vector<GrokDevicePattern>& patterns, GrokDeviceState* state, const GrokOptions& opts, uint32_t row,
                         const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len, uint32_t n, int32_t* d_pattern,
                         int32_t* d_first, int32_t* d_extra, uint32_t extraCap, uint32_t* d_nextra, void* d_scratch, hipStream_t st,
                         int dev) {
    GrokBatchStats& stats = tlsStats;
    const uint32_t nP = uint32_t(patterns.size());
    static const uint32_t envStreams = [] {  // LC_GROK_STREAMS overrides the handle's option (A/B measurements)
        const char* e = getenv("LC_GROK_STREAMS");
        return uint32_t(e ? atoi(e) : 0);
    }();
    const uint32_t nStreams = std::max(1u, std::min<uint32_t>(envStreams ? envStreams : opts.streams, kGrokMaxStreams));
    PlanThread& T = tlsPlan;
    {
        int rc = T.ensure(dev, nStreams);
        if (rc != LC_OK) return rc;
    }
    static const bool trace = getenv("LC_GROK_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto msNow = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };

    // scratch: winner u32[n] | undecided u32[n] from the head, tail words / perEntry / masks at their fixed places
    uint8_t* head = static_cast<uint8_t*>(d_scratch);
    uint32_t* winner = reinterpret_cast<uint32_t*>(head);
    uint32_t* undecided = winner + n;
    uint8_t* tailAt = head + alignUp(size_t(n) * row * 4, 256) + alignUp(n, 256) + 7 * alignUp(size_t(n) * 4, 256);
    uint32_t* tail = T.dTail;                                        // TW_* (the entries' counters behind them: one block, one copy back)
    uint32_t* perEntry = T.dPlanWords;                               // [64] (then firstOf[64], shadow[64][64]: one block, one copy back)
    uint64_t* masks = reinterpret_cast<uint64_t*>(tailAt + 512);

    const uint32_t gridAll = (n + kGrokPlanBlock - 1) / kGrokPlanBlock;
    // ---- phase 1: length order, literal index, all screens, candidate counts
    // Small batches wait for their LONGEST value (every kernel below is a dependent chain per value): chunk-parallel literal pass,
    // screen tables in LDS.  Large batches are about values in flight and equal work per wavefront: lane-per-value in length order.
    static const uint32_t smallBatch = [] {  // LC_GROK_SMALL_BATCH: A/B measurements
        const char* v = getenv("LC_GROK_SMALL_BATCH");
        return v ? uint32_t(atol(v)) : kGrokSmallBatch;
    }();
    const bool small = n <= smallBatch;
    uint32_t* order = reinterpret_cast<uint32_t*>(head + alignUp(size_t(n) * row * 4, 256) + alignUp(n, 256));
    uint32_t* orderWork = reinterpret_cast<uint32_t*>(tailAt + 512 + alignUp(size_t(n) * 8, 256));  // work words behind the masks
    {   // everything the batch needs cleared, in one launch (round 6: seven hipMemsetAsync calls between the kernels of phase 1)
        GrokInitJobs J{};
        auto fill = [&](void* p, size_t bytes, uint32_t value) {
            J.p[J.n] = static_cast<uint32_t*>(p);
            J.words[J.n] = uint32_t(bytes / 4);
            J.value[J.n] = value;
            ++J.n;
        };
        fill(T.dTail, size_t(64 + 64 * GC_WORDS) * 4, 0u);
        fill(T.dPlanWords, size_t(kPlanWords) * 4, 0u);
        fill(winner, size_t(n) * 8, 0xFFFFFFFFu);
        fill(d_first, size_t(n) * row * 4, 0xFFFFFFFFu);
        fill(d_nextra, 4, 0u);
        const uint32_t most = uint32_t(std::max(size_t(n) * 2, size_t(n) * row));
        hipLaunchKernelGGL(grok_init_kernel, dim3(std::min(2048u, (most + kGrokPlanBlock - 1) / kGrokPlanBlock)), dim3(kGrokPlanBlock), 0, st, J);
    }
    {
        int rc = lcLengthOrderOnStream(d_off, d_len, 0, n, orderWork, order, st);
        if (rc != LC_OK) return rc;
    }
    const uint32_t* literalIndex = nullptr;
    {
        int rc = grokLiteralIndex(patterns, state, dev, &literalIndex);
        if (rc != LC_OK) return rc;
    }
    // (round 6) the index's tables in LDS when two workgroups of them fit a CU: grok_literal_lds_kernel.  LC_GROK_LITERAL_LDS=0: through L2.
    const uint32_t litStageBytes = [&]() -> uint32_t {
        const char* v = getenv("LC_GROK_LITERAL_LDS");
        if (!literalIndex || !small || (v && v[0] == '0')) return 0u;
        const uint32_t bytes = uint32_t(state->literalBlob.size() * 4) - state->literalBlob[GL_OFF_MASKS];
        return ((state->literalBlob[GL_OFF_MASKS] & 7u) == 0 && bytes <= 78 * 1024) ? bytes : 0u;
    }();
    if (litStageBytes) {
        static thread_local size_t litAttr[kLcMaxDevices] = {};
        if (litStageBytes > 48 * 1024 && dev < kLcMaxDevices && litStageBytes > litAttr[dev]) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(grok_literal_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(litStageBytes)));
            litAttr[dev] = litStageBytes;
        }
        static thread_local uint32_t cus[kLcMaxDevices] = {};
        if (dev < kLcMaxDevices && !cus[dev]) {
            hipDeviceProp_t prop;
            HIP_TRY(hipGetDeviceProperties(&prop, dev));
            cus[dev] = uint32_t(prop.multiProcessorCount);
        }
        const uint32_t slots = 2u * (dev < kLcMaxDevices && cus[dev] ? cus[dev] : 256u);
        lcNoteKernel("grok_literal_lds_kernel");
        const uint32_t quadBlocks = ((n + 3) / 4 + kGrokPlanBlock / 64 - 1) / (kGrokPlanBlock / 64);  // (a wavefront takes four values at a time)
        hipLaunchKernelGGL(grok_literal_lds_kernel, dim3(std::min(slots, quadBlocks)), dim3(kGrokPlanBlock), litStageBytes, st, d_data, d_off, d_len, n,
                           literalIndex, masks, litStageBytes, static_cast<const uint32_t*>(order));
    } else if (literalIndex && small) {
        lcNoteKernel("grok_literal_chunk_kernel");
        hipLaunchKernelGGL(grok_literal_chunk_kernel, dim3((n + kGrokPlanBlock / 64 - 1) / (kGrokPlanBlock / 64)), dim3(kGrokPlanBlock), 0, st,
                           d_data, d_off, d_len, n, literalIndex, masks);
    } else if (literalIndex) {
        lcNoteKernel("grok_literal_index_kernel");
        hipLaunchKernelGGL(grok_literal_index_kernel, dim3(gridAll), dim3(kGrokBlock), 0, st, d_data, d_off, d_len, n, literalIndex, masks,
                           nP, static_cast<uint32_t*>(nullptr), static_cast<const uint32_t*>(order));
    } else {
        hipLaunchKernelGGL(grok_mask_fill_kernel, dim3(gridAll), dim3(kGrokPlanBlock), 0, st, masks, n,
                           nP >= 64 ? ~0ull : ((1ull << nP) - 1ull));
    }
    const GrokScreenDev* screens = nullptr;
    uint32_t nScreens = 0, screenLds = 0;
    {
        int rc = grokScreenTable(patterns, state, dev, &screens, &nScreens, &screenLds);
        if (rc != LC_OK) return rc;
    }
    if (nScreens) {
        // slices: short enough that a small batch still spreads over the chip, long enough that the table staging is amortised
        uint32_t sliceLen = ((n / 64 + 255) / 256) * 256;
        sliceLen = std::max(256u, std::min(small ? 4096u : 1024u, sliceLen));
        // (round 5) small batches: never more than one value per lane -- the slices are in length order, the first one holds the 4 KiB
        // values, and a screen that half of them carry walked two of them per lane, one after the other (LC_GROK_SLICE: A/B)
        static const uint32_t sliceEnv = [] {
            const char* v = getenv("LC_GROK_SLICE");
            return uint32_t(v ? atoi(v) : 0);
        }();
        if (small) sliceLen = sliceEnv ? std::max(256u, sliceEnv / 256 * 256) : 256u;
        // (round 6) TRIED AND LEFT OFF (LC_GROK_SCREEN_WAVE=1 switches it on): one value per WAVEFRONT (grokScreenWalkWave: a run of bytes
        // that keep the state is crossed 256 bytes at a time).  A relaxed screen takes 130-400 real steps and 80-130 run scans per value
        // (mean 1.1 KB): 50-250 us of a wavefront -- fine for the screens whose literal leaves eight candidates per slice, hopeless for
        // the five whose entries have no literal and see EVERY value: 64 values per wavefront one after the other, 31 ms for a
        // 1000-value batch against 0.44 ms one value per lane (profiles/round6_grok_screen_wave_negative.txt).
        const uint32_t screenWalk = [&] {
            const char* v = getenv("LC_GROK_SCREEN_WAVE");
            return (small && v && v[0] == '1') ? 16u : 0u;
        }() | [&] {  // (round 6) staged tables are staged SCALED (grok_plan_kernel.hpp grokScreenWalkScaled); LC_GROK_SCREEN_SCALED=0: as before
            const char* v = getenv("LC_GROK_SCREEN_SCALED");
            return (v && v[0] == '0') ? 0u : 32u;
        }();
        const uint32_t slices = (n + sliceLen - 1) / sliceLen;
        lcNoteKernel("grok_screen_all_kernel");
        // Round 5: a relaxed whole-pattern screen of 1 000-2 000 states (70-130 KB) does not fit beside other workgroups and walked its
        // table through L2 -- 120 ns a byte, 0.5 ms for a 4 KiB value: the whole screen phase of a small batch waited for three such
        // screens (profiles/round5_grok_timeline.txt).  They now run in a launch of their own, on a worker stream beside the other
        // screens' launch, each workgroup with the table staged into a CU's whole LDS.  MEASURED AND LEFT OFF (LC_GROK_BIG_SCREENS=1
        // switches it on): what the phase waited for was not the table reads of those three screens but the walk itself, in every
        // screen (two values per lane of 4 KiB each); with the walk fixed (grokScreenWalk) a 106 KB workgroup per CU costs more than
        // its faster table reads gain: 16 Ki values 3.92 ms without, 4.13 ms with (profiles/round5_grok_steps.txt).
        const bool bigOff = [] {  // (measured and left off: see above; read per batch -- the GPU tests switch it on)
            const char* v = getenv("LC_GROK_BIG_SCREENS");
            return !(v && v[0] == '1');
        }();
        const uint32_t nBig = (small && !bigOff) ? state->nBigScreens[dev] : 0u;
        if (nBig) {
            const size_t bigLds = size_t(sliceLen) * 4 + state->bigScreenLdsBytes[dev];
            static thread_local size_t attrSet[kLcMaxDevices] = {};
            if (bigLds > 48 * 1024 && dev < kLcMaxDevices && bigLds > attrSet[dev]) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(grok_screen_all_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            int(bigLds)));
                attrSet[dev] = bigLds;
            }
            HIP_TRY(hipEventRecord(T.fork, st));
            HIP_TRY(hipStreamWaitEvent(T.workers[0], T.fork, 0));
            hipLaunchKernelGGL(grok_screen_all_kernel, dim3(slices, nBig), dim3(kGrokPlanBlock), bigLds, T.workers[0], d_data, d_off, d_len, n,
                               sliceLen, screens, reinterpret_cast<unsigned long long*>(masks), static_cast<const uint32_t*>(order), 2u | screenWalk);
            HIP_TRY(hipEventRecord(T.join[0], T.workers[0]));
        }
        static const bool splitScreens = getenv("LC_GROK_SCREEN_SPLIT") != nullptr;  // diagnosis: one launch per screen (a kernel trace then times each)
        if (splitScreens) {
            for (uint32_t k = nBig; k < nScreens; ++k)
                hipLaunchKernelGGL(grok_screen_all_kernel, dim3(slices, 1), dim3(kGrokPlanBlock), size_t(sliceLen) * 4 + (small ? screenLds : 0), st,
                                   d_data, d_off, d_len, n, sliceLen, screens + k, reinterpret_cast<unsigned long long*>(masks),
                                   static_cast<const uint32_t*>(order), (small ? 1u : 0u) | screenWalk);
        } else if (nScreens > nBig) {
            // (grid (screens, slices): the slices of the longest values first -- grok_plan_kernel.hpp; LC_GROK_SCREEN_TRANSPOSED=0: as before)
            const char* tv = getenv("LC_GROK_SCREEN_TRANSPOSED");
            const bool transposed = !(tv && tv[0] == '0') && slices <= 65535u;
            hipLaunchKernelGGL(grok_screen_all_kernel, transposed ? dim3(nScreens - nBig, slices) : dim3(slices, nScreens - nBig), dim3(kGrokPlanBlock),
                               size_t(sliceLen) * 4 + (small ? screenLds : 0), st, d_data, d_off, d_len, n, sliceLen, screens + nBig,
                               reinterpret_cast<unsigned long long*>(masks), static_cast<const uint32_t*>(order),
                               (small ? 1u : 0u) | screenWalk | (transposed ? 128u : 0u));
        }
        if (nBig) HIP_TRY(hipStreamWaitEvent(st, T.join[0], 0));
        if (trace && nBig)
            fprintf(stderr, "grok plan 1: %u big screens in a launch of their own (%u KB of LDS per workgroup)\n", nBig,
                    unsigned((size_t(sliceLen) * 4 + state->bigScreenLdsBytes[dev]) >> 10));
    }
    uint32_t* firstOf = T.dPlanWords + 64;   // [64]
    uint32_t* shadowBy = T.dPlanWords + 128;  // [64][64]
    hipLaunchKernelGGL(grok_count_kernel, dim3(gridAll), dim3(kGrokPlanBlock), 0, st, masks, n, nP, perEntry, firstOf, shadowBy);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(T.hostWords + HW_CAND, T.dPlanWords, size_t(kPlanWords) * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(syncCounted(st));  // sync 1: candidates per entry
    const double tPhase1 = msNow();

    // ---- the batch's active entries and their arrays
    std::vector<PlanEntry> act;
    size_t arenaBytes = 0;
    auto carve = [&](size_t bytes) {
        const size_t at = arenaBytes;
        arenaBytes += alignUp(bytes, 256);
        return at;
    };
    const uint32_t xcap = std::max<uint32_t>(extraCap, n / 4 + 1024);
    const uint32_t xstride = row + 3;
    const size_t xtmpAt = carve(size_t(xcap) * xstride * 4);
    struct Offsets {
        size_t off, len, line, from, nmatch, first, listA, listB, unanchored, ovList, caps, status;
    };
    std::vector<Offsets> offs;
    uint32_t levelOf[64] = {};
    for (uint32_t p = 0; p < nP; ++p) {
        const uint32_t c = T.hostWords[HW_CAND + p];
        if (!c) continue;
        PlanEntry e;
        e.p = p;
        e.cand = c;
        e.columns = patterns[p].columns;
        e.capsRow = 2 * (patterns[p].columns + 1);
        e.rounds = std::max(1u, std::min(kGrokMaxRounds, patterns[p].re->grokRounds.load(std::memory_order_relaxed)));
        for (const GrokScreenDev& sd : state->hostScreens[dev])
            if (sd.bit == p) e.remainderScreen = &sd;
        e.rounds = std::max(2u, e.rounds);  // (round 0 is a phase of its own; round 1 reads the screened list)
        const bool nfa = patterns[p].re->engine == LC_ENGINE_NFA;
        e.cost = double(c) * (nfa ? (patterns[p].anchored ? 8.0 : 40.0) : 1.0);
        // (measured, once a calibration batch has run: ns per candidate of round 0 and of the leftovers)
        const uint32_t c0 = patterns[p].re->grokCost0Ns.load(std::memory_order_relaxed), c1 = patterns[p].re->grokCost1Ns.load(std::memory_order_relaxed);
        e.cost0 = c0 ? double(c0) * c : 0.0;
        e.cost1 = c1 ? double(c1) * c : 0.0;
        // an automaton that walks its tables in global memory (tdfa_l2_kernel) takes as long as its LONGEST candidate -- one dependent
        // read per byte, ~0.7 us per byte, 2.8 ms for a 4 KiB line -- on a grid of a few waves: it goes to the head of its stream,
        // where it runs beside everything else, instead of behind the cheap entries (measured: the step's last 2.8 ms were this)
        auto walksGlobalTables = [](const lc_regex* re) { return re && re->engine == LC_ENGINE_TDFA && !re->hasTdfa && !re->tdfaL2Blob.empty(); };
        // (small batches only -- measured 6.36 -> 4.02 ms on 1000 values, 5.67 -> 6.34 ms on 16 Ki: the runtime maps the worker streams
        // onto a few hardware queues, and with thousands of candidates per entry the wide thread-list kernels, not these, are the
        // long poles; sharing half of the streams among these entries was worse for both sizes)
        if (small && (walksGlobalTables(patterns[p].re) || walksGlobalTables(patterns[p].anchored))) e.cost += 1e12;
        Offsets o;
        o.off = carve(size_t(c) * 4);
        o.len = carve(size_t(c) * 4);
        o.line = carve(size_t(c) * 4);
        o.from = carve(size_t(c) * 4);
        o.nmatch = carve(size_t(c) * 4);
        o.listA = carve(size_t(c) * 4);
        o.listB = carve(size_t(c) * 4);
        o.unanchored = carve(size_t(c) * 4);
        o.ovList = carve(size_t(c) * 4);
        o.first = carve(size_t(c) * e.capsRow * 4);
        o.caps = carve(size_t(c) * e.capsRow * 4);
        o.status = carve(size_t(c) + 16);
        // second pass: see phase 2
        const uint32_t shadowed = c - std::min(c, T.hostWords[HW_FIRST + p]);
        // levels: an entry half of whose candidates have an earlier candidate entry waits for the entries that shadow a quarter of
        // them or more (SYSLOGLINE behind CRONLOG / HTTPD_ERRORLOG; SHOREWALL behind SYSLOGLINE; COMBINEDAPACHELOG behind
        // COMMONAPACHELOG): what those win it never looks at.  Results do not depend on the order -- only the work does.
        // (round 6) ... unless the entry's round 0 is a TABLE WALK (a complete automaton in global memory, or a lazy one): then looking at
        // values an earlier entry will win costs a few microseconds of a wavefront each, while waiting costs the batch a host round
        // trip and the entry's longest walk a second time (0.4 ms in phase 2c, profiles/round6_grok_timeline.txt).  LC_GROK_FLAT=0: levels as in round 5.
        const bool flat = [&] {
            const char* v = getenv("LC_GROK_FLAT");
            if (v && v[0] == '0') return false;
            const lc_regex* first = patterns[p].anchored ? patterns[p].anchored : patterns[p].re;
            if (first->engine == LC_ENGINE_TDFA) return !first->tdfaL2Blob.empty() && (first->preferWave || !first->hasTdfa);
            return first->engine == LC_ENGINE_NFA && first->lazyReady.load(std::memory_order_acquire);
        }();
        if (shadowed * 2 >= c && !flat) {
            uint32_t lv = 1;
            for (uint32_t f = 0; f < p; ++f)
                if (T.hostWords[HW_SHADOW + p * 64 + f] * 4 >= c) lv = std::max(lv, levelOf[f] + 1);
            e.level = std::min(lv, 3u);
        }
        levelOf[p] = e.level;
        e.second = e.level != 0;
        offs.push_back(o);
        act.push_back(e);
    }
    // (the device table lists the entries of the first pass first: its finish kernel runs over a prefix of the table)
    uint32_t nSecond = 0;
    {
        std::vector<size_t> idx(act.size());
        for (size_t a = 0; a < idx.size(); ++a) idx[a] = a;
        std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return act[x].level < act[y].level; });
        std::vector<PlanEntry> act2;
        std::vector<Offsets> offs2;
        for (size_t a : idx) {
            act2.push_back(act[a]);
            offs2.push_back(offs[a]);
            nSecond += act[a].second;
        }
        act.swap(act2);
        offs.swap(offs2);
    }
    stats.activeEntries = uint32_t(act.size());
    GrokSlotMap map;
    for (auto& a : map.activeOfBit) a = -1;
    {
        int rc = T.ensureArena(arenaBytes);
        if (rc != LC_OK) return rc;
    }
    uint8_t* arena = static_cast<uint8_t*>(T.arena);
    int32_t* xtmp = reinterpret_cast<int32_t*>(arena + xtmpAt);
    uint32_t maxCand = 0, nRemScreens = 0, remScreenLds = 0;
    for (size_t a = 0; a < act.size(); ++a) {
        PlanEntry& e = act[a];
        const Offsets& o = offs[a];
        e.dev.off = reinterpret_cast<uint32_t*>(arena + o.off);
        e.dev.len = reinterpret_cast<uint32_t*>(arena + o.len);
        e.dev.line = reinterpret_cast<uint32_t*>(arena + o.line);
        e.dev.from = reinterpret_cast<uint32_t*>(arena + o.from);
        e.dev.nmatch = reinterpret_cast<uint32_t*>(arena + o.nmatch);
        e.dev.first = reinterpret_cast<int32_t*>(arena + o.first);
        e.dev.cnt = T.dCnt + a * GC_WORDS;
        e.dev.cand = e.cand;
        e.dev.capsRow = e.capsRow;
        e.dev.bit = e.p;
        e.listA = reinterpret_cast<uint32_t*>(arena + o.listA);
        e.listB = reinterpret_cast<uint32_t*>(arena + o.listB);
        e.unanchored = reinterpret_cast<uint32_t*>(arena + o.unanchored);
        e.caps = reinterpret_cast<int32_t*>(arena + o.caps);
        e.status = arena + o.status;
        e.dev.status = e.status;
        e.dev.caps = e.caps;
        e.dev.listA = e.listA;
        e.dev.unanchored = e.unanchored;
        e.dev.ovList = reinterpret_cast<uint32_t*>(arena + o.ovList);
        e.dev.columns = e.columns;
        e.dev.anchored = patterns[e.p].anchored ? 1u : 0u;
        T.hostEntries[a] = e.dev;
        T.hostRemScreens[a] = e.remainderScreen ? *e.remainderScreen : GrokScreenDev{nullptr, e.p, 0u, 0u, 0u};
        if (e.remainderScreen) {
            ++nRemScreens;
            remScreenLds = std::max(remScreenLds, e.remainderScreen->ldsBytes);
        }
        map.activeOfBit[e.p] = int8_t(a);
        maxCand = std::max(maxCand, e.cand);
        stats.pairs += e.cand;
    }
    uint32_t* gate = tail + TW_GATE;
    uint32_t* xcount = tail + TW_XCOUNT;
    const uint32_t nAct = uint32_t(act.size());
    // ---- phase 3: the first contributing entry per value; its rows go out.  All of it returns at once when values are still in
    // play somewhere (gate != 0): the host then finishes those entries and queues it again
    const uint32_t gridCand = (maxCand + kGrokPlanBlock - 1) / kGrokPlanBlock;
    auto finishAll = [&](const uint32_t* g) -> int {
        if (nAct) {
            hipLaunchKernelGGL(grok_entry_finish_kernel, dim3(gridCand, nAct), dim3(kGrokPlanBlock), 0, st, T.dEntries, winner, undecided, g, 0u);
        }
        hipLaunchKernelGGL(grok_resolve_kernel, dim3(gridAll), dim3(kGrokPlanBlock), 0, st, n, winner, undecided, d_pattern, g);
        if (nAct) {
            // (row nAct of the grid: the extra rows of the winners -- grok_commit_extra_kernel's launch until round 6)
            hipLaunchKernelGGL(grok_commit_kernel, dim3(std::max(gridCand, 64u), nAct + 1), dim3(kGrokPlanBlock), 0, st, T.dEntries, nAct, d_pattern,
                               d_first, row, g, xtmp, xcount, xcap, xstride, map, d_extra, extraCap, d_nextra, tail + TW_NEXTRA);
        }
        HIP_TRY(hipGetLastError());
        // (tail words, the copy of nextra among them, and the entries' counters: one block)
        HIP_TRY(hipMemcpyAsync(T.hostWords + HW_TAIL, T.dTail, (64 + size_t(nAct) * GC_WORDS) * 4, hipMemcpyDeviceToHost, st));
        if (tlsTail.armed) {
            HIP_TRY(hipMemcpyAsync(tlsTail.hPattern, d_pattern, size_t(n) * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(tlsTail.hFirst, d_first, size_t(n) * row * 4, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(syncCounted(st));
        if (tlsTail.armed) {
            tlsTail.done = true;
            tlsTail.nextra = T.hostWords[HW_TAIL + TW_NEXTRA];
        }
        return LC_OK;
    };
    bool finished = false;  // finishAll has run (phase 2 ended with it: see the end of phase 2c)
    if (nAct) {
        // (the entry table and, behind it, the remainder screens: one block)
        HIP_TRY(hipMemcpyAsync(T.dEntries, T.hostEntries, 64 * sizeof(GrokEntryDev) + nAct * sizeof(GrokScreenDev), hipMemcpyHostToDevice, st));
        (void)nRemScreens;  // (dCnt: cleared by grok_init_kernel at the start of the batch)
        hipLaunchKernelGGL(grok_scatter_kernel, dim3(gridAll), dim3(kGrokPlanBlock), 0, st, masks, n, nP, d_off, d_len, map, T.dEntries,
                           static_cast<const uint32_t*>(order));
        HIP_TRY(hipGetLastError());
        // ---- phase 2 (round 4).  A batch used to queue 14-17 launches per entry -- 560 for the 46 active entries of configs[2] -- most of
        // them over lists that turn out empty, and the host could not queue them as fast as the device finished them (18 us a launch).
        // Now:  2a  round 0, ONE launch per entry (the engine's main kernel, the anchored search where the entry has one), dearest
        //           entry first, dealt round-robin to the worker streams;
        //       2b  grok_post_kernel, ONE launch for all entries: overflowed slots -> the entry's overflow list, slots the anchored
        //           search did not match -> its unanchored list, matches recorded, what is still in play -> listA;   counts -> host
        //       2c  only the entries that have any: the second chance for the overflowed slots, the search proper for the unanchored
        //           ones, each followed by a post launch over that list;
        //       2d  grok_remainder_all_kernel, ONE launch: the remainder of every slot in play against its entry's screen; counts -> host
        //       2e  only the entries with survivors: the search rounds behind the first match, queued ahead as before.
        std::vector<size_t> byCost(nAct);
        for (size_t a = 0; a < nAct; ++a) byCost[a] = a;
        std::sort(byCost.begin(), byCost.end(), [&](size_t x, size_t y) { return act[x].cost > act[y].cost; });
        const uint32_t used = std::min<uint32_t>(nStreams, nAct);
        // (round 6) A worker stream waits for the fork when it first GETS work (W), and only the streams that got any are joined: a
        // phase forks into sixteen streams and usually uses three, and the thirteen idle ones cost the host 26 calls per fork / join
        // and the batch's stream thirteen barrier packets that it works through one after the other (0.09 ms between the end of round
        // 0 and the first kernel behind the join, profiles/round6_grok_timeline.txt).
        uint32_t waitedMask = 0, touchedMask = 0;
        auto fork = [&]() -> int {
            HIP_TRY(hipEventRecord(T.fork, st));
            waitedMask = 0;  // (everybody waits for THIS fork from now on; what was touched before stays to be joined)
            return LC_OK;
        };
        auto W = [&](int s) -> hipStream_t {  // (a failure surfaces at the join: hipGetLastError)
            const uint32_t bit = 1u << uint32_t(s);
            if (!(waitedMask & bit)) {
                (void)hipStreamWaitEvent(T.workers[s], T.fork, 0);
                waitedMask |= bit;
            }
            touchedMask |= bit;
            return T.workers[s];
        };
        auto join = [&](int rc) -> int {
            lcSetDecideSlot(0);
            if (rc == LC_OK && hipGetLastError() != hipSuccess) rc = lcHipFail(hipGetLastError(), "grok entry launch");
            if (rc != LC_OK) {
                for (uint32_t s = 0; s < used; ++s) (void)hipStreamSynchronize(T.workers[s]);
                return rc;
            }
            for (uint32_t s = 0; s < used; ++s) {
                if (!((touchedMask >> s) & 1u)) continue;
                HIP_TRY(hipEventRecord(T.join[s], T.workers[s]));
                HIP_TRY(hipStreamWaitEvent(st, T.join[s], 0));
            }
            touchedMask = 0;
            return LC_OK;
        };
        auto readCounts = [&]() -> int {
            HIP_TRY(hipMemcpyAsync(T.hostWords + HW_CNT, T.dCnt, size_t(nAct) * GC_WORDS * 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(syncCounted(st));
            return LC_OK;
        };
        auto cnt = [&](size_t a, uint32_t word) { return T.hostWords[HW_CNT + a * GC_WORDS + word]; };
        // Which stream an entry goes to.  A phase lasts as long as its most loaded stream, and the entries differ by a factor of ten
        // (an anchored DFA over 300 values: 0.1 ms; the thread-list engine on a format with a free-text field in the middle: 3 ms).
        // Calibration batches (the first three of a handle, then every 64th) time every entry's launches with events; the matcher
        // then deals longest-first to the least loaded stream.  Before the first measurement: the static estimate.
        bool calibrate = false;
        {
            bool allKnown = true;
            for (size_t a = 0; a < nAct; ++a) allKnown = allKnown && act[a].cost0 > 0;
            const uint32_t seen = patterns[act[0].p].re->grokBatches.fetch_add(1, std::memory_order_relaxed);
            calibrate = !allKnown || seen < 3 || (seen & 63u) == 0;
            if (calibrate)
                for (size_t k = 0; k < 4 * nAct; ++k)
                    if (!T.tick[k]) HIP_TRY(hipEventCreate(&T.tick[k]));
        }
        std::vector<char> busy2c(nAct, 0);  // phase 2c: the entries that have anything to do in it
        auto deal = [&](bool leftovers) {  // longest first, each to the least loaded stream
            std::vector<size_t> idx;
            for (size_t a = 0; a < nAct; ++a) idx.push_back(a);
            auto costOf = [&](size_t a) {
                const double known = leftovers ? act[a].cost1 : act[a].cost0;
                return known > 0 ? known : act[a].cost * 50.0;
            };
            std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return costOf(x) > costOf(y); });
            double load[kGrokMaxStreams] = {};
            for (size_t a : idx) {
                if (leftovers ? (!busy2c[a] || act[a].level >= 2) : act[a].level != 0) continue;
                uint32_t best = 0;
                for (uint32_t s2 = 1; s2 < used; ++s2)
                    if (load[s2] < load[best]) best = s2;
                act[a].stream = int(best);
                load[best] += costOf(a) + 20000.0;  // (+ a launch)
            }
            return idx;
        };
        // ---- round 5's plan knobs (each one can be switched off for A/B measurements; read per batch: the GPU tests flip them)
        auto envInt = [](const char* name, int dflt) {
            const char* v = getenv(name);
            return v ? atoi(v) : dflt;
        };
        const int wideFirstMode = envInt("LC_GROK_WIDE_FIRST", 1);      // 0 off, 1 by the entry's history, 2 always
        const bool breadthFirst = envInt("LC_GROK_BREADTH", 1) != 0;    // phase 2c queues the chains' launches round-robin, heavy kernels first
        const int earlyRoundsMode = envInt("LC_GROK_EARLY_ROUNDS", 1);  // 0 off, 1 by the entry's history, 2 always
        const bool remainderLiteral = envInt("LC_GROK_REMAINDER_LITERAL", 1) != 0 && literalIndex != nullptr;
        const bool remainderWon = envInt("LC_GROK_REMAINDER_WON", 1) != 0;      // slots of values an earlier entry has won drop out in front of the screens
        const bool remainderInChain = envInt("LC_GROK_REMAINDER_INCHAIN", 1) != 0;  // the remainder screens per entry, at the end of its chain in phase 2c
        const bool screenWave = small && envInt("LC_GROK_SCREEN_WAVE", 0) != 0;  // remainder screens: one slot per wavefront (grokScreenWalkWave; off: see phase 1)
        const uint32_t remWalk = (screenWave ? 16u : 0u) | (envInt("LC_GROK_SCREEN_SCALED", 1) != 0 ? 32u : 0u);
        auto remGrid = [&](uint32_t slots) { return screenWave ? (slots + kGrokRemWaveSlots - 1) / kGrokRemWaveSlots : (slots + kGrokPlanBlock - 1) / kGrokPlanBlock; };
        // (round 5) the host reads the survivors of the remainder screens together with the finish of the batch (three synchronisations
        // instead of four: see the end of phase 2c).  Round 6: the screens count their survivors onto the gate word themselves.
        const bool lazy = envInt("LC_GROK_LAZY_SYNC3", 1) != 0;
        uint32_t* remGate = lazy ? gate : nullptr;
        // (round 6) the remainder screens of the entries round 0 leaves nothing else to do for are queued BEFORE the host reads round 0's
        // counts, on a stream beside it (the kernels test the entry's overflow / unanchored counts themselves): see phase 2b
        const bool remainderAhead = remainderInChain && envInt("LC_GROK_REMAINDER_AHEAD", 1) != 0;
        const bool postInStream = envInt("LC_GROK_POST_IN_STREAM", 1) != 0;    // round 0's post step behind each entry's kernel, on its stream
        const bool bigRemainder = envInt("LC_GROK_BIG_REMAINDER", 0) != 0;      // an entry with a BIG screen stages it for its remainder screen (measured: slower)
        // An entry whose values needed more than 64 threads in recent batches (GC_WIDE, noted behind the batch) goes WIDE FIRST: its
        // first chance is nfa_wide_kernel over every candidate, and what is left behind it are the decide kernels alone.  Round 4's
        // timeline: the longest entry's first chance 1.0 ms + its second chance (14 values restarted from byte 0) 1.1 ms, back to back
        // on the critical path of the batch; now 1.1 ms.
        auto wantsWideFirst = [&](lc_regex* h) {
            if (!wideFirstMode || h->engine != LC_ENGINE_NFA || !lcNfaWideApplies(h)) return false;
            if (wideFirstMode >= 2) return true;
            return n <= kGrokWideFirstBatch && h->grokOverflowSeen.load(std::memory_order_relaxed) != 0;
        };
        // one engine call of entry e on stream ws: the first-chance kernel ...
        auto runFirst = [&](PlanEntry& e, lc_regex* h, bool wide, const uint32_t* count, const uint32_t* list, bool withFrom, uint32_t* seq,
                            hipStream_t ws) -> int {
            const uint32_t* resume = withFrom ? e.dev.from : nullptr;
            if (wide)
                return lcMatchWideFirstOnStream(0, h, h->engine, dev, d_data, e.dev.off, e.dev.len, 0, e.cand, count, list, resume, e.capsRow / 2,
                                                e.caps, e.status, seq, e.dev.cnt + GC_WIDE, ws);
            return lcMatchFirstOnStream(h, h->engine, dev, d_data, e.dev.off, e.dev.len, 0, e.cand, count, list, resume, e.capsRow / 2, e.caps,
                                        e.status, seq, ws);
        };
        // ... and what is left behind it (note: the entry's GC_WIDE word, set when the wide kernel had anything to do)
        auto runSecond = [&](PlanEntry& e, lc_regex* h, bool wide, const uint32_t* count, const uint32_t* list, bool withFrom, uint32_t seq,
                             uint32_t* note, hipStream_t ws) -> int {
            const uint32_t* resume = withFrom ? e.dev.from : nullptr;
            if (wide)
                return lcMatchWideFirstOnStream(1, h, h->engine, dev, d_data, e.dev.off, e.dev.len, 0, e.cand, count, list, resume, e.capsRow / 2,
                                                e.caps, e.status, &seq, nullptr, ws);
            lcSetWideNote(note);
            const int r = lcMatchSecondChanceOnStream(h, h->engine, dev, d_data, e.dev.off, e.dev.len, 0, e.cand, count, list, resume,
                                                      e.capsRow / 2, e.caps, e.status, seq, ws);
            lcSetWideNote(nullptr);
            return r;
        };
        // 2a
        // Round 6: the entries whose round 0 is a walk of tables in global memory -- a complete tagged DFA, or the LAZY automaton of an
        // entry that does not determinise -- go in ONE launch (tdfa_wave_multi_kernel) with ONE post launch behind it.  Round 0 used to
        // be ~46 kernels of 30-100 us each plus their 46 post kernels, and lasted as long as the host needed to queue them: the last one
        // started 1.4 ms after the first (profiles/round6_grok_timeline_lazy.txt).  LC_GROK_FUSED_ROUND0=0: one launch per entry.
        std::vector<char> fused(nAct, 0);
        {
            std::vector<TdfaWaveJob> jobs;
            std::vector<size_t> jobEntry;
            uint32_t blocks = 0, ldsMax = 0;
            // (all entries' candidates together are what the ONE launch walks; LC_GROK_FUSED_STAGE=0/1 forces it: A/B)
            const bool stagePrograms = envInt("LC_GROK_FUSED_STAGE", stats.pairs <= 32768 ? 1 : 0) != 0;
            for (size_t a = 0; a < nAct && jobs.size() < kTdfaWaveMaxJobs; ++a) {
                PlanEntry& e = act[a];
                const GrokDevicePattern& gp = patterns[e.p];
                if (e.level || !e.cand) continue;
                lc_regex* first = gp.anchored ? gp.anchored : gp.re;
                if (wideFirstMode >= 2 && first->engine == LC_ENGINE_NFA) continue;  // (the knob forces the wide kernel: A/B, parity tests)
                TdfaWaveJob j{};
                uint32_t lds = 0, seq = 0;
                int rcJob = LC_OK;
                if (!lcWaveJobPrepare(first, dev, e.cand, stagePrograms, &j, &lds, &seq, &rcJob)) {
                    if (rcJob != LC_OK) return rcJob;
                    continue;
                }
                j.off = e.dev.off;
                j.len = e.dev.len;
                j.resume = gp.anchored ? nullptr : e.dev.from;
                j.caps = e.caps;
                j.status = e.status;
                j.n = e.cand;
                j.nGroupsOut = e.capsRow / 2;
                j.firstBlock = blocks;
                blocks += (e.cand + kTdfaWaveValues - 1) / kTdfaWaveValues;
                ldsMax = std::max(ldsMax, lds);
                e.seq0 = seq;        // (a lazy automaton's misses are LC_OVERFLOW under this number: the second chance of phase 2c takes them)
                e.wideFirst = false;
                fused[a] = 1;
                jobs.push_back(j);
                jobEntry.push_back(a);
            }
            bool others = false;
            for (size_t a = 0; a < nAct; ++a) others = others || (!act[a].level && !fused[a]);
            int rc = LC_OK;
            if (others) rc = fork();
            if (rc != LC_OK) return rc;
            hipStream_t fusedStream = others ? W(0) : st;
            if (!jobs.empty()) {
                rc = lcLaunchWaveJobs(d_data, jobs.data(), uint32_t(jobs.size()), blocks, ldsMax, T.hostJobs, T.dJobs, dev, fusedStream);
                // the post step of the fused entries: ONE launch over their span of the entry table (the entries in between that are not
                // the fused launch's leave at once: skip mask)
                if (rc == LC_OK) {
                    const size_t lo = jobEntry.front(), hi = jobEntry.back();
                    unsigned long long skip = 0;
                    uint32_t most = 0;
                    for (size_t a = lo; a <= hi; ++a) {
                        if (fused[a]) most = std::max(most, act[a].cand);
                        else skip |= 1ull << a;
                    }
                    hipLaunchKernelGGL(grok_post_kernel, dim3((most + kGrokPlanBlock - 1) / kGrokPlanBlock, uint32_t(hi - lo + 1)), dim3(kGrokPlanBlock), 0,
                                       fusedStream, T.dEntries, uint32_t(lo), static_cast<const uint32_t*>(nullptr),
                                       static_cast<const uint32_t*>(nullptr), uint32_t(GP_ANCHORED_PASS), xtmp, xcap, xstride, xcount, skip);
                }
                if (trace) fprintf(stderr, "grok plan 2a: %zu entries in one launch (%u workgroups, %u B of LDS)\n", jobs.size(), blocks, ldsMax);
            }
            const std::vector<size_t> order0 = deal(false);
            for (size_t i = 0; i < nAct && rc == LC_OK && others; ++i) {
                const size_t a = order0[i];
                PlanEntry& e = act[a];
                const GrokDevicePattern& gp = patterns[e.p];
                if (e.level || fused[a]) continue;  // (level >= 1: waits for the entries that shadow it, phase 2c)
                lcSetDecideSlot(1 + e.stream);
                lc_regex* first = gp.anchored ? gp.anchored : gp.re;
                e.wideFirst = wantsWideFirst(first);
                // (failures inside the forked region become rc: the workers are always joined below)
                if (calibrate && hipEventRecord(T.tick[2 * a], W(e.stream)) != hipSuccess) rc = lcHipFail(hipGetLastError(), "hipEventRecord(calibration)");
                if (rc == LC_OK) rc = runFirst(e, first, e.wideFirst, nullptr, nullptr, !gp.anchored, &e.seq0, W(e.stream));
                if (calibrate) (void)hipEventRecord(T.tick[2 * a + 1], W(e.stream));
                // (round 5) the entry's post step right behind its kernel: the one launch for all entries behind the join (0.12 ms) waited
                // for the slowest entry and then stood between it and the host's read of the counts
                if (rc == LC_OK && postInStream)
                    hipLaunchKernelGGL(grok_post_kernel, dim3((e.cand + kGrokPlanBlock - 1) / kGrokPlanBlock, 1), dim3(kGrokPlanBlock), 0,
                                       W(e.stream), T.dEntries, uint32_t(a), static_cast<const uint32_t*>(nullptr),
                                       static_cast<const uint32_t*>(nullptr), uint32_t(GP_ANCHORED_PASS), xtmp, xcap, xstride, xcount, 0ull);
                if (trace)
                    fprintf(stderr, "grok plan 2a: entry %u cand %u stream %d engine %s%s%s positions %zu slots %d atomic %d | measured round 0 %.3f ms, leftovers %.3f ms\n",
                            e.p, e.cand, e.stream, first->engine == LC_ENGINE_NFA ? "nfa" : first->hasTdfa ? "tdfa-lds" : "tdfa-l2",
                            gp.anchored ? " (anchored)" : "", e.wideFirst ? " (wide first)" : "", first->nfa.positions.size(), first->nfa.slotCount(),
                            first->nfa.atomicCount, e.cost0 * 1e-6, e.cost1 * 1e-6);
            }
            if (others) rc = join(rc);
            else if (rc == LC_OK && hipGetLastError() != hipSuccess) rc = lcHipFail(hipGetLastError(), "grok round 0");
            if (rc != LC_OK) return rc;
        }
        // 2b: round 0's results of the level-0 entries; the values won so far
        const uint32_t gridCand0 = (maxCand + kGrokPlanBlock - 1) / kGrokPlanBlock;
        lcNoteKernel("grok_post_kernel");
        const uint32_t nLevel0 = nAct - nSecond;  // (the table lists the entries by level)
        if (nLevel0 && !postInStream)
            for (uint32_t a = 0; a < nLevel0; ++a)  // (the entries of the fused launch have had their post step behind it)
                if (!fused[a])
                    hipLaunchKernelGGL(grok_post_kernel, dim3((act[a].cand + kGrokPlanBlock - 1) / kGrokPlanBlock, 1), dim3(kGrokPlanBlock), 0, st,
                                       T.dEntries, a, static_cast<const uint32_t*>(nullptr), static_cast<const uint32_t*>(nullptr),
                                       uint32_t(GP_ANCHORED_PASS), xtmp, xcap, xstride, xcount, 0ull);
        uint32_t maxLevel = 0;
        for (size_t a = 0; a < nAct; ++a) maxLevel = std::max(maxLevel, act[a].level);
        // the values won so far, and for the entries of level >= 1 how many of their slots are still open: an entry all of whose
        // candidates an entry of level 0 has won (COMBINEDAPACHELOG behind COMMONAPACHELOG on a corpus without referrers) queues nothing
        // in phase 2c -- its dozen launches over empty lists were the tail of the phase
        const bool boundKnown = nSecond != 0 && envInt("LC_GROK_BOUND", 1) != 0;
        if (boundKnown) {
            hipLaunchKernelGGL(grok_entry_finish_kernel, dim3(gridCand0, nAct), dim3(kGrokPlanBlock), 0, st, T.dEntries, winner, undecided,
                               static_cast<const uint32_t*>(nullptr), 0u);
            hipLaunchKernelGGL(grok_bound_kernel, dim3(gridCand0, nSecond), dim3(kGrokPlanBlock), 0, st, T.dEntries, nLevel0,
                               static_cast<const uint32_t*>(winner));
        }
        // (Tried in round 5 and dropped -- profiles/round5_grok_steps.txt: the chains of level 1, and by the entries' history the
        // leftovers of level 0, queued BEFORE the host reads round 0's counts.  The device has them earlier, but a batch is bound by
        // the rate at which the host can queue launches, and the extra, mostly empty chains cost more than the earlier start gained:
        // 16 Ki values 4.76 -> 5.4 ms, 1000 values 3.9 -> 5.2 ms.)
        // (round 6) The remainder screens, AHEAD of the host's read of round 0's counts.  Round 5 queued them behind it -- the entries
        // with nothing else to do in phase 2c in one launch pair at the fork -- and that pair (0.5 ms: a few thousand remainders of up to
        // 4 KiB, one lane each) started 0.16 ms after round 0 had ended: a copy, a host round trip and the host's planning of phase 2c
        // (profiles/round6_grok_timeline.txt).  Which entries those are is a test of two counters the device has: the launch pair goes out
        // now, on a worker stream beside the copy, and leaves the entries with overflowed or unanchored slots (and the shadowed ones, and
        // the ones whose search rounds their history queues ahead: `skip`) to their chains.  LC_GROK_REMAINDER_AHEAD=0: as before.
        // (an entry's search rounds are queued ahead by its history; decided ONCE per batch -- another runner thread's batch on the same
        // handle may rewrite that history between here and phase 2c)
        std::vector<char> earlyOf(nAct, 0);
        for (size_t a = 0; a < nAct; ++a) {
            const GrokDevicePattern& gp = patterns[act[a].p];
            earlyOf[a] = (earlyRoundsMode && act[a].rounds >= 2 && gp.re->engine == LC_ENGINE_TDFA &&
                          (earlyRoundsMode >= 2 || (!calibrate && gp.re->grokRemainderSeen.load(std::memory_order_relaxed) != 0))) ? 1 : 0;
        }
        bool aheadLaunched = false;
        if (remainderAhead && nLevel0) {
            unsigned long long skip = 0;
            uint32_t most = 0, ldsAhead = 0;
            for (size_t a = 0; a < nAct; ++a) {
                if (act[a].level || earlyOf[a]) skip |= 1ull << a;
                else {
                    most = std::max(most, act[a].cand);
                    if (act[a].remainderScreen) ldsAhead = std::max(ldsAhead, act[a].remainderScreen->ldsBytes);
                }
            }
            if (most) {
                if (!boundKnown)  // the values won so far (the literal pass drops the slots of values an earlier entry has won)
                    hipLaunchKernelGGL(grok_entry_finish_kernel, dim3(gridCand0, nAct), dim3(kGrokPlanBlock), 0, st, T.dEntries, winner, undecided,
                                       static_cast<const uint32_t*>(nullptr), 0u);
                {
                    int rcFork = fork();
                    if (rcFork != LC_OK) return rcFork;
                }
                hipStream_t gs = W(int(used) - 1);
                static const bool splitRem = getenv("LC_GROK_REM_SPLIT") != nullptr;  // diagnosis: one launch pair per entry (a kernel trace then times each)
                for (uint32_t a1 = 0; a1 < (splitRem ? nAct : 1u); ++a1) {
                    const unsigned long long skip1 = splitRem ? ~(1ull << a1) : skip;
                    if (splitRem && ((skip >> a1) & 1ull)) continue;
                    if (remainderLiteral || remainderWon)
                        hipLaunchKernelGGL(grok_remainder_literal_kernel, dim3((most + kGrokPlanBlock / 64 - 1) / (kGrokPlanBlock / 64), nAct),
                                           dim3(kGrokPlanBlock), 0, gs, d_data, T.dEntries,
                                           remainderLiteral ? literalIndex : static_cast<const uint32_t*>(nullptr), skip1,
                                           remainderWon ? static_cast<const uint32_t*>(winner) : static_cast<const uint32_t*>(nullptr), 1u);
                    hipLaunchKernelGGL(grok_remainder_all_kernel, dim3(remGrid(most), nAct), dim3(kGrokPlanBlock), small ? ldsAhead : 0, gs, d_data,
                                       T.dEntries, static_cast<const GrokScreenDev*>(T.dRemScreens), (small ? 1u : 0u) | remWalk | 64u, skip1, remGate);
                }
                aheadLaunched = true;
            }
        }
        HIP_TRY(hipGetLastError());
        {
            int rc = readCounts();  // sync 2
            if (rc != LC_OK) return rc;
        }
        auto learn = [&](std::atomic<uint32_t>& slot, hipEvent_t b, hipEvent_t e2, uint32_t cand) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, b, e2) != hipSuccess) {
                (void)hipGetLastError();
                return;
            }
            const uint32_t ns = uint32_t(std::min(4.0e9, std::max(1.0, double(ms) * 1e6 / std::max(1u, cand))));
            const uint32_t old = slot.load(std::memory_order_relaxed);
            slot.store(old ? (old + ns) / 2 : ns, std::memory_order_relaxed);
        };
        if (calibrate)
            for (size_t a = 0; a < nAct; ++a)
                if (!act[a].level && !fused[a]) learn(patterns[act[a].p].re->grokCost0Ns, T.tick[2 * a], T.tick[2 * a + 1], act[a].cand);
        for (size_t a = 0; a < nAct; ++a)  // (an entry of the fused launch has no kernel of its own to time: a nominal cost keeps the handle "known")
            if (fused[a] && patterns[act[a].p].re->grokCost0Ns.load(std::memory_order_relaxed) == 0)
                patterns[act[a].p].re->grokCost0Ns.store(100, std::memory_order_relaxed);
        // 2c: what round 0 left undone on the entries of level 0 (second chance of overflowed slots; the search proper for what the
        // anchored search did not match -- minus the values an earlier entry has won meanwhile), and, level by level, the shadowed
        // entries on the values nobody before them has won.  ONE fork for all of it: the leftovers of level 0 and the entries of
        // level 1 depend on round 0 only; an entry of level 2 or 3 goes behind the chain of the entry that shadows most of its
        // candidates, on that entry's stream (stream order instead of a barrier per level).
        //
        // Round 5: a chain is built as a list of STEPS (one to three launches each) and the steps of all chains are queued round-robin,
        // dearest chain first.  Round 4 queued chain after chain: a dozen launches each, 15-20 us a launch -- the third shadowed
        // entry's kernel started 0.8 ms after the fork, behind sixty launches of other entries' mostly empty follow-up steps, and it,
        // not the second-chance kernel, ended the phase.  And an entry whose remainders mostly pass its screen (by its history:
        // grokRemainderSeen) gets its search rounds behind the first match queued HERE, at the end of its chain, unscreened: they run
        // in the shadow of this phase instead of in a phase of their own behind the remainder screens and another host round trip.
        using Step = std::function<void(int&)>;
        std::vector<std::vector<Step>> chains(nAct);
        std::vector<size_t> chainOwner(nAct);  // whose list an entry's steps went to (itself; a level >= 2 entry: its shadower's owner)
        for (size_t a = 0; a < nAct; ++a) chainOwner[a] = a;
        unsigned long long earlyMask = 0;  // active entries whose rounds were queued in this phase
        unsigned long long coveredMask = 0;  // ... whose remainder screens were
        unsigned long long groupMask = 0;    // ... of those, the entries of level 0 that have nothing else to do in this phase
        // rounds 1 .. of entry a (FindNextMatch from the end of the previous match; the list lengths stay on the device), one step per
        // round.  screened: round 1 reads the survivors of the remainder screen (phase 2e); else every slot in play (queued ahead)
        auto roundSteps = [&](size_t a, bool screened, std::vector<Step>& out) {
            PlanEntry& e0 = act[a];
            e0.queued = true;
            for (uint32_t r = 1; r < e0.rounds; ++r)
                out.push_back([&, a, r, screened](int& rc) {
                    if (rc != LC_OK) return;
                    PlanEntry& e = act[a];
                    const GrokDevicePattern& gp = patterns[e.p];
                    hipStream_t ws = W(e.stream);
                    lcSetDecideSlot(1 + e.stream);
                    const uint32_t grid = (e.cand + kGrokPlanBlock - 1) / kGrokPlanBlock;
                    const uint32_t* list = r == 1 ? (screened ? e.unanchored : e.listA) : (r & 1) ? e.listA : e.listB;
                    uint32_t* out2 = (r & 1) ? e.listB : e.listA;
                    const uint32_t* countPtr = r == 1 ? e.dev.cnt + (screened ? GC_REMAINDER : GC_ROUND0) : e.dev.cnt + GC_ROUND0 + r - 1;
                    rc = lcMatchOnStream(gp.re, gp.re->engine, dev, d_data, e.dev.off, e.dev.len, 0, e.cand, countPtr, list, e.dev.from,
                                         e.capsRow / 2, e.caps, e.status, ws);
                    if (rc != LC_OK) return;
                    hipLaunchKernelGGL(grok_advance2_kernel, dim3(grid), dim3(kGrokPlanBlock), 0, ws, list, e.cand, countPtr, e.status, e.caps,
                                       e.capsRow, e.columns, e.dev, xtmp, xcap, xstride, xcount, out2, e.dev.cnt + GC_ROUND0 + r,
                                       r + 1 == e.rounds ? gate : static_cast<uint32_t*>(nullptr));
                });
        };
        auto buildChain = [&](size_t a) {
            PlanEntry& e0 = act[a];
            const uint32_t level = e0.level;
            const uint32_t ov = level ? 0u : cnt(a, GC_OVERFLOW);
            const uint32_t un = level ? 0u : cnt(a, GC_UNANCHORED);
            const bool early = earlyOf[a] != 0;
            // (round 0 left slots in play: their remainders are screened at the end of this chain, see remainderSteps)
            const bool inPlay0 = remainderInChain && !level && cnt(a, GC_ROUND0) != 0;
            if (!level && !ov && !un && !early && !inPlay0) return;
            if (!level && !ov && !un && !early) {  // nothing but remainders to screen: with the others of its kind, in ONE launch at the fork
                groupMask |= 1ull << a;
                coveredMask |= 1ull << a;
                return;
            }
            if (level && boundKnown && cnt(a, GC_BOUND) == 0) {  // nothing left for it (values are only ever won, never lost)
                busy2c[a] = 0;
                if (trace) fprintf(stderr, "grok plan 2c: entry %u level %u cand %u: every value already won\n", e0.p, e0.level, e0.cand);
                return;
            }
            size_t owner = a;  // whose list the steps go to
            if (level >= 2) {  // behind its main shadower
                uint32_t best = 0, bestF = 0;
                for (uint32_t f = 0; f < e0.p; ++f)
                    if (T.hostWords[HW_SHADOW + e0.p * 64 + f] > best && map.activeOfBit[f] >= 0) {
                        best = T.hostWords[HW_SHADOW + e0.p * 64 + f];
                        bestF = f;
                    }
                if (best) {
                    const size_t sa = size_t(map.activeOfBit[bestF]);
                    e0.stream = act[sa].stream;
                    owner = chainOwner[sa];
                    chainOwner[a] = owner;
                    chains[owner].push_back([&, a, sa](int& rc) {
                        if (rc != LC_OK) return;
                        hipLaunchKernelGGL(grok_entry_finish_kernel, dim3((act[sa].cand + kGrokPlanBlock - 1) / kGrokPlanBlock, 1), dim3(kGrokPlanBlock),
                                           0, W(act[a].stream), T.dEntries, winner, undecided, static_cast<const uint32_t*>(nullptr), uint32_t(sa));
                    });
                }
            }
            std::vector<Step>& out = chains[owner];
            busy2c[a] = (level || ov || un) ? 1 : 0;
            if (trace) {
                if (level) fprintf(stderr, "grok plan 2c: entry %u level %u cand %u stream %d%s\n", e0.p, e0.level, e0.cand, e0.stream, early ? " + rounds" : "");
                else fprintf(stderr, "grok plan 2c: entry %u overflowed %u unanchored %u%s\n", e0.p, ov, un, early ? " + rounds" : "");
            }
            auto post = [&, a](const uint32_t* in, const uint32_t* inCount, uint32_t flags) {
                PlanEntry& e = act[a];
                hipLaunchKernelGGL(grok_post_kernel, dim3((e.cand + kGrokPlanBlock - 1) / kGrokPlanBlock, 1), dim3(kGrokPlanBlock), 0, W(e.stream),
                                   T.dEntries, uint32_t(a), in, inCount, flags, xtmp, xcap, xstride, xcount, 0ull);
            };
            auto tick = [&, a](int which, int& rc) {
                if (!calibrate || !busy2c[a]) return;
                if (hipEventRecord(T.tick[2 * nAct + 2 * a + size_t(which)], W(act[a].stream)) != hipSuccess && rc == LC_OK && which == 0)
                    rc = lcHipFail(hipGetLastError(), "hipEventRecord(calibration)");
            };
            // the search proper over the slots in `list`, minus the values an earlier entry has won: two steps
            auto searchProperSteps = [&, a, post](const uint32_t* list, const uint32_t* count) {
                out.push_back([&, a, list, count](int& rc) {
                    if (rc != LC_OK) return;
                    PlanEntry& e = act[a];
                    const GrokDevicePattern& gp = patterns[e.p];
                    hipStream_t ws = W(e.stream);
                    lcSetDecideSlot(1 + e.stream);
                    hipLaunchKernelGGL(grok_filter_won_kernel, dim3((e.cand + kGrokPlanBlock - 1) / kGrokPlanBlock), dim3(kGrokPlanBlock), 0, ws, e.dev,
                                       static_cast<const uint32_t*>(winner), list, count, e.listB, e.dev.cnt + GC_FILTERED);
                    rc = runFirst(e, gp.re, false, e.dev.cnt + GC_FILTERED, e.listB, true, &e.seqS, ws);
                });
                out.push_back([&, a, post](int& rc) {
                    if (rc != LC_OK) return;
                    PlanEntry& e = act[a];
                    const GrokDevicePattern& gp = patterns[e.p];
                    lcSetDecideSlot(1 + e.stream);
                    rc = runSecond(e, gp.re, false, e.dev.cnt + GC_FILTERED, e.listB, true, e.seqS, nullptr, W(e.stream));
                    if (rc == LC_OK) post(e.listB, e.dev.cnt + GC_FILTERED, uint32_t(GP_OVERFLOW_FINAL));
                });
            };
            if (level == 0) {
                if (ov) {  // the second chance of round 0's engine over the overflow list (the longest kernel of the phase: first), then the post step
                    out.push_back([&, a, tick](int& rc) {
                        tick(0, rc);
                        if (rc != LC_OK) return;
                        PlanEntry& e = act[a];
                        const GrokDevicePattern& gp = patterns[e.p];
                        lc_regex* first = gp.anchored ? gp.anchored : gp.re;
                        lcSetDecideSlot(1 + e.stream);
                        rc = runSecond(e, first, e.wideFirst, e.dev.cnt + GC_OVERFLOW, e.dev.ovList, !gp.anchored, e.seq0, e.dev.cnt + GC_WIDE,
                                       W(e.stream));
                    });
                    out.push_back([&, a, post](int& rc) {
                        if (rc != LC_OK) return;
                        PlanEntry& e = act[a];
                        post(e.dev.ovList, e.dev.cnt + GC_OVERFLOW, uint32_t(GP_ANCHORED_PASS | GP_OVERFLOW_FINAL));
                    });
                } else if (un) {
                    out.push_back([tick](int& rc) { tick(0, rc); });
                }
                // (only when the anchored search left any value unmatched -- the host knows the count)
                if (patterns[e0.p].anchored && un) searchProperSteps(e0.dev.unanchored, e0.dev.cnt + GC_UNANCHORED);
            } else {
                // every slot whose value nobody before has won: the anchored search (or the search) in full, then the rest
                out.push_back([&, a, tick](int& rc) {
                    tick(0, rc);
                    if (rc != LC_OK) return;
                    PlanEntry& e = act[a];
                    const GrokDevicePattern& gp = patterns[e.p];
                    hipStream_t ws = W(e.stream);
                    lcSetDecideSlot(1 + e.stream);
                    uint32_t* mine = e.dev.ovList;  // (a level > 0 entry has no first-chance pass: its overflow list is free)
                    hipLaunchKernelGGL(grok_filter_won_kernel, dim3((e.cand + kGrokPlanBlock - 1) / kGrokPlanBlock), dim3(kGrokPlanBlock), 0, ws, e.dev,
                                       static_cast<const uint32_t*>(winner), static_cast<const uint32_t*>(nullptr), static_cast<const uint32_t*>(nullptr),
                                       mine, e.dev.cnt + GC_OVERFLOW);
                    lc_regex* first = gp.anchored ? gp.anchored : gp.re;
                    e.wideFirst = wantsWideFirst(first);
                    rc = runFirst(e, first, e.wideFirst, e.dev.cnt + GC_OVERFLOW, mine, !gp.anchored, &e.seq0, ws);
                });
                out.push_back([&, a, post](int& rc) {
                    if (rc != LC_OK) return;
                    PlanEntry& e = act[a];
                    const GrokDevicePattern& gp = patterns[e.p];
                    lc_regex* first = gp.anchored ? gp.anchored : gp.re;
                    lcSetDecideSlot(1 + e.stream);
                    rc = runSecond(e, first, e.wideFirst, e.dev.cnt + GC_OVERFLOW, e.dev.ovList, !gp.anchored, e.seq0, e.dev.cnt + GC_WIDE,
                                   W(e.stream));
                    if (rc == LC_OK) post(e.dev.ovList, e.dev.cnt + GC_OVERFLOW, uint32_t(GP_ANCHORED_PASS | GP_OVERFLOW_FINAL));
                });
                if (patterns[e0.p].anchored) searchProperSteps(e0.dev.unanchored, e0.dev.cnt + GC_UNANCHORED);
            }
            if (busy2c[a]) out.push_back([tick](int& rc) { int ignored = LC_OK; tick(1, ignored); (void)rc; });  // (whichever way the chain ended)
            if (early) {
                earlyMask |= 1ull << a;
                roundSteps(a, false, out);
            } else if (remainderInChain) {
                // Round 5: the remainder screens of THIS entry, here -- on its stream, as soon as its own chain is through -- instead of
                // one launch for all entries behind the join of the phase (that launch, 0.4-0.5 ms for a few hundred 4 KiB remainders of
                // one shadowed entry, sat between two host round trips on every batch's critical path).  Same kernels, the other
                // entries' workgroups leave at once (skip mask).
                coveredMask |= 1ull << a;
                out.push_back([&, a](int& rc) {
                    if (rc != LC_OK) return;
                    PlanEntry& e = act[a];
                    hipStream_t ws = W(e.stream);
                    // (grid.y = 1 over a table that begins at this entry)
                    const GrokEntryDev* mine = static_cast<const GrokEntryDev*>(T.dEntries) + a;
                    if (remainderLiteral || remainderWon)
                        hipLaunchKernelGGL(grok_remainder_literal_kernel, dim3((e.cand + kGrokPlanBlock / 64 - 1) / (kGrokPlanBlock / 64), 1),
                                           dim3(kGrokPlanBlock), 0, ws, d_data, mine,
                                           remainderLiteral ? literalIndex : static_cast<const uint32_t*>(nullptr), 0ull,
                                           remainderWon ? static_cast<const uint32_t*>(winner) : static_cast<const uint32_t*>(nullptr), 0u);
                    const GrokScreenDev* sc = e.remainderScreen;
                    const bool big = small && bigRemainder && sc && sc->bigBytes;
                    const uint32_t lds = big ? sc->bigBytes : (small && sc) ? sc->ldsBytes : 0u;
                    if (big) {
                        static thread_local size_t attrSet[kLcMaxDevices] = {};
                        if (lds > 48 * 1024 && dev < kLcMaxDevices && lds > attrSet[dev]) {
                            if (hipFuncSetAttribute(reinterpret_cast<const void*>(grok_remainder_all_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    int(lds)) != hipSuccess) {
                                rc = lcHipFail(hipGetLastError(), "hipFuncSetAttribute(grok_remainder_all_kernel)");
                                return;
                            }
                            attrSet[dev] = lds;
                        }
                    }
                    hipLaunchKernelGGL(grok_remainder_all_kernel, dim3(remGrid(e.cand), 1), dim3(kGrokPlanBlock), lds, ws,
                                       d_data, mine, static_cast<const GrokScreenDev*>(T.dRemScreens) + a, (big ? 2u : small ? 1u : 0u) | remWalk, 0ull,
                                       remGate);
                });
            }
        };
        bool forked = false;
        int rc2c = LC_OK;
        {
            for (size_t a = 0; a < nAct; ++a)
                busy2c[a] = act[a].level ? ((boundKnown && cnt(a, GC_BOUND) == 0) ? 0 : 1) : (cnt(a, GC_OVERFLOW) || cnt(a, GC_UNANCHORED)) ? 1 : 0;
            const std::vector<size_t> order1 = deal(true);
            for (uint32_t level = 0; level <= maxLevel; ++level)
                for (size_t i = 0; i < nAct; ++i)
                    if (act[order1[i]].level == level) buildChain(order1[i]);
            size_t longest = 0;
            for (size_t a = 0; a < nAct; ++a) longest = std::max(longest, chains[a].size());
            if (longest || groupMask || aheadLaunched) {
                if (!boundKnown && !aheadLaunched)
                    hipLaunchKernelGGL(grok_entry_finish_kernel, dim3(gridCand0, nAct), dim3(kGrokPlanBlock), 0, st, T.dEntries, winner, undecided,
                                       static_cast<const uint32_t*>(nullptr), 0u);  // the values won so far (atomicMin per value: idempotent, finishAll runs it again)
                rc2c = fork();
                if (rc2c != LC_OK) return rc2c;
                forked = true;
                if (groupMask && !aheadLaunched) {  // the entries that only have remainders to screen: one launch pair, beside the chains
                    hipStream_t gs = W(int(used) - 1);
                    if (remainderLiteral || remainderWon)
                        hipLaunchKernelGGL(grok_remainder_literal_kernel, dim3((maxCand + kGrokPlanBlock / 64 - 1) / (kGrokPlanBlock / 64), nAct),
                                           dim3(kGrokPlanBlock), 0, gs, d_data, T.dEntries,
                                           remainderLiteral ? literalIndex : static_cast<const uint32_t*>(nullptr), ~groupMask,
                                           remainderWon ? static_cast<const uint32_t*>(winner) : static_cast<const uint32_t*>(nullptr), 0u);
                    hipLaunchKernelGGL(grok_remainder_all_kernel, dim3(remGrid(maxCand), nAct), dim3(kGrokPlanBlock), small ? remScreenLds : 0, gs, d_data,
                                       T.dEntries, static_cast<const GrokScreenDev*>(T.dRemScreens), (small ? 1u : 0u) | remWalk, ~groupMask, remGate);
                }
                if (breadthFirst) {
                    for (size_t k = 0; k < longest; ++k)
                        for (size_t i = 0; i < nAct; ++i) {
                            const size_t a = order1[i];
                            if (k < chains[a].size()) chains[a][k](rc2c);
                        }
                } else {
                    for (size_t i = 0; i < nAct; ++i)
                        for (Step& s : chains[order1[i]]) s(rc2c);
                }
            }
            if (forked) {
                rc2c = join(rc2c);  // (always: a failure inside the forked region must not leave the worker streams unjoined)
                if (rc2c != LC_OK) return rc2c;
            }
        }
        // 2d (LC_GROK_REMAINDER_INCHAIN=0 only: by default every entry that can have a slot in play screened its remainders in its chain)
        if (!remainderInChain) {
        if (remainderWon)  // (the values won by now, phase 2c included: a slot in play whose value an earlier entry has won is finished)
            hipLaunchKernelGGL(grok_entry_finish_kernel, dim3(gridCand0, nAct), dim3(kGrokPlanBlock), 0, st, T.dEntries, winner, undecided,
                               static_cast<const uint32_t*>(nullptr), 0u);
        if (remainderLiteral || remainderWon) {
            lcNoteKernel("grok_remainder_literal_kernel");
            hipLaunchKernelGGL(grok_remainder_literal_kernel, dim3((maxCand + kGrokPlanBlock / 64 - 1) / (kGrokPlanBlock / 64), nAct), dim3(kGrokPlanBlock),
                               0, st, d_data, T.dEntries, remainderLiteral ? literalIndex : static_cast<const uint32_t*>(nullptr), earlyMask,
                               remainderWon ? static_cast<const uint32_t*>(winner) : static_cast<const uint32_t*>(nullptr), 0u);
        }
        lcNoteKernel("grok_remainder_all_kernel");
        {
            // (the entries whose screens are BIG -- phase 1 -- in a launch of their own beside the others', table in a CU's whole LDS)
            unsigned long long bigMask = 0;
            uint32_t bigLds = 0;
            const bool bigOff = [] {
                const char* v = getenv("LC_GROK_BIG_SCREENS");
                return !(v && v[0] == '1');
            }();
            if (small && !bigOff)
                for (size_t a = 0; a < nAct; ++a)
                    if (act[a].remainderScreen && act[a].remainderScreen->bigBytes && !((earlyMask >> a) & 1ull)) {
                        bigMask |= 1ull << a;
                        bigLds = std::max(bigLds, act[a].remainderScreen->bigBytes);
                    }
            if (bigMask) {
                static thread_local size_t attrSet[kLcMaxDevices] = {};
                if (bigLds > 48 * 1024 && dev < kLcMaxDevices && bigLds > attrSet[dev]) {
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(grok_remainder_all_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                int(bigLds)));
                    attrSet[dev] = bigLds;
                }
                HIP_TRY(hipEventRecord(T.fork, st));
                HIP_TRY(hipStreamWaitEvent(T.workers[0], T.fork, 0));
                hipLaunchKernelGGL(grok_remainder_all_kernel, dim3(remGrid(maxCand), nAct), dim3(kGrokPlanBlock), bigLds, T.workers[0], d_data, T.dEntries,
                                   static_cast<const GrokScreenDev*>(T.dRemScreens), 2u | remWalk, ~bigMask, remGate);
                HIP_TRY(hipEventRecord(T.join[0], T.workers[0]));
            }
            hipLaunchKernelGGL(grok_remainder_all_kernel, dim3(remGrid(maxCand), nAct), dim3(kGrokPlanBlock), small ? remScreenLds : 0, st, d_data, T.dEntries,
                               static_cast<const GrokScreenDev*>(T.dRemScreens), (small ? 1u : 0u) | remWalk, earlyMask | bigMask, remGate);
            if (bigMask) HIP_TRY(hipStreamWaitEvent(st, T.join[0], 0));
        }
        }
        HIP_TRY(hipGetLastError());
        // What is left of phase 2 needs the survivors of the remainder screens -- almost always none.  Round 5: the host no longer waits
        // to learn that.  A one-wave kernel adds the survivors to the gate word, the finish of the batch is queued behind it (it returns at
        // once if the gate is up), and the host reads everything in ONE round trip: gate down = done (three host synchronisations per batch
        // instead of four); gate up = the search rounds of the entries with survivors, then the finish again.  LC_GROK_LAZY_SYNC3=0: as before.
        auto afterCounts = [&] {
            if (calibrate)
                for (size_t a = 0; a < nAct; ++a)
                    if (busy2c[a]) learn(patterns[act[a].p].re->grokCost1Ns, T.tick[2 * nAct + 2 * a], T.tick[2 * nAct + 2 * a + 1], act[a].cand);
            if (trace)
                for (size_t a = 0; a < nAct; ++a)
                    if (cnt(a, GC_ROUND0))
                        fprintf(stderr, "grok plan 2d: entry %u in play %u, remainder passes the screen %u%s\n", act[a].p, cnt(a, GC_ROUND0),
                                cnt(a, GC_REMAINDER), ((earlyMask >> a) & 1ull) ? " (rounds queued ahead)" : "");
            // (the entries' histories: do most of the slots in play pass the remainder screen?  Judged on the batches that ran the screen
            // for the entry -- every calibration batch does)
            for (size_t a = 0; a < nAct; ++a) {
                if ((earlyMask >> a) & 1ull) continue;
                const uint32_t inPlay = cnt(a, GC_ROUND0), survivors = cnt(a, GC_REMAINDER);
                if (inPlay) patterns[act[a].p].re->grokRemainderSeen.store(survivors * 4 >= inPlay * 3 ? 1u : 0u, std::memory_order_relaxed);
            }
        };
        // 2e: the search rounds of the entries with survivors (-> true if there were any)
        auto rounds2e = [&](bool& ran) -> int {
            ran = false;
            bool any = false;
            for (size_t a = 0; a < nAct; ++a) any = any || (cnt(a, GC_REMAINDER) && !act[a].queued);
            if (!any) return LC_OK;
            ran = true;
            int rc = fork();
            if (rc != LC_OK) return rc;
            for (size_t i = 0; i < nAct && rc == LC_OK; ++i) {
                const size_t a = byCost[i];
                PlanEntry& e = act[a];
                if (!cnt(a, GC_REMAINDER) || e.queued) continue;
                if (trace) fprintf(stderr, "grok plan 2e: entry %u in play %u survivors %u\n", e.p, cnt(a, GC_ROUND0), cnt(a, GC_REMAINDER));
                std::vector<Step> steps;
                roundSteps(a, true, steps);
                for (Step& s2 : steps) s2(rc);
            }
            return join(rc);
        };
        if (!lazy) {
            int rc = readCounts();  // sync 3
            if (rc != LC_OK) return rc;
            afterCounts();
            bool ran = false;
            rc = rounds2e(ran);
            if (rc != LC_OK) return rc;
        } else {
            // (the survivors are on the gate word already: grok_remainder_all_kernel counts them there)
            int rc = finishAll(gate);  // sync 3 = the last one, unless the gate is up
            if (rc != LC_OK) return rc;
            finished = true;
            afterCounts();
            if (T.hostWords[HW_TAIL + TW_GATE] != 0) {
                // (the survivors' share leaves the gate again: what stays up is what the queued rounds left in play)
                hipLaunchKernelGGL(grok_survivor_gate_kernel, dim3(1), dim3(64), 0, st, static_cast<const GrokEntryDev*>(T.dEntries), nAct, gate, 0u);
                bool ran = false;
                rc = rounds2e(ran);
                if (rc != LC_OK) return rc;
                if (ran) {
                    rc = finishAll(gate);
                    if (rc != LC_OK) return rc;
                }
            }
        }
    }
    if (!finished) {
        int rc = finishAll(gate);  // the last sync
        if (rc != LC_OK) return rc;
    }
    const double tPhase3 = msNow();
    if (T.hostWords[HW_TAIL + TW_GATE] != 0) {
        // values still in play after the last queued round of some entries: those entries go on, a stretch of rounds at a time (as
        // many again as they had queued, the counts on the device as before), until nothing is left -- an entry pays this once, then
        // asks for that many rounds ahead
        for (size_t a = 0; a < nAct; ++a) {
            PlanEntry& e = act[a];
            uint32_t inPlay = T.hostWords[HW_CNT + a * GC_WORDS + GC_ROUND0 + e.rounds - 1];
            if (!inPlay) continue;
            ++stats.deferredEntries;
            const GrokDevicePattern& gp = patterns[e.p];
            uint32_t r = e.rounds;  // rounds done so far
            while (inPlay) {
                const uint32_t stretch = std::min(kGrokMaxRounds, std::max(4u, r));
                // the list of round r was written by round r - 1: (r & 1) ? listA : listB; counters are reused from slot 0
                HIP_TRY(hipMemsetAsync(e.dev.cnt + GC_ROUND0, 0, size_t(stretch) * 4, st));
                for (uint32_t k = 0; k < stretch; ++k, ++r) {
                    const uint32_t* list = (r & 1) ? e.listA : e.listB;
                    uint32_t* out = (r & 1) ? e.listB : e.listA;
                    const uint32_t bound = k == 0 ? inPlay : e.cand;
                    const uint32_t* countPtr = k == 0 ? nullptr : e.dev.cnt + GC_ROUND0 + k - 1;
                    int rc = lcMatchOnStream(gp.re, gp.re->engine, dev, d_data, e.dev.off, e.dev.len, 0, bound, countPtr, list, e.dev.from,
                                             e.capsRow / 2, e.caps, e.status, st);
                    if (rc != LC_OK) return rc;
                    hipLaunchKernelGGL(grok_advance2_kernel, dim3((bound + kGrokPlanBlock - 1) / kGrokPlanBlock), dim3(kGrokPlanBlock), 0, st,
                                       list, bound, countPtr, e.status, e.caps, e.capsRow, e.columns, e.dev, xtmp, xcap, xstride, xcount, out,
                                       e.dev.cnt + GC_ROUND0 + k, static_cast<uint32_t*>(nullptr));
                }
                HIP_TRY(hipMemcpyAsync(T.hostWords + HW_CNT + a * GC_WORDS + GC_ROUND0, e.dev.cnt + GC_ROUND0, size_t(stretch) * 4,
                                       hipMemcpyDeviceToHost, st));
                HIP_TRY(syncCounted(st));
                inPlay = T.hostWords[HW_CNT + a * GC_WORDS + GC_ROUND0 + stretch - 1];
                if (!inPlay) {  // rounds that had anything to search: up to the first empty list of this stretch
                    uint32_t usedRounds = stretch;
                    while (usedRounds > 1 && T.hostWords[HW_CNT + a * GC_WORDS + GC_ROUND0 + usedRounds - 2] == 0) --usedRounds;
                    r = r - stretch + usedRounds;
                }
            }
            e.ran = r;  // (what this batch needed: the hint below)
        }
        int rc = finishAll(nullptr);
        if (rc != LC_OK) return rc;
    }
    // ---- how many rounds each entry should queue ahead next time: what this batch needed, forgotten slowly
    for (size_t a = 0; a < nAct; ++a) {
        PlanEntry& e = act[a];
        {  // did any of its values need more than 64 threads (GC_WIDE: set by the wide kernel)?  seen now = 8, else forgotten a batch at a time
            lc_regex* first = patterns[e.p].anchored ? patterns[e.p].anchored : patterns[e.p].re;
            const uint32_t v = first->grokOverflowSeen.load(std::memory_order_relaxed);
            if (T.hostWords[HW_CNT + a * GC_WORDS + GC_WIDE]) first->grokOverflowSeen.store(8, std::memory_order_relaxed);
            else if (v) first->grokOverflowSeen.store(v - 1, std::memory_order_relaxed);
        }
        uint32_t needed = 1;  // rounds that had any value to search
        for (uint32_t r = 0; r + 1 < e.rounds && e.ran == 0; ++r)
            if (T.hostWords[HW_CNT + a * GC_WORDS + GC_ROUND0 + r]) needed = r + 2;
        lc_regex* re = patterns[e.p].re;
        const uint32_t hint = re->grokRounds.load(std::memory_order_relaxed);
        if (e.ran > hint) {
            re->grokRounds.store(std::min(e.ran, kGrokMaxRounds), std::memory_order_relaxed);
            re->grokRoundsSlack.store(0, std::memory_order_relaxed);
        } else if (needed < hint && e.ran == 0) {
            if (re->grokRoundsSlack.fetch_add(1, std::memory_order_relaxed) + 1 >= 4) {
                re->grokRounds.store(needed, std::memory_order_relaxed);
                re->grokRoundsSlack.store(0, std::memory_order_relaxed);
            }
        } else {
            re->grokRoundsSlack.store(0, std::memory_order_relaxed);
        }
    }
    if (trace)
        fprintf(stderr, "grok plan: n %u entries %u (second pass %u) pairs %u screens %u | phase1 %.3f ms, entries+finish %.3f ms, total %.3f ms, syncs %u, deferred %u\n",
                n, nAct, nSecond, stats.pairs, nScreens, tPhase1, tPhase3 - tPhase1, msNow(), stats.hostSyncs, stats.deferredEntries);
    const uint32_t xWanted = T.hostWords[HW_TAIL + TW_XCOUNT];
    const uint32_t nextra = T.hostWords[HW_TAIL + TW_NEXTRA];
    if (xWanted > xcap) {
        // the temporary rows themselves did not fit: report an upper bound of what is needed (winners' rows <= all rows)
        HIP_TRY(hipMemcpy(d_nextra, &xWanted, 4, hipMemcpyHostToDevice));
        lcSetLastError("grok: " + std::to_string(xWanted) + " extra match rows needed, buffer holds " + std::to_string(extraCap));
        return LC_ERR_OVERFLOW;
    }
    if (nextra > extraCap) {
        lcSetLastError("grok: " + std::to_string(nextra) + " extra match rows needed, buffer holds " + std::to_string(extraCap));
        return LC_ERR_OVERFLOW;
    }
    return LC_OK;
}
}  // namespace

int lcGrokMatchDevice(const std::vector<GrokDevicePattern>& patterns, GrokDeviceState* state, const GrokOptions& opts, uint32_t row,
                      const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len, uint32_t n, int32_t* d_pattern,
                      int32_t* d_first, int32_t* d_extra, uint32_t extraCap, uint32_t* d_nextra, void* d_scratch, size_t scratchBytes,
                      void* streamPtr) {
    tlsStats = GrokBatchStats{};
    if (n == 0) return LC_OK;
    if (!state || !d_data || !d_off || !d_len || !d_pattern || !d_first || !d_nextra || !d_scratch || (extraCap && !d_extra))
        return LC_ERR_ARG;
    if (scratchBytes < lcGrokScratchBytes(n, row)) {
        lcSetLastError("grok: scratch buffer too small");
        return LC_ERR_ARG;
    }
    if (lc_device_count() <= 0) {
        lcSetLastError("no HIP device");
        return LC_ERR_NO_DEVICE;
    }
    int dev = 0;
    {
        const int rcDev = lcDeviceEntryDevice(d_data, &dev);
        if (rcDev != LC_OK) return rcDev;
    }
    hipStream_t st = static_cast<hipStream_t>(streamPtr);
    static const int forced = [] {  // LC_GROK_SPECULATIVE=0/1 overrides the handle's option (A/B measurements)
        const char* e = getenv("LC_GROK_SPECULATIVE");
        return e ? (e[0] == '0' ? 0 : 1) : -1;
    }();
    const bool speculative = (forced < 0 ? opts.speculative : forced != 0) && patterns.size() <= 64;
    tlsStats.speculative = speculative;
    if (speculative)
        return grokMatchSpeculative(patterns, state, opts, row, d_data, d_off, d_len, n, d_pattern, d_first, d_extra, extraCap, d_nextra,
                                    d_scratch, st, dev);
    return grokMatchSequential(patterns, state, opts, row, d_data, d_off, d_len, n, d_pattern, d_first, d_extra, extraCap, d_nextra,
                               d_scratch, st, dev);
}

int lcGrokSampleDevice(const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len, uint32_t n, uint32_t maxValues, void* streamPtr,
                       std::vector<uint8_t>& data, std::vector<uint32_t>& off, std::vector<uint32_t>& len) {
    data.clear();
    off.clear();
    len.clear();
    const uint32_t take = std::min(n, maxValues);
    if (!take) return LC_OK;
    int dev = 0;
    {
        const int rcDev = lcDeviceEntryDevice(d_data, &dev);
        if (rcDev != LC_OK) return rcDev;
    }
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(streamPtr)));  // (what the caller queued before the call has produced the batch)
    std::vector<uint32_t> o(take), l(take);
    HIP_TRY(hipMemcpy(o.data(), d_off, size_t(take) * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(l.data(), d_len, size_t(take) * 4, hipMemcpyDeviceToHost));
    // one copy of the byte range the values span (batches are packed: the range is the values themselves)
    uint64_t lo = ~uint64_t(0), hi = 0;
    for (uint32_t i = 0; i < take; ++i) {
        lo = std::min<uint64_t>(lo, o[i]);
        hi = std::max<uint64_t>(hi, uint64_t(o[i]) + l[i]);
    }
    if (hi <= lo || hi - lo > (uint64_t(64) << 20)) return LC_OK;  // (nothing, or values strewn over more than is worth a sample)
    std::vector<uint8_t> range(size_t(hi - lo));
    HIP_TRY(hipMemcpy(range.data(), d_data + lo, range.size(), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < take; ++i) {
        off.push_back(uint32_t(data.size()));
        len.push_back(l[i]);
        data.insert(data.end(), range.begin() + long(o[i] - lo), range.begin() + long(o[i] - lo + l[i]));
    }
    return LC_OK;
}

// What a thread that calls lcGrokMatchHost keeps between calls (ProcessLogs hands over group after group): a stream of its own
// (runner threads must not meet on the null stream), pinned staging for the way in and the way out, device buffers; grow-only.
namespace {
struct GrokDev {
    void* p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    void release() {
        if (p && lcRuntimeUsable()) (void)(pinned ? hipHostFree(p) : hipFree(p));
        p = nullptr;
        cap = 0;
    }
    hipError_t ensure(size_t bytes) {
        if (p && cap >= bytes) return hipSuccess;
        release();
        const size_t want = bytes + (bytes >> 2) + 256;
        hipError_t e = pinned ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
};
struct GrokHostThread {
    int device = -1;
    hipStream_t stream = nullptr;
    GrokDev dIn, dPattern, dFirst, dExtra, dNextra, dScratch;  // device
    GrokDev hIn, hOut, hExtra;                                 // pinned host
    GrokHostThread() { hIn.pinned = hOut.pinned = hExtra.pinned = true; }
    ~GrokHostThread() { release(); }
    void release() {
        if (device >= 0 && lcRuntimeUsable()) {
            int cur = 0;
            const bool haveCur = hipGetDevice(&cur) == hipSuccess;
            if (hipSetDevice(device) == hipSuccess) {
                if (stream) {
                    (void)hipStreamSynchronize(stream);
                    (void)hipStreamDestroy(stream);
                }
                for (GrokDev* d : {&dIn, &dPattern, &dFirst, &dExtra, &dNextra, &dScratch, &hIn, &hOut, &hExtra}) d->release();
            }
            if (haveCur) (void)hipSetDevice(cur);
        }
        stream = nullptr;
        device = -1;
    }
    int ensure(int dev) {
        if (device == dev) return LC_OK;
        release();
        lcRegisterExitHook();
        device = dev;
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        return LC_OK;
    }
};
thread_local GrokHostThread tlsGrokHost;
}  // namespace
void lcGrokThreadRelease() {
    tlsGrokHost.release();
    tlsPlan.release();
}

// One group of one runner thread, as handed to lcGrokMatchHost.
struct GrokDeviceState::HostJob {
    const uint8_t* data;
    const uint32_t* off;
    const uint32_t* len;
    uint32_t n;
    int32_t* pattern;
    std::vector<int32_t>* first;      // the group's own rows (filled by its thread from firstSrc)
    std::vector<int32_t>* extraRows;  // [line, seq, row...] sorted, lines relative to the group
    const std::vector<GrokDevicePattern>* patterns;  // the call's view of the processor (anchored forms appear as the warm-up goes on)
    const GrokOptions* opts;
    uint32_t row;
    size_t bytes = 0;                 // sum of the group's value lengths (taken by its own thread before it queues)
    // where the group's values go in the batch's pinned staging (the worker's `place`), filled by the group's own thread (`gather`)
    uint8_t* hData = nullptr;
    uint32_t* hOff = nullptr;
    uint32_t* hLen = nullptr;
    uint32_t byteBase = 0;
    // results in the batch's pinned staging (the worker's `run`), taken out by the group's own thread
    const int32_t* patternSrc = nullptr;
    const int32_t* firstSrc = nullptr;
    const std::string* error = nullptr;  // the batch's message in the worker's batch state (the worker's last error is thread-local to the worker) ...
    std::string errorText;            // ... copied here by the job's own thread while the batch's state is still the batch's
    const GrokBatchStats* stats = nullptr;  // what the batch did (lc_grok_last_batch_stats is per calling thread)
    const std::string* kernels = nullptr;   // ... and which kernels it launched (lc_launched_kernels is per calling thread)
    uint32_t lines() const { return n; }
};

namespace {
// what the worker thread of one (processor, device) keeps between `place` and `run` of a batch
struct HostBatch {
    uint32_t n = 0;
    size_t bytes = 0, offAt = 0, lenAt = 0, inBytes = 0;
    std::string error, kernels;
    GrokBatchStats stats;
};
}  // namespace

struct GrokDeviceState::HostCombiner {
    GrokDeviceState* state = nullptr;
    int device = 0;
    HostBatch batch;
    std::unique_ptr<lccombine::GroupCombiner<HostJob>> combiner;
};

namespace {
using HostJob = GrokDeviceState::HostJob;

// `place`: the batch is fixed -- the values of all groups packed back to back (they may come from anywhere), then offsets and lengths:
// one pinned block, one copy.  Runs on the worker thread; every group's own thread then copies its values in (grokGatherJob).
int grokPlaceHostBatch(GrokDeviceState::HostCombiner& C, std::vector<HostJob*>& jobs) {
    HostBatch& B = C.batch;
    B = HostBatch{};
    GrokHostThread& H = tlsGrokHost;
    auto fail = [&](int rc) {
        B.error = lc_last_error();
        for (HostJob* j : jobs) j->error = &B.error;
        return rc;
    };
    {
        const int rc = H.ensure(C.device);
        if (rc != LC_OK) return fail(rc);
    }
    size_t n64 = 0;
    for (const HostJob* j : jobs) {
        B.bytes += j->bytes;
        n64 += j->n;
    }
    if (B.bytes > 0xFFFFFFF0ull || n64 > 0x7FFFFFFFull) {
        lcSetLastError("grok: more than 4 GiB of values in one batch");
        return fail(LC_ERR_ARG);
    }
    B.n = uint32_t(n64);
    B.offAt = alignUp(B.bytes + 16, 256);
    B.lenAt = B.offAt + alignUp(size_t(B.n) * 4, 256);
    B.inBytes = B.lenAt + size_t(B.n) * 4;
    if (H.hIn.ensure(B.inBytes) != hipSuccess || H.dIn.ensure(B.inBytes) != hipSuccess) {
        lcSetLastError("grok: staging allocation failed");
        return fail(LC_ERR_HIP);
    }
    uint8_t* hIn = static_cast<uint8_t*>(H.hIn.p);
    size_t at = 0;
    uint32_t k = 0;
    for (HostJob* j : jobs) {
        j->hData = hIn + at;
        j->byteBase = uint32_t(at);
        j->hOff = reinterpret_cast<uint32_t*>(hIn + B.offAt) + k;
        j->hLen = reinterpret_cast<uint32_t*>(hIn + B.lenAt) + k;
        at += j->bytes;
        k += j->n;
    }
    std::memset(hIn + at, 0, 16);
    return LC_OK;
}

// `gather`: on the group's OWN thread, all groups of a batch side by side
void grokGatherJob(HostJob& j) {
    uint32_t at = 0;
    for (uint32_t i = 0; i < j.n; ++i) {
        j.hOff[i] = j.byteBase + at;
        j.hLen[i] = j.len[i];
        std::memcpy(j.hData + at, j.data + j.off[i], j.len[i]);
        at += j.len[i];
    }
}

// `run`: the merged batch on the worker thread's stream; sets patternSrc / firstSrc / extraRows of every job
int grokRunHostBatch(GrokDeviceState::HostCombiner& C, std::vector<HostJob*>& jobs) {
    static const bool traceHost = getenv("LC_GROK_TRACE") != nullptr;
    const auto th0 = std::chrono::steady_clock::now();
    auto hostMs = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th0).count(); };
    HostBatch& B = C.batch;
    GrokHostThread& H = tlsGrokHost;
    const std::vector<GrokDevicePattern>& patterns = *jobs.front()->patterns;
    const GrokOptions& opts = *jobs.front()->opts;
    const uint32_t row = jobs.front()->row, n = B.n;
    const size_t offAt = B.offAt, lenAt = B.lenAt, inBytes = B.inBytes;
    int rcAll = LC_OK;
    (void)lc_launched_kernels(nullptr, 0);  // (the worker's log: what THIS batch launches goes to the callers' logs)
    auto body = [&]() -> int {
    const size_t scratch = lcGrokScratchBytes(n, row);
    uint32_t extraCap = n / 4 + 1024;
    const size_t firstBytes = size_t(n) * row * 4;
    HIP_TRY(H.dPattern.ensure(size_t(n) * 4));
    HIP_TRY(H.dFirst.ensure(firstBytes));
    HIP_TRY(H.dNextra.ensure(4));
    HIP_TRY(H.dScratch.ensure(scratch));
    HIP_TRY(H.hOut.ensure(alignUp(size_t(n) * 4, 256) + firstBytes));
    int32_t* hPattern = static_cast<int32_t*>(H.hOut.p);
    int32_t* hFirst = reinterpret_cast<int32_t*>(static_cast<uint8_t*>(H.hOut.p) + alignUp(size_t(n) * 4, 256));
    const uint8_t* dIn = static_cast<const uint8_t*>(H.dIn.p);
    HIP_TRY(hipMemcpyAsync(H.dIn.p, H.hIn.p, inBytes, hipMemcpyHostToDevice, H.stream));
    const double tGather = hostMs();
    uint32_t nExtra = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIP_TRY(H.dExtra.ensure(size_t(extraCap) * (row + 2) * 4));
        // (the speculative path queues the copies of the pattern ids and the first rows behind its last kernels: its second sync
        // covers them)
        tlsTail = GrokTailCopy{hPattern, hFirst, true, false};
        int rc = lcGrokMatchDevice(patterns, C.state, opts, row, dIn, reinterpret_cast<const uint32_t*>(dIn + offAt),
                                   reinterpret_cast<const uint32_t*>(dIn + lenAt), n, static_cast<int32_t*>(H.dPattern.p),
                                   static_cast<int32_t*>(H.dFirst.p), static_cast<int32_t*>(H.dExtra.p), extraCap,
                                   static_cast<uint32_t*>(H.dNextra.p), H.dScratch.p, scratch, H.stream);
        const bool copied = tlsTail.done;
        nExtra = tlsTail.nextra;
        tlsTail = GrokTailCopy{};
        if (rc == LC_ERR_OVERFLOW || !copied) {
            HIP_TRY(hipMemcpyAsync(hFirst, H.dNextra.p, 4, hipMemcpyDeviceToHost, H.stream));
            HIP_TRY(syncCounted(H.stream));
            nExtra = uint32_t(hFirst[0]);
        }
        if (rc == LC_ERR_OVERFLOW && attempt == 0) {
            extraCap = nExtra;  // the number needed is known now
            continue;
        }
        if (rc != LC_OK) return rc;
        if (!copied) {
            HIP_TRY(hipMemcpyAsync(hPattern, H.dPattern.p, size_t(n) * 4, hipMemcpyDeviceToHost, H.stream));
            HIP_TRY(hipMemcpyAsync(hFirst, H.dFirst.p, firstBytes, hipMemcpyDeviceToHost, H.stream));
            HIP_TRY(syncCounted(H.stream));
        }
        break;
    }
    const double tMatch = hostMs();
    const size_t w = row + 2;
    std::vector<uint32_t> idx(nExtra);
    const int32_t* raw = nullptr;
    if (nExtra) {
        HIP_TRY(H.hExtra.ensure(size_t(nExtra) * w * 4));
        HIP_TRY(hipMemcpyAsync(H.hExtra.p, H.dExtra.p, size_t(nExtra) * w * 4, hipMemcpyDeviceToHost, H.stream));
        HIP_TRY(syncCounted(H.stream));
        raw = static_cast<const int32_t*>(H.hExtra.p);
        // rows arrive in atomic order: sort by (line, seq)
        for (uint32_t i = 0; i < nExtra; ++i) idx[i] = i;
        std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
            if (raw[a * w] != raw[b * w]) return raw[a * w] < raw[b * w];
            return raw[a * w + 1] < raw[b * w + 1];
        });
    }
    uint32_t base = 0, x = 0;
    B.stats = tlsStats;
    {
        char names[1024];
        (void)lc_launched_kernels(names, sizeof names);
        B.kernels = names;
    }
    for (HostJob* j : jobs) {
        j->stats = &B.stats;
        j->kernels = &B.kernels;
        j->patternSrc = hPattern + base;
        j->firstSrc = hFirst + size_t(base) * row;
        j->extraRows->clear();
        while (x < nExtra && uint32_t(raw[idx[x] * w]) < base + j->n) {
            const size_t at = j->extraRows->size();
            j->extraRows->resize(at + w);
            std::memcpy(j->extraRows->data() + at, &raw[idx[x] * w], w * 4);
            (*j->extraRows)[at] -= int32_t(base);
            ++x;
        }
        base += j->n;
    }
    if (traceHost)
        fprintf(stderr, "grok host batch: %zu groups %u values %zu bytes | H2D queued %.3f ms, device match %.3f ms, extra rows + hand-out %.3f ms\n",
                jobs.size(), n, B.bytes, tGather, tMatch - tGather, hostMs() - tMatch);
    return LC_OK;
    };
    rcAll = body();
    if (rcAll != LC_OK) {
        B.error = lc_last_error();
        for (HostJob* j : jobs) j->error = &B.error;
        (void)hipStreamSynchronize(H.stream);  // (nothing of a failed batch may still write the staging the next one reuses)
    }
    return rcAll;
}

GrokDeviceState::HostCombiner* grokCombinerFor(GrokDeviceState* state, int dev) {
    std::lock_guard<std::mutex> g(state->m);
    if (state->combiner[dev]) return state->combiner[dev];
    auto* C = new GrokDeviceState::HostCombiner();
    C->state = state;
    C->device = dev;
    lccombine::GroupCombiner<HostJob>::Hooks hooks;
    hooks.threadStart = [dev] { (void)lc_runtime_set_thread_device(dev); };  // the worker runs on the device of the threads it serves
    hooks.threadEnd = [] {
        if (lcRuntimeUsable()) lc_thread_release();  // the worker's plan, streams, pools and staging
    };
    hooks.place = [C](std::vector<HostJob*>& jobs) { return grokPlaceHostBatch(*C, jobs); };
    hooks.gather = [](HostJob& j) { grokGatherJob(j); };
    hooks.run = [C](std::vector<HostJob*>& jobs) { return grokRunHostBatch(*C, jobs); };
    lccombine::CombinerOptions o;
    if (const char* e = getenv("LC_GROK_LINGER_US")) o.lingerUs = unsigned(atoi(e));   // (A/B measurements)
    if (const char* e = getenv("LC_GROK_GAP_US")) o.gapUs = unsigned(atoi(e));
    C->combiner.reset(new lccombine::GroupCombiner<HostJob>(std::move(hooks), o));
    state->combiner[dev] = C;
    return C;
}
}  // namespace

static void grokFreeCombiners(GrokDeviceState* s) {
    for (int d = 0; d < kLcMaxDevices; ++d) {
        GrokDeviceState::HostCombiner* C = nullptr;
        {
            std::lock_guard<std::mutex> g(s->m);
            std::swap(C, s->combiner[d]);
        }
        if (!C) continue;
        C->combiner->stop();  // (groups still inside are answered first; the worker then releases what it holds on the device)
        delete C;
    }
}

int lcGrokCombinerStats(GrokDeviceState* state, uint64_t out[11]) {
    for (int i = 0; i < 11; ++i) out[i] = 0;
    if (!state) return LC_ERR_ARG;
    for (int d = 0; d < kLcMaxDevices; ++d) {
        GrokDeviceState::HostCombiner* C = nullptr;
        {
            std::lock_guard<std::mutex> g(state->m);
            C = state->combiner[d];
        }
        if (!C) continue;
        const lccombine::CombinerStats st = C->combiner->stats();
        out[0] += st.batches;
        out[1] += st.jobs;
        out[2] += st.lines;
        out[3] = std::max<uint64_t>(out[3], st.largestBatchJobs);
        out[4] += st.lingerExpired;
        out[5] += st.usIdle;
        out[6] += st.usLinger;
        out[7] += st.usPlace;
        out[8] += st.usGather;
        out[9] += st.usRun;
        out[10] += st.usTakeOut;
    }
    return LC_OK;
}

int lcGrokMatchHost(const std::vector<GrokDevicePattern>& patterns, GrokDeviceState* state, const GrokOptions& opts, uint32_t row,
                    const uint8_t* data, const uint32_t* off, const uint32_t* len, uint32_t n, int32_t* pattern,
                    const int32_t** firstRows, std::vector<int32_t>& extraRows) {
    *firstRows = nullptr;
    extraRows.clear();
    if (n == 0) return LC_OK;
    if (!state) return LC_ERR_ARG;
    if (lc_device_count() <= 0) {
        lcSetLastError("no HIP device");
        return LC_ERR_NO_DEVICE;
    }
    // Group commit (group_combiner.hpp).  ProcessorRunner hands over ONE ~1000-log group per call, from several threads that share the
    // plugin instance (core/runner/ProcessorRunner.cpp:138-142), and a batch costs the device about the same from 1 000 to 16 000 values
    // (it waits for its longest value, not for the chip).  The groups of the threads that call in together travel as ONE batch, run by
    // the worker thread of (this processor, the calling thread's device); the caller copies its own values in and its own rows out.
    int dev = 0;
    {
        const int rcDev = lcHostEntryDevice(&dev);  // the calling thread's binding decides the device
        if (rcDev != LC_OK) return rcDev;
    }
    thread_local std::vector<int32_t> tlsFirst;
    HostJob job{data, off, len, n, pattern, &tlsFirst, &extraRows, &patterns, &opts, row};
    for (uint32_t i = 0; i < n; ++i) job.bytes += len[i];
    GrokDeviceState::HostCombiner* C = grokCombinerFor(state, dev);
    const int rc = C->combiner->submit(job, [row](HostJob& j, int rcBatch) {
        if (rcBatch != LC_OK) {
            if (j.error) j.errorText = *j.error;
            return;
        }
        std::memcpy(j.pattern, j.patternSrc, size_t(j.n) * 4);
        j.first->resize(size_t(j.n) * row);
        std::memcpy(j.first->data(), j.firstSrc, size_t(j.n) * row * 4);
        if (j.stats) tlsStats = *j.stats;
        if (j.kernels) {
            size_t at = 0;
            while (at < j.kernels->size()) {
                size_t end = j.kernels->find(", ", at);
                if (end == std::string::npos) end = j.kernels->size();
                lcNoteKernel(j.kernels->substr(at, end - at).c_str());
                at = end + 2;
            }
        }
    });
    if (rc != LC_OK) {
        if (rc < 0) {
            lcSetLastError(rc == -1 ? "grok: the processor is shutting down" : "grok: the batch failed on the host (out of memory?)");
            return LC_ERR_ARG;
        }
        if (!job.errorText.empty()) lcSetLastError(job.errorText);
        return rc;
    }
    *firstRows = tlsFirst.data();
    return LC_OK;
}
