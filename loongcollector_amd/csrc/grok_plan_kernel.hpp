// grok_plan_kernel.hpp -- kernels of the SPECULATIVE Grok matcher (grok_device.hip: lcGrokMatchDevice, plan path).
//
// ProcessorGrok.processGrok (plugins/processor/grok/processor_grok.go:148-194) walks the Match list in order and stops at the
// first entry whose matches yield a non-empty named capture.  Whether entry p yields one for value v does not depend on any
// other entry, so the batch form evaluates every (entry, value) pair that can possibly match AT THE SAME TIME and takes the
// minimum over the entries afterwards:
//
//   1. grok_literal_index_kernel (grok_kernel.hpp): one pass, a 64-bit mask per value -- bit p = "contains entry p's literal";
//   2. grok_screen_all_kernel: ONE launch for the screens of all entries -- workgroup (slice of values, entry): the entry's
//      yes/no DFA staged into LDS, walked over the slice's carriers of bit p; a rejected value loses the bit;
//   3. grok_count_kernel -> the host reads how many candidates each entry has (sync 1) and carves per-entry arrays;
//   4. grok_scatter_kernel: every (entry, value) pair becomes a SLOT of the entry (offset, length, line): from here on each
//      entry is a batch of its own -- the regex kernels see "line" = slot, nothing else changes for them;
//   5. per entry, on one of a few streams: search rounds (FindStringMatch / FindNextMatch) with grok_advance2_kernel in
//      between; the number of values still in play stays on the device (count pointers), a fixed number of rounds is queued;
//   6. grok_entry_finish_kernel: atomicMin per value over the entries that contributed / could not decide;
//      grok_resolve_kernel: the winner; grok_commit_kernel / grok_commit_extra_kernel: the winner's rows go out.
//   The host reads one word at the end (sync 2): were values still in play after the last queued round?  (Then, and only then,
//   it finishes those entries round by round and runs step 6 again.)
//
// Index / flag work: HBM-bound, a few bytes per pair; the DFA walks are bound by one dependent LDS read per byte.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/lc_regex_gpu.h"
#include "screen_kernel_layout.h"

constexpr int kGrokPlanBlock = 256;
constexpr uint32_t kGrokRemWaveSlots = 16;  // grok_remainder_all_kernel, wave walk: slots per workgroup
constexpr uint32_t kGrokNone = 0xFFFFFFFFu;
// nmatch word of a slot: low bits = contributing matches so far, high bits = the entry could not decide the value
constexpr uint32_t kGrokSlotOverflow = 0x80000000u;  // thread lists overflowed and nothing settled it
constexpr uint32_t kGrokSlotGaveUp = 0x40000000u;    // the decide kernel ran out of budget (regexp2's match timeout)
constexpr uint32_t kGrokSlotCount = 0x3FFFFFFFu;
// per-entry counter block (u32 words)
enum {
    GC_FILL = 0,        // slots filled by the scatter kernel (== candidates)
    GC_UNANCHORED = 1,  // slots the anchored search of round 0 did not match
    GC_ROUND0 = 2,      // GC_ROUND0 + r: slots still in play after round r
    GC_BOUND = 59,      // entries of level >= 1: slots whose value no entry of level 0 had won after round 0 (an upper bound of what the
                        // entry will have to search in phase 2c: 0 = its chain is not queued at all)
    GC_WIDE = 60,       // set by a wide-first launch of round 0 when some value did need more than 64 threads (nfa_wide_kernel.hpp)
    GC_OVERFLOW = 61,   // slots whose thread lists overflowed in the first-chance kernel of round 0 (they get the second chance)
    GC_REMAINDER = 62,  // slots still in play after round 0 whose REMAINDER passes the entry's screen (grok_remainder_all_kernel)
    GC_FILTERED = 63,   // second-pass entries: slots left after grok_filter_won_kernel
    GC_WORDS = 64
};

// One screen of one Match entry (device table).
struct GrokScreenDev {
    const uint32_t* blob;  // screen_kernel_layout.h
    uint32_t bit;          // Match index
    uint32_t ldsBytes;     // bytes of (accept flags + table) staged into LDS; 0 = walk the table in global memory
    uint32_t bigBytes;     // != 0: too large to be staged beside other workgroups, small enough for a workgroup that has a CU's LDS to
                           // itself -- small batches walk these screens in a launch of their own (grok_device.hip phase 1)
    uint32_t pad;
};

// One ACTIVE entry (device table): the entry's private batch.
struct GrokEntryDev {
    uint32_t* off;     // [cand] byte offset of the slot's value
    uint32_t* len;     // [cand]
    uint32_t* line;    // [cand] the value's index in the caller's batch
    uint32_t* from;    // [cand] resume offset of the next search
    uint32_t* nmatch;  // [cand] kGrokSlot*
    int32_t* first;    // [cand][capsRow] row of the first contributing match
    uint32_t* cnt;     // GC_*
    uint32_t cand, capsRow, bit, pad;
    // round 0 as one launch over all entries (grok_post_kernel) and the remainder screens (grok_remainder_all_kernel)
    const uint8_t* status;  // [cand] what the match kernels reported
    const int32_t* caps;    // [cand][capsRow]
    uint32_t* listA;        // slots in play after round 0
    uint32_t* unanchored;   // slots the anchored search did not match; later: the screened in-play list round 1 reads
    uint32_t* ovList;       // slots the first-chance kernel could not decide (thread-list overflow)
    uint32_t columns;       // named groups
    uint32_t anchored;      // 1: round 0 ran the ANCHORED search (what it does not match goes on to the search proper)
};

// flags of grok_post_kernel
enum {
    GP_ANCHORED_PASS = 1,   // the statuses come from the anchored searches (entries that have one): NOMATCH -> the unanchored list
    GP_OVERFLOW_FINAL = 2   // LC_OVERFLOW is final (the second chance has run): the slot is undecided; else it goes to the overflow list
};

struct GrokSlotMap {
    int8_t activeOfBit[64];  // Match index -> index into the GrokEntryDev table (-1: not active)
};

// ---- every buffer a batch needs cleared, in ONE launch at its start (round 6: they were seven hipMemsetAsync calls strewn over phase 1,
// each a dispatch of its own between the kernels that matter)
struct GrokInitJobs {
    uint32_t* p[8];
    uint32_t words[8];
    uint32_t value[8];
    uint32_t n;
};
__global__ __launch_bounds__(kGrokPlanBlock) void grok_init_kernel(GrokInitJobs J) {
    for (uint32_t j = 0; j < J.n; ++j)
        for (uint32_t i = blockIdx.x * kGrokPlanBlock + threadIdx.x; i < J.words[j]; i += gridDim.x * kGrokPlanBlock) J.p[j][i] = J.value[j];
}

// no literal index for this list (fewer than two literals): every entry is a candidate for every value
__global__ __launch_bounds__(kGrokPlanBlock) void grok_mask_fill_kernel(uint64_t* __restrict__ masks, uint32_t n, uint64_t all) {
    const uint32_t i = blockIdx.x * kGrokPlanBlock + threadIdx.x;
    if (i < n) masks[i] = all;
}

// ---- the literal index for SMALL batches: one value per WAVEFRONT, 64-byte chunks dealt to the lanes.  An occurrence of a literal
// (at most 32 bytes are indexed) that ends inside a chunk starts at most 31 bytes before it, and the Aho-Corasick automaton finds
// every occurrence that lies inside the window it walks whatever state it starts in -- so every lane starts at the root, 31 bytes
// before its chunk, and the masks are OR-ed across the wave.  95 dependent steps for a 4 KiB value instead of 4096: what a small
// batch waits for is its longest value.  (Large batches use the lane-per-value kernel, in length order.)
#include "grok_literal_layout.h"
constexpr uint32_t kGrokChunk = 64, kGrokLookBehind = 31;
__global__ __launch_bounds__(kGrokPlanBlock) void grok_literal_chunk_kernel(const uint8_t* __restrict__ data, const uint32_t* __restrict__ off,
                                                                           const uint32_t* __restrict__ len, uint32_t n,
                                                                           const uint32_t* __restrict__ blob, uint64_t* __restrict__ masks) {
    __shared__ uint8_t cmap[256];
    cmap[threadIdx.x] = reinterpret_cast<const uint8_t*>(blob + GL_HEADER_WORDS)[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t v = blockIdx.x * (kGrokPlanBlock / 64) + (threadIdx.x >> 6);
    if (v >= n) return;  // wave-uniform
    const uint32_t L = len[v], ncls = blob[GL_NCLASSES];
    const uint64_t* outMask = reinterpret_cast<const uint64_t*>(reinterpret_cast<const uint8_t*>(blob) + blob[GL_OFF_MASKS]);
    const uint16_t* table = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(blob) + blob[GL_OFF_TABLE]);
    const uint8_t* p = data + off[v];
    uint64_t mask = 0;
    for (uint32_t c0 = lane * kGrokChunk; c0 < L; c0 += 64 * kGrokChunk) {
        const uint32_t lo = c0 >= kGrokLookBehind ? c0 - kGrokLookBehind : 0;
        const uint32_t hi = c0 + kGrokChunk < L ? c0 + kGrokChunk : L;
        uint32_t state = 0;
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t e = table[state * ncls + cmap[p[i]]];
            state = e & 0x7FFFu;
            if (e & 0x8000u) mask |= outMask[state];
        }
    }
    uint32_t mlo = uint32_t(mask), mhi = uint32_t(mask >> 32);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mlo |= uint32_t(__shfl_xor(int(mlo), d, 64));
        mhi |= uint32_t(__shfl_xor(int(mhi), d, 64));
    }
    if (lane == 0) masks[v] = (uint64_t(mhi) << 32) | mlo | uint64_t(blob[GL_ALWAYS_LO]) | (uint64_t(blob[GL_ALWAYS_HI]) << 32);
}

// ---- the same index with its tables in LDS (round 6).  The index of configs[2] is 547 states x 61 classes (67 KB) + 4 KB of output
// masks: every step of a lane's chunk was a dependent u16 load through L1 / L2 behind a byte load -- 95 steps of ~450 cycles, 0.14 ms
// for a 16 Ki-value batch of which a fifth was the literal pass.  Here the launch has as many workgroups as the chip holds (two per
// CU at 71 KB), each stages [masks | table] once and takes values in turn, one per wavefront; a chunk's bytes are loaded and mapped to
// classes sixteen at a time, in front of the sixteen dependent LDS reads.  stageBytes = the blob from GL_OFF_MASKS to its end.
__global__ __launch_bounds__(kGrokPlanBlock) void grok_literal_lds_kernel(const uint8_t* __restrict__ data, const uint32_t* __restrict__ off,
                                                                         const uint32_t* __restrict__ len, uint32_t n,
                                                                         const uint32_t* __restrict__ blob, uint64_t* __restrict__ masks,
                                                                         uint32_t stageBytes, const uint32_t* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) uint32_t litWords[];
    __shared__ uint8_t cmap[256];
    cmap[threadIdx.x] = reinterpret_cast<const uint8_t*>(blob + GL_HEADER_WORDS)[threadIdx.x];
    {
        const uint32_t* src = blob + blob[GL_OFF_MASKS] / 4;
        for (uint32_t i = threadIdx.x; i < stageBytes / 4; i += kGrokPlanBlock) litWords[i] = src[i];
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, ncls = blob[GL_NCLASSES];
    const uint64_t* outMask = reinterpret_cast<const uint64_t*>(litWords);
    const uint16_t* table = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(litWords) + (blob[GL_OFF_TABLE] - blob[GL_OFF_MASKS]));
    const uint64_t always = uint64_t(blob[GL_ALWAYS_LO]) | (uint64_t(blob[GL_ALWAYS_HI]) << 32);
    // A wavefront takes FOUR values at a time, neighbours in the length order: a lane's work is one 64-byte chunk whatever the value's
    // length, and the mean value (1.1 KB) kept 18 of 64 lanes busy.  Four values of up to 1 KiB share the wavefront (16 lanes each), of
    // up to 2 KiB two by two, longer ones take it in turn.
    const uint32_t nQuads = (n + 3u) / 4u;
    for (uint32_t quad = blockIdx.x * (kGrokPlanBlock / 64) + (threadIdx.x >> 6); quad < nQuads; quad += gridDim.x * (kGrokPlanBlock / 64)) {  // wave-uniform
        uint32_t myLen = 0;
        if (lane < 4u && quad * 4u + lane < n) myLen = len[order ? order[quad * 4u + lane] : quad * 4u + lane];
        uint32_t longest = myLen;
        longest = max(longest, uint32_t(__shfl_xor(int(longest), 1, 64)));
        longest = max(longest, uint32_t(__shfl_xor(int(longest), 2, 64)));
        longest = uint32_t(__builtin_amdgcn_readfirstlane(longest));
        const uint32_t groupLanes = longest <= 16u * kGrokChunk ? 16u : longest <= 32u * kGrokChunk ? 32u : 64u;
        const uint32_t perPass = 64u / groupLanes;
        for (uint32_t first = 0; first < 4u; first += perPass) {
            const uint32_t at = quad * 4u + first + lane / groupLanes;
            const bool have = at < n && lane / groupLanes < perPass;
            const uint32_t v = have ? (order ? order[at] : at) : 0u;
            const uint32_t L = have ? len[v] : 0u;
            const uint8_t* p = data + (have ? off[v] : 0u);
            const uint32_t lg = lane % groupLanes;
            uint64_t mask = 0;
            for (uint32_t c0 = lg * kGrokChunk; c0 < L; c0 += groupLanes * kGrokChunk) {
                const uint32_t lo = c0 >= kGrokLookBehind ? c0 - kGrokLookBehind : 0;
                const uint32_t hi = c0 + kGrokChunk < L ? c0 + kGrokChunk : L;
                uint32_t state = 0;
                uint32_t i = lo;
                // (16 bytes per load: the lanes' chunks are 64 bytes apart, so every load instruction of the wavefront is 64 cache-line
                // requests whatever its width.  gfx950 takes the unaligned 16-byte loads, as in tdfa_stream_kernel.)
                for (; i + 16 <= hi; i += 16) {
                    const uint4 q = *reinterpret_cast<const uint4*>(p + i);
                    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
                    uint32_t cls[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) cls[k] = cmap[(w[k >> 2] >> ((k & 3) * 8)) & 0xFFu];
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const uint32_t e = table[state * ncls + cls[k]];
                        state = e & 0x7FFFu;
                        if (e & 0x8000u) mask |= outMask[state];
                    }
                }
                for (; i < hi; ++i) {
                    const uint32_t e = table[state * ncls + cmap[p[i]]];
                    state = e & 0x7FFFu;
                    if (e & 0x8000u) mask |= outMask[state];
                }
            }
            uint32_t mlo = uint32_t(mask), mhi = uint32_t(mask >> 32);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                if (uint32_t(d) < groupLanes) {  // (wave-uniform: the OR stays inside the value's lanes)
                    mlo |= uint32_t(__shfl_xor(int(mlo), d, 64));
                    mhi |= uint32_t(__shfl_xor(int(mhi), d, 64));
                }
            }
            if (have && lg == 0) masks[v] = (uint64_t(mhi) << 32) | mlo | always;
        }
    }
}

// The walk of a yes/no screen DFA over [p, p + L): one value per lane, 16 bytes per global load.  Round 5: the NEXT 16 bytes are
// loaded while these are walked, their 16 classes are looked up before the chain starts (independent LDS reads), and a chunk that
// lies wholly inside the value takes a copy of the chain without per-byte bounds tests -- what remains per byte is the one dependent
// table read.  (Before: load, wait, then per byte a class lookup, two compares and the table read -- 117 ns a byte with the table in
// LDS, 0.48 ms for a 4 KiB value: the whole screen phase of a small batch; profiles/round5_grok_timeline.txt.)
__device__ __forceinline__ uint32_t grokScreenWalk(const uint8_t* p, uint32_t L, uint32_t start, uint32_t sink, uint32_t ncls,
                                                   const uint8_t* cmap, const uint16_t* table) {
    uint32_t state = start;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(p);
    const uint32_t head = uint32_t(addr & 15);
    const uint4* q = reinterpret_cast<const uint4*>(addr - head);
    const uint32_t total = L ? head + L : 0;
    uint4 cur = total ? q[0] : uint4{0, 0, 0, 0};
    for (uint32_t pos = 0; pos < total && state != sink && state != 0; pos += 16) {
        ++q;
        const uint4 nxt = pos + 16 < total ? *q : uint4{0, 0, 0, 0};
        const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
        uint32_t cls[16];
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) cls[j] = cmap[(w[j >> 2] >> ((j & 3) * 8)) & 0xFFu];
        if (pos >= head && pos + 16 <= total) {
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) state = table[state * ncls + cls[j]];
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) {
                const uint32_t bi = pos + j;
                if (bi >= head && bi < total) state = table[state * ncls + cls[j]];
            }
        }
        cur = nxt;
    }
    return state;
}

// Round 6: the same walk over a table whose entries were SCALED when the table was staged into LDS -- an entry is the byte offset of the
// next state's row (next x classes x 2) instead of its index.  The dependent chain of a byte was  ds_read_u16 -> v_mul_lo_u32 (quarter
// rate) -> v_lshlrev -> v_add3 -> ds_read_u16 : 110 ns a byte, 0.44 ms for a 4 KiB value -- the screen phase of every small batch,
// and once more the remainder screens of phase 2c.  Scaled it is  ds_read_u16 -> v_add3 -> ds_read_u16.  Tables of up to 64 KiB
// (offsets fit 16 bits: every staged table).  `state` in and out: row byte offsets (0 = dead).
__device__ __forceinline__ uint32_t grokScreenWalkScaled(const uint8_t* p, uint32_t L, uint32_t startOff, uint32_t sinkOff, const uint8_t* cmap,
                                                         const uint16_t* ldsTable) {
    uint32_t state = startOff;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(p);
    const uint32_t head = uint32_t(addr & 15);
    const uint4* q = reinterpret_cast<const uint4*>(addr - head);
    const uint32_t total = L ? head + L : 0;
    const uint8_t* tableBytes = reinterpret_cast<const uint8_t*>(ldsTable);
    uint4 cur = total ? q[0] : uint4{0, 0, 0, 0};
    for (uint32_t pos = 0; pos < total && state != sinkOff && state != 0; pos += 16) {
        ++q;
        const uint4 nxt = pos + 16 < total ? *q : uint4{0, 0, 0, 0};
        const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
        uint32_t cls2[16];  // class x 2: the byte offset inside a row
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) cls2[j] = uint32_t(cmap[(w[j >> 2] >> ((j & 3) * 8)) & 0xFFu]) << 1;
        if (pos >= head && pos + 16 <= total) {
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) state = *reinterpret_cast<const uint16_t*>(tableBytes + state + cls2[j]);
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) {
                const uint32_t bi = pos + j;
                if (bi >= head && bi < total) state = *reinterpret_cast<const uint16_t*>(tableBytes + state + cls2[j]);
            }
        }
        cur = nxt;
    }
    return state;
}
// staging a screen's [accept flags | table] block into LDS; scale: the table's entries become row byte offsets (see above)
__device__ __forceinline__ void grokStageScreen(uint32_t* dst, const uint32_t* src, uint32_t stageBytes, uint32_t tableAt, uint32_t rowBytes,
                                                bool scale, uint32_t tid) {
    uint32_t i = tid;
    auto conv = [&](uint32_t w, uint32_t wordIndex) {
        if (!scale || wordIndex * 4 < tableAt) return w;
        return ((w & 0xFFFFu) * rowBytes) | (((w >> 16) * rowBytes) << 16);
    };
    for (; i + 3 * kGrokPlanBlock < stageBytes / 4; i += 4 * kGrokPlanBlock) {  // (four loads in flight: a big table is 100+ KB)
        const uint32_t a = src[i], b = src[i + kGrokPlanBlock], c = src[i + 2 * kGrokPlanBlock], d = src[i + 3 * kGrokPlanBlock];
        dst[i] = conv(a, i);
        dst[i + kGrokPlanBlock] = conv(b, i + kGrokPlanBlock);
        dst[i + 2 * kGrokPlanBlock] = conv(c, i + 2 * kGrokPlanBlock);
        dst[i + 3 * kGrokPlanBlock] = conv(d, i + 3 * kGrokPlanBlock);
    }
    for (; i < stageBytes / 4; i += kGrokPlanBlock) dst[i] = conv(src[i], i);
}

// The same walk by a whole WAVEFRONT (round 6).  A screen is a relaxed whole-pattern automaton: between the few bytes that move it,
// it sits in states that loop on nearly everything ("X.*Y.*Z").  One value per lane paid a dependent table read for every one of a
// value's bytes -- 110 ns each, 0.44 ms for a 4 KiB value: the screen phase of a small batch, and again the remainder screens of phase
// 2c (profiles/round5_grok_timeline.txt).  Here the state is wave-uniform, the wave holds 256 bytes of the value as classes (4 per
// lane), and a byte that keeps the state starts a scan: every lane asks the table whether ITS four bytes keep it too, a ballot finds
// the first one that does not -- 256 bytes per step instead of one.  Same states as grokScreenWalk, byte for byte.
template <typename TablePtr>
__device__ __forceinline__ uint32_t grokScreenWalkWave(const uint8_t* p, uint32_t L, uint32_t start, uint32_t sink, uint32_t ncls,
                                                       const uint8_t* cmap, TablePtr table, uint32_t lane) {
    uint32_t state = start;
    if (L == 0) return state;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(p);
    const uint32_t head = uint32_t(addr & 3);
    const uint32_t* words = reinterpret_cast<const uint32_t*>(addr - head);
    const uint32_t end = head + L, nWords = (end + 3) / 4;
    auto classesOf = [&](uint32_t w4) {
        return uint32_t(cmap[w4 & 0xFFu]) | (uint32_t(cmap[(w4 >> 8) & 0xFFu]) << 8) | (uint32_t(cmap[(w4 >> 16) & 0xFFu]) << 16) |
               (uint32_t(cmap[w4 >> 24]) << 24);
    };
    uint32_t chunk = 0, idx = head;
    uint32_t cur = classesOf(lane < nWords ? words[lane] : 0u);
    while (idx < end && state != sink && state != 0) {
        if ((idx >> 8) != chunk) {
            chunk = idx >> 8;
            const uint32_t w = (chunk << 6) + lane;
            cur = classesOf(w < nWords ? words[w] : 0u);
        }
        const uint32_t wsel = uint32_t(__builtin_amdgcn_readlane(int(cur), int((idx >> 2) & 63u)));
        const uint32_t cls = (wsel >> ((idx & 3u) * 8)) & 0xFFu;
        const uint32_t row = state * ncls;
        const uint32_t next = uint32_t(__builtin_amdgcn_readfirstlane(int(table[row + cls])));
        if (next != state) {
            state = next;
            ++idx;
            continue;
        }
        // a byte that keeps the state: where does the run of such bytes end?
        uint32_t stop = end;
        for (;;) {
            const uint32_t chunkBase = chunk << 8;
            uint32_t firstHit = 4;
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                const uint32_t bi = chunkBase + lane * 4 + uint32_t(j);
                const uint32_t c = (cur >> (8 * j)) & 0xFFu;
                if (bi > idx && bi < end && uint32_t(table[row + c]) != state) firstHit = uint32_t(j);
            }
            const uint64_t hit = __ballot(firstHit < 4);
            if (hit) {
                const int l = __ffsll(static_cast<long long>(hit)) - 1;
                stop = chunkBase + uint32_t(l) * 4 + uint32_t(__builtin_amdgcn_readlane(int(firstHit), l));
                break;
            }
            if (chunkBase + 256 >= end) break;  // the run reaches the end of the value
            ++chunk;
            const uint32_t w = (chunk << 6) + lane;
            cur = classesOf(w < nWords ? words[w] : 0u);
        }
        idx = stop;
    }
    return state;
}

// ---- all screens in one launch.  grid = (slices, screens); dynamic LDS = candidate list [sliceLen] u32 + staged table.
// A slice is a run of values in LENGTH order (order[]: the lanes of a wavefront then walk values of about the same length); the
// workgroup compacts the slice's carriers of the entry's bit into LDS (ballot), then walks them one value per lane.  Table entries
// are read as u16 next-state.  stage != 0 (small batches: what counts is the latency of the longest value): the table is staged
// into LDS behind the list; stage == 0 (large batches: what counts is values in flight per CU): it is read through L2.
__global__ __launch_bounds__(kGrokPlanBlock) void grok_screen_all_kernel(const uint8_t* __restrict__ data, const uint32_t* __restrict__ off,
                                                                        const uint32_t* __restrict__ len, uint32_t n, uint32_t sliceLen,
                                                                        const GrokScreenDev* __restrict__ screens,
                                                                        unsigned long long* __restrict__ masks,
                                                                        const uint32_t* __restrict__ order, uint32_t stageAndWalk) {
    // stage: 0 = tables through L2, 1 = the screens' ldsBytes staged, 2 = the launch of the BIG screens: bigBytes staged
    // + 16: one value per WAVEFRONT (grokScreenWalkWave) instead of one per lane
    const uint32_t stage = stageAndWalk & 15u;
    const bool waveWalk = (stageAndWalk & 16u) != 0;
    extern __shared__ uint32_t ldsWords[];
    __shared__ uint8_t cmap[256];
    __shared__ uint32_t sCount;
    // (round 6) + 128: the grid is (screens, slices) instead of (slices, screens) -- workgroups are dispatched x first, the slices are in
    // LENGTH order, and a launch lasts as long as its longest value: slice 0 of EVERY screen goes out first, instead of the last
    // screens' slice 0 behind two thousand workgroups of short values
    const bool transposed = (stageAndWalk & 128u) != 0;
    const uint32_t sliceIdx = transposed ? blockIdx.y : blockIdx.x;
    const GrokScreenDev sc = screens[transposed ? blockIdx.x : blockIdx.y];
    const uint32_t tid = threadIdx.x;
    const uint32_t lo = sliceIdx * sliceLen;
    const uint32_t hi = lo + sliceLen < n ? lo + sliceLen : n;
    if (tid == 0) sCount = 0;
    __syncthreads();
    uint32_t* cand = ldsWords;
    // 1. carriers of this slice
    for (uint32_t base = lo; base < hi; base += kGrokPlanBlock) {
        const uint32_t i = base + tid;
        const uint32_t v = i < hi ? (order ? order[i] : i) : 0;
        const bool has = i < hi && ((masks[v] >> sc.bit) & 1ull);
        const uint64_t b = __ballot(has);
        if (b) {
            uint32_t at = 0;
            if ((tid & 63u) == 0) at = atomicAdd(&sCount, uint32_t(__popcll(b)));
            at = __shfl(at, 0, 64);
            if (has) cand[at + __popcll(b & ((1ull << (tid & 63u)) - 1ull))] = v;
        }
    }
    __syncthreads();
    const uint32_t count = sCount;
    if (count == 0) return;
    // 2. the entry's automaton
    const uint32_t* blob = sc.blob;
    cmap[tid] = reinterpret_cast<const uint8_t*>(blob + SC_HEADER_WORDS)[tid];
    const uint32_t ncls = blob[SC_NCLASSES], sink = blob[SC_SINK], start = blob[SC_START];
    const uint8_t* accept = reinterpret_cast<const uint8_t*>(blob) + blob[SC_OFF_ACCEPT];
    const uint16_t* table = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(blob) + blob[SC_OFF_TABLE]);
    const uint32_t stageBytes = stage == 2 ? sc.bigBytes : stage ? sc.ldsBytes : 0u;
    const bool staged = stageBytes != 0;
    // (round 6) a staged table of up to 64 KiB is staged SCALED: entries = row byte offsets (grokScreenWalkScaled); the wave walk and
    // LC_GROK_SCREEN_SCALED=0 (the launcher clears bit 5 of stageAndWalk) keep the indices
    const uint32_t tableAt = blob[SC_OFF_TABLE] - blob[SC_OFF_ACCEPT], rowBytes = ncls * 2;
    const bool scaled = staged && !waveWalk && (stageAndWalk & 32u) != 0 && (tableAt & 3u) == 0 && blob[SC_NSTATES] * rowBytes <= 0xFFFFu;
    if (staged) grokStageScreen(ldsWords + sliceLen, reinterpret_cast<const uint32_t*>(accept), stageBytes, tableAt, rowBytes, scaled, tid);
    __syncthreads();
    const uint8_t* lAccept = staged ? reinterpret_cast<const uint8_t*>(ldsWords + sliceLen) : accept;
    // (an LDS pointer the compiler can see is one: ds_read instead of flat loads in the walk)
    const uint16_t* ldsTable = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(ldsWords + sliceLen) +
                                                                 (blob[SC_OFF_TABLE] - blob[SC_OFF_ACCEPT]));
    // 3. walk
    if (waveWalk) {  // one value per wavefront (grokScreenWalkWave): small batches wait for their longest value
        const uint32_t lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        for (uint32_t k = wave; k < count; k += kGrokPlanBlock / 64) {
            const uint32_t v = __builtin_amdgcn_readfirstlane(cand[k]);
            const uint32_t o = __builtin_amdgcn_readfirstlane(off[v]), L = __builtin_amdgcn_readfirstlane(len[v]);
            const uint32_t state = staged ? grokScreenWalkWave(data + o, L, start, sink, ncls, cmap, ldsTable, lane)
                                          : grokScreenWalkWave(data + o, L, start, sink, ncls, cmap, table, lane);
            const bool pass = state == sink || (state != 0 && lAccept[state]);
            if (!pass && lane == 0) atomicAnd(&masks[v], ~(1ull << sc.bit));
        }
        return;
    }
    for (uint32_t k = tid; k < count; k += kGrokPlanBlock) {
        const uint32_t v = cand[k];
        uint32_t state;
        if (scaled) state = grokScreenWalkScaled(data + off[v], len[v], start * rowBytes, sink == 0xFFFFFFFFu ? 0xFFFFFFFFu : sink * rowBytes, cmap, ldsTable) / rowBytes;
        else state = staged ? grokScreenWalk(data + off[v], len[v], start, sink, ncls, cmap, ldsTable)
                            : grokScreenWalk(data + off[v], len[v], start, sink, ncls, cmap, table);
        const bool pass = state == sink || (state != 0 && lAccept[state]);
        if (!pass) atomicAnd(&masks[v], ~(1ull << sc.bit));
    }
}

// candidates per entry after the screens; firstOf[p] = values whose FIRST candidate entry is p (the rest of p's candidates have
// an earlier candidate that may win them: potential waste of evaluating p on them)
// shadowBy[p * 64 + f] (optional): candidates of p whose value's FIRST candidate is the earlier entry f -- who shadows whom, for the
// order in which the entries are evaluated (grok_device.hip: levels)
__global__ __launch_bounds__(kGrokPlanBlock) void grok_count_kernel(const uint64_t* __restrict__ masks, uint32_t n, uint32_t nPatterns,
                                                                   uint32_t* __restrict__ perEntry, uint32_t* __restrict__ firstOf,
                                                                   uint32_t* __restrict__ shadowBy) {
    // (round 6) The counts gather in LDS and leave the workgroup as ONE add per cell: a hot cell (a general format behind a specific one:
    // a thousand values) used to take a thousand device-wide adds on one word, one after the other at the L2 -- 0.06 ms for a launch
    // that reads 128 KB.
    __shared__ uint32_t sPer[64], sFirst[64], sShadow[64 * 64];
    for (uint32_t i = threadIdx.x; i < 64 * 64; i += kGrokPlanBlock) sShadow[i] = 0;
    if (threadIdx.x < 64) {
        sPer[threadIdx.x] = 0;
        sFirst[threadIdx.x] = 0;
    }
    __syncthreads();
    const uint32_t v = blockIdx.x * kGrokPlanBlock + threadIdx.x;
    const uint64_t m = v < n ? masks[v] : 0;
    const uint32_t lowest = m ? uint32_t(__ffsll(static_cast<long long>(m))) - 1u : 64u;
    if (shadowBy && m) {
        uint64_t rest = m & (m - 1);  // the candidates behind the first one (a value has one to three)
        while (rest) {
            const uint32_t p = uint32_t(__ffsll(static_cast<long long>(rest))) - 1u;
            rest &= rest - 1;
            atomicAdd(&sShadow[p * 64 + lowest], 1u);
        }
    }
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t mine = 0, mineFirst = 0;
    for (uint32_t p = 0; p < nPatterns; ++p) {  // lane p holds the wavefront's counts of pattern p
        const uint64_t has = __ballot((m >> p) & 1ull);
        const uint64_t isFirst = __ballot(lowest == p);
        if (lane == p) {
            mine = uint32_t(__popcll(has));
            mineFirst = uint32_t(__popcll(isFirst));
        }
    }
    if (mine) atomicAdd(&sPer[lane], mine);
    if (mineFirst) atomicAdd(&sFirst[lane], mineFirst);
    __syncthreads();
    if (threadIdx.x < 64) {
        if (sPer[threadIdx.x]) atomicAdd(&perEntry[threadIdx.x], sPer[threadIdx.x]);
        if (sFirst[threadIdx.x]) atomicAdd(&firstOf[threadIdx.x], sFirst[threadIdx.x]);
    }
    if (shadowBy)
        for (uint32_t i = threadIdx.x; i < 64 * 64; i += kGrokPlanBlock)
            if (sShadow[i]) atomicAdd(&shadowBy[i], sShadow[i]);
}

// every (entry, value) pair that passed becomes a slot of the entry
__global__ __launch_bounds__(kGrokPlanBlock) void grok_scatter_kernel(const uint64_t* __restrict__ masks, uint32_t n, uint32_t nPatterns,
                                                                     const uint32_t* __restrict__ off, const uint32_t* __restrict__ len,
                                                                     GrokSlotMap map, const GrokEntryDev* __restrict__ entries,
                                                                     const uint32_t* __restrict__ order) {
    // (in length order, longest first: an entry's slots come out roughly sorted too -- homogeneous wavefronts for its
    // lane-per-value kernels, the long values of the wave-per-value kernels first)
    const uint32_t i = blockIdx.x * kGrokPlanBlock + threadIdx.x;
    const uint32_t v = i < n ? (order ? order[i] : i) : n;
    const uint64_t m = v < n ? masks[v] : 0;
    const uint32_t o = v < n ? off[v] : 0, L = v < n ? len[v] : 0;
    // (round 6) the wavefront's places in ALL entries behind one round trip: lane p adds the wavefront's number of candidates of
    // pattern p to that entry's fill count (round 5: one atomic with a return value per pattern, 46 round trips one after the other)
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t mine = 0;
    for (uint32_t p = 0; p < nPatterns; ++p) {
        const uint64_t b = __ballot((m >> p) & 1ull);
        if (lane == p) mine = uint32_t(__popcll(b));
    }
    uint32_t myAt = 0;
    if (mine) {
        const int a = map.activeOfBit[lane];
        if (a >= 0) myAt = atomicAdd(&entries[a].cnt[GC_FILL], mine);
    }
    for (uint32_t p = 0; p < nPatterns; ++p) {
        const bool has = (m >> p) & 1ull;
        const uint64_t b = __ballot(has);
        if (!b) continue;
        const int a = map.activeOfBit[p];
        if (a < 0) continue;
        const GrokEntryDev& e = entries[a];
        const uint32_t at = __shfl(myAt, int(p), 64);
        if (has) {
            const uint32_t slot = at + __popcll(b & ((1ull << lane) - 1ull));
            if (slot < e.cand) {
                e.off[slot] = o;
                e.len[slot] = L;
                e.line[slot] = v;
                e.from[slot] = 0;
                e.nmatch[slot] = 0;
            }
        }
    }
}

// the slots the anchored search of round 0 did not match go on to the search proper
// (in: the slots that were searched, nullptr = all below the bound; inCount: the list's length on the device)
__global__ __launch_bounds__(kGrokPlanBlock) void grok_unmatched2_kernel(const uint32_t* __restrict__ in, uint32_t bound,
                                                                        const uint32_t* __restrict__ inCount,
                                                                        const uint8_t* __restrict__ status, uint32_t* __restrict__ out,
                                                                        uint32_t* __restrict__ count) {
    uint32_t nIn = bound;
    if (inCount) {
        const uint32_t dyn = *inCount;
        nIn = dyn < nIn ? dyn : nIn;
    }
    const uint32_t k = blockIdx.x * kGrokPlanBlock + threadIdx.x;
    if (k >= nIn) return;
    const uint32_t slot = in ? in[k] : k;
    if (status[slot] != LC_MATCH) out[atomicAdd(count, 1u)] = slot;
}

// Second pass (entries whose candidates are largely shadowed by earlier entries): only the slots whose value no earlier entry
// has won by now are searched.  An entry's result on a value that an earlier entry contributes to is never looked at, so
// dropping those slots changes nothing -- and whatever the first pass has not settled yet simply stays in.
// in / inCount (optional): the slots to look at (a list and its length on the device); nullptr = every slot of the entry
__global__ __launch_bounds__(kGrokPlanBlock) void grok_filter_won_kernel(GrokEntryDev e, const uint32_t* __restrict__ winner,
                                                                        const uint32_t* __restrict__ in, const uint32_t* __restrict__ inCount,
                                                                        uint32_t* __restrict__ out, uint32_t* __restrict__ count) {
    uint32_t nIn = e.cand;
    if (inCount) {
        const uint32_t dyn = *inCount;
        nIn = dyn < nIn ? dyn : nIn;
    }
    const uint32_t i = blockIdx.x * kGrokPlanBlock + threadIdx.x;
    const uint32_t k = i < nIn ? (in ? in[i] : i) : 0;
    const bool keep = i < nIn && winner[e.line[k]] >= e.bit;  // (kGrokNone included)
    const uint64_t b = __ballot(keep);
    if (!b) return;
    uint32_t at = 0;
    if ((threadIdx.x & 63u) == 0) at = atomicAdd(count, uint32_t(__popcll(b)));
    at = __shfl(at, 0, 64);
    if (keep) out[at + __popcll(b & ((1ull << (threadIdx.x & 63u)) - 1ull))] = k;
}

// Entries of level >= 1 (grid.y = entries from entryBase): how many of their slots hold a value nobody has won so far -> GC_BOUND.
__global__ __launch_bounds__(kGrokPlanBlock) void grok_bound_kernel(const GrokEntryDev* __restrict__ entries, uint32_t entryBase,
                                                                   const uint32_t* __restrict__ winner) {
    const GrokEntryDev& e = entries[entryBase + blockIdx.y];
    const uint32_t k = blockIdx.x * kGrokPlanBlock + threadIdx.x;
    const bool open = k < e.cand && winner[e.line[k]] >= e.bit;
    const uint64_t b = __ballot(open);
    if (b && (threadIdx.x & 63u) == 0) atomicAdd(&e.cnt[GC_BOUND], uint32_t(__popcll(b)));
}

// FindNextMatch (processor_grok.go:176) searches what is left of the value behind a match -- for a format that ends 150 bytes into a
// 4 KiB value that is 3.9 KiB in which nothing will be found, on the slowest engine there is (an unanchored search on the thread-list
// kernel starts an attempt at every byte the format can begin with).  A further match lies entirely inside [from, len), and every
// match of the entry contains a match of the entry's screen (the relaxed whole pattern, or the prefix; no assertions), so the slots
// whose remainder the screen rejects are done: grok_remainder_all_kernel below, one lane per slot, the yes/no DFA walked from `from`.

// a row of (begin, end) pairs from one array to another, four pairs in flight (the arrays never overlap; a plain loop waits for
// every load before its store: a 60-word row was 60 round trips to L2, grok_commit_kernel 0.08 ms)
__device__ __forceinline__ void grokCopyPairs(int2* __restrict__ dst, const int2* __restrict__ src, uint32_t pairs) {
    uint32_t g = 0;
    for (; g + 4 <= pairs; g += 4) {
        const int2 a = src[g], b = src[g + 1], c = src[g + 2], d = src[g + 3];
        dst[g] = a;
        dst[g + 1] = b;
        dst[g + 2] = c;
        dst[g + 3] = d;
    }
    for (; g < pairs; ++g) dst[g] = src[g];
}

// One list append per WAVEFRONT: the lanes that want a place (all lanes of the wavefront call this, converged, with the same counter)
// take consecutive places behind one atomic add of their number.  -> the lane's place (undefined for a lane that does not want one)
__device__ __forceinline__ uint32_t grokAppend(uint32_t* counter, bool want) {
    const uint64_t b = __ballot(want);
    if (!b) return 0u;
    const uint32_t lane = threadIdx.x & 63u;
    const int leader = __ffsll(static_cast<long long>(b)) - 1;
    uint32_t base = 0;
    if (int(lane) == leader) base = atomicAdd(counter, uint32_t(__popcll(b)));
    base = __shfl(base, leader, 64);
    return base + uint32_t(__popcll(b & ((1ull << lane) - 1ull)));
}

// ---- round 0 in ONE launch for all entries (grid.y = entries from entryBase): what grok_unmatched2_kernel + grok_advance2_kernel did
// per entry.  in / inCount (single-entry launches only): the slots to look at and their number on the device; nullptr = every slot.
__global__ __launch_bounds__(kGrokPlanBlock) void grok_post_kernel(const GrokEntryDev* __restrict__ entries, uint32_t entryBase,
                                                                  const uint32_t* __restrict__ in, const uint32_t* __restrict__ inCount,
                                                                  uint32_t flags, int32_t* __restrict__ xtmp, uint32_t xcap, uint32_t xstride,
                                                                  uint32_t* __restrict__ xcount, unsigned long long skip) {
    if ((skip >> (entryBase + blockIdx.y)) & 1ull) return;  // (bit a: entry a of the table is not this launch's)
    const GrokEntryDev& e = entries[entryBase + blockIdx.y];
    uint32_t nIn = e.cand;
    if (inCount) {
        const uint32_t dyn = *inCount;
        nIn = dyn < nIn ? dyn : nIn;
    }
    if (blockIdx.x * kGrokPlanBlock >= nIn) return;  // (workgroup-uniform)
    // (round 6) No lane leaves before the lists are appended to: a wavefront appends with ONE atomic per list (grokAppend).  Round 0 of
    // a general format leaves ten thousand slots in play, and ten thousand atomics with a return value on one word took 0.14 ms between
    // the end of round 0 and everything behind it (profiles/round6_grok_timeline.txt).
    const uint32_t k = blockIdx.x * kGrokPlanBlock + threadIdx.x;
    const bool live = k < nIn;
    const uint32_t slot = live ? (in ? in[k] : k) : 0u;
    const uint8_t st = live ? e.status[slot] : uint8_t(LC_NOMATCH);
    const bool toOverflow = live && st == LC_OVERFLOW && !(flags & GP_OVERFLOW_FINAL);
    {
        const uint32_t at = grokAppend(&e.cnt[GC_OVERFLOW], toOverflow);
        if (toOverflow) e.ovList[at] = slot;
    }
    const bool undecidedSlot = live && !toOverflow && (st == LC_OVERFLOW || st == LC_GAVE_UP);
    if (undecidedSlot) e.nmatch[slot] |= st == LC_OVERFLOW ? kGrokSlotOverflow : kGrokSlotGaveUp;
    const bool matched = live && st == LC_MATCH;
    const bool toUnanchored = live && !toOverflow && !undecidedSlot && !matched && (flags & GP_ANCHORED_PASS) && e.anchored;
    {
        const uint32_t at = grokAppend(&e.cnt[GC_UNANCHORED], toUnanchored);
        if (toUnanchored) e.unanchored[at] = slot;
    }
    bool inPlay = false;
    if (matched) {
        // (round 6: a row is read as (begin, end) PAIRS -- capsRow is even, rows are 8-byte aligned -- and without a branch between the
        // loads: `c[2g] >= 0 && c[2g+1] > c[2g]` was twenty dependent round trips to L2 per slot, the post step of a general format 0.1 ms)
        const uint32_t capsRow = e.capsRow, pairs = capsRow >> 1;
        const int2* c2 = reinterpret_cast<const int2*>(e.caps + size_t(slot) * capsRow);
        uint32_t contributes = 0;
        auto test = [&](const int2 be) { contributes |= uint32_t(be.x >= 0) & uint32_t(be.y > be.x); };
        uint32_t g = 1;
        for (; g + 4 <= pairs; g += 4) {  // (four loads in flight)
            const int2 p0 = c2[g], p1 = c2[g + 1], p2 = c2[g + 2], p3 = c2[g + 3];
            test(p0);
            test(p1);
            test(p2);
            test(p3);
        }
        for (; g < pairs; ++g) test(c2[g]);
        const int2 whole = c2[0];
        if (contributes) {
            const uint32_t seq = e.nmatch[slot]++ & kGrokSlotCount;
            if (seq == 0) {
                grokCopyPairs(reinterpret_cast<int2*>(e.first + size_t(slot) * capsRow), c2, pairs);
            } else {
                const uint32_t at = atomicAdd(xcount, 1u);
                if (at < xcap) {
                    int32_t* dst = xtmp + size_t(at) * xstride;
                    dst[0] = int32_t(e.line[slot]);
                    dst[1] = int32_t(seq);
                    dst[2] = int32_t(e.bit);
                    const int32_t* c = reinterpret_cast<const int32_t*>(c2);
                    for (uint32_t s = 0; s < capsRow; ++s) dst[3 + s] = c[s];
                }
            }
        }
        const uint32_t b = uint32_t(whole.x), en = uint32_t(whole.y);
        const uint32_t next = en > b ? en : en + 1;
        if (next < e.len[slot]) {
            e.from[slot] = next;
            inPlay = true;
        }
    }
    const uint32_t at = grokAppend(&e.cnt[GC_ROUND0], inPlay);
    if (inPlay) e.listA[at] = slot;
}

// ---- round 5: in front of the remainder screens, the LITERAL index over the remainders (grid.y = entries, one slot in play per
// wavefront, chunk-parallel like grok_literal_chunk_kernel).  A value is a candidate of an entry only if it contains the entry's
// required literal, and every match of the entry contains it: a further match needs an occurrence INSIDE the remainder.  The typical
// slot in play -- a format matched at the head of a 4 KiB value, free text behind -- has none, and is settled by 95 dependent steps
// of the literal automaton instead of 4 000 of the screen's.  Such a slot gets from = len (nothing left to search: the screen kernel
// below rejects it without a walk).  Entries without a literal (the index's ALWAYS bits) are left to the screen.  skip: bit a = the
// entry's rounds were queued ahead by its history (grok_device.hip) -- nothing to do here.
// winner (optional): the values won so far -- a slot whose value an EARLIER entry has won is finished too (nothing this entry finds in it is
// ever looked at); blob == nullptr: that test alone.
__global__ __launch_bounds__(kGrokPlanBlock) void grok_remainder_literal_kernel(const uint8_t* __restrict__ data,
                                                                               const GrokEntryDev* __restrict__ entries,
                                                                               const uint32_t* __restrict__ blob, unsigned long long skip,
                                                                               const uint32_t* __restrict__ winner, uint32_t quietOnly) {
    __shared__ uint8_t cmap[256];
    if ((skip >> blockIdx.y) & 1ull) return;
    const GrokEntryDev& e = entries[blockIdx.y];
    // quietOnly: the launch queued BEFORE the host has read round 0's counts takes the entries round 0 left nothing else to do for; an
    // entry with overflowed or unanchored slots screens its remainders at the end of its chain in phase 2c (grok_device.hip)
    if (quietOnly && (e.cnt[GC_OVERFLOW] | e.cnt[GC_UNANCHORED])) return;
    uint32_t nIn = e.cnt[GC_ROUND0];
    nIn = nIn < e.cand ? nIn : e.cand;
    if (blockIdx.x * (kGrokPlanBlock / 64) >= nIn) return;
    const uint64_t always = blob ? uint64_t(blob[GL_ALWAYS_LO]) | (uint64_t(blob[GL_ALWAYS_HI]) << 32) : ~0ull;
    const bool hasLiteral = !((always >> e.bit) & 1ull);
    if (!hasLiteral && !winner) return;
    if (hasLiteral) cmap[threadIdx.x] = reinterpret_cast<const uint8_t*>(blob + GL_HEADER_WORDS)[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t k = blockIdx.x * (kGrokPlanBlock / 64) + (threadIdx.x >> 6);
    if (k >= nIn) return;  // wave-uniform
    const uint32_t slot = e.listA[k];
    const uint32_t from = e.from[slot], len = e.len[slot];
    if (from >= len) return;
    if (winner && winner[e.line[slot]] < e.bit) {
        if (lane == 0) e.from[slot] = len;
        return;
    }
    if (!hasLiteral) return;
    const uint32_t L = len - from, ncls = blob[GL_NCLASSES];
    const uint64_t* outMask = reinterpret_cast<const uint64_t*>(reinterpret_cast<const uint8_t*>(blob) + blob[GL_OFF_MASKS]);
    const uint16_t* table = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(blob) + blob[GL_OFF_TABLE]);
    const uint8_t* p = data + e.off[slot] + from;
    uint32_t hit = 0;
    for (uint32_t c0 = lane * kGrokChunk; c0 < L; c0 += 64 * kGrokChunk) {
        const uint32_t lo = c0 >= kGrokLookBehind ? c0 - kGrokLookBehind : 0;
        const uint32_t hi = c0 + kGrokChunk < L ? c0 + kGrokChunk : L;
        uint32_t state = 0;
        uint32_t i = lo;
        // (round 6: sixteen bytes per load and their classes looked up in front of the chain, as in grok_literal_lds_kernel -- what is left
        // per byte is the table read)
        for (; i + 16 <= hi; i += 16) {
            const uint4 q = *reinterpret_cast<const uint4*>(p + i);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
            uint32_t cls[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) cls[k] = cmap[(w[k >> 2] >> ((k & 3) * 8)) & 0xFFu];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t t = table[state * ncls + cls[k]];
                state = t & 0x7FFFu;
                if ((t & 0x8000u) && ((outMask[state] >> e.bit) & 1ull)) hit = 1;
            }
        }
        for (; i < hi; ++i) {
            const uint32_t t = table[state * ncls + cmap[p[i]]];
            state = t & 0x7FFFu;
            if ((t & 0x8000u) && ((outMask[state] >> e.bit) & 1ull)) hit = 1;
        }
    }
    if (!__any(hit != 0) && lane == 0) e.from[slot] = len;
}

// ---- the remainder screens of ALL entries in one launch (grid.y = entries): see above.  screens[a].blob ==
// nullptr: the entry has no screen, every slot in play goes on.  In: e.listA / GC_ROUND0; out: e.unanchored / GC_REMAINDER.
// skip: see grok_remainder_literal_kernel.
__global__ __launch_bounds__(kGrokPlanBlock) void grok_remainder_all_kernel(const uint8_t* __restrict__ data,
                                                                           const GrokEntryDev* __restrict__ entries,
                                                                           const GrokScreenDev* __restrict__ screens, uint32_t stageAndWalk,
                                                                           unsigned long long skip, uint32_t* __restrict__ gate) {
    // gate (optional): every survivor counts onto the batch's gate word (round 6: grok_survivor_gate_kernel's launch behind the screens)
    const uint32_t stage = stageAndWalk & 15u;          // (as grok_screen_all_kernel)
    const bool waveWalk = (stageAndWalk & 16u) != 0;    // one slot per WAVEFRONT: the workgroup takes the slots of its 256-slot window in turn
    extern __shared__ uint32_t ldsWords[];
    __shared__ uint8_t cmap[256];
    if ((skip >> blockIdx.y) & 1ull) return;
    const GrokEntryDev& e = entries[blockIdx.y];
    if ((stageAndWalk & 64u) && (e.cnt[GC_OVERFLOW] | e.cnt[GC_UNANCHORED])) return;  // quietOnly: see grok_remainder_literal_kernel
    const GrokScreenDev sc = screens[blockIdx.y];
    const uint32_t tid = threadIdx.x;
    uint32_t nIn = e.cnt[GC_ROUND0];
    nIn = nIn < e.cand ? nIn : e.cand;
    // (wave walk: a workgroup takes kGrokRemWaveSlots slots, four at a time -- the host sizes the grid by the same number)
    const uint32_t perBlock = (waveWalk && sc.blob) ? kGrokRemWaveSlots : uint32_t(kGrokPlanBlock);
    if (!sc.blob && waveWalk && blockIdx.x * kGrokPlanBlock >= nIn) return;
    if (blockIdx.x * perBlock >= nIn) return;
    const uint32_t k = blockIdx.x * kGrokPlanBlock + tid;
    if (!sc.blob) {
        if (k < nIn) {
            const uint32_t slot = e.listA[k];
            if (e.from[slot] < e.len[slot]) {
                e.unanchored[atomicAdd(&e.cnt[GC_REMAINDER], 1u)] = slot;
                if (gate) atomicAdd(gate, 1u);
            }
        }
        return;
    }
    const uint32_t* blob = sc.blob;
    cmap[tid] = reinterpret_cast<const uint8_t*>(blob + SC_HEADER_WORDS)[tid];
    const uint32_t ncls = blob[SC_NCLASSES], sink = blob[SC_SINK], start = blob[SC_START];
    const uint8_t* accept = reinterpret_cast<const uint8_t*>(blob) + blob[SC_OFF_ACCEPT];
    const uint16_t* table = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(blob) + blob[SC_OFF_TABLE]);
    const uint32_t stageBytes = stage == 2 ? sc.bigBytes : stage ? sc.ldsBytes : 0u;  // (2: the launch of the entries with BIG screens)
    const bool staged = stageBytes != 0;
    const uint32_t tableAt = blob[SC_OFF_TABLE] - blob[SC_OFF_ACCEPT], rowBytes = ncls * 2;
    const bool scaled = staged && !waveWalk && (stageAndWalk & 32u) != 0 && (tableAt & 3u) == 0 && blob[SC_NSTATES] * rowBytes <= 0xFFFFu;  // (as grok_screen_all_kernel)
    if (staged) grokStageScreen(ldsWords, reinterpret_cast<const uint32_t*>(accept), stageBytes, tableAt, rowBytes, scaled, tid);
    __syncthreads();
    const uint8_t* lAccept = staged ? reinterpret_cast<const uint8_t*>(ldsWords) : accept;
    const uint16_t* ldsTable = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(ldsWords) + (blob[SC_OFF_TABLE] - blob[SC_OFF_ACCEPT]));
    if (waveWalk) {
        const uint32_t lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const uint32_t k0 = blockIdx.x * kGrokRemWaveSlots, k1 = k0 + kGrokRemWaveSlots < nIn ? k0 + kGrokRemWaveSlots : nIn;
        for (uint32_t kk = k0 + wave; kk < k1; kk += kGrokPlanBlock / 64) {
            const uint32_t slot = __builtin_amdgcn_readfirstlane(e.listA[kk]);
            const uint32_t L = __builtin_amdgcn_readfirstlane(e.len[slot]), from = __builtin_amdgcn_readfirstlane(e.from[slot]);
            const uint32_t o = __builtin_amdgcn_readfirstlane(e.off[slot]);
            const uint32_t rem = L > from ? L - from : 0;
            const uint32_t state = staged ? grokScreenWalkWave(data + o + from, rem, start, sink, ncls, cmap, ldsTable, lane)
                                          : grokScreenWalkWave(data + o + from, rem, start, sink, ncls, cmap, table, lane);
            if ((state == sink || (state != 0 && lAccept[state])) && lane == 0) {
                e.unanchored[atomicAdd(&e.cnt[GC_REMAINDER], 1u)] = slot;
                if (gate) atomicAdd(gate, 1u);
            }
        }
        return;
    }
    if (k >= nIn) return;
    const uint32_t slot = e.listA[k];
    const uint32_t L = e.len[slot], from = e.from[slot];
    const uint32_t rem = L > from ? L - from : 0;
    uint32_t state;
    if (scaled) state = grokScreenWalkScaled(data + e.off[slot] + from, rem, start * rowBytes, sink == 0xFFFFFFFFu ? 0xFFFFFFFFu : sink * rowBytes, cmap, ldsTable) / rowBytes;
    else state = staged ? grokScreenWalk(data + e.off[slot] + from, rem, start, sink, ncls, cmap, ldsTable)
                        : grokScreenWalk(data + e.off[slot] + from, rem, start, sink, ncls, cmap, table);
    if (state == sink || (state != 0 && lAccept[state])) {
        e.unanchored[atomicAdd(&e.cnt[GC_REMAINDER], 1u)] = slot;
        if (gate) atomicAdd(gate, 1u);
    }
}

// After one search round over the slots in `in` (nullptr: all slots below the bound); see grok_advance_kernel (grok_kernel.hpp)
// for the rules.  inCount (optional): the list's length on the device.  Extra rows go to xtmp as [line, seq, bit, row...].
__global__ __launch_bounds__(kGrokPlanBlock) void grok_advance2_kernel(
    const uint32_t* __restrict__ in, uint32_t bound, const uint32_t* __restrict__ inCount, const uint8_t* __restrict__ status,
    const int32_t* __restrict__ caps, uint32_t capsRow, uint32_t columns, GrokEntryDev e, int32_t* __restrict__ xtmp, uint32_t xcap,
    uint32_t xstride, uint32_t* __restrict__ xcount, uint32_t* __restrict__ out, uint32_t* __restrict__ outCount,
    uint32_t* __restrict__ gate) {
    uint32_t nIn = bound;
    if (inCount) {
        const uint32_t dyn = *inCount;
        nIn = dyn < nIn ? dyn : nIn;
    }
    const uint32_t k = blockIdx.x * kGrokPlanBlock + threadIdx.x;
    if (k >= nIn) return;
    const uint32_t slot = in ? in[k] : k;
    const uint8_t st = status[slot];
    if (st == LC_OVERFLOW || st == LC_GAVE_UP) {
        e.nmatch[slot] |= st == LC_OVERFLOW ? kGrokSlotOverflow : kGrokSlotGaveUp;
        return;
    }
    if (st != LC_MATCH) return;
    const int32_t* c = caps + size_t(slot) * capsRow;
    bool contributes = false;
    for (uint32_t g = 1; g <= columns; ++g) contributes |= c[2 * g] >= 0 && c[2 * g + 1] > c[2 * g];
    if (contributes) {
        const uint32_t seq = e.nmatch[slot]++ & kGrokSlotCount;
        int32_t* dst = nullptr;
        if (seq == 0) {
            dst = e.first + size_t(slot) * capsRow;
        } else {
            const uint32_t at = atomicAdd(xcount, 1u);
            if (at < xcap) {
                dst = xtmp + size_t(at) * xstride;
                dst[0] = int32_t(e.line[slot]);
                dst[1] = int32_t(seq);
                dst[2] = int32_t(e.bit);
                dst += 3;
            }
        }
        if (dst)
            for (uint32_t s = 0; s < capsRow; ++s) dst[s] = c[s];
    }
    const uint32_t b = uint32_t(c[0]), en = uint32_t(c[1]);
    const uint32_t next = en > b ? en : en + 1;
    if (next < e.len[slot]) {
        e.from[slot] = next;
        out[atomicAdd(outCount, 1u)] = slot;
        if (gate) atomicAdd(gate, 1u);
    }
}

// One wavefront: the survivors of the remainder screens (GC_REMAINDER of every active entry) go onto the gate word (add != 0) or come
// off it again (add == 0) -- see the end of phase 2c in grok_device.hip.
__global__ __launch_bounds__(64) void grok_survivor_gate_kernel(const GrokEntryDev* __restrict__ entries, uint32_t nAct, uint32_t* __restrict__ gate,
                                                                uint32_t add) {
    const uint32_t a = threadIdx.x;
    const uint32_t c = a < nAct ? entries[a].cnt[GC_REMAINDER] : 0u;
    if (c) {
        if (add) atomicAdd(gate, c);
        else atomicSub(gate, c);
    }
}

// grid = (ceil(max cand / block), active entries).  gate (optional): leave at once while values are still in play somewhere.
__global__ __launch_bounds__(kGrokPlanBlock) void grok_entry_finish_kernel(const GrokEntryDev* __restrict__ entries,
                                                                          uint32_t* __restrict__ winner, uint32_t* __restrict__ undecided,
                                                                          const uint32_t* __restrict__ gate, uint32_t entryBase) {
    if (gate && *gate) return;
    const GrokEntryDev& e = entries[entryBase + blockIdx.y];
    // (second-pass entries: a slot the filter dropped has nmatch == 0 and says nothing)
    const uint32_t k = blockIdx.x * kGrokPlanBlock + threadIdx.x;
    if (k >= e.cand) return;
    const uint32_t nm = e.nmatch[k];
    const uint32_t line = e.line[k];
    if (nm & (kGrokSlotOverflow | kGrokSlotGaveUp))
        atomicMin(&undecided[line], (e.bit << 1) | ((nm & kGrokSlotOverflow) ? 0u : 1u));
    else if (nm)
        atomicMin(&winner[line], e.bit);
}

// pattern[v] = the first entry that contributed; an entry that could not decide the value stops the list there
// (-2: thread-list overflow nobody settled; -3: gave up = the reference's matchTimeOut, processor_grok.go:156-160)
__global__ __launch_bounds__(kGrokPlanBlock) void grok_resolve_kernel(uint32_t n, const uint32_t* __restrict__ winner,
                                                                     const uint32_t* __restrict__ undecided, int32_t* __restrict__ pattern,
                                                                     const uint32_t* __restrict__ gate) {
    if (gate && *gate) return;
    const uint32_t v = blockIdx.x * kGrokPlanBlock + threadIdx.x;
    if (v >= n) return;
    const uint32_t w = winner[v], u = undecided[v];
    int32_t p = w == kGrokNone ? -1 : int32_t(w);
    if (u != kGrokNone && (w == kGrokNone || (u >> 1) <= w)) p = (u & 1u) ? -3 : -2;
    pattern[v] = p;
}

// the winner's first row goes out (d_first was preset to -1: only the entry's own columns are written); grid.y = active entries + 1.
// Row nAct of the grid (round 6: grok_commit_extra_kernel's launch): the further matches of the winners,
// [line, seq, bit, row...] -> [line, seq, row... padded with -1]; nextra = rows wanted, counted into the caller's word and into the
// batch's tail words (nextraCopy: what the host reads back with them).
__global__ __launch_bounds__(kGrokPlanBlock) void grok_commit_kernel(const GrokEntryDev* __restrict__ entries, uint32_t nAct,
                                                                    const int32_t* __restrict__ pattern, int32_t* __restrict__ first,
                                                                    uint32_t row, const uint32_t* __restrict__ gate,
                                                                    const int32_t* __restrict__ xtmp, const uint32_t* __restrict__ xcount,
                                                                    uint32_t xcap, uint32_t xstride, GrokSlotMap map,
                                                                    int32_t* __restrict__ extra, uint32_t extraCap,
                                                                    uint32_t* __restrict__ nextra, uint32_t* __restrict__ nextraCopy) {
    if (gate && *gate) return;
    if (blockIdx.y == nAct) {
        uint32_t total = *xcount;
        total = total < xcap ? total : xcap;
        for (uint32_t r = blockIdx.x * kGrokPlanBlock + threadIdx.x; r < total; r += gridDim.x * kGrokPlanBlock) {
            const int32_t* src = xtmp + size_t(r) * xstride;
            const int32_t line = src[0], bit = src[2];
            if (pattern[line] != bit) continue;
            const uint32_t at = atomicAdd(nextra, 1u);
            atomicAdd(nextraCopy, 1u);
            if (at >= extraCap) continue;
            const uint32_t capsRow = entries[map.activeOfBit[bit]].capsRow;
            int32_t* dst = extra + size_t(at) * (row + 2);
            dst[0] = line;
            dst[1] = src[1];
            for (uint32_t s = 0; s < row; ++s) dst[2 + s] = s < capsRow ? src[3 + s] : -1;
        }
        return;
    }
    const GrokEntryDev& e = entries[blockIdx.y];
    const uint32_t k = blockIdx.x * kGrokPlanBlock + threadIdx.x;
    if (k >= e.cand) return;
    const uint32_t nm = e.nmatch[k];
    if (!nm || (nm & (kGrokSlotOverflow | kGrokSlotGaveUp))) return;
    const uint32_t line = e.line[k];
    if (pattern[line] != int32_t(e.bit)) return;
    const int32_t* src = e.first + size_t(k) * e.capsRow;
    int32_t* dst = first + size_t(line) * row;
    // (capsRow and row are even and the plan's own arrays 256-byte aligned; the caller's `first` is int32 by contract: tested)
    if ((reinterpret_cast<uintptr_t>(first) & 7u) == 0) grokCopyPairs(reinterpret_cast<int2*>(dst), reinterpret_cast<const int2*>(src), e.capsRow >> 1);
    else
        for (uint32_t s = 0; s < e.capsRow; ++s) dst[s] = src[s];
}
