// grok_kernel.hpp -- bookkeeping kernels of the Grok matcher (grok_device.hip: the sequential path of lcGrokMatchDevice).
//
// The matching itself is done by the regex kernels (tdfa_match_kernel / nfa_match_kernel) in their "subset of lines,
// resumed search" mode.  What is restated here is the control flow of ProcessorGrok.processGrok
// (plugins/processor/grok/processor_grok.go:148-194) for a whole batch at once:
//     for each Match pattern, in order:            <- host loop, one pass per pattern over the values still undecided
//         m = FindStringMatch(value)               <- round 0 of the pattern
//         while m != nil: collect named non-empty groups; m = FindNextMatch(m)      <- further rounds (resume offsets)
//         if anything was collected: this pattern wins, stop
// Pure index/flag work, a few bytes per value per round: HBM-bound, no LDS, no MFMA.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/lc_regex_gpu.h"
#include "grok_literal_layout.h"

constexpr int kGrokBlock = 256;

// every value starts undecided, in play for Match[0], searching from its first byte
__global__ __launch_bounds__(kGrokBlock) void grok_init_kernel(uint32_t n, int32_t* __restrict__ pattern,
                                                              uint32_t* __restrict__ tried, uint32_t* __restrict__ from,
                                                              uint32_t* __restrict__ nmatch) {
    const uint32_t i = blockIdx.x * kGrokBlock + threadIdx.x;
    if (i >= n) return;
    pattern[i] = -1;
    tried[i] = i;
    from[i] = 0;
    nmatch[i] = 0;
}

// Necessary-condition prefilter: a value that does not contain the pattern's required literal (regex_handle.cpp
// requiredLiteral: the longest byte string every match must contain) cannot match, so only the others are handed to the
// automaton.  With an ordered list of 50 log formats this turns "every pattern scans every undecided value" into about
// one automaton run per value.  One value per wavefront: lane l tests the start offsets l, l+64, ...; the value is
// read through L1/L2 (it was just touched by the previous pattern's pass), the literal sits in kernel arguments.
struct GrokLiteral {
    uint32_t len;      // 1..32
    uint8_t bytes[32];
};

__global__ __launch_bounds__(kGrokBlock) void grok_literal_filter_kernel(const uint32_t* __restrict__ in, uint32_t nIn,
                                                                        const uint8_t* __restrict__ data,
                                                                        const uint32_t* __restrict__ off,
                                                                        const uint32_t* __restrict__ len, GrokLiteral lit,
                                                                        uint32_t* __restrict__ out,
                                                                        uint32_t* __restrict__ counters) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t k = blockIdx.x * (kGrokBlock / 64) + (threadIdx.x >> 6);
    if (k >= nIn) return;  // wave-uniform
    const uint32_t line = in[k];
    const uint8_t* p = data + off[line];
    const uint32_t L = len[line];
    bool found = false;
    if (L >= lit.len) {
        const uint32_t last = L - lit.len;
        for (uint32_t base = 0; base <= last && !__any(found); base += 64) {
            const uint32_t s = base + lane;
            if (s <= last && p[s] == lit.bytes[0]) {
                uint32_t j = 1;
                while (j < lit.len && p[s + j] == lit.bytes[j]) ++j;
                found = j == lit.len;
            }
        }
    }
    if (__any(found) && lane == 0) out[atomicAdd(&counters[0], 1u)] = line;
}

// ---- all required literals in ONE pass.  50 Match entries used to mean 50 literal passes over the values still undecided
// (10 % of a configs[2] step).  Instead: the Aho-Corasick automaton of all literals, as a DFA over byte classes with its table
// in global memory (a few hundred states: L1/L2-resident), walked once per value, one value per lane; the result is a 64-bit
// mask per value -- bit p = the value contains Match[p]'s literal (always set for entries without one) -- and each entry's
// filter becomes a read of 8 bytes per value.
// Blob: grok_literal_layout.h, built by grok_literal_index.cpp.

__global__ __launch_bounds__(kGrokBlock) void grok_literal_index_kernel(const uint8_t* __restrict__ data, const uint32_t* __restrict__ off,
                                                                       const uint32_t* __restrict__ len, uint32_t n,
                                                                       const uint32_t* __restrict__ blob, uint64_t* __restrict__ masks,
                                                                       uint32_t nPatterns, uint32_t* __restrict__ perPattern,
                                                                       const uint32_t* __restrict__ order) {
    __shared__ uint8_t cmap[256];
    cmap[threadIdx.x] = reinterpret_cast<const uint8_t*>(blob + GL_HEADER_WORDS)[threadIdx.x];
    __syncthreads();
    // order (optional): the values sorted by length, so that the lanes of a wavefront walk values of about the same length
    const uint32_t slot = blockIdx.x * kGrokBlock + threadIdx.x;
    if (slot >= n) {  // (whole trailing wavefronts leave here; a partial one takes part in the ballots below with an empty mask)
        if ((blockIdx.x * kGrokBlock + (threadIdx.x & ~63u)) >= n) return;
    }
    const uint32_t line = slot < n ? (order ? order[slot] : slot) : n;
    const uint32_t L = line < n ? len[line] : 0u, ncls = blob[GL_NCLASSES];
    const uint64_t* outMask = reinterpret_cast<const uint64_t*>(reinterpret_cast<const uint8_t*>(blob) + blob[GL_OFF_MASKS]);
    const uint16_t* table = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(blob) + blob[GL_OFF_TABLE]);
    const bool live = line < n;
    uint64_t mask = live ? uint64_t(blob[GL_ALWAYS_LO]) | (uint64_t(blob[GL_ALWAYS_HI]) << 32) : 0;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(data) + (live ? off[line] : 0u);
    const uint32_t head = uint32_t(addr & 15);
    const uint4* q = reinterpret_cast<const uint4*>(addr - head);
    const uint32_t total = L ? head + L : 0;
    uint32_t state = 0;
    for (uint32_t pos = 0; pos < total; pos += 16) {
        const uint4 v = *q++;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) {
            const uint32_t bi = pos + j;
            if (bi >= head && bi < total) {
                const uint32_t e = table[state * ncls + cmap[(w[j >> 2] >> ((j & 3) * 8)) & 0xFFu]];
                state = e & 0x7FFFu;
                if (e & 0x8000u) mask |= outMask[state];
            }
        }
    }
    if (live) masks[line] = mask;
    // how many values carry each entry's literal at all: an entry nobody carries is skipped without a single launch
    if (!perPattern) return;
    for (uint32_t p = 0; p < nPatterns; ++p) {
        const uint64_t has = __ballot((mask >> p) & 1u);
        if (has && (threadIdx.x & 63u) == 0) atomicAdd(&perPattern[p], uint32_t(__popcll(has)));
    }
}

// Match[bit]'s literal filter once the masks exist
__global__ __launch_bounds__(kGrokBlock) void grok_mask_filter_kernel(const uint32_t* __restrict__ in, uint32_t nIn,
                                                                     const uint64_t* __restrict__ masks, uint32_t bit,
                                                                     uint32_t* __restrict__ out, uint32_t* __restrict__ counters) {
    const uint32_t k = blockIdx.x * kGrokBlock + threadIdx.x;
    if (k >= nIn) return;
    const uint32_t line = in[k];
    if ((masks[line] >> bit) & 1u) out[atomicAdd(&counters[0], 1u)] = line;
}

// After the anchored search of round 0: the values it did NOT match go on to the search proper
__global__ __launch_bounds__(kGrokBlock) void grok_unmatched_kernel(const uint32_t* __restrict__ in, uint32_t nIn,
                                                                   const uint8_t* __restrict__ status, uint32_t* __restrict__ out,
                                                                   uint32_t* __restrict__ count) {
    const uint32_t k = blockIdx.x * kGrokBlock + threadIdx.x;
    if (k >= nIn) return;
    const uint32_t line = in[k];
    if (status[line] != LC_MATCH) out[atomicAdd(count, 1u)] = line;
}

// Keeps the values whose screen search matched (status bytes written by the TDFA kernel for the values listed in `in`).
__global__ __launch_bounds__(kGrokBlock) void grok_status_filter_kernel(const uint32_t* __restrict__ in, uint32_t nIn,
                                                                       const uint8_t* __restrict__ status,
                                                                       uint32_t* __restrict__ out,
                                                                       uint32_t* __restrict__ counters) {
    const uint32_t k = blockIdx.x * kGrokBlock + threadIdx.x;
    if (k >= nIn) return;
    const uint32_t line = in[k];
    if (status[line] == LC_MATCH) out[atomicAdd(&counters[0], 1u)] = line;
}

// After one round of searches over the values listed in `in`:
//   * a match that holds a non-empty named capture is recorded (first one -> `first`, later ones -> `extra`);
//   * FindNextMatch: the value stays in play from the end of this match (one further after an empty match) unless
//     that is the end of the value -- a match starting there is empty and contributes nothing;
//   * LC_OVERFLOW (NFA engine out of threads, nothing settled it): the value is undecidable here, pattern = -2, dropped from
//     play; LC_GAVE_UP (the decide kernel ran out of budget -- regexp2's match timeout, processor_grok.go:156-160: the reference
//     returns matchTimeOut and tries no further pattern): pattern = -3, dropped from play.
// counters[0] = values in `out`, counters[1] = rows wanted in `extra`.
__global__ __launch_bounds__(kGrokBlock) void grok_advance_kernel(
    const uint32_t* __restrict__ in, uint32_t nIn, const uint8_t* __restrict__ status, const int32_t* __restrict__ caps,
    uint32_t capsRow, uint32_t row, uint32_t columns, const uint32_t* __restrict__ len, uint32_t* __restrict__ from,
    uint32_t* __restrict__ nmatch, int32_t* __restrict__ pattern, int32_t* __restrict__ first,
    int32_t* __restrict__ extra, uint32_t extraCap, uint32_t* __restrict__ out, uint32_t* __restrict__ counters) {
    const uint32_t k = blockIdx.x * kGrokBlock + threadIdx.x;
    if (k >= nIn) return;
    const uint32_t line = in[k];
    const uint8_t st = status[line];
    if (st == LC_OVERFLOW || st == LC_GAVE_UP) {
        pattern[line] = st == LC_OVERFLOW ? -2 : -3;
        return;
    }
    if (st != LC_MATCH) return;
    const int32_t* c = caps + size_t(line) * capsRow;  // this pattern's own row: whole match + its columns
    bool contributes = false;
    for (uint32_t g = 1; g <= columns; ++g) contributes |= c[2 * g] >= 0 && c[2 * g + 1] > c[2 * g];
    if (contributes) {
        const uint32_t seq = nmatch[line]++;
        int32_t* dst = nullptr;
        if (seq == 0) {
            dst = first + size_t(line) * row;
        } else {
            const uint32_t at = atomicAdd(&counters[1], 1u);
            if (at < extraCap) {
                dst = extra + size_t(at) * (row + 2);
                dst[0] = int32_t(line);
                dst[1] = int32_t(seq);
                dst += 2;
            }
        }
        if (dst) {
            for (uint32_t s = 0; s < 2 * (columns + 1); ++s) dst[s] = c[s];
            if (seq)  // rows of `first` were preset to -1 by the caller; rows of `extra` are not
                for (uint32_t s = 2 * (columns + 1); s < row; ++s) dst[s] = -1;
        }
    }
    const uint32_t b = uint32_t(c[0]), e = uint32_t(c[1]);
    const uint32_t next = e > b ? e : e + 1;
    if (next < len[line]) {
        from[line] = next;
        out[atomicAdd(&counters[0], 1u)] = line;
    }
}

// After the last round of pattern p over the values in `tried`: values that collected something are decided, the others
// go on to the next pattern with a fresh search.  counters[2] = values in `next`.
__global__ __launch_bounds__(kGrokBlock) void grok_finish_kernel(const uint32_t* __restrict__ tried, uint32_t nTried,
                                                                int32_t p, const uint32_t* __restrict__ nmatch,
                                                                int32_t* __restrict__ pattern, uint32_t* __restrict__ from,
                                                                uint32_t* __restrict__ next,
                                                                uint32_t* __restrict__ counters) {
    const uint32_t k = blockIdx.x * kGrokBlock + threadIdx.x;
    if (k >= nTried) return;
    const uint32_t line = tried[k];
    if (pattern[line] <= -2) return;
    if (nmatch[line]) {
        pattern[line] = p;
    } else {
        from[line] = 0;
        next[atomicAdd(&counters[2], 1u)] = line;
    }
}
