// nfa_kernel.hpp -- gfx950 kernel of the follow-NFA engine (included by gpu_runtime.hip only).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/lc_regex_gpu.h"
#include "device_tables.h"

// One line per wavefront.  Lane t holds the t-th live thread of the Pike VM in backtracking-priority order:
// (position, capture offsets in VGPRs).  Per input byte (wave-uniform, fetched with v_readlane from a 256-byte
// chunk the wave loaded with one coalesced dword per lane):
//   1. every lane walks its position's follow list (LDS) and keeps the paths whose target accepts the byte class
//      and whose assertions hold; ds_min on best[target] elects, per target, the candidate of highest priority
//      (rank = lexicographic (lane, path index));
//   2. winners are compacted with a wave prefix sum into the new thread order and scattered through LDS;
//   3. lane j pulls its captures from its source lane with ds_bpermute and stamps the tagged slots.
constexpr int kNfaBlock = 256;  // 4 wavefronts = 4 lines per workgroup, sharing one LDS copy of the tables
constexpr int kNfaWaves = kNfaBlock / 64;

// LDS hand-off between lanes of ONE wavefront: order the wave's own DS operations and stop the compiler from
// moving loads/stores across the hand-off (no s_barrier needed, the wave is the only party).
__device__ __forceinline__ void waveLdsSync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t waveExclusiveScan(uint32_t v, uint32_t lane, uint32_t& total) {
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= uint32_t(d)) incl += up;
    }
    total = __shfl(incl, 63, 64);
    return incl - v;
}

template <int NS>
__global__ __launch_bounds__(kNfaBlock) void nfa_match_kernel(const uint8_t* __restrict__ data,
                                                              const uint32_t* __restrict__ off,
                                                              const uint32_t* __restrict__ len, uint32_t sepBytes,
                                                              uint32_t nLines, const uint32_t* __restrict__ nLinesPtr,
                                                              const uint32_t* __restrict__ order,
                                                              const uint32_t* __restrict__ resume,
                                                              const uint32_t* __restrict__ blob,
                                                              uint32_t blobBytes, uint32_t nGroupsOut,
                                                              int32_t* __restrict__ caps,
                                                              uint8_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    if (nLinesPtr) {  // line count produced on the device (split kernels) -- no host round trip between the launches
        const uint32_t dyn = *nLinesPtr;
        nLines = dyn < nLines ? dyn : nLines;
    }
    {
        const uint4* src = reinterpret_cast<const uint4*>(blob);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = tid; i < blobBytes / 16; i += kNfaBlock) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(smem);
    const uint32_t nPos = hdr[NF_NPOS];
    const uint32_t nSlots = hdr[NF_NSLOTS];
    const uint8_t* classMap = smem + hdr[NF_OFF_CLASSMAP];
    const uint2* posMask = reinterpret_cast<const uint2*>(smem + hdr[NF_OFF_POSMASK]);
    const uint2* stable = reinterpret_cast<const uint2*>(smem + hdr[NF_OFF_STABLE]);
    const uint32_t* behindBits = reinterpret_cast<const uint32_t*>(smem + hdr[NF_OFF_BEHIND]);
    const uint32_t* aheadBits = reinterpret_cast<const uint32_t*>(smem + hdr[NF_OFF_AHEAD]);
    const uint32_t edgeClass = hdr[NF_NCLASSES];  // table index standing for start / end of input
    const uint32_t* followStart = reinterpret_cast<const uint32_t*>(smem + hdr[NF_OFF_FOLLOWSTART]);
    const uint4* paths = reinterpret_cast<const uint4*>(smem + hdr[NF_OFF_PATHS]);

    const uint32_t wave = tid >> 6, lane = tid & 63;
    // per-wave scratch: best[nPos] then 4 x 64 words (newPos, newSrc, newTagsLo, newTagsHi)
    const uint32_t scratchWords = ((nPos + 3) & ~3u) + 256;
    uint32_t* best = reinterpret_cast<uint32_t*>(smem + blobBytes) + wave * scratchWords;
    uint32_t* newPos = best + ((nPos + 3) & ~3u);
    uint32_t* newSrc = newPos + 64;
    uint32_t* newTagsLo = newSrc + 64;
    uint32_t* newTagsHi = newTagsLo + 64;
    for (uint32_t i = lane; i < nPos; i += 64) best[i] = 0xFFFFFFFFu;
    waveLdsSync();

    const uint32_t slot = blockIdx.x * kNfaWaves + wave;
    if (slot >= nLines) return;  // wave-uniform
    const uint32_t line = order ? order[slot] : slot;
    const uint32_t o = off[line];
    const uint32_t L = len ? len[line] : off[line + 1] - o - sepBytes;

    int32_t cap[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) cap[s] = -1;
    uint32_t nThreads = 1;
    uint32_t myPos = nPos;  // lane 0: the start pseudo-position
    bool overflow = false;

    const uintptr_t addr = reinterpret_cast<uintptr_t>(data) + o;
    const uint32_t head = uint32_t(addr & 3);
    const uint32_t* words = reinterpret_cast<const uint32_t*>(addr - head);
    const uint32_t nWords = L ? (head + L + 3) / 4 : 0;
    uint32_t prevCls = edgeClass;
    uint32_t from = 0;  // search patterns: resume the search here (the next match of an iterate-all-matches caller)
    if (resume) {
        from = resume[line];
        from = from < L ? from : L;
        if (from) {  // one thread, on the wrapper's prefix position, which has just consumed the byte before `from`
            myPos = 0;
            prevCls = classMap[data[size_t(o) + from - 1]];
        }
    }
    uint32_t curWord;
    {
        const uint32_t w = (((head + from) >> 8) << 6) + lane;  // the 256-byte chunk the first byte lives in
        curWord = (w < nWords) ? words[w] : 0;
    }

    for (uint32_t i = from; i < L && nThreads; ++i) {
        const uint32_t idx = head + i;
        if (i != from && (idx & 255u) == 0) {  // next 256-byte chunk: one coalesced dword per lane
            const uint32_t w = (idx >> 2) + lane;
            curWord = (w < nWords) ? words[w] : 0;
        }
        const uint32_t wsel = __builtin_amdgcn_readlane(curWord, (idx >> 2) & 63u);
        const int b = int((wsel >> ((idx & 3u) * 8)) & 0xFFu);
        const uint32_t cls = classMap[b];
        const bool liveLane = lane < nThreads;
        // steady state: every live thread sits on a position whose only move on this byte class is its own
        // unconditional, tag-free self loop (inside a field such as [^ ]* that is almost every byte) -> the thread
        // list, its order and the captures are unchanged; skip the whole election/compaction/transfer machinery.
        {
            const uint2 st = stable[myPos];
            const uint32_t bit = cls < 32 ? (st.x >> cls) & 1u : (st.y >> (cls - 32)) & 1u;
            if (__all(!liveLane || bit)) {
                prevCls = cls;
                continue;
            }
        }
        const uint32_t ctrue = behindBits[prevCls] | aheadBits[cls];  // look assertions that hold at this offset
        prevCls = cls;

        const uint32_t fs = liveLane ? followStart[myPos] : 0;
        const uint32_t cnt = liveLane ? followStart[myPos + 1] - fs : 0;
        uint32_t totalCand;
        const uint32_t rankBase = waveExclusiveScan(cnt, lane, totalCand);
        // pass 1: filter candidates, elect per-target winners
        uint64_t passMask = 0;
        for (uint32_t k = 0; __any(k < cnt); ++k) {
            if (k < cnt) {
                const uint4 p = paths[fs + k];
                if (p.x != NF_TARGET_MATCH && (p.y & ~ctrue) == 0) {
                    const uint2 pm = posMask[p.x];
                    const uint32_t bit = cls < 32 ? (pm.x >> cls) & 1u : (pm.y >> (cls - 32)) & 1u;
                    if (bit) {
                        passMask |= uint64_t(1) << k;
                        atomicMin(&best[p.x], rankBase + k);
                    }
                }
            }
        }
        waveLdsSync();
        // pass 2: winners, in (lane, k) order
        uint64_t winMask = 0;
        for (uint64_t m = passMask; __any(m != 0);) {
            if (m) {
                const uint32_t k = uint32_t(__ffsll((long long)m)) - 1;
                m &= m - 1;
                const uint4 p = paths[fs + k];
                if (best[p.x] == rankBase + k) winMask |= uint64_t(1) << k;
            }
        }
        uint32_t totalWins;
        uint32_t slot = waveExclusiveScan(uint32_t(__popcll(winMask)), lane, totalWins);
        if (totalWins > 64) {
            overflow = true;
            break;
        }
        for (uint64_t m = winMask; __any(m != 0);) {
            if (m) {
                const uint32_t k = uint32_t(__ffsll((long long)m)) - 1;
                m &= m - 1;
                const uint4 p = paths[fs + k];
                newPos[slot] = p.x;
                newSrc[slot] = lane;
                newTagsLo[slot] = p.z;
                newTagsHi[slot] = p.w;
                best[p.x] = 0xFFFFFFFFu;
                ++slot;
            }
        }
        waveLdsSync();
        nThreads = totalWins;
        uint32_t src = lane;
        uint64_t tags = 0;
        if (lane < nThreads) {
            myPos = newPos[lane];
            src = newSrc[lane];
            tags = uint64_t(newTagsLo[lane]) | (uint64_t(newTagsHi[lane]) << 32);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (uint32_t(s) < nSlots) {
                const int32_t v = __shfl(cap[s], int(src), 64);
                cap[s] = ((tags >> s) & 1) ? int32_t(i) : v;
            }
        }
        waveLdsSync();  // newPos/newSrc are rewritten by the next byte's scatter
    }

    // acceptance at end of input: first thread (priority order) with a MATCH path whose assertions hold
    bool accept = false;
    uint64_t endTags = 0;
    if (!overflow && lane < nThreads) {
        const uint32_t ctrue = behindBits[prevCls] | aheadBits[edgeClass];
        const uint32_t fs = followStart[myPos], fe = followStart[myPos + 1];
        for (uint32_t q = fs; q < fe; ++q) {
            const uint4 p = paths[q];
            if (p.x == NF_TARGET_MATCH && (p.y & ~ctrue) == 0) {
                accept = true;
                endTags = uint64_t(p.z) | (uint64_t(p.w) << 32);
                break;
            }
        }
    }
    const uint64_t acc = __ballot(accept);
    const bool matched = acc != 0;
    const uint32_t winner = matched ? uint32_t(__ffsll((long long)acc)) - 1 : 0;
    int32_t* out = caps + size_t(line) * 2 * nGroupsOut;
    if (lane == winner) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (uint32_t(s) < 2 * nGroupsOut) {
                int32_t v = -1;
                if (matched && uint32_t(s) < nSlots) v = ((endTags >> s) & 1) ? int32_t(L) : cap[s];
                out[s] = v;
            }
        }
        status[line] = overflow ? LC_OVERFLOW : (matched ? LC_MATCH : LC_NOMATCH);
    }
    for (uint32_t s = NS + lane; s < 2 * nGroupsOut; s += 64) out[s] = -1;
}

