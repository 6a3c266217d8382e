// nfa_wide_kernel.hpp -- second chance for lines the NFA kernel gave up on (included by gpu_runtime.hip only).
//
// nfa_match_kernel keeps one Pike-VM thread per lane: a line that needs more than 64 live threads at some byte is
// reported LC_OVERFLOW.  On BASELINE configs[2] that happens to 0.04 % of the values (CISCOFW305011 on an IPv6 address
// peaks at 71 threads).  This kernel decides those: same algorithm, same tables, but every lane holds TWO threads
// (thread T lives in lane T & 63, slot T >> 6), so a line may have 128 live threads.  It is launched behind every
// launch of the NFA kernel on the same stream and costs a launch of empty workgroups unless that kernel raised its
// overflow flag (launch sequence number, atomicMax -- the same protocol as the TDFA kernel's long-line flag).
//
// One line per 64-lane workgroup; the tables are read in place from HBM/L2 (no LDS staging, no start-byte skip; the steady-state
// shortcuts of nfa_match_kernel since round 4); LDS holds the election marks and the hand-off arrays only.  Patterns with
// atomic groups are not handled here (their ordered commit pass is serial in lane 0 and has its own 64-entry work
// arrays): their lines stay LC_OVERFLOW.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/lc_regex_gpu.h"
#include "device_tables.h"
#include "nfa_kernel.hpp"

constexpr int kNfaWideThreads = 128;

// exclusive scan over the 128 thread slots (slot 0 of all lanes first, then slot 1)
__device__ __forceinline__ void nfaWideScan(uint32_t v0, uint32_t v1, uint32_t lane, uint32_t& e0, uint32_t& e1, uint32_t& total) {
    uint32_t t0, t1;
    e0 = waveExclusiveScan(v0, lane, t0);
    e1 = waveExclusiveScan(v1, lane, t1) + t0;
    total = t0 + t1;
}

// STAGE (round 5): the program is staged into LDS (blobBytes of it, in front of the scratch) -- a value comes here to be walked to its
// end at the latency of a byte step, and from L2 the step's dependent table reads are most of that latency (nfa_kernel.hpp BLOCK).  The
// launcher stages when program + scratch fit the CU's LDS and the batch is small.
template <int NS, bool STAGE = false>
__global__ __launch_bounds__(64) void nfa_wide_kernel(const uint8_t* __restrict__ data, const uint32_t* __restrict__ off,
                                                      const uint32_t* __restrict__ len, uint32_t sepBytes, uint32_t nLines,
                                                      const uint32_t* __restrict__ nLinesPtr,
                                                      const uint32_t* __restrict__ order,
                                                      const uint32_t* __restrict__ resume,
                                                      const uint32_t* __restrict__ blob, uint32_t blobBytes, uint32_t nGroupsOut,
                                                      int32_t* __restrict__ caps, uint8_t* __restrict__ status,
                                                      uint32_t* __restrict__ overflowFlag, uint32_t launchSeq, uint32_t first,
                                                      uint32_t* __restrict__ wideNote) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // first != 0 (round 5, "wide first"): this kernel is the FIRST chance of the launch -- every line is walked, whatever its status
    // says, and a line that needs more than 128 threads raises the overflow flag itself (the decide kernels behind look at it).  A
    // host that knows the pattern overflows 64 threads on its data (the Grok matcher: by the entry's history) saves the narrow kernel's
    // walk up to the overflow and the restart from byte 0 -- on configs[2] the longest entry's 1.0 ms + 1.1 ms became 1.1 ms.
    // wideNote (optional): set to 1 when some line of this launch did need more than 64 threads (the history stays honest).
    if (!first && __atomic_load_n(overflowFlag, __ATOMIC_RELAXED) < launchSeq) return;  // the NFA kernel decided every line
    if (nLinesPtr) {
        const uint32_t dyn = *nLinesPtr;
        nLines = dyn < nLines ? dyn : nLines;
    }
    const uint32_t slot = blockIdx.x, lane = threadIdx.x;
    if (slot >= nLines) return;
    const uint32_t line = order ? order[slot] : slot;
    if (!first && status[line] != LC_OVERFLOW) return;

    const uint8_t* tbl = reinterpret_cast<const uint8_t*>(blob);
    uint32_t scratchBase = 0;
    if constexpr (STAGE) {  // (only the workgroups that have a value to decide get here)
        const uint4* src = reinterpret_cast<const uint4*>(blob);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        const uint32_t nQuads = blobBytes / 16;
        uint32_t i = lane;
        for (; i + 7 * 64 < nQuads; i += 8 * 64) {
            uint4 q[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) q[k] = src[i + uint32_t(k) * 64];
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[i + uint32_t(k) * 64] = q[k];
        }
        for (; i < nQuads; i += 64) dst[i] = src[i];
        waveLdsSync();
        tbl = smem;
        scratchBase = blobBytes;
    }
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(tbl);
    const uint32_t nPos = hdr[NF_NPOS];
    const uint32_t nSlots = hdr[NF_NSLOTS];
    const uint8_t* classMap = tbl + hdr[NF_OFF_CLASSMAP];
    const uint32_t* behindBits = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_BEHIND]);
    const uint32_t* aheadBits = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_AHEAD]);
    const uint32_t edgeClass = hdr[NF_NCLASSES];
    constexpr int TW = NS > 128 ? 10 : (NS > 64 ? 4 : 2);
    NfaTables tb;
    tb.followStart = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_FOLLOWSTART]);
    tb.paths = reinterpret_cast<const uint2*>(tbl + hdr[NF_OFF_PATHS]);
    tb.aux = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_AUX]);
    tb.auxShift = TW == 10 ? 4 : (TW == 4 ? 3 : 2);
    tb.posMask = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_POSMASK]);
    tb.maskShift = hdr[NF_MASK_WORDS] == 4 ? 2 : 1;
    // follow lists by byte class (device_tables.h NF_OFF_CSTART; nfa_kernel.hpp): always read from global memory
    const uint32_t* cstart = hdr[NF_OFF_CSTART] ? reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(blob) + hdr[NF_OFF_CSTART]) : nullptr;
    const uint32_t* cpaths = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(blob) + hdr[NF_OFF_CPATHS]);
    const uint32_t nClasses = hdr[NF_NCLASSES];

    // LDS: best[nPos] then newPos / newSrc / newAux for 128 threads
    uint32_t* best = reinterpret_cast<uint32_t*>(smem + scratchBase);
    uint32_t* newPos = best + ((nPos + 3) & ~3u);
    uint32_t* newSrc = newPos + kNfaWideThreads;
    uint32_t* newAux = newSrc + kNfaWideThreads;
    uint32_t* ownerMark = newAux + kNfaWideThreads;  // 64 words: candidate -> owning thread, per election round
    for (uint32_t i = lane; i < nPos; i += 64) best[i] = 0xFFFFFFFFu;
    waveLdsSync();

    const uint32_t o = __builtin_amdgcn_readfirstlane(off[line]);
    const uint32_t L = __builtin_amdgcn_readfirstlane(len ? len[line] : off[line + 1] - o - sepBytes);
    uint32_t from = 0;
    uint32_t prevCls = edgeClass;
    // thread T = lane + 64 * k: position pos[k], capture offsets cap[k][]
    uint32_t pos[2] = {nPos, nPos};  // lane 0, slot 0: the start pseudo-position
    int32_t cap[2][NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) cap[0][s] = cap[1][s] = -1;
    if (resume) {
        from = __builtin_amdgcn_readfirstlane(resume[line]);
        from = from < L ? from : L;
        if (from) {
            pos[0] = 0;
            prevCls = __builtin_amdgcn_readfirstlane(classMap[data[size_t(o) + from - 1]]);
        }
    }
    uint32_t nThreads = 1;
    bool overflow = false, sawWide = false;
    // Round 4: a value lands here because ONE stretch of it (an IPv6 address) needs more than 64 threads; the rest of it is the same
    // walk as in nfa_match_kernel and takes the same shortcuts -- bytes come from a 256-byte chunk held one dword per lane, a thread
    // list that is the suffix thread alone ends the walk, and a steady byte (every live thread on a tag-free self loop) starts a scan
    // of the chunk for the end of its run.  (Before: one dependent global load per byte, 3.6 us a byte, 15 ms for a 4 KiB value.)
    const bool hasSuffix = hdr[NF_SUFFIX] != 0;
    const uint32_t* stable = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_STABLE]);
    const uint32_t maskShift = tb.maskShift;
    const NfaQuasi quasi = nfaQuasiOf(tbl, hdr);
    const uintptr_t addr = reinterpret_cast<uintptr_t>(data) + o;
    const uint32_t head = uint32_t(addr & 3);
    const uint32_t* words = reinterpret_cast<const uint32_t*>(addr - head);
    const uint32_t nWords = L ? (head + L + 3) / 4 : 0;
    uint32_t curWord;
    {
        const uint32_t w = (((head + from) >> 8) << 6) + lane;
        curWord = (w < nWords) ? words[w] : 0;
    }

    for (uint32_t i = from; i < L && nThreads && !overflow; ++i) {
        if (hasSuffix && nThreads == 1 && __builtin_amdgcn_readfirstlane(pos[0]) == nPos - 1) break;  // decided (nfa_kernel.hpp)
        const uint32_t idx = head + i;
        if (i != from && (idx & 255u) == 0) {
            const uint32_t w = (idx >> 2) + lane;
            curWord = (w < nWords) ? words[w] : 0;
        }
        const uint32_t wsel = __builtin_amdgcn_readlane(curWord, (idx >> 2) & 63u);
        const uint32_t cls = __builtin_amdgcn_readfirstlane(classMap[(wsel >> ((idx & 3u) * 8)) & 0xFFu]);
        const uint32_t cw = cls >> 5, cb = cls & 31u;
        if (nThreads <= 64) {  // steady byte: nothing moves (nfa_kernel.hpp); with few threads, look for the end of the run
            uint32_t clsNext = 0xFFFFFFFFu;
            if (quasi.idx && i + 1 < L && ((idx + 1) >> 8) == (idx >> 8)) {
                const uint32_t wn = __builtin_amdgcn_readlane(curWord, ((idx + 1) >> 2) & 63u);
                clsNext = __builtin_amdgcn_readfirstlane(classMap[(wn >> (((idx + 1) & 3u) * 8)) & 0xFFu]);
            }
            const bool bit = lane < nThreads && nfaQuiet(stable, maskShift, quasi, pos[0], cls, clsNext);
            if (__all(lane >= nThreads || bit)) {
                prevCls = cls;
                if (nThreads <= kNfaSteadyScanThreads) {
                    const uint32_t chunkBase = idx & ~255u, end = head + L;
                    uint32_t firstHit = 4;
                    const uint32_t nextWord = __shfl_down(curWord, 1, 64);
                    uint32_t dNext = (lane < 63 && chunkBase + lane * 4 + 4 < end) ? uint32_t(classMap[nextWord & 0xFFu]) : 0xFFFFFFFFu;
#pragma unroll
                    for (int j = 3; j >= 0; --j) {
                        const uint32_t bi = chunkBase + lane * 4 + uint32_t(j);
                        const uint32_t c = classMap[(curWord >> (8 * j)) & 0xFFu];
                        bool steadyAll = true;
                        for (uint32_t t = 0; t < nThreads; ++t)
                            steadyAll = steadyAll && nfaQuiet(stable, maskShift, quasi, __builtin_amdgcn_readlane(pos[0], t), c,
                                                              bi + 1 < end ? dNext : 0xFFFFFFFFu);
                        if (bi > idx && bi < end && !steadyAll) firstHit = uint32_t(j);
                        dNext = c;
                    }
                    const uint64_t hit = __ballot(firstHit < 4);
                    uint32_t stop = chunkBase + 256 < end ? chunkBase + 256 : end;
                    if (hit) {
                        const int l = __ffsll((long long)hit) - 1;
                        stop = chunkBase + uint32_t(l) * 4 + uint32_t(__shfl(int(firstHit), l, 64));
                    }
                    stop = __builtin_amdgcn_readfirstlane(stop);
                    if (stop > idx + 1) {
                        const uint32_t w = __builtin_amdgcn_readlane(curWord, ((stop - 1) >> 2) & 63u);
                        prevCls = __builtin_amdgcn_readfirstlane(classMap[(w >> (((stop - 1) & 3u) * 8)) & 0xFFu]);
                        i = stop - head - 1;
                    }
                }
                continue;
            }
        }
        const uint32_t ctrue = __builtin_amdgcn_readfirstlane(behindBits[prevCls] | aheadBits[cls]);
        prevCls = cls;
        const bool live0 = lane < nThreads, live1 = lane + 64 < nThreads;
        uint32_t fs0 = 0, fs1 = 0, cnt0 = 0, cnt1 = 0;
        if (live0) {
            const uint32_t* r0 = cstart ? cstart + (pos[0] * nClasses + cls) : tb.followStart + pos[0];
            fs0 = r0[0];
            cnt0 = r0[1] - fs0;
        }
        if (live1) {
            const uint32_t* r1 = cstart ? cstart + (pos[1] * nClasses + cls) : tb.followStart + pos[1];
            fs1 = r1[0];
            cnt1 = r1[1] - fs1;
        }
        uint32_t rank0, rank1, totalCand;
        nfaWideScan(cnt0, cnt1, lane, rank0, rank1, totalCand);
        totalCand = __builtin_amdgcn_readfirstlane(totalCand);
        uint32_t totalWins = 0;
        for (uint32_t r0 = 0; r0 < totalCand && !overflow; r0 += 64) {
            const uint32_t cand = r0 + lane;  // one candidate (thread, path) per lane, in priority order
            // its owner: the nearest mark at or below it (nfa_kernel.hpp, round 5) -- threads 0..127 leave their marks, slot 0 first
            ownerMark[lane] = 0xFFFFFFFFu;
            waveLdsSync();
            if (live0 && cnt0) {
                const uint32_t lo = rank0 > r0 ? rank0 : r0;
                if (lo < rank0 + cnt0 && lo < r0 + 64) ownerMark[lo - r0] = lane;
            }
            if (live1 && cnt1) {
                const uint32_t lo = rank1 > r0 ? rank1 : r0;
                if (lo < rank1 + cnt1 && lo < r0 + 64) ownerMark[lo - r0] = lane + 64;
            }
            waveLdsSync();
            uint32_t src = 0, q = 0;
            {
                const uint32_t m = ownerMark[lane];
                const uint64_t marks = __ballot(m != 0xFFFFFFFFu) & ((uint64_t(2) << lane) - 1);
                if (cand < totalCand) src = uint32_t(__shfl(int(m), 63 - __clzll((long long)marks), 64));
                const int l = int(src & 63u);
                const uint32_t tbA = uint32_t(__shfl(int(rank0), l, 64)), tbB = uint32_t(__shfl(int(rank1), l, 64));
                const uint32_t tfA = uint32_t(__shfl(int(fs0), l, 64)), tfB = uint32_t(__shfl(int(fs1), l, 64));
                q = (src < 64 ? tfA : tfB) + (cand - (src < 64 ? tbA : tbB));
            }
            bool pass = false;
            uint4 p{0, 0, 0, 0};
            if (cand < totalCand) {
                if (cstart) {
                    p = nfaPath(tb, cpaths[q]);
                    pass = (p.y & ~ctrue) == 0;
                } else {
                    p = nfaPath(tb, q);
                    if (p.x != NF_TARGET_MATCH && (p.y & ~ctrue) == 0) pass = nfaMaskBit(tb.posMask, tb.maskShift, p.x, cw, cb);
                }
                if (pass) atomicMin(&best[p.x], cand);
            }
            waveLdsSync();
            const bool win = pass && best[p.x] == cand;
            const uint64_t wins = __ballot(win);
            const uint32_t nWins = uint32_t(__popcll(wins));
            if (totalWins + nWins > uint32_t(kNfaWideThreads)) {
                overflow = true;
                break;
            }
            if (win) {
                const uint32_t w = totalWins + uint32_t(__popcll(wins & ((uint64_t(1) << lane) - 1)));
                newPos[w] = p.x;
                newSrc[w] = src;
                newAux[w] = p.z;
            }
            totalWins += nWins;
        }
        if (overflow) break;
        waveLdsSync();
        for (uint32_t w = lane; w < totalWins; w += 64) best[newPos[w]] = 0xFFFFFFFFu;  // clear the election marks
        waveLdsSync();
        const uint32_t prevThreads = nThreads;
        nThreads = totalWins;
        sawWide = sawWide || totalWins > 64;
        if (hdr[NF_SUFFIX]) {  // nothing ranked below a thread on the wrapper's suffix position can win (nfa_kernel.hpp)
            const uint64_t s0 = __ballot(lane < nThreads && newPos[lane] == nPos - 1);
            const uint64_t s1 = __ballot(lane + 64 < nThreads && newPos[(lane + 64) & 127u] == nPos - 1);
            if (s0) nThreads = uint32_t(__ffsll((long long)s0));
            else if (s1) nThreads = 64u + uint32_t(__ffsll((long long)s1));
        }
        // every lane takes over threads `lane` and `lane + 64` of the new list; captures come from the source thread
        int srcLane[2] = {int(lane), int(lane)};
        bool srcHi[2] = {false, false};
        uint32_t tags[2][TW] = {};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t T = lane + 64u * uint32_t(k);
            pos[k] = 0;
            if (T < nThreads) {
                pos[k] = newPos[T];
                const uint32_t src = newSrc[T];
                srcLane[k] = int(src & 63u);
                srcHi[k] = src >= 64;
                const uint32_t* a = tb.aux + (newAux[T] << tb.auxShift) + 1;
#pragma unroll
                for (int j = 0; j < TW; ++j) tags[k][j] = a[j];
            }
        }
        if (prevThreads <= 64 && nThreads <= 64) {
            // (round 5) the list fits one slot per lane before and after the step -- what a value does on all but the few bytes that
            // brought it here: every source is a slot-0 thread and no slot-1 thread is written; a quarter of the shuffles
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (uint32_t(s) < nSlots) {
                    const int32_t a0 = __shfl(cap[0][s], srcLane[0], 64);
                    cap[0][s] = ((tags[0][s >> 5] >> (s & 31)) & 1u) ? int32_t(i) : a0;
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) {  // slot by slot: all four reads of the old values happen before the two writes
                if (uint32_t(s) < nSlots) {
                    const int32_t a0 = __shfl(cap[0][s], srcLane[0], 64), a1 = __shfl(cap[1][s], srcLane[0], 64);
                    const int32_t b0 = __shfl(cap[0][s], srcLane[1], 64), b1 = __shfl(cap[1][s], srcLane[1], 64);
                    cap[0][s] = ((tags[0][s >> 5] >> (s & 31)) & 1u) ? int32_t(i) : (srcHi[0] ? a1 : a0);
                    cap[1][s] = ((tags[1][s >> 5] >> (s & 31)) & 1u) ? int32_t(i) : (srcHi[1] ? b1 : b0);
                }
            }
        }
        waveLdsSync();
    }

    // acceptance at end of input: first thread in priority order with a MATCH path whose assertions hold
    bool acc[2] = {false, false};
    uint32_t endAux[2] = {0, 0};
    if (!overflow) {
        const uint32_t ctrue = behindBits[prevCls] | aheadBits[edgeClass];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (lane + 64u * uint32_t(k) >= nThreads) continue;
            const uint32_t fs = tb.followStart[pos[k]], fe = tb.followStart[pos[k] + 1];
            for (uint32_t q = fs; q < fe; ++q) {
                const uint4 p = nfaPath(tb, q);
                if (p.x == NF_TARGET_MATCH && (p.y & ~ctrue) == 0) {
                    acc[k] = true;
                    endAux[k] = p.z;
                    break;
                }
            }
        }
    }
    const uint64_t a0 = __ballot(acc[0]), a1 = __ballot(acc[1]);
    const bool matched = (a0 | a1) != 0;
    const int wk = a0 ? 0 : 1;  // slot 0 threads outrank slot 1 threads
    const uint32_t winner = matched ? uint32_t(__ffsll((long long)(a0 ? a0 : a1))) - 1 : 0;
    int32_t* out = caps + size_t(line) * 2 * nGroupsOut;
    if (lane == winner) {
        uint32_t endTags[TW];
        const uint32_t ea = wk == 0 ? endAux[0] : endAux[1];
#pragma unroll
        for (int j = 0; j < TW; ++j) endTags[j] = tb.aux[(ea << tb.auxShift) + 1 + j];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (uint32_t(s) < 2 * nGroupsOut) {
                int32_t v = -1;
                if (matched && uint32_t(s) < nSlots)
                    v = ((endTags[s >> 5] >> (s & 31)) & 1u) ? int32_t(L) : (wk == 0 ? cap[0][s] : cap[1][s]);
                out[s] = v;
            }
        }
        status[line] = overflow ? LC_OVERFLOW : (matched ? LC_MATCH : LC_NOMATCH);
        if (first && overflow) atomicMax(overflowFlag, launchSeq);
        if (wideNote && (sawWide || overflow || !first)) __atomic_store_n(wideNote, 1u, __ATOMIC_RELAXED);  // (second chance: the narrow kernel overflowed on this line)
    }
    for (uint32_t s = NS + lane; s < 2 * nGroupsOut; s += 64) out[s] = -1;
}
