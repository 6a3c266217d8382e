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
constexpr uint32_t kNfaSteadyScanThreads = 6;  // live threads up to which a steady byte triggers the scan for the end of its run

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

// OR over the 64 lanes of a wavefront, wave-uniform result: six DPP steps (two quad permutes, two row rotations, the two row
// broadcasts of GFX9) and one lane read -- no LDS round trip
template <int CTRL>
__device__ __forceinline__ uint32_t waveOrStep(uint32_t v) {
    return v | uint32_t(__builtin_amdgcn_update_dpp(int(v), int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ uint32_t waveOr(uint32_t v) {
    v = waveOrStep<0xB1>(v);   // quad_perm:[1,0,3,2]
    v = waveOrStep<0x4E>(v);   // quad_perm:[2,3,0,1]
    v = waveOrStep<0x124>(v);  // row_ror:4
    v = waveOrStep<0x128>(v);  // row_ror:8
    v = waveOrStep<0x142>(v);  // row_bcast:15
    v = waveOrStep<0x143>(v);  // row_bcast:31
    return uint32_t(__builtin_amdgcn_readlane(int(v), 63));
}

// ---- capture transfer of one step (round 5).  New thread j takes over the offsets of its source thread and stamps the slots its path
// tags.  Done slot by slot with one lane permute each, that was most of a byte step's instructions (CISCOFW313005, 40 slots: 328 VALU
// and 45 LDS-pipe instructions per byte, profiles/round5_nfa_step_counters.txt).  But the threads alive at one offset descend from
// common paths: nearly all of their offsets are the SAME in every thread -- the fields already behind them -- and only the slots of
// the field being decided differ.  So the wave keeps, per slot, whether all 64 lanes hold one value (uniform): a uniform slot needs no
// permute -- nothing to do when no new thread stamps it, one move when all do; it turns "diverged" when only some stamp it, and a
// diverged slot is permuted as before and turns uniform again as soon as the live threads agree.  `div` is wave-uniform (SGPRs).
template <int NS>
__device__ __forceinline__ void nfaTransferCaptures(int32_t (&cap)[NS], uint64_t& div, uint32_t src, const uint32_t (&tags)[2], uint32_t lane,
                                                    uint32_t nThreads, int32_t offset) {
    static_assert(NS <= 64, "two tag words");
    const bool live = lane < nThreads;
    const uint64_t orT = uint64_t(waveOr(live ? tags[0] : 0u)) | (NS > 32 ? uint64_t(waveOr(live ? tags[1] : 0u)) << 32 : 0ull);
    const uint64_t andT = ~(uint64_t(waveOr(live ? ~tags[0] : 0u)) | (NS > 32 ? uint64_t(waveOr(live ? ~tags[1] : 0u)) << 32 : 0ull));
    const uint64_t act = orT | div;
#pragma unroll
    for (int g = 0; g < (NS + 7) / 8; ++g) {
        if (((act >> (8 * g)) & 0xFFull) == 0) continue;  // (wave-uniform: a scalar branch over eight slots)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int s = 8 * g + k;
            if (s >= NS) break;
            const uint64_t bit = uint64_t(1) << s;
            if (!(act & bit)) continue;
            const bool tagged = (tags[s >> 5] >> (s & 31)) & 1u;
            if (div & bit) {
                const int32_t v = __shfl(cap[s], int(src), 64);
                cap[s] = tagged ? offset : v;
                const int32_t f = __builtin_amdgcn_readfirstlane(cap[s]);  // (lane 0 is alive whenever anything is)
                if (__all(!live || cap[s] == f)) {
                    cap[s] = f;
                    div &= ~bit;
                }
            } else if (andT & bit) {
                cap[s] = offset;  // every new thread stamps it: every lane takes it, the slot stays uniform
            } else {
                if (tagged) cap[s] = offset;  // (the old value is the same in every lane: nothing to fetch)
                div |= bit;
            }
        }
    }
}

// ---------------------------------------------------------------- atomic groups / possessive quantifiers
// A thread of a pattern with atomic groups also carries its unsettled atomic-segment memberships (its LINEAGE), and a
// step has to look at ALL epsilon paths in priority order, viable or not: a path that leaves a group commits it and
// closes the segment for everything of lower priority, except continuations of that very exit (tdfa.cpp commitAtomic
// states the rules; tests/helpers/nfa_atomic_interp.py is this routine in Python).  That pass is inherently ordered, so
// lane 0 runs it serially over LDS while the other lanes wait; it is only entered on steps where some live thread has a
// lineage or sits on a position whose paths enter/leave a group -- a few bytes around each number / quoted string of a
// log line -- every other step takes the vector path below.
constexpr int kNfaLineage = 6;       // memberships a thread can carry between steps (more: LC_OVERFLOW)
constexpr int kNfaLineageWork = 10;  // ... while a step is being worked out
constexpr uint32_t kNfaExited = 0x80000000u;
// lineage entry: bit 31 = has left the group, bits 30..23 = group instance, bits 22..0 = segment id
__device__ __forceinline__ uint32_t nfaLinKey(uint32_t e) { return e & 0x7FFFFFFFu; }
__device__ __forceinline__ uint32_t nfaLinGroup(uint32_t e) { return (e >> 23) & 0xFFu; }

// per-wave LDS scratch of the atomic path, in words
constexpr uint32_t kNfaAtomicScratchWords = 64 /*tPos*/ + 64 /*tNlin*/ + 64 * kNfaLineage /*tLin*/ + 64 /*newNlin*/ +
                                            64 * kNfaLineageWork /*newLin*/ + 64 /*closedKey*/ + 64 /*closedBy*/;
static_assert(kNfaAtomicScratchWords == 1344, "keep regex_handle.hpp lcNfaLdsBytes in step");

// A path record is 8 bytes: x = target (16 bits, 0xFFFF = MATCH) | index of its (cond, tags) entry in the aux table
// (16 bits, 0 = no assertion and no tag), y = (first event << 8) | event count (patterns with atomic groups).  Unpacked
// here into x = target, y = cond bits, z = aux index (the tags are fetched from the aux entry by the lane that ends up
// owning the new thread: 2 words, or 4 for patterns with more than 64 capture slots).
struct NfaTables {
    const uint32_t* followStart;
    const uint2* paths;
    const uint32_t* aux;      // auxWords per entry: cond, tag words
    uint32_t auxShift;        // log2(auxWords)
    const uint32_t* posMask;  // maskWords per position
    uint32_t maskShift;       // log2(maskWords)
};
__device__ __forceinline__ uint4 nfaPath(const NfaTables& tb, uint32_t q) {
    const uint32_t x = tb.paths[q].x;
    uint4 p{x & 0xFFFFu, 0u, x >> 16, 0u};
    if (p.x == 0xFFFFu) p.x = NF_TARGET_MATCH;
    if (p.z) p.y = tb.aux[p.z << tb.auxShift];
    return p;
}
// bit `cls` of the class mask of position p (cw = cls >> 5, cb = cls & 31: wave-uniform, computed once per byte)
__device__ __forceinline__ bool nfaMaskBit(const uint32_t* masks, uint32_t maskShift, uint32_t p, uint32_t cw, uint32_t cb) {
    return (masks[(p << maskShift) + cw] >> cb) & 1u;
}

// doomed spawns (device_tables.h NF_OFF_QUASI): on class c followed by class d the thread on position p only repeats itself
struct NfaQuasi {
    const uint16_t* idx = nullptr;  // nullptr: the pattern has no such rows
    const uint32_t* rows = nullptr;
    uint32_t nCls = 0;
};
__device__ __forceinline__ NfaQuasi nfaQuasiOf(const uint8_t* tbl, const uint32_t* hdr) {
    NfaQuasi q;
    const uint32_t at = hdr[NF_OFF_QUASI];
    if (at) {
        q.idx = reinterpret_cast<const uint16_t*>(tbl + at + 8);
        q.rows = reinterpret_cast<const uint32_t*>(tbl + reinterpret_cast<const uint32_t*>(tbl + at)[1]);
        q.nCls = hdr[NF_NCLASSES];
    }
    return q;
}
// steady on class c: the stable mask says so, or (d = class of the NEXT byte, 0xFFFFFFFF = none / unknown) every thread the byte
// would spawn is gone again behind the next one
__device__ __forceinline__ bool nfaQuiet(const uint32_t* stable, uint32_t maskShift, const NfaQuasi& q, uint32_t p, uint32_t c, uint32_t d) {
    if (nfaMaskBit(stable, maskShift, p, c >> 5, c & 31u)) return true;
    if (!q.idx || d == 0xFFFFFFFFu) return false;
    const uint32_t r = q.idx[p];
    if (!r) return false;
    const uint32_t* row = q.rows + (((r - 1) * q.nCls + c) << maskShift);
    return (row[d >> 5] >> (d & 31u)) & 1u;
}

struct NfaAtomicCtx {
    NfaTables tb;
    const uint32_t* events;
    uint32_t *tPos, *tNlin, *tLin, *newNlin, *newLin, *closedKey, *closedBy;
    uint32_t *newPos, *newSrc, *newAux;
    uint32_t* best;  // per-position marker array of the wave (all 0xFFFFFFFF between steps)
};

// exit visit through which path events [from, n) leave group g (0: they do not, or an assertion stops them first)
__device__ inline uint32_t nfaExitVisitFor(const uint32_t* ev, uint32_t from, uint32_t n, uint32_t g, uint32_t ctrue) {
    int depth = 0;
    for (uint32_t i = from; i < n; ++i) {
        const int code = int(int16_t(ev[i] & 0xFFFFu));
        if (code >= 20000) {
            if (!((ctrue >> (code - 20000)) & 1u)) return 0;
        } else if (code == int(g) + 1) {
            ++depth;
        } else if (code == -(int(g) + 1)) {
            if (depth == 0) return ev[i] >> 16;
            --depth;
        }
    }
    return 0;
}

// Lane 0 only.  One commit pass over the threads in c.tPos/tNlin/tLin; `final` = end of input (targets are MATCH paths,
// survivors are not merged).  Returns the number of survivors written to new*, or 0xFFFFFFFF on overflow.
__device__ inline uint32_t nfaAtomicStep(const NfaAtomicCtx& c, uint32_t nThreads, uint32_t cls, uint32_t ctrue,
                                         uint32_t step, bool final) {
    uint32_t nClosed = 0, nKept = 0;
    auto findClosed = [&](uint32_t key) -> int {
        for (uint32_t q = 0; q < nClosed; ++q)
            if (c.closedKey[q] == key) return int(q);
        return -1;
    };
    for (uint32_t t = 0; t < nThreads; ++t) {
        const uint32_t fs = c.tb.followStart[c.tPos[t]], fe = c.tb.followStart[c.tPos[t] + 1];
        const uint32_t nlin = c.tNlin[t];
        for (uint32_t q = fs; q < fe; ++q) {
            const uint4 p = nfaPath(c.tb, q);
            const uint32_t pe = c.tb.paths[q].y;
            const uint32_t* ev = c.events + (pe >> 8);
            const uint32_t nev = pe & 0xFFu;
            bool targetOk;
            if (final) {
                targetOk = p.x == NF_TARGET_MATCH;
            } else if (p.x == NF_TARGET_MATCH) {
                targetOk = false;
            } else {
                targetOk = nfaMaskBit(c.tb.posMask, c.tb.maskShift, p.x, cls >> 5, cls & 31u);
            }
            bool dead = false;
            for (uint32_t j = 0; j < nlin && !dead; ++j) {
                const uint32_t e = c.tLin[t * kNfaLineage + j];
                const int cl = findClosed(nfaLinKey(e));
                if (cl < 0) continue;
                const uint32_t by = c.closedBy[cl];
                if ((e & kNfaExited) || (by >> 16) != t || nfaExitVisitFor(ev, 0, nev, nfaLinGroup(e), ctrue) != (by & 0xFFFFu))
                    dead = true;
            }
            if (dead) continue;
            uint32_t work[kNfaLineageWork];
            uint32_t wn = nlin;
            for (uint32_t j = 0; j < nlin; ++j) work[j] = c.tLin[t * kNfaLineage + j];
            bool ok = true;
            for (uint32_t i = 0; i < nev && !dead; ++i) {
                const int code = int(int16_t(ev[i] & 0xFFFFu));
                if (code >= 20000) {
                    if (!((ctrue >> (code - 20000)) & 1u)) {
                        ok = false;
                        break;
                    }
                } else if (code > 0) {
                    const uint32_t g = uint32_t(code - 1);
                    const uint32_t key = (g << 23) | ((((step + 1) << 6) | t) & 0x7FFFFFu);
                    const int cl = findClosed(key);
                    if (cl >= 0) {
                        const uint32_t by = c.closedBy[cl];
                        if ((by >> 16) != t || nfaExitVisitFor(ev, i + 1, nev, g, ctrue) != (by & 0xFFFFu)) {
                            dead = true;
                            break;
                        }
                    }
                    if (wn == kNfaLineageWork) return 0xFFFFFFFFu;
                    work[wn++] = key;
                } else {
                    const uint32_t g = uint32_t(-code - 1);
                    for (uint32_t j = wn; j-- > 0;)
                        if (nfaLinGroup(work[j]) == g && !(work[j] & kNfaExited)) {
                            if (findClosed(nfaLinKey(work[j])) < 0) {
                                if (nClosed == 64) return 0xFFFFFFFFu;
                                c.closedKey[nClosed] = nfaLinKey(work[j]);
                                c.closedBy[nClosed] = (t << 16) | (ev[i] >> 16);
                                ++nClosed;
                            }
                            work[j] |= kNfaExited;
                            break;
                        }
                }
            }
            if (dead || !ok || !targetOk) continue;
            // Survivors without memberships mirror each other trivially: among them the first one on a position wins
            // (the ordinary Pike rule), decided right here with the per-position marker array.
            if (!final && wn == 0) {
                if (c.best[p.x] != 0xFFFFFFFFu) continue;
                c.best[p.x] = nKept;
            }
            if (nKept == 64) {
                for (uint32_t k = 0; k < nKept; ++k)
                    if (c.newPos[k] < 0x40000000u) c.best[c.newPos[k]] = 0xFFFFFFFFu;
                return 0xFFFFFFFFu;
            }
            c.newPos[nKept] = final ? (0x40000000u + nKept) : p.x;  // final: survivors never merge
            c.newSrc[nKept] = t;
            c.newAux[nKept] = p.z;
            c.newNlin[nKept] = wn;
            for (uint32_t j = 0; j < wn; ++j) c.newLin[nKept * kNfaLineageWork + j] = work[j];
            ++nKept;
        }
    }
    // memberships nobody can act on any more: an exited entry stays only while a higher-priority survivor is inside
    for (uint32_t i = 0; i < nKept; ++i) {
        uint32_t* lin = c.newLin + i * kNfaLineageWork;
        uint32_t n = c.newNlin[i], w = 0;
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t e = lin[k];
            bool keep = !(e & kNfaExited);
            for (uint32_t j = 0; j < i && !keep; ++j) {
                const uint32_t* lj = c.newLin + j * kNfaLineageWork;
                for (uint32_t m = 0; m < c.newNlin[j]; ++m)
                    if (lj[m] == nfaLinKey(e)) keep = true;  // same key, not exited
            }
            if (keep) lin[w++] = e;
        }
        c.newNlin[i] = w;
    }
    // a survivor that mirrors a higher-priority one on the same position is redundant (tdfa.cpp `mirrors`)
    auto holds = [&](uint32_t who, uint32_t key, bool insideOnly) {
        const uint32_t* l = c.newLin + who * kNfaLineageWork;
        for (uint32_t m = 0; m < c.newNlin[who]; ++m)
            if (nfaLinKey(l[m]) == key && (!insideOnly || !(l[m] & kNfaExited))) return true;
        return false;
    };
    auto mirrors = [&](uint32_t hi, uint32_t lo) {
        if (c.newPos[hi] != c.newPos[lo]) return false;
        const uint32_t* lh = c.newLin + hi * kNfaLineageWork;
        for (uint32_t m = 0; m < c.newNlin[hi]; ++m) {
            const uint32_t key = nfaLinKey(lh[m]);
            if (holds(lo, key, false)) continue;
            for (uint32_t k = 0; k < hi; ++k)
                if (holds(k, key, true)) return false;
        }
        const uint32_t* ll = c.newLin + lo * kNfaLineageWork;
        for (uint32_t m = 0; m < c.newNlin[lo]; ++m) {
            if (ll[m] & kNfaExited) continue;
            const uint32_t key = nfaLinKey(ll[m]);
            if (holds(hi, key, true)) continue;
            for (uint32_t k = lo + 1; k < nKept; ++k)
                if (holds(k, key, false)) return false;
        }
        return true;
    };
    if (!final) {
        // Only pairs that involve a membership need the full test (the others were settled while appending); a dropped
        // survivor is parked on position kNfaDropped (it holds nothing, so it cannot influence a later test).
        constexpr uint32_t kNfaDropped = 0x7FFFFFFFu;
        for (uint32_t i = 0; i < nKept; ++i) c.best[c.newPos[i]] = 0xFFFFFFFFu;  // clear the markers set while appending
        for (bool changed = true; changed;) {
            changed = false;
            for (uint32_t i = 1; i < nKept; ++i) {
                if (c.newPos[i] == kNfaDropped) continue;
                for (uint32_t j = 0; j < i; ++j) {
                    if (c.newPos[j] != c.newPos[i] || (!c.newNlin[i] && !c.newNlin[j])) continue;
                    if (mirrors(j, i)) {
                        c.newPos[i] = kNfaDropped;
                        c.newNlin[i] = 0;
                        changed = true;
                        break;
                    }
                }
            }
        }
        uint32_t w = 0;
        for (uint32_t k = 0; k < nKept; ++k) {
            if (c.newPos[k] == kNfaDropped) continue;
            if (w != k) {
                c.newPos[w] = c.newPos[k];
                c.newSrc[w] = c.newSrc[k];
                c.newAux[w] = c.newAux[k];
                c.newNlin[w] = c.newNlin[k];
                for (uint32_t m = 0; m < c.newNlin[k]; ++m)
                    c.newLin[w * kNfaLineageWork + m] = c.newLin[k * kNfaLineageWork + m];
            }
            ++w;
        }
        nKept = w;
    }
    for (uint32_t i = 0; i < nKept; ++i)
        if (c.newNlin[i] > uint32_t(kNfaLineage)) return 0xFFFFFFFFu;
    return nKept;
}

// BLOCK: 256 (four values per workgroup share one LDS copy of the program: large batches) or 128 / 64 (round 5).  A SMALL batch -- a
// Grok entry's few hundred candidates -- waits for its longest value, i.e. for the latency of a byte step, and a step is a chain of
// dependent table reads (follow-list bounds -> path -> its conditions -> the target's class mask -> the tags): from L2 that chain is
// most of the step (measured: 300 ns a byte for CISCOFW313005, 3 219 positions, 127 KB of program).  With fewer values per workgroup
// the per-wave election marks shrink and programs up to ~145 KB fit the CU's 160 KB of LDS next to them: the launcher picks the
// largest BLOCK whose LDS need fits (gpu_runtime.hip launchNfa).
template <int NS, bool ATOMIC, bool GLOBAL, int BLOCK = kNfaBlock>
__global__ __launch_bounds__(BLOCK) void nfa_match_kernel(const uint8_t* __restrict__ data,
                                                              const uint32_t* __restrict__ off,
                                                              const uint32_t* __restrict__ len, uint32_t sepBytes,
                                                              uint32_t nLines, const uint32_t* __restrict__ nLinesPtr,
                                                              const uint32_t* __restrict__ order,
                                                              const uint32_t* __restrict__ resume,
                                                              const uint32_t* __restrict__ blob,
                                                              uint32_t blobBytes, uint32_t nGroupsOut,
                                                              int32_t* __restrict__ caps,
                                                              uint8_t* __restrict__ status,
                                                              uint32_t* __restrict__ overflowFlag, uint32_t launchSeq,
                                                              const uint32_t* __restrict__ pendingFlag) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x;
    // behind nfa_dfs_kernel (nfa_decide_kernel.hpp): only the lines it left pending are this launch's business -- usually none
    if (pendingFlag && __atomic_load_n(pendingFlag, __ATOMIC_RELAXED) < launchSeq) return;
    if (nLinesPtr) {  // line count produced on the device (split kernels) -- no host round trip between the launches
        const uint32_t dyn = *nLinesPtr;
        nLines = dyn < nLines ? dyn : nLines;
    }
    // the program: staged into LDS once per workgroup -- or, for programs that do not fit next to the scratch (Grok
    // patterns with several IPv6/hostname alternations run to 100+ KiB), read in place from HBM through L2
    const uint8_t* tbl;
    uint32_t scratchBase;
    if constexpr (GLOBAL) {
        tbl = reinterpret_cast<const uint8_t*>(blob);
        scratchBase = 0;
    } else {
        const uint4* src = reinterpret_cast<const uint4*>(blob);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        const uint32_t nQuads = blobBytes / 16;
        // (eight loads in flight per lane: a workgroup of one wavefront stages 127 KB in 16 round trips, not 124)
        uint32_t i = tid;
        for (; i + 7 * BLOCK < nQuads; i += 8 * BLOCK) {
            uint4 q[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) q[k] = src[i + uint32_t(k) * BLOCK];
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[i + uint32_t(k) * BLOCK] = q[k];
        }
        for (; i < nQuads; i += BLOCK) dst[i] = src[i];
        __syncthreads();
        tbl = smem;
        scratchBase = blobBytes;
    }
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(tbl);
    const uint32_t nPos = hdr[NF_NPOS];
    const uint32_t nSlots = hdr[NF_NSLOTS];
    const uint8_t* classMap = tbl + hdr[NF_OFF_CLASSMAP];
    const uint32_t* stable = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_STABLE]);
    const uint32_t* behindBits = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_BEHIND]);
    const uint32_t* aheadBits = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_AHEAD]);
    const uint32_t edgeClass = hdr[NF_NCLASSES];  // table index standing for start / end of input
    // tag words per aux entry (regex_handle.cpp packNfaBlob: entries of 4 / 8 / 16 words for up to 64 / 128 / 320 slots)
    constexpr int TW = NS > 128 ? 10 : (NS > 64 ? 4 : 2);
    NfaTables tb;
    tb.followStart = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_FOLLOWSTART]);
    tb.paths = reinterpret_cast<const uint2*>(tbl + hdr[NF_OFF_PATHS]);
    tb.aux = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_AUX]);
    tb.auxShift = TW == 10 ? 4 : (TW == 4 ? 3 : 2);
    tb.posMask = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_POSMASK]);
    tb.maskShift = hdr[NF_MASK_WORDS] == 4 ? 2 : 1;
    const uint32_t maskShift = tb.maskShift;
    const uint32_t* followStart = tb.followStart;
    // follow lists by byte class (device_tables.h NF_OFF_CSTART): in global memory also when the program is staged
    const uint32_t* cstart = hdr[NF_OFF_CSTART] ? reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(blob) + hdr[NF_OFF_CSTART]) : nullptr;
    const uint32_t* cpaths = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(blob) + hdr[NF_OFF_CPATHS]);
    const uint32_t nClasses = hdr[NF_NCLASSES];

    // (round 5: what is wave-uniform is SAID to be -- readfirstlane -- or the compiler, for which anything derived from threadIdx or read
    // from LDS is divergent, keeps the line's offsets, the byte position and the loop conditions in VGPRs and steers the walk with exec masks)
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // per-wave scratch: best[nPos] then 4 x 64 words (newPos, newSrc, newAux, spare)
    const uint32_t scratchWords = ((nPos + 3) & ~3u) + 256;
    uint32_t* best = reinterpret_cast<uint32_t*>(smem + scratchBase) + wave * scratchWords;
    uint32_t* newPos = best + ((nPos + 3) & ~3u);
    uint32_t* newSrc = newPos + 64;
    uint32_t* newAux = newSrc + 64;
    uint32_t* ownerMark = newAux + 64;  // (the fourth 64-word array: candidate -> owning thread, per election round)
    for (uint32_t i = lane; i < nPos; i += 64) best[i] = 0xFFFFFFFFu;
    waveLdsSync();
    // atomic path: its per-wave scratch sits behind the scratch of all waves
    NfaAtomicCtx actx{};
    const uint32_t* atomicPos = nullptr;
    const uint32_t* touchyMask = nullptr;
    constexpr uint32_t kWaves = BLOCK / 64;
    if constexpr (ATOMIC) {
        uint32_t* a = reinterpret_cast<uint32_t*>(smem + scratchBase) + kWaves * scratchWords + wave * kNfaAtomicScratchWords;
        actx.tb = tb;
        actx.events = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_EVENTS]);
        actx.tPos = a;
        actx.tNlin = a + 64;
        actx.tLin = a + 128;
        actx.newNlin = actx.tLin + 64 * kNfaLineage;
        actx.newLin = actx.newNlin + 64;
        actx.closedKey = actx.newLin + 64 * kNfaLineageWork;
        actx.closedBy = actx.closedKey + 64;
        actx.newPos = newPos;
        actx.newSrc = newSrc;
        actx.newAux = newAux;
        actx.best = best;
        atomicPos = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_ATOMICPOS]);
        touchyMask = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_TOUCHY]);
    }

    const uint32_t slot = blockIdx.x * kWaves + wave;
    if (slot >= nLines) return;  // wave-uniform (the block never synchronises again)
    const uint32_t line = __builtin_amdgcn_readfirstlane(order ? order[slot] : slot);
    if (pendingFlag && status[line] != 4 /* LC_PENDING */) return;  // settled by the depth-first walk
    const uint32_t o = __builtin_amdgcn_readfirstlane(off[line]);
    const uint32_t L = __builtin_amdgcn_readfirstlane(len ? len[line] : off[line + 1] - o - sepBytes);

    int32_t cap[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) cap[s] = -1;
    uint64_t capDiverged = 0;  // slots in which the lanes do not all hold one value (nfaTransferCaptures)
    uint32_t nThreads = 1;
    uint32_t myPos = nPos;  // lane 0: the start pseudo-position
    bool overflow = false;
    uint32_t lin[ATOMIC ? kNfaLineage : 1];  // this thread's unsettled atomic-segment memberships
    uint32_t nlin = 0;
    if (ATOMIC && L >= (1u << 17) - 2) overflow = true;  // segment ids are (offset << 6 | thread) in 23 bits

    const uintptr_t addr = reinterpret_cast<uintptr_t>(data) + o;
    const uint32_t head = uint32_t(addr & 3);
    const uint32_t* words = reinterpret_cast<const uint32_t*>(addr - head);
    const uint32_t nWords = L ? (head + L + 3) / 4 : 0;
    uint32_t prevCls = edgeClass;
    uint32_t from = 0;  // search patterns: resume the search here (the next match of an iterate-all-matches caller)
    if (resume) {
        from = __builtin_amdgcn_readfirstlane(resume[line]);
        from = from < L ? from : L;
        if (from) {  // one thread, on the wrapper's prefix position, which has just consumed the byte before `from`
            myPos = 0;
            prevCls = __builtin_amdgcn_readfirstlane(classMap[data[size_t(o) + from - 1]]);
        }
    }
    const bool searchSkip = hdr[NF_SEARCH] != 0;
    // the last position is the search wrapper's greedy suffix (search and anchored search, device_tables.h NF_SUFFIX)
    const bool hasSuffix = hdr[NF_SUFFIX] != 0;
    const NfaQuasi quasi = nfaQuasiOf(tbl, hdr);
    // steady classes of the search wrapper's prefix position (words 2, 3 only exist for patterns with > 64 byte classes)
    const uint32_t stable0[4] = {stable[0], stable[1], maskShift == 2 ? stable[2] : 0u, maskShift == 2 ? stable[3] : 0u};
    uint32_t curWord;
    {
        const uint32_t w = (((head + from) >> 8) << 6) + lane;  // the 256-byte chunk the first byte lives in
        curWord = (w < nWords) ? words[w] : 0;
    }

    for (uint32_t i = from; i < L && nThreads && !overflow; ++i) {
        // The thread list is the suffix thread alone: it takes every byte that is left and ends on MATCH, no other thread can appear
        // (threads only come from threads) and no capture moves -- the value is decided, whatever its length.  A log format that ends
        // 150 bytes into a 4 KiB value used to walk the other 3.9 KiB.
        if (hasSuffix && nThreads == 1 && __builtin_amdgcn_readfirstlane(myPos) == nPos - 1 &&
            (!ATOMIC || __builtin_amdgcn_readfirstlane(nlin) == 0))
            break;
        const uint32_t idx = head + i;
        if (i != from && (idx & 255u) == 0) {  // next 256-byte chunk: one coalesced dword per lane
            const uint32_t w = (idx >> 2) + lane;
            curWord = (w < nWords) ? words[w] : 0;
        }
        // Search patterns, only the wrapper's prefix thread alive (the usual state on a value this pattern does not
        // match): nothing can happen until a byte the pattern can START with, so all 64 lanes look for that byte in the
        // loaded chunk at once (4 bytes each) instead of stepping the automaton byte by byte.
        if (searchSkip && nThreads == 1 && __builtin_amdgcn_readfirstlane(myPos) == 0) {
            const uint32_t chunkBase = idx & ~255u, end = head + L;
            uint32_t firstHit = 4;
            const uint32_t nextWord = __shfl_down(curWord, 1, 64);  // (lane 63: its own word -- its last byte is never looked ahead from)
            uint32_t dNext = (lane < 63 && chunkBase + lane * 4 + 4 < end) ? uint32_t(classMap[nextWord & 0xFFu]) : 0xFFFFFFFFu;
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                const uint32_t bi = chunkBase + lane * 4 + uint32_t(j);
                const uint32_t c = classMap[(curWord >> (8 * j)) & 0xFFu];
                const uint32_t sw = c < 64 ? (c < 32 ? stable0[0] : stable0[1]) : (c < 96 ? stable0[2] : stable0[3]);
                uint32_t steady = (sw >> (c & 31u)) & 1u;
                // (a byte the format can begin with, followed by one its second position cannot take: the attempt is gone again)
                if (!steady && quasi.idx && bi >= idx && bi + 1 < end) steady = nfaQuiet(stable, maskShift, quasi, 0u, c, dNext) ? 1u : 0u;
                if (bi >= idx && bi < end && !steady) firstHit = uint32_t(j);
                dNext = c;
            }
            const uint64_t hit = __ballot(firstHit < 4);
            uint32_t stop = chunkBase + 256 < end ? chunkBase + 256 : end;
            if (hit) {
                const int l = __ffsll((long long)hit) - 1;
                stop = chunkBase + uint32_t(l) * 4 + uint32_t(__shfl(int(firstHit), l, 64));
            }
            stop = __builtin_amdgcn_readfirstlane(stop);
            if (stop > idx) {  // bytes [idx, stop) only feed the prefix loop; remember the class of the last one
                const uint32_t w = __builtin_amdgcn_readlane(curWord, ((stop - 1) >> 2) & 63u);
                prevCls = __builtin_amdgcn_readfirstlane(classMap[(w >> (((stop - 1) & 3u) * 8)) & 0xFFu]);
                i = stop - head - 1;
                continue;
            }
        }
        const uint32_t wsel = __builtin_amdgcn_readlane(curWord, (idx >> 2) & 63u);
        const int b = int((wsel >> ((idx & 3u) * 8)) & 0xFFu);
        const uint32_t cls = __builtin_amdgcn_readfirstlane(classMap[b]);
        const uint32_t cw = cls >> 5, cb = cls & 31u;
        const bool liveLane = lane < nThreads;
        // steady state: every live thread sits on a position whose only move on this byte class is its own
        // unconditional, tag-free self loop (inside a field such as [^ ]* that is almost every byte) -> the thread
        // list, its order and the captures are unchanged; skip the whole election/compaction/transfer machinery.
        {
            // (the class of the byte behind this one, for the doomed-spawn test: known when it lies in the loaded chunk)
            uint32_t clsNext = 0xFFFFFFFFu;
            if (quasi.idx && i + 1 < L && ((idx + 1) >> 8) == (idx >> 8)) {
                const uint32_t wn = __builtin_amdgcn_readlane(curWord, ((idx + 1) >> 2) & 63u);
                clsNext = __builtin_amdgcn_readfirstlane(classMap[(wn >> (((idx + 1) & 3u) * 8)) & 0xFFu]);
            }
            const bool bit = liveLane && nfaQuiet(stable, maskShift, quasi, myPos, cls, clsNext);
            if (__all(!liveLane || bit)) {
                prevCls = cls;
                // A steady byte is usually the first of a RUN of them (inside a field, inside GREEDYDATA): with a few live threads all
                // 64 lanes test the rest of the loaded 256-byte chunk at once -- 4 bytes each against every live thread's steady mask --
                // and the walk resumes at the first byte that is not steady for some thread (or at the end of the chunk).
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
                            steadyAll = steadyAll && nfaQuiet(stable, maskShift, quasi, __builtin_amdgcn_readlane(myPos, t), c,
                                                              bi + 1 < end ? dNext : 0xFFFFFFFFu);  // (the value's last byte has no next one)
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
                    if (stop > idx + 1) {  // bytes (idx, stop) are steady too; remember the class of the last one
                        const uint32_t w = __builtin_amdgcn_readlane(curWord, ((stop - 1) >> 2) & 63u);
                        prevCls = __builtin_amdgcn_readfirstlane(classMap[(w >> (((stop - 1) & 3u) * 8)) & 0xFFu]);
                        i = stop - head - 1;
                    }
                }
                continue;
            }
        }
        const uint32_t ctrue = __builtin_amdgcn_readfirstlane(behindBits[prevCls] | aheadBits[cls]);  // look assertions that hold at this offset
        prevCls = cls;

        if constexpr (ATOMIC) {
            bool touchy = false;
            if (liveLane) touchy = nlin != 0 || nfaMaskBit(touchyMask, maskShift, myPos, cw, cb);
            if (__any(touchy)) {  // ordered commit pass, lane 0 (see nfaAtomicStep)
                if (liveLane) {
                    actx.tPos[lane] = myPos;
                    actx.tNlin[lane] = nlin;
#pragma unroll
                    for (int j = 0; j < kNfaLineage; ++j) actx.tLin[lane * kNfaLineage + j] = lin[j];
                }
                waveLdsSync();
                uint32_t kept = 0;
                if (lane == 0) kept = nfaAtomicStep(actx, nThreads, cls, ctrue, i, false);
                kept = __builtin_amdgcn_readfirstlane(kept);
                waveLdsSync();
                if (kept == 0xFFFFFFFFu) {
                    overflow = true;
                    break;
                }
                nThreads = kept;
                if (hasSuffix) {  // see the vector path below; a membership could still get the suffix thread killed
                    const uint64_t suf = __ballot(lane < nThreads && newPos[lane] == nPos - 1 && actx.newNlin[lane] == 0);
                    if (suf) nThreads = uint32_t(__ffsll((long long)suf));
                }
                uint32_t src = lane;
                uint32_t tags[TW] = {};
                nlin = 0;
                if (lane < nThreads) {
                    myPos = newPos[lane];
                    src = newSrc[lane];
                    const uint32_t* a = tb.aux + (newAux[lane] << tb.auxShift) + 1;
#pragma unroll
                    for (int k = 0; k < TW; ++k) tags[k] = a[k];
                    nlin = actx.newNlin[lane];
#pragma unroll
                    for (int j = 0; j < kNfaLineage; ++j) lin[j] = actx.newLin[lane * kNfaLineageWork + j];
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (uint32_t(s) < nSlots) {
                        const int32_t v = __shfl(cap[s], int(src), 64);
                        cap[s] = ((tags[s >> 5] >> (s & 31)) & 1u) ? int32_t(i) : v;
                    }
                }
                waveLdsSync();
                continue;
            }
        }

        // (round 5) this thread's paths whose target takes THIS byte: the class list where the blob has one, else the whole follow list
        uint32_t fs = 0, cnt = 0;
        if (liveLane) {
            const uint32_t* rowStart = cstart ? cstart + (myPos * nClasses + cls) : followStart + myPos;
            fs = rowStart[0];
            cnt = rowStart[1] - fs;
        }
        uint32_t totalCand;
        const uint32_t rankBase = waveExclusiveScan(cnt, lane, totalCand);
        totalCand = __builtin_amdgcn_readfirstlane(totalCand);
        // One CANDIDATE (thread, path) per lane, 64 per round, in priority order (rank = lexicographic (thread, path index)
        // = the candidate's index).  A thread with a long follow list -- the search prefix of a pattern that can start in
        // dozens of ways -- no longer walks it serially in one lane while 60 lanes idle.
        uint32_t totalWins = 0;
        for (uint32_t r0 = 0; r0 < totalCand; r0 += 64) {
            const uint32_t cand = r0 + lane;
            // Which thread owns candidate `cand` (round 5).  The ranges [rankBase, rankBase + cnt) tile [0, totalCand) in thread order:
            // every thread whose range reaches into this round leaves its number at the first candidate of the round it owns, and a
            // candidate's owner is the nearest mark at or below it -- one LDS store, one load, a ballot and three lane reads per round.
            // (Before: a loop over all live threads with three lane reads each; with 40 threads alive and follow lists of a hundred
            // paths -- an IP address behind a lazy field -- that loop WAS the byte step: 4 us a byte on CISCOFW313005.)
            ownerMark[lane] = 0xFFFFFFFFu;
            waveLdsSync();
            if (liveLane && cnt) {
                const uint32_t lo = rankBase > r0 ? rankBase : r0;
                if (lo < rankBase + cnt && lo < r0 + 64) ownerMark[lo - r0] = lane;
            }
            waveLdsSync();
            uint32_t src = 0, q = 0;
            {
                const uint32_t m = ownerMark[lane];
                const uint64_t marks = __ballot(m != 0xFFFFFFFFu) & ((uint64_t(2) << lane) - 1);  // (lane 63: 2 << 63 = 0, minus 1 = all)
                if (cand < totalCand) src = uint32_t(__shfl(int(m), 63 - __clzll((long long)marks), 64));
                const uint32_t tb = uint32_t(__shfl(int(rankBase), int(src), 64)), tf = uint32_t(__shfl(int(fs), int(src), 64));
                q = tf + (cand - tb);
            }
            bool pass = false;
            uint4 p{0, 0, 0, 0};
            if (cand < totalCand) {
                if (cstart) {  // (listed = the target takes the byte; MATCH paths are not listed)
                    p = nfaPath(tb, cpaths[q]);
                    pass = (p.y & ~ctrue) == 0;
                } else {
                    p = nfaPath(tb, q);
                    if (p.x != NF_TARGET_MATCH && (p.y & ~ctrue) == 0) pass = nfaMaskBit(tb.posMask, maskShift, p.x, cw, cb);
                }
                if (pass) atomicMin(&best[p.x], cand);  // per target, the candidate of highest priority
            }
            waveLdsSync();
            const bool win = pass && best[p.x] == cand;
            const uint64_t wins = __ballot(win);
            const uint32_t nWins = uint32_t(__popcll(wins));
            if (totalWins + nWins > 64) {
                overflow = true;
                break;
            }
            if (win) {
                const uint32_t slot = totalWins + uint32_t(__popcll(wins & ((uint64_t(1) << lane) - 1)));
                newPos[slot] = p.x;
                newSrc[slot] = src;
                newAux[slot] = p.z;
            }
            totalWins += nWins;
        }
        if (overflow) break;
        waveLdsSync();
        if (lane < totalWins) best[newPos[lane]] = 0xFFFFFFFFu;  // clear the election marks (targets are distinct)
        waveLdsSync();
        nThreads = totalWins;
        // Search patterns: a thread on the wrapper's suffix position ((?s:.*), the last position) takes every byte and
        // ends on MATCH whatever follows, so nothing of lower priority can win any more -- above all the wrapper's lazy
        // prefix thread, which would otherwise keep starting new attempts at every byte of the rest of the line.
        if (hasSuffix) {
            const uint64_t suf = __ballot(lane < nThreads && newPos[lane] == nPos - 1);
            if (suf) nThreads = uint32_t(__ffsll((long long)suf));
        }
        uint32_t src = lane;
        uint32_t tags[TW] = {};
        if (lane < nThreads) {
            myPos = newPos[lane];
            src = newSrc[lane];
            const uint32_t* a = tb.aux + (newAux[lane] << tb.auxShift) + 1;
#pragma unroll
            for (int k = 0; k < TW; ++k) tags[k] = a[k];
        }
        if constexpr (NS <= 64 && !ATOMIC) {
            nfaTransferCaptures<NS>(cap, capDiverged, src, tags, lane, nThreads, int32_t(i));
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (uint32_t(s) < nSlots) {
                    const int32_t v = __shfl(cap[s], int(src), 64);
                    cap[s] = ((tags[s >> 5] >> (s & 31)) & 1u) ? int32_t(i) : v;
                }
            }
        }
        waveLdsSync();  // newPos/newSrc are rewritten by the next byte's scatter
    }

    // acceptance at end of input: first thread (priority order) with a MATCH path whose assertions hold
    bool accept = false;
    uint32_t endAux = 0;
    bool atomicEnd = false;
    if constexpr (ATOMIC) {
        // any membership left, or a MATCH path that crosses a group boundary: the ordered commit decides the winner
        const bool touchy = !overflow && lane < nThreads && (nlin != 0 || ((atomicPos[myPos >> 5] >> (myPos & 31)) & 1u));
        if (__any(touchy)) {
            atomicEnd = true;
            if (lane < nThreads) {
                actx.tPos[lane] = myPos;
                actx.tNlin[lane] = nlin;
#pragma unroll
                for (int j = 0; j < kNfaLineage; ++j) actx.tLin[lane * kNfaLineage + j] = lin[j];
            }
            waveLdsSync();
            uint32_t kept = 0;
            if (lane == 0) kept = nfaAtomicStep(actx, nThreads, 0, behindBits[prevCls] | aheadBits[edgeClass], L, true);
            kept = __shfl(kept, 0, 64);
            waveLdsSync();
            if (kept == 0xFFFFFFFFu) {
                overflow = true;
            } else if (kept) {  // the first survivor is the match; its source thread holds the captures
                const uint32_t win = newSrc[0];
                if (lane == win) {
                    accept = true;
                    endAux = newAux[0];
                }
            }
        }
    }
    if (!atomicEnd && !overflow && lane < nThreads) {
        const uint32_t ctrue = behindBits[prevCls] | aheadBits[edgeClass];
        const uint32_t fs = followStart[myPos], fe = followStart[myPos + 1];
        for (uint32_t q = fs; q < fe; ++q) {
            const uint4 p = nfaPath(tb, q);
            if (p.x == NF_TARGET_MATCH && (p.y & ~ctrue) == 0) {
                accept = true;
                endAux = p.z;
                break;
            }
        }
    }
    const uint64_t acc = __ballot(accept);
    const bool matched = acc != 0;
    const uint32_t winner = matched ? uint32_t(__ffsll((long long)acc)) - 1 : 0;
    int32_t* out = caps + size_t(line) * 2 * nGroupsOut;
    if (lane == winner) {
        uint32_t endTags[TW];
#pragma unroll
        for (int k = 0; k < TW; ++k) endTags[k] = tb.aux[(endAux << tb.auxShift) + 1 + k];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (uint32_t(s) < 2 * nGroupsOut) {
                int32_t v = -1;
                if (matched && uint32_t(s) < nSlots) v = ((endTags[s >> 5] >> (s & 31)) & 1u) ? int32_t(L) : cap[s];
                out[s] = v;
            }
        }
        status[line] = overflow ? LC_OVERFLOW : (matched ? LC_MATCH : LC_NOMATCH);
        // tell the second-chance launch (nfa_wide_kernel.hpp) behind this one that there is something to do
        if (overflow && overflowFlag) atomicMax(overflowFlag, launchSeq);
    }
    for (uint32_t s = NS + lane; s < 2 * nGroupsOut; s += 64) out[s] = -1;
}

