// nfa_decide_kernel.hpp -- last resort of the follow-NFA engine: decides the lines the thread-list kernels gave up on
// (included by gpu_runtime.hip only).
//
// nfa_match_kernel / nfa_wide_kernel simulate the program breadth first and hold a bounded number of live threads (64 /
// 128 per line; with atomic groups 64 threads of at most 6 memberships): a line that needs more is left LC_OVERFLOW.
// boost::regex_match (core/common/StringTools.cpp:183-211) and regexp2 (processor_grok.go:156) have no such bound -- they
// backtrack.  This kernel does what they do, on the same tables: a depth-first, priority-ordered walk of the follow NFA.
// Every transition consumes one byte, so depth d of the walk is offset from+d of the line: one frame per depth holds
// (position, next path to try).  The first path that reaches MATCH at the end of the line is the match (leftmost-first);
// its capture offsets are read off the frames.
//
// Atomic groups are a CUT, as in any backtracker: a path that leaves group instance S drops every choice made inside S --
// frames opened after S's entry frame are exhausted, the entry frame loses its alternatives that enter the same group,
// and the leaving frame keeps only the continuations of that very exit (same exit visit).  These are nfaAtomicStep's
// rules (nfa_kernel.hpp) in their natural depth-first form; tests/helpers/nfa_dfs_interp.py is this routine in Python
// and is checked on the CPU against the golden vectors, the oracle and the breadth-first interpreters.
//
// A memo nibble per (position, offset) keeps the walk linear in positions x length where a plain backtracker is
// exponential: 1 = "fails from here", 1 + j = "fails after committing the j innermost enclosing atomic groups" (the cut is
// replayed on arrival).  When the memo of a line does not fit the worker's slice of the scratch pool the walk runs
// without it under a step budget; a line that exhausts the budget is reported LC_GAVE_UP -- the counterpart of boost's
// complexity exception, which the reference counts as a parse failure (StringTools.cpp:200-205).
//
// Launch protocol: nfa_decide_plan_kernel (one workgroup) collects the LC_OVERFLOW lines of the launch into a list at the
// head of the pool and sizes the workers' slices; nfa_decide_kernel (one wavefront per worker; lane 0 walks, all lanes
// clear the memo) takes lines off that list.  Both return at once unless the NFA kernel raised its overflow flag for this
// launch sequence number, so on ordinary batches they cost two empty launches.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/lc_regex_gpu.h"
#include "device_tables.h"
#include "nfa_kernel.hpp"

constexpr int kDecideWorkers = 128;          // wavefronts of the decide launch
constexpr uint32_t kDecideHeaderBytes = 256;  // DecidePlan at the head of the pool
constexpr uint32_t kDecideFrameWords = 5;     // + closedCap words of closed records
constexpr uint32_t kDecideNodeWords = 3;
constexpr uint32_t kDecideExhausted = 0x80000000u;
constexpr uint32_t kDecideNone = 0xFFFFFFFFu;
constexpr uint64_t kDecideBudgetMemo = uint64_t(1) << 28;    // steps; with the memo the walk is bounded by paths x length anyway
constexpr uint64_t kDecideBudgetNoMemo = uint64_t(1) << 22;  // plain backtracking: give up like boost's complexity guard

struct DecidePlan {
    uint32_t count;       // residue lines in the list
    uint32_t next;        // next list entry to hand out
    uint32_t workers;     // wavefronts that take part
    uint32_t listCap;     // entries the list can hold
    uint64_t sliceBytes;  // scratch per worker
    uint64_t slicesAt;    // byte offset of slice 0 in the pool
    uint32_t maxFrames;   // longest residue line + 1
    uint32_t gaveUp;      // lines reported LC_GAVE_UP by this launch (statistics)
};
static_assert(sizeof(DecidePlan) <= kDecideHeaderBytes, "plan must fit the pool header");

struct DecideShape {  // per pattern, computed by the host (regex_handle.cpp lcDecideShape)
    uint32_t closedCap;  // closed records a frame can collect: enter + exit events of the longest path
    uint32_t maxEnter;   // enter events of the longest path
};

__host__ __device__ inline uint64_t decideFixedBytes(uint32_t nFrames, DecideShape sh) {
    const uint64_t frameWords = kDecideFrameWords + sh.closedCap;
    const uint64_t nodeWords = uint64_t(kDecideNodeWords) * sh.maxEnter;
    return (uint64_t(nFrames) * (frameWords + nodeWords) + nodeWords) * 4 + 64;
}
__host__ __device__ inline uint64_t decideMemoBytes(uint32_t nFrames, uint32_t nPos) {
    return ((uint64_t(nFrames) * nPos + 1) / 2 + 15) & ~uint64_t(15);
}

// line length as this launch sees it (resume offsets shorten the walk, not the line)
__device__ __forceinline__ void decideLineSpan(const uint32_t* off, const uint32_t* len, uint32_t sepBytes, const uint32_t* resume,
                                               uint32_t line, uint32_t& o, uint32_t& L, uint32_t& from) {
    o = off[line];
    L = len ? len[line] : off[line + 1] - o - sepBytes;
    from = 0;
    if (resume) {
        from = resume[line];
        from = from < L ? from : L;
    }
}

__global__ __launch_bounds__(256) void nfa_decide_plan_kernel(const uint32_t* __restrict__ off, const uint32_t* __restrict__ len,
                                                              uint32_t sepBytes, uint32_t nLines,
                                                              const uint32_t* __restrict__ nLinesPtr,
                                                              const uint32_t* __restrict__ order,
                                                              const uint32_t* __restrict__ resume, uint32_t nPos,
                                                              DecideShape shape, uint8_t* __restrict__ status,
                                                              const uint32_t* __restrict__ overflowFlag, uint32_t launchSeq,
                                                              uint8_t* __restrict__ pool, uint64_t poolBytes) {
    DecidePlan* plan = reinterpret_cast<DecidePlan*>(pool);
    __shared__ uint32_t sCount, sMaxFrames, sGaveUp;
    if (threadIdx.x == 0) {
        sCount = 0;
        sMaxFrames = 0;
        sGaveUp = 0;
    }
    __syncthreads();
    const bool active = __atomic_load_n(overflowFlag, __ATOMIC_RELAXED) >= launchSeq;
    if (nLinesPtr) {
        const uint32_t dyn = *nLinesPtr;
        nLines = dyn < nLines ? dyn : nLines;
    }
    // the list may take a quarter of the pool
    const uint64_t capBytes = (poolBytes - kDecideHeaderBytes) / 4;
    const uint32_t listCap = uint32_t(capBytes / 4 < nLines ? capBytes / 4 : nLines);
    uint32_t* list = reinterpret_cast<uint32_t*>(pool + kDecideHeaderBytes);
    if (active) {
        for (uint32_t slot = threadIdx.x; slot < nLines; slot += 256) {
            const uint32_t line = order ? order[slot] : slot;
            if (status[line] != LC_OVERFLOW) continue;
            const uint32_t idx = atomicAdd(&sCount, 1u);
            if (idx >= listCap) {  // more undecided lines than the pool can even list: reported, never guessed
                status[line] = LC_GAVE_UP;
                atomicAdd(&sGaveUp, 1u);
                continue;
            }
            list[idx] = line;
            uint32_t o, L, from;
            decideLineSpan(off, len, sepBytes, resume, line, o, L, from);
            atomicMax(&sMaxFrames, L - from + 1);
        }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const uint32_t count = sCount < listCap ? sCount : listCap;
    plan->count = count;
    plan->next = 0;
    plan->listCap = listCap;
    plan->maxFrames = sMaxFrames;
    plan->gaveUp = sGaveUp;
    uint64_t slicesAt = kDecideHeaderBytes + ((uint64_t(count) * 4 + 255) & ~uint64_t(255));
    plan->slicesAt = slicesAt;
    uint32_t workers = 0;
    uint64_t slice = 0;
    if (count) {
        const uint64_t avail = poolBytes - slicesAt;
        const uint64_t fixed = decideFixedBytes(sMaxFrames, shape);
        const uint64_t withMemo = fixed + decideMemoBytes(sMaxFrames, nPos);
        workers = count < uint32_t(kDecideWorkers) ? count : uint32_t(kDecideWorkers);
        if (withMemo * workers > avail) {  // fewer workers with the memo beat many without it
            const uint64_t fit = avail / withMemo;
            if (fit >= 1) workers = uint32_t(fit < workers ? fit : workers);
            else {
                const uint64_t fitPlain = avail / fixed;
                workers = uint32_t(fitPlain < workers ? fitPlain : workers);  // 0: not even one frame stack fits
            }
        }
        slice = workers ? (avail / workers) & ~uint64_t(255) : 0;
    }
    plan->workers = workers;
    plan->sliceBytes = slice;
}

// LC_ENGINE_DECIDE: hand every line of the launch to the decide kernel
__global__ __launch_bounds__(256) void nfa_decide_mark_all_kernel(uint32_t nLines, const uint32_t* __restrict__ nLinesPtr,
                                                                  const uint32_t* __restrict__ order, uint8_t* __restrict__ status,
                                                                  uint32_t* __restrict__ overflowFlag, uint32_t launchSeq) {
    if (nLinesPtr) {
        const uint32_t dyn = *nLinesPtr;
        nLines = dyn < nLines ? dyn : nLines;
    }
    const uint32_t slot = blockIdx.x * 256 + threadIdx.x;
    if (slot == 0) atomicMax(overflowFlag, launchSeq);
    if (slot < nLines) status[order ? order[slot] : slot] = LC_OVERFLOW;
}

struct DecideTables {
    const uint8_t* classMap;
    const uint32_t* followStart;
    const uint2* paths;
    const uint32_t* aux;
    uint32_t auxWords;
    const uint32_t* posMask;
    uint32_t maskShift;
    const uint32_t* behindBits;
    const uint32_t* aheadBits;
    const uint32_t* events;
    uint32_t nPos, nSlots, edgeClass;
    bool atomic;
};

// closed record of a frame: bits 0..7 group, 8..9 kind, 16..31 exit visit
enum { kClosedOld = 0, kClosedBySelf = 1, kClosedByOther = 2 };
__device__ __forceinline__ uint32_t decideClosedRec(uint32_t g, uint32_t kind, uint32_t visit) { return g | (kind << 8) | (visit << 16); }

struct DecideWalk {
    // scratch of one worker
    uint32_t* frames;   // nFrames x frameWords: pos, q | exhausted, chainTop, arenaMark, (chainDepth << 16) | (cutLevel << 8) | nClosed, closed[]
    uint32_t* nodes;    // arena: g | cutDone << 8 | cdepth << 16, frame depth, parent
    uint8_t* memo;      // nibbles [pos][frame], or nullptr
    uint32_t frameWords, closedCap, nFrames;

    __device__ __forceinline__ uint32_t* frame(uint32_t d) const { return frames + size_t(d) * frameWords; }
    __device__ __forceinline__ uint32_t memoGet(uint32_t p, uint32_t f) const {
        const size_t idx = size_t(p) * nFrames + f;
        return (memo[idx >> 1] >> ((idx & 1) * 4)) & 0xFu;
    }
    __device__ __forceinline__ void memoSet(uint32_t p, uint32_t f, uint32_t v) const {
        const size_t idx = size_t(p) * nFrames + f;
        const uint32_t sh = (idx & 1) * 4;
        memo[idx >> 1] = uint8_t((memo[idx >> 1] & ~(0xFu << sh)) | (v << sh));
    }
    __device__ __forceinline__ void addClosed(uint32_t d, uint32_t rec) const {
        uint32_t* f = frame(d);
        const uint32_t n = f[4] & 0xFFu;
        for (uint32_t k = 0; k < n; ++k)
            if (f[kDecideFrameWords + k] == rec) return;
        if (n < closedCap) {  // (cannot overflow: one record per enter / exit event of the paths of this frame)
            f[kDecideFrameWords + n] = rec;
            f[4] = (f[4] & ~0xFFu) | (n + 1);
        }
    }
    __device__ __forceinline__ void raiseCutLevel(uint32_t d, uint32_t cdepthN) const {
        uint32_t* f = frame(d);
        const uint32_t lvl = (f[4] >> 16) - cdepthN + 1;
        const uint32_t cur = (f[4] >> 8) & 0xFFu;
        if (lvl > cur) f[4] = (f[4] & ~0xFF00u) | ((lvl < 255u ? lvl : 255u) << 8);
    }
    // drop the choices made inside the group instance held by node n; d = the frame the cut is seen from
    __device__ void cut(uint32_t n, uint32_t d, bool real, uint32_t visit) const {
        uint32_t* node = nodes + size_t(n) * kDecideNodeWords;
        const uint32_t g = node[0] & 0xFFu, cdepthN = node[0] >> 16, d0 = node[1];
        if (d0 < d) {
            if (!(node[0] & 0x100u)) {
                for (uint32_t dd = d0 + 1; dd < d; ++dd) {
                    frame(dd)[1] |= kDecideExhausted;
                    raiseCutLevel(dd, cdepthN);
                }
                addClosed(d0, decideClosedRec(g, kClosedByOther, 0));
                node[0] |= 0x100u;
            }
            raiseCutLevel(d, cdepthN);
            if (real) addClosed(d, decideClosedRec(g, kClosedOld, visit));
            else frame(d)[1] |= kDecideExhausted;
        } else {
            addClosed(d, real ? decideClosedRec(g, kClosedBySelf, visit) : decideClosedRec(g, kClosedByOther, 0));
        }
    }
};

// One line, one lane.  Returns LC_MATCH / LC_NOMATCH / LC_GAVE_UP; on a match the capture row has been written.
__device__ inline uint32_t decideLine(const DecideTables& t, const DecideWalk& w, const uint8_t* __restrict__ text, uint32_t L,
                                      uint32_t from, uint32_t nGroupsOut, int32_t* __restrict__ out, uint64_t budget) {
    auto classAt = [&](uint32_t i) -> uint32_t { return t.classMap[text[i]]; };
    auto holdsAt = [&](uint32_t i) -> uint32_t {
        const uint32_t prev = i == 0 ? t.edgeClass : classAt(i - 1);
        const uint32_t nxt = i == L ? t.edgeClass : classAt(i);
        return t.behindBits[prev] | t.aheadBits[nxt];
    };
    uint32_t* f0 = w.frame(0);
    f0[0] = from ? 0u : t.nPos;  // a resumed search: the wrapper's prefix position has just consumed the byte before `from`
    f0[1] = 0;
    f0[2] = kDecideNone;
    f0[3] = 0;
    f0[4] = 0;
    int64_t d = 0;
    uint64_t steps = 0;
    while (d >= 0) {
        uint32_t* f = w.frame(uint32_t(d));
        const uint32_t i = from + uint32_t(d);
        const uint32_t p = f[0];
        const uint32_t fs = t.followStart[p], fe = t.followStart[p + 1];
        const uint32_t k = f[1] & ~kDecideExhausted;
        if ((f[1] & kDecideExhausted) || fs + k >= fe) {  // nothing left here: remember how this (position, offset) ended
            if (w.memo && p < t.nPos) {
                const uint32_t lvl = (f[4] >> 8) & 0xFFu;
                if (lvl <= 14) w.memoSet(p, uint32_t(d), 1 + lvl);
            }
            --d;
            continue;
        }
        f[1] = k + 1;  // (not exhausted, or we would not be here)
        if (++steps > budget) return LC_GAVE_UP;
        uint32_t arenaTop = f[3];
        uint32_t cur = f[2], curDepth = f[4] >> 16;
        const uint2 rec = t.paths[fs + k];
        uint32_t tgt = rec.x & 0xFFFFu;
        const uint32_t auxIdx = rec.x >> 16;
        const uint32_t holds = holdsAt(i);
        bool dead = false, ok = true;
        if (t.atomic) {
            const uint32_t* ev = t.events + (rec.y >> 8);
            const uint32_t nev = rec.y & 0xFFu;
            const uint32_t nClosed = f[4] & 0xFFu;
            for (uint32_t c = 0; c < nClosed && !dead; ++c) {
                const uint32_t cr = f[kDecideFrameWords + c];
                if (((cr >> 8) & 3u) == kClosedOld && nfaExitVisitFor(ev, 0, nev, cr & 0xFFu, holds) != (cr >> 16)) dead = true;
            }
            if (dead) continue;
            for (uint32_t e = 0; e < nev && !dead; ++e) {
                const int code = int(int16_t(ev[e] & 0xFFFFu));
                if (code >= 20000) {
                    if (!((holds >> (code - 20000)) & 1u)) {
                        ok = false;
                        break;
                    }
                } else if (code > 0) {
                    const uint32_t g = uint32_t(code - 1);
                    const uint32_t nc = f[4] & 0xFFu;  // (cuts earlier on this very path may have added records)
                    for (uint32_t c = 0; c < nc && !dead; ++c) {
                        const uint32_t cr = f[kDecideFrameWords + c];
                        if ((cr & 0xFFu) != g) continue;
                        const uint32_t kind = (cr >> 8) & 3u;
                        if (kind == kClosedByOther || (kind == kClosedBySelf && nfaExitVisitFor(ev, e + 1, nev, g, holds) != (cr >> 16)))
                            dead = true;
                    }
                    if (dead) break;
                    uint32_t* node = w.nodes + size_t(arenaTop) * kDecideNodeWords;
                    node[0] = g | ((curDepth + 1) << 16);
                    node[1] = uint32_t(d);
                    node[2] = cur;
                    cur = arenaTop++;
                    ++curDepth;
                } else {
                    const uint32_t g = uint32_t(-code - 1);
                    uint32_t n = cur;
                    while (n != kDecideNone && (w.nodes[size_t(n) * kDecideNodeWords] & 0xFFu) != g) n = w.nodes[size_t(n) * kDecideNodeWords + 2];
                    if (n == kDecideNone) continue;
                    w.cut(n, uint32_t(d), true, ev[e] >> 16);
                    cur = w.nodes[size_t(n) * kDecideNodeWords + 2];
                    curDepth = (w.nodes[size_t(n) * kDecideNodeWords] >> 16) - 1;
                }
            }
        } else if (auxIdx) {
            ok = (t.aux[size_t(auxIdx) * t.auxWords] & ~holds) == 0;
        }
        if (dead || !ok) continue;
        if (i == L) {
            if (tgt != 0xFFFFu) continue;
            // the match: capture offsets are the last stamp of every slot along the frames
            const uint32_t nOut = 2 * nGroupsOut;
            for (uint32_t s = 0; s < nOut; ++s) out[s] = -1;
            for (uint32_t dd = 0; dd <= uint32_t(d); ++dd) {
                const uint32_t* ff = w.frame(dd);
                const uint32_t qi = t.followStart[ff[0]] + (ff[1] & ~kDecideExhausted) - 1;
                const uint32_t a = t.paths[qi].x >> 16;
                if (!a) continue;
                const uint32_t* tags = t.aux + size_t(a) * t.auxWords + 1;
                for (uint32_t wd = 0; wd * 32 < t.nSlots; ++wd) {
                    uint32_t bits = tags[wd];
                    while (bits) {
                        const uint32_t s = wd * 32 + uint32_t(__ffs(int(bits)) - 1);
                        bits &= bits - 1;
                        if (s < nOut && s < t.nSlots) out[s] = int32_t(from + dd);
                    }
                }
            }
            return LC_MATCH;
        }
        if (tgt == 0xFFFFu) continue;
        {
            const uint32_t cls = classAt(i);
            if (!nfaMaskBit(t.posMask, t.maskShift, tgt, cls >> 5, cls & 31u)) continue;
        }
        const uint32_t m = w.memo ? w.memoGet(tgt, uint32_t(d) + 1) : 0u;
        if (m == 0) {
            uint32_t* nf = w.frame(uint32_t(d) + 1);
            nf[0] = tgt;
            nf[1] = 0;
            nf[2] = cur;
            nf[3] = arenaTop;
            nf[4] = curDepth << 16;
            ++d;
        } else if (m > 1) {  // fails from there after committing its m-1 innermost enclosing groups: replay the cut
            uint32_t n = cur;
            for (uint32_t up = 0; up + 2 < m && n != kDecideNone; ++up) n = w.nodes[size_t(n) * kDecideNodeWords + 2];
            if (n != kDecideNone) w.cut(n, uint32_t(d), false, 0);
        }
    }
    return LC_NOMATCH;
}

__global__ __launch_bounds__(64) void nfa_decide_kernel(const uint8_t* __restrict__ data, const uint32_t* __restrict__ off,
                                                        const uint32_t* __restrict__ len, uint32_t sepBytes,
                                                        const uint32_t* __restrict__ resume,
                                                        const uint32_t* __restrict__ blob, DecideShape shape,
                                                        uint32_t nGroupsOut, int32_t* __restrict__ caps,
                                                        uint8_t* __restrict__ status,
                                                        const uint32_t* __restrict__ overflowFlag, uint32_t launchSeq,
                                                        uint8_t* __restrict__ pool) {
    if (__atomic_load_n(overflowFlag, __ATOMIC_RELAXED) < launchSeq) return;  // the thread-list kernels decided every line
    DecidePlan* plan = reinterpret_cast<DecidePlan*>(pool);
    const uint32_t count = plan->count;
    if (count == 0) return;
    const uint32_t* list = reinterpret_cast<const uint32_t*>(pool + kDecideHeaderBytes);
    const uint32_t lane = threadIdx.x;
    const uint32_t workers = plan->workers;
    if (workers == 0) {  // the pool cannot hold one frame stack for the longest line: report, never guess
        for (uint32_t idx = blockIdx.x * 64 + lane; idx < count; idx += gridDim.x * 64) {
            status[list[idx]] = LC_GAVE_UP;
            atomicAdd(&plan->gaveUp, 1u);
        }
        return;
    }
    if (blockIdx.x >= workers) return;
    uint8_t* slice = pool + plan->slicesAt + uint64_t(blockIdx.x) * plan->sliceBytes;
    const uint64_t sliceBytes = plan->sliceBytes;

    const uint8_t* tbl = reinterpret_cast<const uint8_t*>(blob);
    const uint32_t* hdr = blob;
    DecideTables t;
    t.nPos = hdr[NF_NPOS];
    t.nSlots = hdr[NF_NSLOTS];
    t.edgeClass = hdr[NF_NCLASSES];
    t.classMap = tbl + hdr[NF_OFF_CLASSMAP];
    t.followStart = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_FOLLOWSTART]);
    t.paths = reinterpret_cast<const uint2*>(tbl + hdr[NF_OFF_PATHS]);
    t.aux = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_AUX]);
    t.auxWords = hdr[NF_AUX_WORDS];
    t.posMask = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_POSMASK]);
    t.maskShift = hdr[NF_MASK_WORDS] == 4 ? 2 : 1;
    t.behindBits = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_BEHIND]);
    t.aheadBits = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_AHEAD]);
    t.atomic = hdr[NF_ATOMIC] != 0;
    t.events = t.atomic ? reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_EVENTS]) : nullptr;

    for (;;) {
        uint32_t idx = 0;
        if (lane == 0) idx = atomicAdd(&plan->next, 1u);
        idx = __shfl(idx, 0, 64);
        if (idx >= count) break;
        const uint32_t line = list[idx];
        uint32_t o, L, from;
        decideLineSpan(off, len, sepBytes, resume, line, o, L, from);
        const uint32_t nFrames = L - from + 1;
        DecideWalk w;
        w.closedCap = shape.closedCap;
        w.frameWords = kDecideFrameWords + shape.closedCap;
        w.nFrames = nFrames;
        const uint64_t fixed = decideFixedBytes(nFrames, shape);
        w.frames = reinterpret_cast<uint32_t*>(slice);
        w.nodes = w.frames + size_t(nFrames) * w.frameWords;
        w.memo = nullptr;
        uint32_t verdict = LC_GAVE_UP;
        if (fixed <= sliceBytes) {
            const uint64_t memoBytes = decideMemoBytes(nFrames, t.nPos);
            if (fixed + memoBytes <= sliceBytes) {
                w.memo = slice + ((fixed + 15) & ~uint64_t(15));
                uint4* m4 = reinterpret_cast<uint4*>(w.memo);
                for (uint64_t q = lane; q < memoBytes / 16; q += 64) m4[q] = uint4{0, 0, 0, 0};
            }
            __threadfence_block();
            waveLdsSync();
            if (lane == 0)
                verdict = decideLine(t, w, data + o, L, from, nGroupsOut, caps + size_t(line) * 2 * nGroupsOut,
                                     w.memo ? kDecideBudgetMemo : kDecideBudgetNoMemo);
        }
        if (lane == 0) {
            if (verdict != LC_MATCH) {
                int32_t* out = caps + size_t(line) * 2 * nGroupsOut;
                for (uint32_t s = 0; s < 2 * nGroupsOut; ++s) out[s] = -1;
            }
            status[line] = uint8_t(verdict);
            if (verdict == LC_GAVE_UP) atomicAdd(&plan->gaveUp, 1u);
        }
        waveLdsSync();
    }
}


// ------------------------------------------------------------------------------------------------ the walk as the FIRST engine
// nfa_dfs_kernel: ONE LINE PER LANE, plain backtracking (no memo) under a small step budget -- what regexp2 / boost do for
// every line, 64 lines per wavefront.  The thread-list kernels keep a line's whole thread set in one wavefront and pay a
// chain of dependent table reads per byte (2-4 us per byte-step when the tables sit in L2: 0.2 GB/s on the Grok corpus);
// a backtracker touches one path at a time, so a lane is enough for a line and the chip holds half a million lines' walks
// at once -- the dependent reads of one lane hide behind the other lanes'.  Lines whose walk exceeds the budget (the
// patterns a backtracker is exponential on) or whose frame stack finds no room in the pool are marked LC_PENDING and the
// launch's pending flag is raised: the thread-list kernels behind this launch take exactly those lines (and never blow up).
// Frame stacks are carved from the pool with one atomicAdd per line.
constexpr uint8_t LC_PENDING = 4;  // transient: left by nfa_dfs_kernel for the thread-list kernels of the same launch

struct DfsPoolHeader {
    unsigned long long cursor;  // bytes handed out (zeroed before every launch)
    uint32_t pending;           // lines left to the thread-list kernels by the last launch (statistics)
    uint32_t decided;
};

__global__ __launch_bounds__(64) void nfa_dfs_kernel(const uint8_t* __restrict__ data, const uint32_t* __restrict__ off,
                                                     const uint32_t* __restrict__ len, uint32_t sepBytes, uint32_t nLines,
                                                     const uint32_t* __restrict__ nLinesPtr, const uint32_t* __restrict__ order,
                                                     const uint32_t* __restrict__ resume, const uint32_t* __restrict__ blob,
                                                     DecideShape shape, uint32_t nGroupsOut, int32_t* __restrict__ caps,
                                                     uint8_t* __restrict__ status, uint32_t* __restrict__ pendingFlag,
                                                     uint32_t launchSeq, uint8_t* __restrict__ pool, uint64_t poolBytes,
                                                     uint32_t stepsPerByte) {
    if (nLinesPtr) {
        const uint32_t dyn = *nLinesPtr;
        nLines = dyn < nLines ? dyn : nLines;
    }
    const uint32_t slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= nLines) return;
    const uint32_t line = order ? order[slot] : slot;
    DfsPoolHeader* hdrPool = reinterpret_cast<DfsPoolHeader*>(pool);

    const uint8_t* tbl = reinterpret_cast<const uint8_t*>(blob);
    const uint32_t* hdr = blob;
    DecideTables t;
    t.nPos = hdr[NF_NPOS];
    t.nSlots = hdr[NF_NSLOTS];
    t.edgeClass = hdr[NF_NCLASSES];
    t.classMap = tbl + hdr[NF_OFF_CLASSMAP];
    t.followStart = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_FOLLOWSTART]);
    t.paths = reinterpret_cast<const uint2*>(tbl + hdr[NF_OFF_PATHS]);
    t.aux = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_AUX]);
    t.auxWords = hdr[NF_AUX_WORDS];
    t.posMask = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_POSMASK]);
    t.maskShift = hdr[NF_MASK_WORDS] == 4 ? 2 : 1;
    t.behindBits = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_BEHIND]);
    t.aheadBits = reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_AHEAD]);
    t.atomic = hdr[NF_ATOMIC] != 0;
    t.events = t.atomic ? reinterpret_cast<const uint32_t*>(tbl + hdr[NF_OFF_EVENTS]) : nullptr;

    uint32_t o, L, from;
    decideLineSpan(off, len, sepBytes, resume, line, o, L, from);
    const uint32_t nFrames = L - from + 1;
    const uint64_t need = (decideFixedBytes(nFrames, shape) + 63) & ~uint64_t(63);
    uint32_t verdict = LC_GAVE_UP;
    const uint64_t at = kDecideHeaderBytes + atomicAdd(&hdrPool->cursor, static_cast<unsigned long long>(need));
    int32_t* out = caps + size_t(line) * 2 * nGroupsOut;
    if (at + need <= poolBytes) {
        DecideWalk w;
        w.closedCap = shape.closedCap;
        w.frameWords = kDecideFrameWords + shape.closedCap;
        w.nFrames = nFrames;
        w.frames = reinterpret_cast<uint32_t*>(pool + at);
        w.nodes = w.frames + size_t(nFrames) * w.frameWords;
        w.memo = nullptr;
        verdict = decideLine(t, w, data + o, L, from, nGroupsOut, out, uint64_t(stepsPerByte) * nFrames + 4096);
    }
    if (verdict == LC_GAVE_UP) {
        status[line] = LC_PENDING;
        atomicMax(pendingFlag, launchSeq);
        atomicAdd(&hdrPool->pending, 1u);
        return;
    }
    if (verdict != LC_MATCH)
        for (uint32_t s = 0; s < 2 * nGroupsOut; ++s) out[s] = -1;
    status[line] = uint8_t(verdict);
}
