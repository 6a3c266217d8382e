// bt_kernel.hpp -- launch form of the device backtracking engine (bt_vm.hpp; included by gpu_runtime.hip only).
//
// One value per LANE; a lane owns a slice of the launch's scratch in HBM (captures, loop registers, stack) and takes values in turn
// (grid-stride), so the scratch is sized by the lanes in flight, not by the batch (gpu_runtime.hip launchBt: cached per calling thread
// and stream for small launches, stream-ordered from a pool of the library's own for large ones).  The program (a few KB: byte classes + 16 B per
// instruction) is staged into LDS when it fits 48 KB -- every step is a dependent read of it.  The lanes of a wavefront walk
// different paths: the engine is the drop-in answer for patterns no automaton can run (back-references), not a throughput path; what
// bounds it is the latency of a step (program word from LDS, value byte and stack entry from L2 / HBM).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/lc_regex_gpu.h"
#include "bt_vm.hpp"

constexpr uint32_t kBtBlock = 64;
// Two passes share one scratch pool (1 GB at most).  Pass 1: up to 131 072 lanes in flight (two wavefronts per SIMD: more hide more of a
// step's latency on lines that backtrack, but their lines no longer fit L2 together -- measured, profiles/round6_bt_engine.txt) with 8 KB
// of scratch each -- ~1 000 stack entries, a log line's counted repeats and captures need a few dozen.  A value that fills its slice is
// left LC_PENDING and raises the pool's flag word; pass 2 (an eighth of pass 1's lanes, at most 8 192, x 64 KB: ~8 000 entries) takes
// the pending values and returns at once when the flag is down.
constexpr uint32_t kBtMaxLanes = 131072, kBtSliceWords = 2048;
constexpr uint32_t kBtRetryLanes = 8192, kBtRetrySliceWords = 16384;
constexpr uint32_t kBtPoolHeaderWords = 64;      // the flag word, in front of the slices
constexpr uint8_t kBtPending = 4;                // (LC_PENDING of the other engines' protocols: transient, inside one match call)
constexpr uint32_t kBtStageMaxBytes = 48 * 1024;  // programs up to this size are walked from LDS
// Steps per value before it is reported LC_GAVE_UP.  boost's BOOST_REGEX_MAX_STATE_COUNT is 100 000 000 states per match; a lane
// that took that many steps would hold its wavefront for minutes, so the device bound is lower (LC_BT_BUDGET overrides it): values
// between the two bounds are reported, counted (lc_gave_up_values_total) and treated as parse failures -- never guessed.
constexpr uint32_t kBtDefaultBudget = 1u << 22;

__global__ __launch_bounds__(kBtBlock) void bt_match_kernel(const uint8_t* __restrict__ data, const uint32_t* __restrict__ off,
                                                            const uint32_t* __restrict__ len, uint32_t sepBytes, uint32_t nLines,
                                                            const uint32_t* __restrict__ nLinesPtr, const uint32_t* __restrict__ order,
                                                            const uint32_t* __restrict__ resume, const uint32_t* __restrict__ blob,
                                                            uint32_t blobWords, uint32_t stageWords, uint32_t nGroupsOut,
                                                            int32_t* __restrict__ caps, uint8_t* __restrict__ status,
                                                            uint32_t* __restrict__ pool, uint32_t sliceWords, uint32_t budget, uint32_t retryPass) {
    extern __shared__ uint32_t btStaged[];
    if (retryPass && __atomic_load_n(pool, __ATOMIC_RELAXED) == 0u) return;  // nothing was left pending
    uint32_t* scratch = pool + kBtPoolHeaderWords;
    for (uint32_t i = threadIdx.x; i < stageWords; i += kBtBlock) btStaged[i] = blob[i];
    __syncthreads();
    const uint32_t* prog = stageWords ? btStaged : blob;
    (void)blobWords;
    if (nLinesPtr) nLines = *nLinesPtr < nLines ? *nLinesPtr : nLines;
    const uint32_t lane = blockIdx.x * kBtBlock + threadIdx.x, lanes = gridDim.x * kBtBlock;
    uint32_t* mine = scratch + size_t(lane) * sliceWords;
    const uint32_t nCaps = prog[BT_NCAPS];
    for (uint32_t slot = lane; slot < nLines; slot += lanes) {
        const uint32_t line = order ? order[slot] : slot;
        if (retryPass && status[line] != kBtPending) continue;
        const uint32_t o = off[line];
        const uint32_t L = len ? len[line] : off[line + 1] - o - sepBytes;
        uint32_t from = 0;
        if (resume) {
            from = resume[line];
            from = from < L ? from : L;
        }
        const int r = btRun(prog, data + o, L, from, mine, sliceWords, budget);
        int32_t* out = caps + size_t(line) * 2 * nGroupsOut;
        for (uint32_t s = 0; s < 2 * nGroupsOut; ++s) out[s] = (r == 1 && s + 2 < nCaps) ? int32_t(mine[s + 2]) : -1;  // (slot 0/1: the whole match)
        if (r == -2 && !retryPass) {
            status[line] = kBtPending;
            atomicOr(pool, 1u);
        } else {
            status[line] = r == 1 ? LC_MATCH : r == 0 ? LC_NOMATCH : LC_GAVE_UP;
        }
    }
}
