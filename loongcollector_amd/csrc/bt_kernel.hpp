// bt_kernel.hpp -- launch form of the device backtracking engine (bt_vm.hpp; included by gpu_runtime.hip only).
//
// One value per LANE; a lane owns a slice of the launch's scratch pool in HBM (captures, loop registers, stack) and takes values in
// turn (grid-stride), so the pool is sized by the lanes in flight, not by the batch.  The program (a few KB: byte classes + 16 B per
// instruction) is staged into LDS when it fits 48 KB -- every step is a dependent read of it.  The lanes of a wavefront walk
// different paths: the engine is the drop-in answer for patterns no automaton can run (back-references), not a throughput path; what
// bounds it is the latency of a step (program word from LDS, value byte and stack entry from L2 / HBM).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/lc_regex_gpu.h"
#include "bt_vm.hpp"

constexpr uint32_t kBtBlock = 64;
constexpr uint32_t kBtMaxLanes = 8192;           // lanes of a launch (128 wavefronts)
constexpr uint32_t kBtSliceWords = 16384;        // 64 KB of scratch per lane: ~8 000 stack entries
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
                                                            uint32_t* __restrict__ scratch, uint32_t sliceWords, uint32_t budget) {
    extern __shared__ uint32_t btStaged[];
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
        status[line] = r == 1 ? LC_MATCH : r == 0 ? LC_NOMATCH : LC_GAVE_UP;
    }
}
