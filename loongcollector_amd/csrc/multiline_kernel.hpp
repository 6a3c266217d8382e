// multiline_kernel.hpp -- the multiline processors' record boundaries on the device (included by multiline_device.hip only).
//
// After the match kernels have answered "does item i match the start / continue / end pattern" for every line of a read buffer
// (or every event of a group), ml_flags_kernel folds the three status arrays into one flag byte per item and counts the
// undecided ones, and ml_bounds_kernel -- ONE workgroup, multiline_scan.hpp's four phases -- turns the flags into the records the
// reference's sequential walk would emit.  Only the records and eight counters travel back.  Integer work on a few bytes per
// line; the point is that nothing but the records crosses PCIe and that the host does not touch the lines at all.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/lc_multiline.h"
#include "../../include/lc_regex_gpu.h"
#include "multiline_scan.hpp"

__global__ __launch_bounds__(256) void ml_flags_kernel(const uint32_t* __restrict__ nPtr, uint32_t maxItems,
                                                       const uint8_t* __restrict__ sStart, const uint8_t* __restrict__ sCont,
                                                       const uint8_t* __restrict__ sEnd, const uint32_t* __restrict__ off,
                                                       uint8_t* __restrict__ flags, uint32_t* __restrict__ counts) {
    uint32_t n = nPtr ? *nPtr : maxItems;
    n = n < maxItems ? n : maxItems;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t fl = 0, over = 0, gave = 0;
    if (i < n) {
        const uint8_t a = sStart ? sStart[i] : uint8_t(LC_NOMATCH), b = sCont ? sCont[i] : uint8_t(LC_NOMATCH),
                      c = sEnd ? sEnd[i] : uint8_t(LC_NOMATCH);
        fl = (a == LC_MATCH ? ML_F_START : 0u) | (b == LC_MATCH ? ML_F_CONT : 0u) | (c == LC_MATCH ? ML_F_END : 0u) |
             ((off && off[i + 1] - off[i] == 1u) ? ML_F_EMPTY : 0u);
        over = (a == LC_OVERFLOW) + (b == LC_OVERFLOW) + (c == LC_OVERFLOW);
        gave = (a == LC_GAVE_UP) + (b == LC_GAVE_UP) + (c == LC_GAVE_UP);
        flags[i] = uint8_t(fl);
    }
    if (__any(over != 0 || gave != 0)) {  // (rare: the decide pass switched off, or out of budget)
        if (over) atomicAdd(&counts[ML_CNT_OVERFLOW], over);
        if (gave) atomicAdd(&counts[ML_CNT_GAVE_UP], gave);
    }
}

// what phase C / D write through
struct MlDeviceWriter {
    lc_ml_record_t* records;
    uint32_t recCap;
    const uint32_t* off;  // byte mode: the offsets[n+1] + separator table of the split kernel; NULL: records carry item indices
    uint32_t n, nbytes, lastItemIsLast;
    MlJob* jobs;
    uint32_t* nJobs;
    __device__ void record(uint32_t slot, uint32_t first, uint32_t last, uint32_t matched, uint32_t emitter) const {
        if (slot >= recCap) return;
        const uint32_t fl = matched | ((emitter >= n || (emitter + 1 == n && lastItemIsLast)) ? LC_ML_LAST : 0u);
        if (off) {
            const uint32_t b = off[first];
            // (the flush emits the rest of the source value, a trailing line feed included: :290-292)
            records[slot] = lc_ml_record_t{b, (matched & 1u) && emitter >= n ? nbytes - b : off[last + 1] - 1u - b, fl};
        } else {
            records[slot] = lc_ml_record_t{first, last - first + 1u, fl};
        }
    }
    __device__ void job(const MlJob& j) const { jobs[atomicAdd(nJobs, 1u)] = j; }
};

// counts: ML_CNT_WORDS words; [ML_CNT_OVERFLOW], [ML_CNT_GAVE_UP] are ml_flags_kernel's, the others are written here
__global__ __launch_bounds__(kMlThreads) void ml_bounds_kernel(const uint32_t* __restrict__ nPtr, uint32_t maxItems,
                                                               const uint8_t* __restrict__ flags, const uint32_t* __restrict__ off,
                                                               uint32_t nbytes, uint32_t mode, lc_ml_record_t* __restrict__ records,
                                                               uint32_t recCap, uint32_t* __restrict__ counts) {
    __shared__ MlSummary summaries[2 * kMlThreads];
    __shared__ MlEntry entries[kMlThreads];
    __shared__ MlJob jobs[kMlThreads + 1];
    __shared__ uint32_t nJobs, flushMatchedFirst, flushRecBase, totals[ML_CNT_WORDS];
    uint32_t n = nPtr ? *nPtr : maxItems;
    n = n < maxItems ? n : maxItems;
    const uint32_t t = threadIdx.x, slices = mlSliceCount(n);
    if (t == 0) nJobs = 0;
    if (t < slices) mlPhaseA(t, n, mode, flags, &summaries[2 * t]);
    __syncthreads();
    if (t == 0) {
        MlJob flush;
        uint32_t fm;
        mlPhaseB(n, mode, summaries, entries, totals, flush, fm);
        flushMatchedFirst = fm;
        flushRecBase = flush.recBase;
        if (flush.first != kMlInherit) jobs[nJobs++] = flush;
        counts[ML_CNT_ITEMS] = totals[ML_CNT_ITEMS];
        counts[ML_CNT_UNMATCHED] = totals[ML_CNT_UNMATCHED];
        counts[ML_CNT_MATCHED_LOGS] = totals[ML_CNT_MATCHED_LOGS];
        counts[ML_CNT_RECORDS] = totals[ML_CNT_RECORDS];
        counts[ML_CNT_FINAL_PARTIAL] = totals[ML_CNT_FINAL_PARTIAL];
        counts[ML_CNT_FINAL_START] = totals[ML_CNT_FINAL_START];
    }
    __syncthreads();
    // isLastLog of the last item: its end is the end of the source value (an unterminated last line: the split kernel's table
    // ends one past the buffer)
    MlDeviceWriter w{records, recCap, off, n, nbytes, (off && n && off[n] == nbytes + 1u) ? 1u : 0u, jobs, &nJobs};
    if (t < slices) mlPhaseC(t, n, mode, flags, entries[t], w);
    if (t == 0 && flushMatchedFirst != kMlInherit) w.record(flushRecBase, flushMatchedFirst, n - 1, 1u, n);
    __syncthreads();
    const uint32_t queued = nJobs;
    for (uint32_t j = 0; j < queued; ++j) {
        const MlJob job = jobs[j];
        for (uint32_t k = job.first + t; k <= job.last; k += kMlThreads) w.record(job.recBase + (k - job.first), k, k, k > job.first ? ML_REC_RUN : 0u, job.emitter);
    }
}
