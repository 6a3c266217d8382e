// multiline_device.hip -- the multiline processors' device trips (include/lc_multiline.h; kernels in multiline_kernel.hpp).
//
//   lc_multiline_bounds_device   flags + records for items whose start / continue / end answers are already on the device
//   lcMultilineSplitTrip         ProcessorSplitMultilineLogStringNative::ProcessEvent for ONE source value:
//                                    ONE upload of the read buffer -> split kernels (line table on the device) -> one status-only
//                                    match launch per configured pattern over the same device copy -> flags -> record scan;
//                                    the records (12 bytes each) and eight counters are written straight into pinned memory.
//                                    One host synchronisation per buffer; the host never looks at a line.
//   lcMultilineViewsTrip         the same for the events of a group (ProcessorMergeMultilineLogNative::MergeLogsByRegex): the
//                                    values are gathered into the pinned staging once, whatever the number of patterns.
// Round 2 did this with a host byte loop for the line feeds, one gather + upload + launch + download PER PATTERN, and the
// reference's sequential walk on the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lc_multiline.h"
#include "../../include/lc_regex_gpu.h"
#include "multiline_gpu.hpp"
#include "multiline_kernel.hpp"
#include "regex_handle.hpp"
#include "runtime_internal.hpp"
#include "trip_buffers.hpp"

extern "C" int lc_multiline_bounds_device(uint32_t mode, const uint8_t* d_start, const uint8_t* d_cont, const uint8_t* d_end,
                                          const uint32_t* d_nitems, uint32_t max_items, const uint32_t* d_off, uint32_t nbytes,
                                          uint8_t* d_flags, lc_ml_record_t* records, uint32_t record_cap, uint32_t* counts,
                                          void* stream) {
    if (!counts || (max_items && !d_flags) || (record_cap && !records)) return LC_ERR_ARG;
    if (lc_device_count() <= 0) {
        lcSetLastError("no HIP device");
        return LC_ERR_NO_DEVICE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    LC_HIP_TRY(hipMemsetAsync(counts, 0, ML_CNT_WORDS * 4, st));
    if (max_items) {
        lcNoteKernel("ml_flags_kernel");
        hipLaunchKernelGGL(ml_flags_kernel, dim3((max_items + 255) / 256), dim3(256), 0, st, d_nitems, max_items, d_start, d_cont, d_end,
                           d_off, d_flags, counts);
    }
    lcNoteKernel("ml_bounds_kernel");
    hipLaunchKernelGGL(ml_bounds_kernel, dim3(1), dim3(kMlThreads), 0, st, d_nitems, max_items, d_flags, d_off, nbytes, mode, records,
                       record_cap, counts);
    LC_HIP_TRY(hipGetLastError());
    return LC_OK;
}

namespace {

// per runner thread: a stream, pinned staging, device buffers; grow-only
typedef TripBuf MlBuf;
struct MlThread {
    hipStream_t stream = nullptr;
    int device = -1;
    MlBuf hIn, hOut;                                             // pinned
    MlBuf dData, dOff, dStatus, dFlags, dScratch, dSmall;        // device
    uint32_t recordGuess = 1024;
    MlThread() { hIn.pinned = hOut.pinned = true; }
    ~MlThread() {
        if (lcRuntimeUsable() && (stream || hIn.p || dData.p)) lcMultilineThreadRelease();
    }
};
thread_local MlThread tlsMl;

int prepare(MlThread& T, int& dev) {
    if (lc_device_count() <= 0) {
        lcSetLastError("no HIP device: the multiline processors have no CPU path");
        return LC_ERR_NO_DEVICE;
    }
    {
        const int rcDev = lcHostEntryDevice(&dev);  // the thread's binding (runtime_internal.hpp)
        if (rcDev != LC_OK) return rcDev;
    }
    if (T.stream && T.device != dev) lcMultilineThreadRelease();  // (another device: old stream and buffers go)
    if (!T.stream) {
        LC_HIP_TRY(hipStreamCreateWithFlags(&T.stream, hipStreamNonBlocking));
        T.device = dev;
        lcRegisterExitHook();
    }
    return LC_OK;
}

uint32_t modeOf(const lc_multiline& m, bool flush) {
    return (m.start ? ML_HAS_START : 0u) | (m.cont ? ML_HAS_CONT : 0u) | (m.end ? ML_HAS_END : 0u) | (m.discardUnmatched ? ML_DISCARD : 0u) |
           (flush ? ML_FLUSH : 0u);
}

// flags + records behind the status launches; one synchronisation (two when the first guess of the record count was too small)
int boundsAndFetch(MlThread& T, uint32_t mode, const uint8_t* const status[3], const uint32_t* dN, uint32_t maxItems, const uint32_t* dOff,
                   uint32_t nbytes, std::vector<lc_ml_record_t>& out, uint32_t counts[ML_CNT_WORDS]) {
    for (int attempt = 0; attempt < 2; ++attempt) {
        const uint32_t cap = attempt == 0 ? std::min<uint32_t>(maxItems + 1, std::max<uint32_t>(T.recordGuess, 256)) : counts[ML_CNT_RECORDS];
        LC_HIP_TRY(T.hOut.ensure(64 + size_t(cap) * sizeof(lc_ml_record_t)));
        uint32_t* hCounts = static_cast<uint32_t*>(T.hOut.p);
        lc_ml_record_t* hRecs = reinterpret_cast<lc_ml_record_t*>(static_cast<uint8_t*>(T.hOut.p) + 64);
        // (the flag kernel counts with device atomics: the counters live in device memory -- atomics on pinned host memory need PCIe
        // atomic support -- and come down behind the kernels; the records are plain stores into the pinned block)
        LC_HIP_TRY(T.dSmall.ensure(256));
        uint32_t* dCounts = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(T.dSmall.p) + 128);
        const int rc = lc_multiline_bounds_device(mode, status[0], status[1], status[2], dN, maxItems, dOff, nbytes,
                                                  static_cast<uint8_t*>(T.dFlags.p), hRecs, cap, dCounts, T.stream);
        if (rc != LC_OK) return rc;
        LC_HIP_TRY(hipMemcpyAsync(hCounts, dCounts, ML_CNT_WORDS * 4, hipMemcpyDeviceToHost, T.stream));
        LC_HIP_TRY(hipStreamSynchronize(T.stream));
        std::memcpy(counts, hCounts, ML_CNT_WORDS * 4);
        if (counts[ML_CNT_RECORDS] <= cap) {
            out.assign(hRecs, hRecs + counts[ML_CNT_RECORDS]);
            T.recordGuess = counts[ML_CNT_RECORDS] + counts[ML_CNT_RECORDS] / 4 + 64;
            return LC_OK;
        }
    }
    lcSetLastError("multiline: record count changed between two launches over the same flags");
    return LC_ERR_HIP;
}

int checkUndecided(const uint32_t counts[ML_CNT_WORDS]) {
    if (counts[ML_CNT_OVERFLOW]) {  // "not decided" (decide pass switched off) must not drive the automaton as "no match"
        lcSetLastError("multiline: lines left undecided by the match kernels (LC_OVERFLOW)");
        return LC_ERR_UNSUPPORTED;
    }
    // BoostRegexSearch failed with an exception (StringTools.cpp:277-282): false, and counted
    if (counts[ML_CNT_GAVE_UP]) lcNoteGaveUp(counts[ML_CNT_GAVE_UP]);
    return LC_OK;
}

}  // namespace

void lcMultilineThreadRelease() {
    MlThread& T = tlsMl;
    if (T.stream) {
        (void)hipStreamSynchronize(T.stream);
        (void)hipStreamDestroy(T.stream);
        T.stream = nullptr;
    }
    for (MlBuf* b : {&T.hIn, &T.hOut, &T.dData, &T.dOff, &T.dStatus, &T.dFlags, &T.dScratch, &T.dSmall}) b->release();
    T.device = -1;
}

int lcMultilineSplitTrip(lc_multiline* m, const uint8_t* data, uint32_t nbytes, std::vector<lc_ml_record_t>& out,
                         uint32_t counts[ML_CNT_WORDS]) {
    out.clear();
    std::memset(counts, 0, ML_CNT_WORDS * 4);
    MlThread& T = tlsMl;
    int dev = 0;
    int rc = prepare(T, dev);
    if (rc != LC_OK) return rc;
    if (nbytes == 0) return LC_OK;  // GetNextLine finds no line; the flush needs multiStart < size (:288)
    if (nbytes >= 0xFFFFFFF0u) return LC_ERR_ARG;
    const uint32_t maxLines = nbytes;  // (a line has at least its separator, except the last)
    const size_t splitScratch = lc_split_scratch_bytes(nbytes);
    const size_t statusStride = (size_t(maxLines) + 16 + 15) & ~size_t(15);
    LC_HIP_TRY(T.hIn.ensure(size_t(nbytes) + 16));
    LC_HIP_TRY(T.dData.ensure(size_t(nbytes) + 16));
    LC_HIP_TRY(T.dOff.ensure((size_t(maxLines) + 2) * 4));
    LC_HIP_TRY(T.dStatus.ensure(3 * statusStride));
    LC_HIP_TRY(T.dFlags.ensure(size_t(maxLines) + 16));
    LC_HIP_TRY(T.dScratch.ensure(splitScratch));
    LC_HIP_TRY(T.dSmall.ensure(256));
    std::memcpy(T.hIn.p, data, nbytes);
    std::memset(static_cast<uint8_t*>(T.hIn.p) + nbytes, 0, 16);
    const uint8_t* dData = static_cast<const uint8_t*>(T.dData.p);
    uint32_t* dOff = static_cast<uint32_t*>(T.dOff.p);
    uint32_t* dN = static_cast<uint32_t*>(T.dSmall.p);
    int32_t* dCapsDummy = reinterpret_cast<int32_t*>(static_cast<uint8_t*>(T.dSmall.p) + 64);
    rc = lc_upload_pinned(T.hIn.p, T.dData.p, size_t(nbytes) + 16, T.stream);
    // GetNextLine :382-392: lines are separated by '\n'; a trailing '\n' does not open an empty last line (the split kernels' rule)
    if (rc == LC_OK) rc = lc_split_lines_device(dData, nbytes, uint8_t('\n'), dOff, maxLines + 2, dN, T.dScratch.p, splitScratch, T.stream);
    lc_regex_t* const res[3] = {m->start, m->cont, m->end};
    const uint8_t* status[3] = {nullptr, nullptr, nullptr};
    for (int k = 0; k < 3 && rc == LC_OK; ++k) {
        if (!res[k]) continue;
        uint8_t* st = static_cast<uint8_t*>(T.dStatus.p) + size_t(k) * statusStride;
        status[k] = st;
        rc = lcMatchOnStream(res[k], res[k]->engine, dev, dData, dOff, nullptr, 1, maxLines, dN, nullptr, nullptr, 0, dCapsDummy, st, T.stream);
    }
    if (rc != LC_OK) return rc;
    rc = boundsAndFetch(T, modeOf(*m, true), status, dN, maxLines, dOff, nbytes, out, counts);
    if (rc != LC_OK) return rc;
    return checkUndecided(counts);
}

int lcMultilineViewsTrip(const lc_multiline* m, const uint8_t* const* ptrs, const uint32_t* lens, uint32_t n, bool flush, bool keepUnmatched,
                         std::vector<lc_ml_record_t>& out, uint32_t counts[ML_CNT_WORDS]) {
    out.clear();
    std::memset(counts, 0, ML_CNT_WORDS * 4);
    MlThread& T = tlsMl;
    int dev = 0;
    int rc = prepare(T, dev);
    if (rc != LC_OK) return rc;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; ++i) total += lens[i];
    // adjacent values are copied WITH the one byte between them (the run fast path below): at most total + (n - 1) bytes land
    if (total + n >= 0xFFFFFFE0ull) return LC_ERR_ARG;
    // staging: [bytes of all values and of the joins between adjacent ones, padded to 16][off: n words][len: n words]
    const size_t dataBytes = (size_t(total) + size_t(n) + 31) & ~size_t(15);
    const size_t upBytes = dataBytes + size_t(n) * 8 + 16;
    const size_t statusStride = (size_t(n) + 16 + 15) & ~size_t(15);
    LC_HIP_TRY(T.hIn.ensure(upBytes));
    LC_HIP_TRY(T.dData.ensure(upBytes));
    LC_HIP_TRY(T.dStatus.ensure(3 * statusStride));
    LC_HIP_TRY(T.dFlags.ensure(size_t(n) + 16));
    LC_HIP_TRY(T.dSmall.ensure(256));
    uint8_t* h = static_cast<uint8_t*>(T.hIn.p);
    uint32_t* hOff = reinterpret_cast<uint32_t*>(h + dataBytes);
    uint32_t* hLen = hOff + n;
    uint32_t at = 0;
    for (uint32_t i = 0; i < n; ++i) {
        // the events of one read buffer lie back to back in the source buffer: runs of adjacent views go in one memcpy
        uint32_t j = i, run = lens[i];
        hOff[i] = at;
        hLen[i] = lens[i];
        while (j + 1 < n && ptrs[j + 1] == ptrs[j] + lens[j] + 1 && uint64_t(run) + 1 + lens[j + 1] < 0x7FFFFFFFu) {
            ++j;
            hOff[j] = at + run + 1;
            hLen[j] = lens[j];
            run += 1 + lens[j];
        }
        if (run) std::memcpy(h + at, ptrs[i], run);
        at += run;
        i = j;
    }
    if (size_t(at) > dataBytes) return LC_ERR_OVERFLOW;  // (cannot happen: at <= total + n - 1)
    std::memset(h + at, 0, dataBytes - at);
    int32_t* dCapsDummy = reinterpret_cast<int32_t*>(static_cast<uint8_t*>(T.dSmall.p) + 64);
    const uint8_t* dData = static_cast<const uint8_t*>(T.dData.p);
    const uint32_t* dOff = reinterpret_cast<const uint32_t*>(dData + dataBytes);
    const uint32_t* dLen = dOff + n;
    if (n) rc = lc_upload_pinned(T.hIn.p, T.dData.p, upBytes, T.stream);
    lc_regex_t* const res[3] = {m->start, m->cont, m->end};
    const uint8_t* status[3] = {nullptr, nullptr, nullptr};
    // the patterns are the jobs of ONE lc_regex_match_device_multi call over the same device copy (patterns on the tagged-DFA engine
    // share a single launch)
    std::vector<lc_match_job> jobs;
    for (int k = 0; k < 3 && n; ++k) {
        if (!res[k]) continue;
        uint8_t* st = static_cast<uint8_t*>(T.dStatus.p) + size_t(k) * statusStride;
        status[k] = st;
        jobs.push_back({res[k], dData, dOff, dLen, 0u, n, 0u, dCapsDummy, st});
    }
    if (rc == LC_OK && !jobs.empty()) rc = lc_regex_match_device_multi(jobs.data(), uint32_t(jobs.size()), T.stream);
    if (rc != LC_OK) return rc;
    // keepUnmatched: the caller applies UnmatchedContentTreatment itself (it counts EVENTS, multiline_events.cpp) -- a parameter, not a
    // write to the processor's shared state (runner threads share the instance)
    rc = boundsAndFetch(T, modeOf(*m, flush) & ~(keepUnmatched ? uint32_t(ML_DISCARD) : 0u), status, nullptr, n, nullptr, 0, out, counts);
    if (rc != LC_OK) return rc;
    return checkUndecided(counts);
}
