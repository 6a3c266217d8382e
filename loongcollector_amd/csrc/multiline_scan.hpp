// multiline_scan.hpp -- record boundaries of the multiline processors as a SCAN over per-item flag bytes (shared by the device
// kernel, multiline_kernel.hpp, and its host model, lc_multiline_bounds_model: the same code runs in both).
//
// What is restated.  ProcessorSplitMultilineLogStringNative::ProcessEvent (core/plugin/processor/inner/
// ProcessorSplitMultilineLogStringNative.cpp:161-298) and ProcessorMergeMultilineLogNative::MergeLogsByRegex
// (ProcessorMergeMultilineLogNative.cpp:161-330) walk their items (lines of a read buffer / events of a group) with the same
// automaton: one bit of state (isPartialLog), one datum (multiStart / begin: where the log under construction began), three
// answers per item (does it match the start / continue / end pattern).  The answers come from the match kernels for all
// items at once, so the walk no longer has to be sequential either:
//
//   phase A  every thread takes a contiguous SLICE of items and runs the automaton over it twice -- entering with
//            isPartialLog = 0 and = 1 -- with `multiStart` symbolic ("inherited"): state on exit, multiStart on exit (a concrete
//            item, or still inherited), what the slice emits.  At most one emission per slice refers to the inherited value.
//   phase B  one thread chains the slice summaries: entry state, entry multiStart and first output slot of every slice; the
//            flush after the last item (:288-298 / :316-323).
//   phase C  every thread re-runs its slice with the real entry values and writes its records to their slots.  A run of
//            unmatched items that began in an earlier slice (HandleUnmatchLogs over [multiStart, cur], :341-380 / :360-392) is
//            queued, and
//   phase D  the whole workgroup writes the queued runs, one item per thread.
//
// Records come out in item order, exactly the order the reference emits them in (every emission covers items behind all
// earlier ones).  `emitter` is the item being processed when the reference emits (its isLastLog decides the position length of
// the new event, ProcessorSplitMultilineLogStringNative.cpp:174,327-329).
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define LC_ML_HD __host__ __device__
#else
#define LC_ML_HD
#endif

// mode bits: which patterns the processor works with (Has*Pattern), UnmatchedContentTreatment, flush at the end of the input
constexpr uint32_t ML_HAS_START = 1u, ML_HAS_CONT = 2u, ML_HAS_END = 4u, ML_DISCARD = 8u, ML_FLUSH = 16u;
// flag byte of an item
constexpr uint32_t ML_F_START = 1u, ML_F_CONT = 2u, ML_F_END = 4u;
// (lines of a read buffer only) the line is empty: HandleUnmatchLogs' loop `while (begin < sourceVal.size())` (:350) ends BEFORE an
// empty last line of the range it was handed -- that line is neither emitted nor counted
constexpr uint32_t ML_F_EMPTY = 8u;
constexpr uint32_t kMlInherit = 0xFFFFFFFFu;
// record flag bits next to bit 0 (matched): the item belongs to the same HandleUnmatchLogs call as the record before it
constexpr uint32_t ML_REC_RUN = 2u;
constexpr int kMlThreads = 512;      // slices per launch (one thread each)
constexpr uint32_t kMlMinSlice = 16;  // items per slice at least: phase B is serial over the slices
// counts[] of a launch
enum { ML_CNT_ITEMS = 0, ML_CNT_UNMATCHED, ML_CNT_MATCHED_LOGS, ML_CNT_RECORDS, ML_CNT_OVERFLOW, ML_CNT_GAVE_UP, ML_CNT_FINAL_PARTIAL,
       ML_CNT_FINAL_START, ML_CNT_WORDS };

// one item through the automaton; sink.matched(first, last, emitter) / sink.unmatched(first, last, emitter, last line is empty)
template <class Sink>
LC_ML_HD inline void mlStep(uint32_t mode, uint32_t fl, uint32_t i, uint32_t& partial, uint32_t& ms, Sink& sink) {
    const bool hasStart = (mode & ML_HAS_START) != 0, hasCont = (mode & ML_HAS_CONT) != 0, hasEnd = (mode & ML_HAS_END) != 0;
    const bool S = (fl & ML_F_START) != 0, C = (fl & ML_F_CONT) != 0, E = (fl & ML_F_END) != 0;
    const uint32_t empty = (fl & ML_F_EMPTY) ? 1u : 0u;
    if (!partial) {
        if (hasStart ? S : C) {                                   // :176-184 / :219-230
            ms = i;
            partial = 1;
        } else if (hasEnd && !hasStart && hasCont && E) {        // continue + end (:187-192 / :231-239)
            sink.matched(i, i, i);                                // (multiStart = end + 1 is never read before it is set again)
        } else {
            sink.unmatched(i, i, i, empty);
        }
        return;
    }
    if (hasCont && C) return;                                     // :199-203 / :244-249
    if (hasEnd) {
        if (hasCont) {                                            // :206-228 / :253-265
            if (E) sink.matched(ms, i, i);
            else sink.unmatched(ms, i, i, empty);
            partial = 0;
        } else if (E) {                                           // start + end, or end alone (:229-246 / :266-280)
            sink.matched(ms, i, i);
            if (hasStart) partial = 0;
            else ms = i + 1;
        }
    } else if (!hasCont) {                                        // start only (:250-260 / :283-294)
        if (S) {
            sink.matched(ms, i - 1, i);
            ms = i;
        }
    } else {                                                      // start + continue, and no continuation (:261-282 / :295-311)
        sink.matched(ms, i - 1, i);
        if (!S) {
            sink.unmatched(i, i, i, empty);
            partial = 0;
        } else {
            ms = i;
        }
    }
}

struct MlSummary {
    uint32_t stateOut;   // bit 0: isPartialLog on exit; bit 1: an unmatched run of this slice began at the inherited multiStart
    uint32_t msOut;      // multiStart on exit (kMlInherit: unchanged)
    uint32_t recs, unmatched, logs;  // records written / unmatched items / matched logs, WITHOUT the inherited part of that run
};
struct MlEntry {
    uint32_t partial, ms, recBase;
};
struct MlJob {
    uint32_t first, last, recBase, emitter;
};

struct MlCountSink {
    uint32_t sliceStart, discard;
    uint32_t recs = 0, unm = 0, logs = 0, inherited = 0;
    LC_ML_HD void matched(uint32_t, uint32_t, uint32_t) {
        ++recs;
        ++logs;
    }
    LC_ML_HD void unmatched(uint32_t first, uint32_t last, uint32_t, uint32_t lastEmpty) {
        uint32_t cnt;
        if (first == kMlInherit) {
            inherited = 1;
            cnt = last - sliceStart + 1;
        } else {
            cnt = last - first + 1;
        }
        cnt -= lastEmpty;
        unm += cnt;
        if (!discard) recs += cnt;
    }
};

LC_ML_HD inline uint32_t mlSliceItems(uint32_t n) {
    const uint32_t per = (n + uint32_t(kMlThreads) - 1) / uint32_t(kMlThreads);
    return per < kMlMinSlice ? kMlMinSlice : per;
}
LC_ML_HD inline uint32_t mlSliceCount(uint32_t n) {
    const uint32_t per = mlSliceItems(n);
    return (n + per - 1) / per;
}

// phase A for slice t: out[0] = entering with isPartialLog 0, out[1] = with 1
LC_ML_HD inline void mlPhaseA(uint32_t t, uint32_t n, uint32_t mode, const uint8_t* flags, MlSummary out[2]) {
    const uint32_t per = mlSliceItems(n), b = t * per, e = b + per < n ? b + per : n;
    for (uint32_t s = 0; s < 2; ++s) {
        MlCountSink sink{b, (mode & ML_DISCARD) ? 1u : 0u};
        uint32_t partial = s, ms = kMlInherit;
        for (uint32_t i = b; i < e; ++i) mlStep(mode, flags[i], i, partial, ms, sink);
        out[s] = MlSummary{partial | (sink.inherited << 1), ms, sink.recs, sink.unm, sink.logs};
    }
}

// phase B: entries of all slices, the totals, the flush.  counts[ML_CNT_ITEMS .. ML_CNT_RECORDS] and the final state are
// written; a flush that is an unmatched run is returned as a job (flushJob.first != kMlInherit).
LC_ML_HD inline void mlPhaseB(uint32_t n, uint32_t mode, const MlSummary* summaries /* [slices][2] */,
                              MlEntry* entries, uint32_t* counts, MlJob& flushJob, uint32_t& flushMatchedFirst) {
    const uint32_t per = mlSliceItems(n), slices = mlSliceCount(n);
    const bool discard = (mode & ML_DISCARD) != 0;
    uint32_t partial = ((mode & (ML_HAS_START | ML_HAS_CONT)) == 0 && (mode & ML_HAS_END)) ? 1u : 0u;  // end alone (:161-165 / :174-178)
    uint32_t ms = 0, recBase = 0, unm = 0, logs = 0;
    for (uint32_t t = 0; t < slices; ++t) {
        entries[t] = MlEntry{partial, ms, recBase};
        const MlSummary& s = summaries[2 * t + partial];
        const uint32_t extra = (s.stateOut & 2u) ? t * per - ms : 0u;  // the part of the inherited run that lies before the slice
        recBase += s.recs + (discard ? 0u : extra);
        unm += s.unmatched + extra;
        logs += s.logs;
        partial = s.stateOut & 1u;
        if (s.msOut != kMlInherit) ms = s.msOut;
    }
    flushJob.first = kMlInherit;
    flushMatchedFirst = kMlInherit;
    counts[ML_CNT_FINAL_PARTIAL] = partial;
    counts[ML_CNT_FINAL_START] = ms;
    if ((mode & ML_FLUSH) && partial && ms < n) {  // :288-298 / :316-323
        if (!(mode & ML_HAS_END)) {
            flushMatchedFirst = ms;  // one matched record [ms, n-1] at slot recBase
            flushJob.recBase = recBase;
            ++recBase;
            ++logs;
        } else {
            // (the flush hands over [multiStart, size): an empty last line there is followed by its line feed, so the loop of :350
            // still sees it -- unlike an empty line at the end of a range that stops at ITS end)
            const uint32_t cnt = n - ms;
            unm += cnt;
            if (!discard && cnt) {
                flushJob = MlJob{ms, ms + cnt - 1, recBase, n};
                recBase += cnt;
            }
        }
    }
    counts[ML_CNT_ITEMS] = n;
    counts[ML_CNT_UNMATCHED] = unm;
    counts[ML_CNT_MATCHED_LOGS] = logs;
    counts[ML_CNT_RECORDS] = recBase;
}

// phase C for slice t.  W: void record(slot, first, last, matched, emitter); bool job(MlJob) -- queue a run for phase D
template <class W>
struct MlWriteSink {
    W& w;
    uint32_t sliceStart, discard, recBase;
    LC_ML_HD void matched(uint32_t first, uint32_t last, uint32_t emitter) { w.record(recBase++, first, last, 1u, emitter); }
    LC_ML_HD void unmatched(uint32_t first, uint32_t last, uint32_t emitter, uint32_t lastEmpty) {
        if (discard) return;
        const uint32_t cnt = last - first + 1 - lastEmpty;
        if (!cnt) return;
        last -= lastEmpty;
        if (first < sliceStart) {  // (at most one per slice: the queue holds one job per slice + the flush)
            w.job(MlJob{first, last, recBase, emitter});
        } else {
            for (uint32_t k = 0; k < cnt; ++k) w.record(recBase + k, first + k, first + k, k ? ML_REC_RUN : 0u, emitter);
        }
        recBase += cnt;
    }
};
template <class W>
LC_ML_HD inline void mlPhaseC(uint32_t t, uint32_t n, uint32_t mode, const uint8_t* flags, const MlEntry& in, W& w) {
    const uint32_t per = mlSliceItems(n), b = t * per, e = b + per < n ? b + per : n;
    MlWriteSink<W> sink{w, b, (mode & ML_DISCARD) ? 1u : 0u, in.recBase};
    uint32_t partial = in.partial, ms = in.ms;
    for (uint32_t i = b; i < e; ++i) mlStep(mode, flags[i], i, partial, ms, sink);
}
