/*
 * lc_multiline.h -- C ABI of the multiline splitter on the device.
 *
 * Replaces the record-boundary logic of core/plugin/processor/inner/ProcessorSplitMultilineLogStringNative.cpp:
 *   lc_multiline_create      <- ProcessorSplitMultilineLogStringNative::Init :36-84 (the three patterns compiled as written)
 *                               + MultilineOptions::Init (core/file_server/MultilineOptions.cpp:100-222): custom mode,
 *                               UnmatchedContentTreatment, IsMultiline
 *   lc_multiline_split_host  <- ProcessEvent :126-300 + HandleUnmatchLogs :341-380 for ONE source value (a read buffer of
 *                               '\n'-separated lines): which lines form one log, which are unmatched
 * The buffer goes up ONCE; its lines are found by the split kernels; the per-line question of the reference --
 * BoostRegexSearch(line, pattern) = regex_search with match_continuous (StringTools.cpp:263-289) -- is answered for ALL lines
 * by one status-only launch per configured pattern (LC_SYNTAX_PREFIX) over the same device copy; the start/continue/end walk
 * runs as a scan over three flag bits per line on the device (lc_multiline_bounds_device) and only the records come back.
 * Evaluating every pattern on every line is a superset of what the reference evaluates lazily; matches have no side
 * effects, so the records are the same.  lc_multiline_process_group builds the output events (CreateNewEvent :302-339).
 */
#ifndef LC_MULTILINE_H
#define LC_MULTILINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lc_multiline lc_multiline_t;

typedef struct lc_ml_record {
    uint32_t begin, length;  /* byte range inside the source value (line feeds inside a multi-line log included) */
    uint32_t matched;        /* bit 0: 1 = a log delimited by the patterns, 0 = an unmatched line kept as a single-line log; LC_ML_RUN;
                              * LC_ML_LAST: emitted with isLastLog = true (CreateNewEvent :327-329: its position length runs
                              * to the end of the source event) */
} lc_ml_record_t;
#define LC_ML_LAST 0x80000000u
#define LC_ML_RUN 0x2u /* an unmatched line handed to the same HandleUnmatchLogs call (:341-380) as the record before it */

/* config_json: {"StartPattern": "...", "ContinuePattern": "...", "EndPattern": "...",
 *               "UnmatchedContentTreatment": "single_line" | "discard"}   (keys of the Multiline object, custom mode) */
int lc_multiline_create(const char* config_json, size_t config_len, lc_multiline_t** out, char* err, size_t errcap);
void lc_multiline_free(lc_multiline_t* m);
/* MultilineOptions::IsMultiline() (:203-205, decided on the patterns with a trailing '$' / ".*" stripped): 0 means the
 * input plugin would not install the multiline splitter at all */
int lc_multiline_is_multiline(const lc_multiline_t* m);
/* "" or one line per pattern that was ignored because it is not a valid regex (the reference warns and carries on,
 * MultilineOptions.cpp:109-118).  lc_multiline_create fails for a pattern that is valid but not runnable on the device
 * (there is no CPU path), and for a config without a usable StartPattern or EndPattern (the reference never builds the
 * processor for one: InputFile.cpp:225). */
const char* lc_multiline_warnings(const lc_multiline_t* m);
/* which patterns the processor works with (bit 0 start, bit 1 continue, bit 2 end): Has*Pattern() = the string as written
 * is not empty (ProcessorSplitMultilineLogStringNative.h:68-70) */
int lc_multiline_patterns(const lc_multiline_t* m);

/* counters[3] = input lines, unmatched lines, matched logs (mMatchedLinesTotal = input - unmatched).
 * *records is malloc'ed (release with lc_multiline_free_records). */
int lc_multiline_split_host(lc_multiline_t* m, const uint8_t* data, uint32_t nbytes, lc_ml_record_t** records,
                            uint32_t* nrecords, uint32_t counters[3]);
void lc_multiline_free_records(lc_ml_record_t* r);

/* The record scan on its own, for items (lines of a buffer, events of a group) whose start / continue / end answers are already on
 * the device -- what lc_multiline_split_host and the merge processor run behind their status-only match launches
 * (csrc/multiline_kernel.hpp; the walk of ProcessEvent :161-298 / MergeLogsByRegex :161-330 as a scan, csrc/multiline_scan.hpp).
 *   mode      LC_ML_HAS_* of the patterns in use | LC_ML_DISCARD (UnmatchedContentTreatment discard) | LC_ML_FLUSH (the input ends
 *             here: the log under construction is emitted / handed to HandleUnmatchLogs, :288-298)
 *   d_start, d_cont, d_end   status bytes of the three match launches (NULL for a pattern that is not in use)
 *   d_nitems  optional device-side item count (<= max_items)
 *   d_off     the offsets[n+1] + one separator byte table of lc_split_lines_device: records are BYTE ranges of the source value and
 *             carry LC_ML_LAST as CreateNewEvent needs it; NULL: records are ITEM ranges (begin = first item, length = items)
 *   d_flags   max_items bytes of scratch; records / counts may be device or pinned host memory; at most record_cap records are
 *             written (counts[LC_ML_CNT_RECORDS] says how many there are)
 * Asynchronous on `stream`. */
#define LC_ML_HAS_START 1u
#define LC_ML_HAS_CONT 2u
#define LC_ML_HAS_END 4u
#define LC_ML_DISCARD 8u
#define LC_ML_FLUSH 16u
enum { LC_ML_CNT_ITEMS = 0, LC_ML_CNT_UNMATCHED, LC_ML_CNT_MATCHED_LOGS, LC_ML_CNT_RECORDS, LC_ML_CNT_OVERFLOW, LC_ML_CNT_GAVE_UP,
       LC_ML_CNT_FINAL_PARTIAL, LC_ML_CNT_FINAL_START, LC_ML_CNT_WORDS };
int lc_multiline_bounds_device(uint32_t mode, const uint8_t* d_start, const uint8_t* d_cont, const uint8_t* d_end,
                               const uint32_t* d_nitems, uint32_t max_items, const uint32_t* d_off, uint32_t nbytes, uint8_t* d_flags,
                               lc_ml_record_t* records, uint32_t record_cap, uint32_t* counts, void* stream);
/* The same scan on the host, slice by slice as the kernel's threads do it (the shared code of csrc/multiline_scan.hpp): `flags`
 * holds one byte per item (bit 0 start, bit 1 continue, bit 2 end matched; bit 3: with `off`, the line is empty).  For the CPU test suite -- it pins the scan against
 * the oracle's sequential walk without a device; no product path calls it. */
int lc_multiline_bounds_model(uint32_t mode, const uint8_t* flags, uint32_t nitems, const uint32_t* off, uint32_t nbytes,
                              lc_ml_record_t* records, uint32_t record_cap, uint32_t counts[LC_ML_CNT_WORDS]);

/* ProcessorSplitMultilineLogStringNative::Process :95-112 on a whole logtail::PipelineEventGroup* (lc_group_native() of a
 * fixture group, or the agent's own group): every log event that holds exactly the SourceKey content is replaced by one event
 * per record -- CreateNewEvent :302-339: a LogEvent (or, with EnableRawContent, a RawEvent) whose content is a VIEW into the
 * source value, the source event's timestamp, position = (source offset + record begin, record length + 1, or the rest of the
 * source event for a record emitted with isLastLog), and the group's LOG_FILE_OFFSET_KEY content when that metadata is set.
 * Other events pass through untouched (:133-156).  "SourceKey" (default "content") and "EnableRawContent" are read from the
 * config given to lc_multiline_create (:41-65).  counters[3] accumulate matched lines, unmatched lines, matched events. */
int lc_multiline_process_group(lc_multiline_t* m, void* pipeline_event_group);
int lc_multiline_counters(const lc_multiline_t* m, uint64_t counters[3]);

/* processor_merge_multiline_log_native (core/plugin/processor/inner/ProcessorMergeMultilineLogNative.cpp) -- what file
 * pipelines run behind the line splitter nowadays: ONE LINE PER EVENT comes in, events are merged into logs.
 *   lc_merge_multiline_create         <- Init :33-78: {"SourceKey", "MergeType": "regex" | "flag", + the Multiline keys above}
 *   lc_merge_multiline_process_group  <- Process :80-92 on a logtail::PipelineEventGroup*:
 *        "regex": MergeLogsByRegex :161-330 -- the start / continue / end answers for ALL events come from one status-only device
 *                 launch per pattern over the events' values; the state machine, MergeEvents :332-358 (the first event's value
 *                 is extended in place over the following ones, a line feed written back between them) and HandleUnmatchLogs
 *                 :360-392 (single_line: kept one by one; discard: dropped) are the reference's
 *        "flag":  MergeLogsByFlag :113-159 (events carrying the "P" content are partial; runs only when the group has the
 *                 HAS_PART_LOG metadata, which it then clears) -- no regex involved
 *   counters[2] = merged events, unmatched events (:74-75). */
typedef struct lc_merge_multiline lc_merge_multiline_t;
int lc_merge_multiline_create(const char* config_json, size_t config_len, lc_merge_multiline_t** out, char* err, size_t errcap);
void lc_merge_multiline_free(lc_merge_multiline_t* p);
/* "regex": the patterns the processor matches with (bit 0 start, bit 1 continue, bit 2 end; 0 in flag mode), and the warnings of its
 * Init.  NOT the splitter's rule: this processor matches with MultilineOptions' own regexes (Get*PatternReg(), :219-224,244-262), i.e.
 * with what ParseRegex (MultilineOptions.cpp:250-266) leaves of a pattern after stripping one trailing '$' and all trailing ".*" --
 * "END$" accepts a line that merely begins with END, ".*" is no pattern at all -- and without ContinuePattern when all three are given
 * (:185-200: a warning, listed by lc_merge_multiline_warnings).  ProcessorSplitMultilineLogStringNative compiles the strings as
 * written (:66-76) and keeps all three; the two processors differ on such configs in the reference, and they differ here. */
int lc_merge_multiline_patterns(const lc_merge_multiline_t* p);
const char* lc_merge_multiline_warnings(const lc_merge_multiline_t* p);
int lc_merge_multiline_process_group(lc_merge_multiline_t* p, void* pipeline_event_group);
int lc_merge_multiline_counters(const lc_merge_multiline_t* p, uint64_t counters[2]);

#ifdef __cplusplus
}
#endif
#endif /* LC_MULTILINE_H */
