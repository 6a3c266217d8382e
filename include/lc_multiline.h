/*
 * lc_multiline.h -- C ABI of the multiline splitter on the device.
 *
 * Replaces the record-boundary logic of core/plugin/processor/inner/ProcessorSplitMultilineLogStringNative.cpp:
 *   lc_multiline_create      <- ProcessorSplitMultilineLogStringNative::Init :36-84 (the three patterns compiled as written)
 *                               + MultilineOptions::Init (core/file_server/MultilineOptions.cpp:100-222): custom mode,
 *                               UnmatchedContentTreatment, IsMultiline
 *   lc_multiline_split_host  <- ProcessEvent :126-300 + HandleUnmatchLogs :341-380 for ONE source value (a read buffer of
 *                               '\n'-separated lines): which lines form one log, which are unmatched
 * The per-line question of the reference -- BoostRegexSearch(line, pattern) = regex_search with match_continuous
 * (StringTools.cpp:263-289) -- is answered for ALL lines of the buffer by one device launch per configured pattern
 * (LC_SYNTAX_PREFIX, status bytes only); the start/continue/end state machine then runs over three flag bytes per line.
 * Evaluating every pattern on every line is a superset of what the reference evaluates lazily; matches have no side
 * effects, so the records are the same.  Building the output events (CreateNewEvent :302-339) stays with the caller.
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
    uint32_t matched;        /* 1: a log delimited by the patterns; 0: an unmatched line kept as a single-line log */
} lc_ml_record_t;

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

#ifdef __cplusplus
}
#endif
#endif /* LC_MULTILINE_H */
