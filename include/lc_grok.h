/*
 * lc_grok.h -- C ABI of the MI355X-native Grok processor.
 *
 * Replaces, behind plain pointers and sizes, the Go plugin plugins/processor/grok/processor_grok.go of the reference:
 *   lc_grok_create          <- ProcessorGrok.Init                     :62-102  (defaults + CustomPatternDir + CustomPatterns,
 *                                                                               buildPatterns :239-279, compileMatchs :335-359)
 *   lc_grok_match_device /  <- the loop of ProcessLogs -> processLog -> processGrok   :108-194: for every log the ordered
 *   lc_grok_match_host         Match list, FindStringMatch + FindNextMatch, first pattern that yields a named non-empty
 *                              capture wins.  One batch = the SourceKey values of many logs.
 *   lc_grok_process_logs_json <- ProcessLogs including the KeepSource / IgnoreParseFailure policy :115-146 (tests, tools)
 * The reference-side binding (cgo) is shown in INTEGRATION.md.
 *
 * What is matched on the device: each Match entry is expanded (%{SYNTAX:alias} -> (?P<alias>...)) and compiled with
 * LC_SYNTAX_SEARCH | LC_SYNTAX_NAMED_ONLY | LC_SYNTAX_NO_DOTALL | LC_SYNTAX_NO_MULTILINE | LC_SYNTAX_REGEXP2 -- the
 * semantics of regexp2.Compile(pattern, regexp2.RE2) as far as byte-oriented engines can give them (bytes
 * not runes, no time-outs).  Round 6: an entry with a back-reference BY NAME (\k<name>) or a general look-around runs on the
 * device backtracking engine (LC_ENGINE_BT) and the handle walks its list entry by entry; numbered back-references stay
 * refused (regexp2 numbers unnamed groups first).  A Match entry the device engines cannot run makes lc_grok_create FAIL
 * with the reason: there is no CPU path.
 */
#ifndef LC_GROK_H
#define LC_GROK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lc_grok lc_grok_t;
typedef struct lc_grok_result lc_grok_result_t;

/* config_json: the plugin's JSON detail, same keys as the Go struct (processor_grok.go:42-53):
 *   {"CustomPatternDir": [...], "CustomPatterns": {...}, "SourceKey": "content", "Match": [...],
 *    "IgnoreParseFailure": true, "KeepSource": true, "TimeoutMilliSeconds": 0, ...}
 * Returns LC_OK (0) or an LC_ERR_* code (lc_regex_gpu.h); err receives the message ("no pattern found for X",
 * "cannot build patterns because cyclic exist...", "Match[3]: tdfa: state limit exceeded; nfa: ..."). */
int lc_grok_create(const char* config_json, size_t config_len, lc_grok_t** out, char* err, size_t errcap);
void lc_grok_free(lc_grok_t* g);
/* (No reference counterpart: compileMatchs, processor_grok.go:335-359, compiles each entry once and that is all.)
 * Entries that run on the NFA engine also get an ANCHORED search (LC_SYNTAX_SEARCH | LC_SYNTAX_PREFIX: a tagged DFA with its tables
 * in L2), tried first on every value -- log formats match from the first byte.  Those automata take seconds to build, so they are
 * compiled behind lc_grok_create on a warm-up thread and join the matcher as they arrive; results never depend on them.  This call
 * returns once the thread is done (benchmarks; tests that want to pin which path runs).  Config key "AnchoredFirst": false = none. */
void lc_grok_wait_ready(lc_grok_t* g);

int lc_grok_match_count(const lc_grok_t* g);                       /* len(Match) */
const char* lc_grok_expanded(const lc_grok_t* g, int i);           /* Match[i] after denormalizePattern (:282-316) */
const char* lc_grok_processed(const lc_grok_t* g, const char* name); /* processedPatterns[name], NULL if unknown */
/* denormalizePattern on any expression against this handle's library, without compiling it (malloc'ed, release with
 * lc_grok_free_string; NULL + err on "no pattern found for X" / "invalid pattern X") */
char* lc_grok_denormalize(lc_grok_t* g, const char* pattern, char* err, size_t errcap);
/* The literal index of the Match list (introspection for tests): the required literals of all entries as one Aho-Corasick DFA
 * (csrc/grok_literal_layout.h); the device walks it once per value and gets a 64-bit mask "which entries' literals does this
 * value contain".  *words = NULL when the list is not indexable (more than 64 entries, fewer than two literals).  The pointer
 * stays valid until lc_grok_free. */
int lc_grok_literal_index(lc_grok_t* g, const uint32_t** words, size_t* nwords);
int lc_grok_engine(const lc_grok_t* g, int i);                     /* LC_ENGINE_TDFA / LC_ENGINE_NFA chosen for Match[i] */
/* diagnostics: how Match[i] is run -- out[0] engine of the search form, [1] states of its tagged DFA (0: none), [2] 1 = tables in
 * LDS / 2 = in global memory, [3] prefix-screen states, [4] relaxed-screen states, [5] anchored search present (they arrive behind
 * lc_grok_create: lc_grok_wait_ready), [6] its states, [7] LDS / global, [8] bytes of its global tables, [9] its offset registers,
 * [10] its byte classes, [11] bytes of the search form's global tables. */
int lc_grok_entry_info(lc_grok_t* g, int i, uint32_t out[12]);

/* Emitted keys.  A match of Match[i] writes one capture column per NAMED group; columns that share a name are one field
 * (regexp2 merges same-named groups, the value is the one captured furthest along).  Keys are already mapped back through
 * nameToAlias (:326-332), e.g. group `english_word` -> key "english-word". */
int lc_grok_key_count(const lc_grok_t* g);                         /* distinct emitted keys over all Match entries */
const char* lc_grok_key(const lc_grok_t* g, int key);
int lc_grok_column_count(const lc_grok_t* g, int i);               /* named groups of Match[i] */
int lc_grok_column_key(const lc_grok_t* g, int i, int column);     /* key index of that column */
int lc_grok_row_ints(const lc_grok_t* g);                          /* ints per capture row: 2 * (1 + max column count) */

/* ---- device-resident batch (inputs and outputs in HBM; nothing is copied) ------------------------------------------
 * d_data: the values' bytes; 16-byte aligned, its allocation ending on a 16-byte boundary at or behind the last value (the kernels
 *             read aligned 4- and 16-byte units around a value's ends: lc_regex_gpu.h, "the contract for d_data"; any hipMalloc /
 *             torch allocation qualifies).
 * d_off/d_len: uint32[n] byte offsets / lengths of the SourceKey values inside d_data.
 * d_pattern  int32[n]            winning Match index, -1 = matchFail, -2 = undecidable on the device (the NFA engine ran out
 *                                of threads on this value and nothing settled it), -3 = an entry gave up on the value (the
 *                                decide kernel ran out of budget: the reference's matchTimeOut, processor_grok.go:156-160 --
 *                                no further entry is tried, the value counts as a failed parse)
 * d_first    int32[n][row]       capture row of the FIRST match that contributed a non-empty named capture:
 *                                [whole.b, whole.e, col0.b, col0.e, ...], -1 = column did not take part
 * d_extra    int32[cap][2+row]   further contributing matches of the same value (FindNextMatch): [line, seq>=1, row...]
 * d_nextra   uint32[1]           rows written to d_extra; > cap means d_extra was too small (LC_ERR_OVERFLOW is returned
 *                                and the call must be repeated with a d_extra of at least that many rows; the default path keeps
 *                                the further matches of every candidate entry in temporary rows until the winner is known, so
 *                                the number reported with LC_ERR_OVERFLOW can exceed the rows finally written)
 * d_scratch  lc_grok_scratch_bytes(g, n) bytes
 * The call enqueues on `stream` (and on a few worker streams of its own that fork from it and join it) and returns after the
 * last kernel has finished.  Default path: THREE host synchronisations per batch (candidates per entry; round 0's counts; results) --
 * a fourth when a value's remainder behind a first match passes its entry's screen; the number of values still in play after each
 * search round stays on the device; an entry that needs more rounds than it queued ahead costs one more per extra round, once
 * (csrc/grok_device.hip).  Config key "Speculative": false (or a list of more than 64 entries)
 * selects the sequential walk of the list with a host round trip per filter / screen / round. */
size_t lc_grok_scratch_bytes(const lc_grok_t* g, uint32_t n);
int lc_grok_match_device(lc_grok_t* g, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len, uint32_t n,
                         int32_t* d_pattern, int32_t* d_first, int32_t* d_extra, uint32_t extra_cap, uint32_t* d_nextra,
                         void* d_scratch, size_t scratch_bytes, void* stream);

/* what the calling thread's last lc_grok_match_device / lc_grok_match_host batch did:
 * out[0] host synchronisations, [1] Match entries with at least one candidate, [2] (entry, value) pairs evaluated,
 * [3] entries that needed more rounds than queued ahead, [4] 1 = speculative path */
void lc_grok_last_batch_stats(uint32_t out[5]);

/* ---- host batch: copies in, matches on the device, returns the fields of every value in emission order ---------------
 * Callable from any number of runner threads on ONE handle, one group per call, synchronous -- the contract of
 * core/runner/ProcessorRunner.cpp:138-142.  Groups of threads that call in together travel as ONE device batch (group commit:
 * csrc/group_combiner.hpp; a batch costs the device about the same from 1 000 to 16 000 values): a worker thread per (handle, device)
 * runs the batches, one at a time; a batch starts when the device is free and the threads seen in the last three batches have arrived,
 * or 100 us after the last arrival (LC_GROK_GAP_US), at most 500 us after the device became free (LC_GROK_LINGER_US).  One thread
 * alone never waits.  lc_grok_combiner_stats: out = {batches, groups, values, most groups in one batch, batches started by the
 * linger's timeout; then the worker's microseconds: idle, lingering, laying out the staging, the callers' gather,
 * the device trip, the callers taking their rows} since the handle was created. */
int lc_grok_combiner_stats(lc_grok_t* g, uint64_t out[11]);

/* The LAZY automata of the entries that do not determinise (include/lc_regex_gpu.h lc_regex_lazy_train; config key "LazyTdfa", default
 * true).  A background thread of the handle builds them along the handle's own traffic: lc_grok_match_host / lc_grok_match_device OFFER
 * a copy of up to 4 096 values of a batch (a window that moves through the batch from offer to offer) when the trainer is idle -- the
 * first 64 batches at once, later ones every 200 ms at most.
 * Results never depend on it.  lc_grok_lazy_settle blocks until the trainer has nothing to do (LC_OK) or the timeout is over (LC_ERR_ARG):
 * benchmarks and tests call it between warm-up batches.  lc_grok_lazy_stats: out = {automata with a lazy front in use, builds, values
 * offered (summed over the automata), values kept in their samples, batches the trainer has taken}. */
int lc_grok_lazy_settle(lc_grok_t* g, uint32_t timeout_ms);
int lc_grok_lazy_stats(lc_grok_t* g, uint64_t out[5]);
int lc_grok_match_host(lc_grok_t* g, const uint8_t* data, const uint32_t* off, const uint32_t* len, uint32_t n,
                       int32_t* pattern /* [n], as d_pattern */, lc_grok_result_t** result);
/* fields of value i: indices field_off[i] .. field_off[i+1]) into key[] / begin[] / end[] (byte range inside value i) */
void lc_grok_result_arrays(const lc_grok_result_t* r, const uint32_t** field_off, const uint32_t** key,
                           const uint32_t** begin, const uint32_t** end);
void lc_grok_result_free(lc_grok_result_t* r);

/* ---- whole-plugin behaviour on JSON logs (tests / tools): [[["content","..."],["k","v"]], ...] in, same shape out.
 * *out_json is malloc'ed; release with lc_grok_free_string. */
int lc_grok_process_logs_json(lc_grok_t* g, const char* logs_json, size_t len, char** out_json);
void lc_grok_free_string(char* s);

#ifdef __cplusplus
}
#endif
#endif /* LC_GROK_H */
