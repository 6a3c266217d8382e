/*
 * lc_regex_gpu.h -- C ABI of the MI355X-native regex parse engine (liblc_regex_gpu.so).
 *
 * This is the drop-in boundary for the matching arithmetic of LoongCollector's
 * processor_parse_regex_native.  Each entry point names the reference interface it replaces
 * (paths relative to the reference tree).  Plain pointers and sizes only; no C++/torch types.
 *
 *   reference                                                             this ABI
 *   --------------------------------------------------------------------  ---------------------------
 *   boost::regex ctor, one per runner thread                              lc_regex_compile
 *     core/plugin/processor/ProcessorParseRegexNative.cpp:64-67
 *   IsRegexValid(regex)   core/common/ParamExtractor.cpp:199-209          lc_regex_compile (rc != 0)
 *   what.size() (= mark_count()+1)   ProcessorParseRegexNative.cpp:227    lc_regex_mark_count
 *   BoostRegexMatch(buf,len,reg,exception,what,match_default)            lc_regex_match_device /
 *     core/common/StringTools.cpp:183-211, called per event at            lc_regex_match_host
 *     ProcessorParseRegexNative.cpp:194 inside the loop at :115-124       (whole event group per call)
 *   what[i+1].begin()/length()       ProcessorParseRegexNative.cpp:249-251  caps[line][2*i], [2*i+1]
 *
 * Semantics: boost::regex_match with Perl syntax and match_default -- the WHOLE line must match,
 * leftmost-first (backtracking) sub-match rules, byte-oriented, '.' matches '\n', '^'/'$' match at
 * embedded line separators.  Output per line: status (LC_MATCH / LC_NOMATCH) and, for every capture
 * group g = 1..ngroups, the byte range [begin,end) relative to the start of the line; a group that did
 * not participate is reported as (-1,-1) (the reference turns that into an empty value positioned at
 * end-of-input, ProcessorParseRegexNative.cpp:250 -- see lc_processor.h for the stitch).
 *
 * The engine has NO CPU execution path: if no HIP device is usable the match calls return
 * LC_ERR_NO_DEVICE and the caller must fail loudly.
 */
#ifndef LC_REGEX_GPU_H
#define LC_REGEX_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LC_REGEX_ABI_VERSION 1

typedef struct lc_regex lc_regex_t;

/* syntax flags (0 = boost::regex(str) defaults) */
enum {
    LC_SYNTAX_ICASE = 1u << 0,          /* (?i) */
    LC_SYNTAX_NO_DOTALL = 1u << 1,      /* (?-s): '.' does not match '\n' */
    LC_SYNTAX_NO_MULTILINE = 1u << 2,   /* (?-m): '^'/'$' only at the ends of the line */
    LC_SYNTAX_EXTENDED = 1u << 3,       /* (?x) */
    LC_SYNTAX_NAMED_ONLY = 1u << 4,     /* unnamed groups do not capture (Grok / regexp2 ExplicitCapture-style) */
    LC_SYNTAX_SEARCH = 1u << 5,         /* leftmost-first SEARCH instead of whole-line match (Go processor_regex without
                                           FullMatch, regex.go:105-129): compiled as (?s:.*?)(re)(?s:.*), so group 1 is
                                           the whole match and the pattern's own groups are 2..mark_count */
    LC_SYNTAX_PREFIX = 1u << 7,         /* the pattern must match a PREFIX of the line: boost::regex_search with
                                           match_continuous, what the multiline splitter asks per line for its start /
                                           continue / end patterns (StringTools.cpp:263-289 called from
                                           ProcessorSplitMultilineLogStringNative.cpp:184-272).  Compiled as (re)(?s:.*);
                                           captures of the leftmost-first prefix match are reported as usual.
                                           Together with LC_SYNTAX_SEARCH: the ANCHORED search -- the match must start at
                                           the first byte, groups as in a search (group 1 = the whole match).  It finds what
                                           the search finds whenever the search's leftmost match starts at byte 0, from a
                                           far smaller automaton (no "anywhere" prefix): the Grok matcher tries it first */
    LC_SYNTAX_REGEXP2 = 1u << 6         /* escape dialect of github.com/dlclark/regexp2 with the RE2 option (Go Grok,
                                           processor_grok.go:343): \s = [\t\n\f\r ]; \< \> \` \' are literals */
};

/* device engines */
enum {
    LC_ENGINE_AUTO = 0,  /* TDFA if it fits the limits, else NFA */
    LC_ENGINE_TDFA = 1,  /* tagged DFA: one line per lane, tables in LDS */
    LC_ENGINE_NFA = 2,   /* follow NFA: one line per wavefront, one lane per live thread */
    LC_ENGINE_DECIDE = 3, /* match calls only (not lc_regex_compile): skip the thread-list kernels and settle EVERY line with the
                            depth-first decide kernel that normally only sees the lines they overflow on.  Slow (one lane per
                            line); for cross-checking the engines against each other and for diagnosis */
    LC_ENGINE_BT = 4      /* round 6: the device BACKTRACKING engine (csrc/bt_vm.hpp): one line per lane, an instruction program and
                            an explicit stack in HBM -- what boost::regex_match itself does (StringTools.cpp:183-211).  Chosen by
                            LC_ENGINE_AUTO for patterns no automaton runs: back-references (\1 .. \N, \k<name>, \g{-1}), general
                            look-arounds ((?=a+b), (?<=ab|cd)), trees the position automaton cannot express ((a*)*); may be asked
                            for any pattern it can run, which is how the tests cross-check it against the automata.  A line that
                            runs out of its step budget or stack is LC_GAVE_UP.  Not under LC_SYNTAX_NAMED_ONLY / LC_SYNTAX_REGEXP2
                            (the Go plugins' dialects) */
};

/* per-line status bytes */
enum {
    LC_NOMATCH = 0,
    LC_MATCH = 1,
    LC_OVERFLOW = 2, /* transient, NFA engine only: more live threads than the thread-list kernels hold (128; 64 with atomic
                        groups).  Every match entry point launches the depth-first decide kernel behind them, which settles
                        these lines -- a caller only ever sees this value if the decide pass was switched off (LC_NFA_NO_DECIDE) */
    LC_GAVE_UP = 3   /* the decide kernel ran out of its step budget (plain backtracking, when the memo of an extremely long
                        line does not fit the scratch pool) or of scratch: the counterpart of boost's complexity-exceeded
                        exception, which the reference counts as a parse failure (core/common/StringTools.cpp:200-205).
                        Processors treat it as a failed parse AND count it separately (never silently) */
};

/* return codes */
enum {
    LC_OK = 0,
    LC_ERR_SYNTAX = 1,       /* invalid regex (reference: IsRegexValid false -> Init fails) */
    LC_ERR_UNSUPPORTED = 2,  /* valid Perl regex but not executable bit-exactly on the device engines */
    LC_ERR_NO_DEVICE = 3,
    LC_ERR_HIP = 4,
    LC_ERR_ARG = 5,
    LC_ERR_OVERFLOW = 6      /* a caller-provided output buffer was too small; the needed size has been reported */
};

typedef struct lc_regex_info {
    int engine;            /* LC_ENGINE_TDFA, LC_ENGINE_NFA or LC_ENGINE_BT */
    int mark_count;        /* capture groups */
    uint32_t positions;    /* follow-NFA positions (byte-consuming steps) */
    uint32_t states;       /* TDFA states (0 for NFA engine) */
    uint32_t classes;      /* byte equivalence classes */
    uint32_t registers;    /* TDFA offset registers per line */
    uint32_t table_bytes;  /* bytes staged into LDS per workgroup */
} lc_regex_info_t;

/* Compile `pattern` (pattern_len bytes) for the requested engine.  On failure returns LC_ERR_SYNTAX /
 * LC_ERR_UNSUPPORTED and writes a NUL-terminated message into err (if errcap > 0). */
int lc_regex_compile(const char* pattern, size_t pattern_len, uint32_t syntax_flags, int engine,
                     lc_regex_t** out, char* err, size_t errcap);
void lc_regex_free(lc_regex_t* re);

int lc_regex_mark_count(const lc_regex_t* re);
/* name of group g (1-based) or NULL */
const char* lc_regex_group_name(const lc_regex_t* re, int g);
int lc_regex_info(const lc_regex_t* re, lc_regex_info_t* out);

/* Read-only view of a compiled host table (introspection for tests and docs; the pointer stays valid until
 * lc_regex_free).  `which` is one of LC_TABLE_*.  Returns LC_ERR_ARG for a table the engine does not have. */
enum {
    LC_TABLE_CLASSMAP = 0,    /* u8[256] */
    LC_TABLE_TDFA_TRANS = 1,  /* u32[states*classes] */
    LC_TABLE_TDFA_OPSSTART = 2, /* u32[lists+1] */
    LC_TABLE_TDFA_OPS = 3,    /* u16[] */
    LC_TABLE_TDFA_FINALID = 4, /* u16[states] */
    LC_TABLE_TDFA_FINALMAP = 5, /* u8[nfinal*slots] */
    LC_TABLE_TDFA_HEADER = 6, /* u32[8]: states, classes, registers, slots, start state, 0,0,0 */
    LC_TABLE_NFA_BLOB = 7,    /* the packed NFA program uploaded to the device (see csrc/device_tables.h) */
    LC_TABLE_TDFA_STARTAFTER = 8, /* u32[classes]: search patterns only -- state a resumed search starts in, by the class
                                    of the byte before the resume point (lc_regex_match_device_from) */
    LC_TABLE_TDFA_BLOB = 9,      /* the packed TDFA tables uploaded to the device (csrc/device_tables.h) */
    LC_TABLE_TDFA_WIDE_BLOB = 10, /* small automata only: the same with byte-indexed rows, for the 1024-lane kernel */
    LC_TABLE_TDFA_L2_BLOB = 11,  /* automata too large for the LDS kernels: the tables as the global-memory kernel reads them
                                    (csrc/tdfa_l2_layout.h); such a handle has no LC_TABLE_TDFA_BLOB */
    LC_TABLE_LAZY_TDFA_BLOB = 12, /* thread-list handles that have been trained (lc_regex_lazy_train): the partial automaton in the same
                                    layout, TL_MISS != 0; valid until the handle's next training call */
    LC_TABLE_BT_BLOB = 13        /* LC_ENGINE_BT handles: the backtracking program uploaded to the device (csrc/bt_vm.hpp) */
};
int lc_regex_table(const lc_regex_t* re, int which, const void** data, size_t* bytes);

/* A LAZY / partial tagged DFA in front of the thread-list engine.  A pattern whose automaton does not determinise within the limits
 * (LC_ENGINE_NFA handles: log formats of 170 000+ states) runs the thread-list kernels, two orders of magnitude slower per byte than a
 * table walk -- but the values it sees visit a few thousand states.  lc_regex_lazy_train adds `n` values in HOST memory to the handle's
 * sample (bounded: 8 192 values, 12 MiB) and (re)builds the automaton ALONG them: only the transitions the sample takes are computed,
 * every other one leads to a MISS state (csrc/tdfa.cpp buildTdfaLazy).  Every match call on the handle then walks all values through the
 * partial automaton first; only values that step on a MISS are walked by the thread-list kernels of the same call, from their first
 * byte.  Results never depend on the sample: what the partial automaton decides it decides as the complete one would, and as the
 * thread-list engine does.  Handles of other engines: LC_OK, nothing done.  Safe beside match calls on other threads.
 * out (optional) = {states, transitions computed, sample values, sample values the tables still miss, 1 = in use}.
 * LC_LAZY_TDFA=0 (environment, read per call) takes the partial automata out of every call (A/B, parity tests).
 * Replaces nothing in the reference: boost and regexp2 backtrack (core/common/StringTools.cpp:183-211, processor_grok.go:156-176). */
int lc_regex_lazy_train(lc_regex_t* re, const uint8_t* data, const uint32_t* off, const uint32_t* len, uint32_t n, uint64_t out[5]);

/* A SCREEN for a pattern whose own tagged DFA is too large for LDS: a TDFA-engine handle for the longest prefix of the
 * pattern's top-level concatenation (captures dropped) that stays within max_states / max_table_bytes, compiled as a
 * search.  Every match of the pattern contains a match of that prefix, so lines the screen rejects (status only,
 * ngroups = 0) need not be handed to the NFA engine.  NULL if no useful prefix exists.  Free with lc_regex_free. */
lc_regex_t* lc_regex_compile_screen(const char* pattern, size_t pattern_len, uint32_t syntax_flags, uint32_t max_states,
                                    size_t max_table_bytes);
/* A second screen, over the WHOLE pattern (what stands in front of an entry of the Grok plugin's ordered Match loop,
 * plugins/processor/grok/processor_grok.go:148-194, when the entry runs on the NFA engine -- the reference asks regexp2 for every
 * (value, entry) pair, :156): captures, assertions and atomic brackets dropped, long counters opened, and every
 * sub-expression that is still too large replaced by "one of the bytes it can start with, then any of the bytes it can
 * contain" until the automaton fits.  The relaxed pattern matches wherever the pattern does, so what it rejects cannot match.
 * NULL if even the coarsest relaxation is too large or accepts the empty string.  Free with lc_regex_free. */
lc_regex_t* lc_regex_compile_relaxed_screen(const char* pattern, size_t pattern_len, uint32_t syntax_flags, uint32_t max_states,
                                            size_t max_table_bytes);
/* One pass of a relaxed screen over device-resident values (one value per lane, the yes/no DFA's table read through L2):
 *   d_lines (optional) uint32[n]: the values to look at (NULL: 0..n-1);  d_len: their lengths (required);
 *   d_out   uint32[n]: receives the values the screen ACCEPTS, in no particular order;
 *   d_count uint32: incremented by the number of accepted values (the caller zeroes it).
 * Only for handles made by lc_regex_compile_relaxed_screen (LC_ERR_UNSUPPORTED otherwise). */
int lc_regex_screen_device(lc_regex_t* screen, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len, uint32_t n,
                           const uint32_t* d_lines, uint32_t* d_out, uint32_t* d_count, void* stream);

/* The longest byte string every match of the pattern must contain (*len = 0: none is certain).  A value without it cannot
 * match; the Grok matcher uses it to skip the automaton for most (value, Match pattern) pairs. */
const uint8_t* lc_regex_required_literal(const lc_regex_t* re, size_t* len);

/* Groups written "(?=(S*))" -- a look-ahead that always holds and only captures (Grok: "(?=%{GREEDYDATA:message})").  The
 * tables stamp where such a group begins; its end is the end of the run of S bytes that starts there, which the match
 * entry points fill in after the automaton (a reader of lc_regex_table has to do the same).  Returns how many such groups
 * the pattern has; writes up to `cap` group indices (as in the caps rows) and 32-byte sets (bit b of byte b/8 = byte b). */
int lc_regex_run_captures(const lc_regex_t* re, int32_t* groups, uint8_t* sets, int cap);

/* Atomic groups / possessive quantifiers of the pattern: *kept = instances the engines honour (ordered commit pass), *elided = groups
 * that were turned into plain groups at compile time because they provably change no match and no capture (csrc/atomic_elide.cpp:
 * whole-line language unchanged and prefix-free, nothing captured or asserted inside).  Diagnostics and tests. */
void lc_regex_atomic_groups(const lc_regex_t* re, uint32_t* kept, uint32_t* elided);

/* Compiled automata across process restarts.  Determinising a large pattern costs seconds of a host core (an anchored Grok format: 2-8 s;
 * finding out that one does NOT fit the limits: as long), and the agent pays it at every start and every pipeline reload in a new
 * process.  With a cache directory set, every tagged-DFA / screen-DFA construction is looked up first -- under a hash of its input (the
 * follow NFA, the limits, the stamp of this build of the library) -- and stored afterwards: tables, or the verdict of a construction
 * that ran into its limits.  Tables are bit-identical to a fresh construction; a missing, truncated or foreign file is ignored.
 * dir = NULL or "" switches the cache off (default).  Process-wide.  Environment: LC_TABLE_CACHE_DIR.  Grok config key: "CacheDir".
 * lc_runtime_table_cache_stats: {hits, misses, files stored, failures recalled} of this process.
 * lc_runtime_table_cache_stamp: the stamp that is part of every key -- a hash of the library SOURCES that shape the tables (the
 * constructions, their structures, the file format), so that files written by a library with another construction are never read.
 * A file also carries its key and a checksum of its payload, and every index in it is range-checked on load: anything else is a miss.
 * A process under the construction's A/B switches (LC_TDFA_NO_DSE, LC_TDFA_NO_MINIMIZE) bypasses the cache. */
int lc_runtime_set_table_cache_dir(const char* dir);
void lc_runtime_table_cache_stats(uint64_t out[4]);
const char* lc_runtime_table_cache_stamp(void);

/* Ask that SMALL batches of this handle (<= 16 Ki lines) walk one value per WAVEFRONT with the tables in global memory
 * (tdfa_wave_kernel: quiet runs crossed 256 bytes at a time) even when the automaton fits LDS -- what the Grok matcher asks for
 * its entries, whose batches are a few hundred long values.  Packs the global-memory table form if the handle has none yet.
 * Returns 1 when the handle will take that kernel, 0 when it cannot (no tagged DFA, tables too large).  Results are unchanged. */
int lc_regex_prefer_wave_tdfa(lc_regex_t* re);

/* Number of visible HIP devices (0 when there is none / no driver). */
int lc_device_count(void);

/* Hardware queues behind the HIP streams (ROCm: GPU_MAX_HW_QUEUES, default 4, read ONCE when the HIP runtime initialises).  The
 * library never changes the process environment on its own -- not at load time, not from a processor's Init.  A host that
 * wants n queues (a process hosting Grok processors gains from 16) either exports the variable or calls
 * this BEFORE the first HIP call of the process; a value the host has already exported is kept.  LC_OK / LC_ERR_ARG. */
int lc_runtime_prefer_hw_queues(int n);

/* Which GPU a thread's work goes to ("In-agent: map runner thread -> GPU (threadNo % nGPU)").
 * The reference calls Process from process_thread_count runner threads (core/runner/ProcessorRunner.cpp:138-142, the thread index
 * ProcessorRunner.h:40 selects mReg[threadNo], ProcessorParseRegexNative.cpp:255-257); the index does not cross the C slot and an
 * agent never selects a HIP device, so the library does:
 *   - HOST entry points (lc_processor_process and the other processors, lc_regex_match_host*, lc_grok_match_host, lc_multiline_*,
 *     lc_filter_*, lc_pipeline_*): the first such call of a thread BINDS the thread to a device by the process-wide policy and makes
 *     it the thread's current HIP device; staging, streams and table uploads of the thread live there.
 *       LC_BIND_ROUND_ROBIN (default): device = (thread ordinal) % lc_device_count(), ordinal = order of the threads' first host entry;
 *                                      a thread whose current device is already non-zero was placed by its host and keeps it.
 *       LC_BIND_FIXED:                 every thread -> `device` (a process that owns one GPU, e.g. one rank of a launcher).
 *       LC_BIND_INHERIT:               never switch: whatever device is current for the thread (rounds 1-4 behaviour).
 *     Environment: LC_BIND_POLICY = inherit | rr | fixed:<d> (read once, overridden by lc_runtime_set_bind_policy).
 *   - DEVICE-pointer entry points (lc_regex_match_device*, lc_grok_match_device, ...) never switch devices: they run on the caller's
 *     current device and return LC_ERR_ARG when d_data belongs to another device.
 * lc_runtime_set_bind_policy: process-wide, affects threads not yet bound.  lc_runtime_bind_thread(policy): bind the calling thread now
 * (policy < 0: the process-wide one); returns the device index, or -LC_ERR_*.  lc_runtime_set_thread_device(d): explicit binding
 * (LC_OK / LC_ERR_ARG / LC_ERR_NO_DEVICE).  lc_runtime_thread_device(): the calling thread's bound device, -1 when unbound.
 * lc_runtime_device_for_ordinal: the round-robin rule itself (pure; -1 when ndevices <= 0). */
enum { LC_BIND_INHERIT = 0, LC_BIND_ROUND_ROBIN = 1, LC_BIND_FIXED = 2 };
int lc_runtime_set_bind_policy(int policy, int device);
int lc_runtime_bind_policy(void);
int lc_runtime_bind_thread(int policy);
int lc_runtime_set_thread_device(int device);
int lc_runtime_thread_device(void);
int lc_runtime_device_for_ordinal(uint32_t ordinal, int ndevices);

/* Device buffers handed to the lc_*_device entry points -- the contract for d_data:
 *   the kernels read d_data in naturally ALIGNED units of up to 16 bytes (dword / dwordx4 loads, LDS-DMA rows), i.e. they may touch
 *   the bytes between a line's first byte and the 16-byte boundary below it, and between its last byte and the 16-byte boundary above
 *   it.  Those bytes are never interpreted, but they must be readable: d_data must be 16-byte aligned and its ALLOCATION must end on
 *   a 16-byte boundary at or behind the last line's end.  Every hipMalloc / hipMallocAsync / torch allocation satisfies this (256-byte
 *   granularity); a sub-range carved by the caller out of a larger allocation does too as long as the range starts 16-byte aligned.
 *   Lines themselves may start and end anywhere.  d_off / d_len / d_caps are 4-byte aligned arrays of their element type.
 *
 * Match n lines that already live in device memory on the current HIP device.
 *   d_data   : line bytes (any layout); line i = d_data[d_off[i] .. d_off[i]+d_len[i])
 *   d_len    : may be NULL, then d_off has n+1 entries and len[i] = d_off[i+1]-d_off[i]-sep_bytes
 *   ngroups  : number of (begin,end) pairs written per line (normally lc_regex_mark_count)
 *   d_caps   : int32[n][2*ngroups]   (groups beyond mark_count are written as -1,-1)
 *   d_status : uint8[n]
 *   stream   : hipStream_t (NULL = default stream).  Asynchronous: returns after enqueueing.
 */
int lc_regex_match_device(lc_regex_t* re, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len,
                          uint32_t sep_bytes, uint32_t n, uint32_t ngroups, int32_t* d_caps, uint8_t* d_status,
                          void* stream);

/* Same, but with an explicit engine (LC_ENGINE_TDFA / LC_ENGINE_NFA; LC_ENGINE_AUTO = the handle's own choice).
 * A handle compiled with LC_ENGINE_AUTO carries both programs whenever both fit, so the two kernels can be
 * cross-checked on identical input. */
int lc_regex_match_device_engine(lc_regex_t* re, int engine, const uint8_t* d_data, const uint32_t* d_off,
                                 const uint32_t* d_len, uint32_t sep_bytes, uint32_t n, uint32_t ngroups,
                                 int32_t* d_caps, uint8_t* d_status, void* stream);

/* The general device entry: a SUBSET of the lines, each search optionally RESUMED inside its line.
 *   d_lines  (optional) uint32[n]: indices of the lines to match; results land at caps[line]/status[line], lines not
 *            listed are not touched.  d_nlines (optional): the count lives on the device (<= n).
 *   d_from   (optional) uint32[], indexed by LINE: where the search resumes inside the line (0 = fresh search).  Only for
 *            patterns compiled with LC_SYNTAX_SEARCH.  The resumed search sees the byte before the resume point (look-
 *            behinds, \b, ^) exactly as regexp2's FindNextMatch / Go's FindAll do; offsets stay relative to the line.
 * This is what an iterate-all-matches, ordered-pattern-list caller (the Grok processor, processor_grok.go:148-194) is
 * built from. */
int lc_regex_match_device_from(lc_regex_t* re, int engine, const uint8_t* d_data, const uint32_t* d_off,
                               const uint32_t* d_len, uint32_t sep_bytes, uint32_t n, const uint32_t* d_lines,
                               const uint32_t* d_nlines, const uint32_t* d_from, uint32_t ngroups, int32_t* d_caps,
                               uint8_t* d_status, void* stream);

/* Several batches, each with its own compiled regex, in ONE call (BASELINE configs[3]: many pipelines share one GPU, and
 * ProcessQueueManager::PopItem hands their event groups out round-robin, core/collection_pipeline/queue/ProcessQueueManager.cpp).
 * Batches whose pattern runs on the TDFA engine are packed into a single kernel launch: every workgroup stages the tables of
 * ITS batch into LDS (the per-pipeline switch costs what staging 1-3 KB costs), so that 64 groups of 1000 lines fill the chip
 * the way one 64 000-line batch does.  The other batches (NFA engine, run captures) are launched one by one on the same
 * stream.  Results are exactly those of one lc_regex_match_device call per job.  Asynchronous on `stream`. */
typedef struct lc_match_job {
    lc_regex_t* re;
    const uint8_t* d_data;
    const uint32_t* d_off;
    const uint32_t* d_len; /* or NULL: len = off[i+1] - off[i] - sep_bytes */
    uint32_t sep_bytes, n, ngroups;
    int32_t* d_caps;
    uint8_t* d_status;
} lc_match_job;
int lc_regex_match_device_multi(const lc_match_job* jobs, uint32_t njobs, void* stream);

/* Same as lc_regex_match_device_engine with offsets[n+1] + sep_bytes, but the line count is read from device memory
 * (*d_nlines, clamped to max_lines) when the kernel starts: lets lc_split_lines_device and the match run back to back
 * on one stream with no host round trip.  Lines beyond *d_nlines are not touched. */
int lc_regex_match_device_dyn(lc_regex_t* re, int engine, const uint8_t* d_data, const uint32_t* d_off,
                              uint32_t sep_bytes, const uint32_t* d_nlines, uint32_t max_lines, uint32_t ngroups,
                              int32_t* d_caps, uint8_t* d_status, void* stream);

/* Same as lc_regex_match_device_engine, for batches whose line lengths vary: a counting sort on the device groups
 * lines of similar length (longest first) and the match kernel visits them in that order, so that the 64 lines of a
 * wavefront finish together (measured 2x on 128-2048 B lines).  Results land at the lines' ORIGINAL indices.
 * d_nlines: optional device-side line count (NULL = n).  d_scratch: lc_sched_scratch_bytes(n) bytes. */
size_t lc_sched_scratch_bytes(uint32_t max_lines);
int lc_regex_match_device_ragged(lc_regex_t* re, int engine, const uint8_t* d_data, const uint32_t* d_off,
                                 const uint32_t* d_len, uint32_t sep_bytes, uint32_t n, const uint32_t* d_nlines,
                                 uint32_t ngroups, int32_t* d_caps, uint8_t* d_status, void* d_scratch,
                                 size_t scratch_bytes, void* stream);

/* Line splitting on the device: the step BEFORE the parse processor in the reference pipeline,
 * ProcessorSplitLogStringNative::ProcessEvent / GetNextLine
 * (core/plugin/processor/inner/ProcessorSplitLogStringNative.cpp:101-174).  Scans d_data[0..nbytes) for split_char
 * and writes the line-offset table d_off[0..nlines] in the "offsets[n+1] + sep_bytes=1" form the match entry points
 * take (len[i] = off[i+1]-off[i]-1, also for an unterminated last line) and the line count to *d_nlines.  Same line
 * set as the reference: empty lines are lines, a trailing split_char does not open a new line, an empty buffer has
 * none.  If *d_nlines >= off_capacity the table was truncated (caller error).  d_scratch: lc_split_scratch_bytes().
 * Asynchronous on `stream`. */
size_t lc_split_scratch_bytes(uint64_t nbytes);
int lc_split_lines_device(const uint8_t* d_data, uint64_t nbytes, uint8_t split_char, uint32_t* d_off,
                          uint32_t off_capacity, uint32_t* d_nlines, void* d_scratch, size_t scratch_bytes,
                          void* stream);

/* The step AFTER the parser on the parser's own output, still on the device: ProcessorFilterNative::IsMatched
 * (core/plugin/processor/ProcessorFilterNative.cpp:258-286) for FilterKey / FilterRegex rules whose keys the parser has just
 * produced -- the reference's benchmark pipeline filters on the captured user_agent
 * (test/benchmark/local/test_cases/performance_file_to_blackhole_loongcollector/loongcollector.yaml:22-27).  The value of a parsed
 * key is the span of its capture group; each rule is regex_match(span) with the rule's regex as a yes/no automaton.
 *   lc_regex_prepare_span_filter: the rule's regex (compiled with lc_regex_compile, default syntax = regex_match) gets that
 *       automaton; LC_ERR_UNSUPPORTED when its tagged DFA does not exist or is too large (the caller then filters on the host side)
 *   lc_span_filter_device: after lc_split_lines_device + lc_regex_match_device_dyn on the same stream.  A line survives when the
 *       parser matched it and every rule matches its group's span (a group that did not take part has the empty value).
 *       d_packed[k] = [line, offset, length, 2*ngroups capture offsets] for survivor k (order unspecified), at most
 *       packed_cap_rows rows are written; d_counts[0] = lines, [1] = survivors, [2] = lines the parser did not match,
 *       [3] = lines left LC_OVERFLOW / LC_GAVE_UP.  Asynchronous on `stream`. */
/* nbytes (rounded up to 16) from a PINNED host block (hipHostMalloc, 16-byte aligned) to device memory by a kernel that reads the
 * block through its PCIe mapping -- not by the copy engine, whose queue the uploads of all runner threads would share. */
int lc_upload_pinned(const void* pinned_src, void* d_dst, size_t nbytes, void* stream);
typedef struct lc_span_filter {
    lc_regex_t* re;
    uint32_t group; /* 1-based capture group of the parse regex */
} lc_span_filter_t;
int lc_regex_prepare_span_filter(lc_regex_t* re);
int lc_span_filter_device(const lc_span_filter_t* filters, uint32_t nfilters, const uint8_t* d_data, const uint32_t* d_off,
                          uint32_t sep_bytes, const uint32_t* d_nlines, uint32_t max_lines, uint32_t ngroups, const int32_t* d_caps,
                          const uint8_t* d_status, int32_t* d_packed, uint32_t packed_cap_rows, uint32_t* d_counts, void* stream);

/* Same, for host buffers: lines are gathered through pinned staging buffers and copied with
 * hipMemcpyAsync on two streams so that chunk k+1 uploads while chunk k is being matched and chunk k-1
 * downloads.  Synchronous: results are in caps/status on return. */
int lc_regex_match_host(lc_regex_t* re, const uint8_t* data, const uint32_t* off, const uint32_t* len, uint32_t n,
                        uint32_t ngroups, int32_t* caps, uint8_t* status);

/* Same, for lines given as n independent views (pointer + length), which is what a PipelineEventGroup holds after
 * ProcessorSplitLogStringNative: StringViews into the group's SourceBuffer.  Views that sit back to back in memory
 * are staged with one memcpy per chunk; scattered views are gathered line by line. */
int lc_regex_match_host_views(lc_regex_t* re, const uint8_t* const* lines, const uint32_t* len, uint32_t n,
                              uint32_t ngroups, int32_t* caps, uint8_t* status);

/* Values that the consumers without a parse-failure counter of their own (filter leaves, multiline start / continue / end flags,
 * the Go regex plugin) took as "no match" because the matcher gave up on them (LC_GAVE_UP: where boost would have thrown its
 * complexity exception and BoostRegexMatch / BoostRegexSearch would have returned false).  Process-wide, monotonically growing. */
uint64_t lc_gave_up_values_total(void);

/* last HIP error string of the calling thread (for LC_ERR_HIP) */
const char* lc_last_error(void);

/* Frees the device resources the CALLING thread accumulated inside the match entry points (pinned staging slots, streams,
 * the decide kernel's scratch pool).  They are otherwise released when the thread ends; a host that recycles runner threads
 * (ProcessorRunner, core/runner/ProcessorRunner.cpp:138-142) may call this when a thread goes idle.  Safe to call any time.
 * A thread that was dealt to a device by LC_BIND_ROUND_ROBIN also gives its ORDINAL back (here, and when it ends): the next thread
 * that enters takes the lowest free one, so helper threads that come and go do not skew the deal of the runner threads. */
void lc_thread_release(void);

/* Statistics of the calling thread's decide passes since the last call: lines[0] = lines the thread-list kernels left
 * undecided and the decide kernel settled, lines[1] = of those, reported LC_GAVE_UP.  Synchronises the thread's streams. */
int lc_decide_stats(uint64_t lines[2]);

/* The calling thread's LAST NFA-engine launch: lines[0] = bytes of frame stack its depth-first first pass (one line per lane,
 * nfa_dfs_kernel) carved from the pool, lines[1] = lines it left to the thread-list kernels (step budget exceeded, or no room in
 * the pool).  Synchronises the thread's last such launch.  Diagnostics (tools/grok_bench.py, tests). */
int lc_dfs_stats(uint64_t lines[2]);

/* Switches the NFA engine's optional depth-first first pass (one line per lane, plain backtracking under a step budget; what it
 * cannot settle goes to the thread-list kernels of the same launch, so results never change) on (1) / off (0) for the whole
 * process, or back to the LC_NFA_DFS environment default (-1).  Off by default: measured slower on 8-64 Ki-line batches. */
void lc_nfa_set_dfs(int on);

/* Names of the match kernels the calling thread has launched since the last call (comma separated, duplicates folded) are
 * copied to buf (NUL terminated, at most cap bytes); returns the length of the full list and clears it.  Diagnostics only:
 * __graft_entry__.smoke() prints it so that the GPU-box log names the native code that ran. */
size_t lc_launched_kernels(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
