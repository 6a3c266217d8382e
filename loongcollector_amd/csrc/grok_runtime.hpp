// grok_runtime.hpp -- device half of the Grok matcher (implemented in grok_device.hip).
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

struct lc_regex;

// One compiled Match entry as the device loop needs it.
struct GrokDevicePattern {
    lc_regex* re;        // compiled with LC_SYNTAX_SEARCH | LC_SYNTAX_NAMED_ONLY: group 1 = whole match, 2.. = named groups
    uint32_t columns;    // named groups
    lc_regex* screen;    // optional screen for the pattern's prefix (regex_handle.hpp lcCompilePrefixScreen), or null
    lc_regex* relaxed = nullptr;  // optional screen for the whole pattern, relaxed (lcCompileRelaxedScreen), or null
    lc_regex* anchored = nullptr; // optional: the same pattern as an ANCHORED search (LC_SYNTAX_SEARCH | LC_SYNTAX_PREFIX), same groups --
                                  // on the TDFA engine where the automaton builds, else on the NFA engine: tried first on values
                                  // searched from their first byte
};

// How a handle wants its batches matched (ProcessorGrokGpu carries one; nothing here changes a result).
struct GrokOptions {
    // true (default): every (entry, value) pair that passes the literal index and the entry's screen is evaluated at the same time,
    // entries spread over a few streams, the first contributing entry per value taken at the end: 2 host syncs per batch.
    // false: the Match list is walked entry by entry over the values still undecided (lists of more than 64 entries always are).
    bool speculative = true;
    // sequential path: an entry that has a relaxed screen runs its prefix screen first only when more than this many values carry
    // its literal
    uint32_t prefixScreenAbove = 65536;
    uint32_t streams = 16;  // worker streams of the speculative path (1..16)
};

// What a Grok handle keeps on the device(s) between batches: the literal index of its Match list, the table of its screens.
// Built on first use per device, freed with the handle.
struct GrokDeviceState;
GrokDeviceState* lcGrokStateCreate();
void lcGrokStateFree(GrokDeviceState* s);

// per-batch statistics of the calling thread's last lcGrokMatchDevice call (tests, tools/grok_bench.py)
struct GrokBatchStats {
    uint32_t hostSyncs;      // hipStreamSynchronize calls of the batch
    uint32_t activeEntries;  // entries that had at least one candidate
    uint32_t pairs;          // (entry, value) pairs evaluated
    uint32_t deferredEntries;  // entries that needed more search rounds than were queued ahead
    uint32_t speculative;    // 1: speculative path, 0: sequential path
};
GrokBatchStats lcGrokLastBatchStats();

size_t lcGrokScratchBytes(uint32_t n, uint32_t rowInts);

// grok_literal_index.cpp: the literal the index keeps for a pattern (its required literal; the last 32 bytes of a longer one) and
// the index blob of a list of such literals ("" = the entry has none: its bit is always set).  Empty blob: not indexable.
std::string lcGrokLiteralOf(const lc_regex* re);
std::vector<uint32_t> lcBuildGrokLiteralBlob(const std::vector<std::string>& literals);

// See include/lc_grok.h: lc_grok_match_device.  rowInts = 2 * (1 + max columns).
int lcGrokMatchDevice(const std::vector<GrokDevicePattern>& patterns, GrokDeviceState* state, const GrokOptions& opts,
                      uint32_t rowInts, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len, uint32_t n,
                      int32_t* d_pattern, int32_t* d_first, int32_t* d_extra, uint32_t extraCap, uint32_t* d_nextra, void* d_scratch,
                      size_t scratchBytes, void* stream);

// pinned-host convenience used by lc_grok_match_host: copies in (one block, the calling thread's own stream), runs
// lcGrokMatchDevice, copies out.  *firstRows = int32[n][rowInts] in the thread's pinned staging, valid until the thread's next call;
// extraRows receives [line, seq, row...] records sorted by (line, seq).
int lcGrokMatchHost(const std::vector<GrokDevicePattern>& patterns, GrokDeviceState* state, const GrokOptions& opts, uint32_t rowInts,
                    const uint8_t* data, const uint32_t* off, const uint32_t* len, uint32_t n, int32_t* pattern,
                    const int32_t** firstRows, std::vector<int32_t>& extraRows);
// up to `maxValues` of a DEVICE-resident batch's first values, copied to host memory (offsets rebased to the copy): the lazy automata's
// trainer (processor_grok_gpu.cpp) takes its sample of a device-fed handle this way.  Synchronises `stream` (rare: see OfferDeviceBatch).
int lcGrokSampleDevice(const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len, uint32_t n, uint32_t maxValues, void* stream,
                       std::vector<uint8_t>& data, std::vector<uint32_t>& off, std::vector<uint32_t>& len);
// the group commit behind lcGrokMatchHost (group_combiner.hpp), summed over the devices of `state`:
// out = {batches, groups, values, most groups in one batch, batches started by the linger's timeout}
int lcGrokCombinerStats(GrokDeviceState* state, uint64_t out[11]);
