// grok_runtime.hpp -- device half of the Grok matcher (implemented in gpu_runtime.hip).
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

struct lc_regex;

// One compiled Match entry as the device loop needs it.
struct GrokDevicePattern {
    lc_regex* re;        // compiled with LC_SYNTAX_SEARCH | LC_SYNTAX_NAMED_ONLY: group 1 = whole match, 2.. = named groups
    uint32_t columns;    // named groups
    lc_regex* screen;    // optional TDFA screen for the pattern's prefix (regex_handle.hpp lcCompilePrefixScreen), or null
    lc_regex* relaxed = nullptr;  // optional TDFA screen for the whole pattern, relaxed (lcCompileRelaxedScreen), or null
    lc_regex* anchored = nullptr; // optional: the same pattern as an ANCHORED search (LC_SYNTAX_SEARCH | LC_SYNTAX_PREFIX) on the TDFA
                                  // engine, same groups: tried first on values searched from their first byte
};

size_t lcGrokScratchBytes(uint32_t n, uint32_t rowInts);

// grok_literal_index.cpp: the literal the index keeps for a pattern (its required literal; the last 32 bytes of a longer one) and
// the index blob of a list of such literals ("" = the entry has none: its bit is always set).  Empty blob: not indexable.
#include <string>
std::string lcGrokLiteralOf(const lc_regex* re);
std::vector<uint32_t> lcBuildGrokLiteralBlob(const std::vector<std::string>& literals);

// See include/lc_grok.h: lc_grok_match_device.  rowInts = 2 * (1 + max columns).
int lcGrokMatchDevice(const std::vector<GrokDevicePattern>& patterns, uint32_t rowInts, const uint8_t* d_data,
                      const uint32_t* d_off, const uint32_t* d_len, uint32_t n, int32_t* d_pattern, int32_t* d_first,
                      int32_t* d_extra, uint32_t extraCap, uint32_t* d_nextra, void* d_scratch, size_t scratchBytes,
                      void* stream);

// pinned-host convenience used by lc_grok_match_host: copies in, runs lcGrokMatchDevice, copies out.
// extraRows receives [line, seq, row...] records sorted by (line, seq).
int lcGrokMatchHost(const std::vector<GrokDevicePattern>& patterns, uint32_t rowInts, const uint8_t* data,
                    const uint32_t* off, const uint32_t* len, uint32_t n, int32_t* pattern, std::vector<int32_t>& first,
                    std::vector<int32_t>& extraRows);
