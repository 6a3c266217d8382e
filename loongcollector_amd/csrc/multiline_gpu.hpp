// multiline_gpu.hpp -- the compiled multiline configuration shared by multiline_gpu.cpp (record boundaries of one source value)
// and multiline_events.cpp (the processors on whole event groups).
#pragma once

#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/lc_multiline.h"
#include "../../include/lc_regex_gpu.h"

struct lc_multiline {
    lc_regex_t *start = nullptr, *cont = nullptr, *end = nullptr;
    bool discardUnmatched = false;
    bool isMultiline = false;
    std::string warnings;  // patterns that were ignored (MultilineOptions::Init only warns about an invalid regex)
    // ProcessorSplitMultilineLogStringNative's own parameters (:41-65) and counters (:78-82)
    std::string sourceKey = "content";
    bool enableRawContent = false;
    std::atomic<uint64_t> matchedLinesTotal{0}, unmatchedLinesTotal{0}, matchedEventsTotal{0};
    ~lc_multiline() {
        lc_regex_free(start);
        lc_regex_free(cont);
        lc_regex_free(end);
    }
};

// lc_multiline_create for the merge processor: the patterns in the form MultilineOptions' own regexes have (ParseRegex :250-266 strips a
// trailing '$' and ".*"s; Init :170-200 drops ContinuePattern when all three are given) -- multiline_gpu.cpp
int lcMultilineCreateForMerge(const char* config_json, size_t config_len, lc_multiline_t** out, char* err, size_t errcap);

// multiline_device.hip -- the device trips (one upload, one synchronisation); counts: ML_CNT_* of multiline_scan.hpp
int lcMultilineSplitTrip(lc_multiline* m, const uint8_t* data, uint32_t nbytes, std::vector<lc_ml_record_t>& out, uint32_t counts[8]);
// records carry ITEM indices (begin = first item, length = number of items); flush = false leaves the log under construction open
// and reports it in counts[ML_CNT_FINAL_PARTIAL] / counts[ML_CNT_FINAL_START]
int lcMultilineViewsTrip(const lc_multiline* m, const uint8_t* const* ptrs, const uint32_t* lens, uint32_t n, bool flush, bool keepUnmatched,
                         std::vector<lc_ml_record_t>& out, uint32_t counts[8]);
