// tests/native/pipeline_double.cpp -- TEST INFRASTRUCTURE ONLY: the HOST side of the fused split -> parse -> filter pipeline
// (csrc/processor_pipeline_gpu.cpp: which groups travel fused, the events built from the survivors' rows, the three processors' counters
// reconstructed from the trip's counts; and the chained path through the parser's and the filter's own classes) on a box without a GPU.
//
// On top of tests/native/filter_double.cpp (the HIP runtime on host memory, lc_regex_compile / lc_regex_match_device_multi from the CPU
// oracle): the parser's match call and the four device steps of the fused trip, each restated from its contract in include/lc_regex_gpu.h
//   lc_split_lines_device      the offsets[n+1] table of ProcessorSplitLogStringNative's line set (:303-314)
//   lc_regex_match_device_dyn  full match of every line, capture offsets relative to the line
//   lc_span_filter_device      survivors = matched lines whose rule groups' spans full-match the rules; rows in REVERSE line order here
//                              (the contract leaves the order open; the host sorts)
// tests/test_pipeline_host_double.py builds  c_processor_slot.cpp + processor_pipeline_gpu.cpp + processor_parse_regex_gpu.cpp +
// processor_filter_gpu.cpp + event_model.cpp + this file  ->  tests/_build/libpipeline_double.so  and runs the PRODUCT's host code beside
// the reference's own three processors, chained (oracle/_ref/libref_processor.so).  The device side is the -m gpu tests' business.
#include "filter_double.cpp"

#include "../../loongcollector_amd/csrc/processor_parse_regex_gpu.hpp"

extern "C" int lc_regex_mark_count(const lc_regex_t* re) { return re ? re->marks : -1; }
static void matchOne(const lc_regex_t* re, const uint8_t* s, uint32_t len, uint32_t ngroups, int32_t* caps, uint8_t* status, std::vector<int32_t>& what) {
    const int r = orx_fullmatch(re->prog, s, len, what.data());
    *status = r == 1 ? LC_MATCH : r == 0 ? LC_NOMATCH : LC_GAVE_UP;
    for (uint32_t g = 0; g < ngroups; ++g) {
        const bool have = r == 1 && int(g) < re->marks;
        caps[size_t(g) * 2] = have ? what[(g + 1) * 2] : -1;
        caps[size_t(g) * 2 + 1] = have ? what[(g + 1) * 2 + 1] : -1;
    }
}
extern "C" int lc_regex_match_host_views(lc_regex_t* re, const uint8_t* const* lines, const uint32_t* len, uint32_t n, uint32_t ngroups,
                                         int32_t* caps, uint8_t* status) {
    if (!re || (n && (!lines || !len || !status))) return LC_ERR_ARG;
    std::vector<int32_t> what(size_t(re->marks + 1) * 2);
    for (uint32_t i = 0; i < n; ++i) matchOne(re, lines[i], len[i], ngroups, caps + size_t(i) * ngroups * 2, status + i, what);
    return LC_OK;
}
extern "C" size_t lc_split_scratch_bytes(uint64_t) { return 64; }
extern "C" int lc_split_lines_device(const uint8_t* d_data, uint64_t nbytes, uint8_t split_char, uint32_t* d_off, uint32_t off_capacity,
                                     uint32_t* d_nlines, void*, size_t, void*) {
    uint32_t n = 0;
    uint64_t at = 0;
    while (at < nbytes) {
        const void* hit = std::memchr(d_data + at, split_char, nbytes - at);
        const uint64_t end = hit ? uint64_t(static_cast<const uint8_t*>(hit) - d_data) : nbytes;
        if (n < off_capacity) d_off[n] = uint32_t(at);
        ++n;
        at = end + 1;
    }
    if (n < off_capacity) d_off[n] = uint32_t(at);  // len[i] = off[i+1] - off[i] - 1, also for an unterminated last line
    *d_nlines = n;
    return LC_OK;
}
extern "C" int lc_regex_match_device_dyn(lc_regex_t* re, int, const uint8_t* d_data, const uint32_t* d_off, uint32_t sep_bytes,
                                         const uint32_t* d_nlines, uint32_t max_lines, uint32_t ngroups, int32_t* d_caps, uint8_t* d_status, void*) {
    const uint32_t n = *d_nlines < max_lines ? *d_nlines : max_lines;
    std::vector<int32_t> what(size_t(re->marks + 1) * 2);
    for (uint32_t i = 0; i < n; ++i)
        matchOne(re, d_data + d_off[i], d_off[i + 1] - d_off[i] - sep_bytes, ngroups, d_caps + size_t(i) * ngroups * 2, d_status + i, what);
    return LC_OK;
}
extern "C" int lc_regex_prepare_span_filter(lc_regex_t* re) { return re ? LC_OK : LC_ERR_ARG; }
extern "C" int lc_span_filter_device(const lc_span_filter_t* filters, uint32_t nfilters, const uint8_t* d_data, const uint32_t* d_off,
                                     uint32_t sep_bytes, const uint32_t* d_nlines, uint32_t max_lines, uint32_t ngroups, const int32_t* d_caps,
                                     const uint8_t* d_status, int32_t* d_packed, uint32_t packed_cap_rows, uint32_t* d_counts, void*) {
    const uint32_t n = *d_nlines < max_lines ? *d_nlines : max_lines;
    const uint32_t rowInts = 3 + 2 * ngroups;
    uint32_t survivors = 0, failed = 0, undecided = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t i = n - 1 - k;  // (reverse order: the host must not rely on the rows' order)
        if (d_status[i] == LC_OVERFLOW || d_status[i] == LC_GAVE_UP) {
            ++undecided;
            continue;
        }
        if (d_status[i] != LC_MATCH) {
            ++failed;
            continue;
        }
        const int32_t* caps = d_caps + size_t(i) * ngroups * 2;
        bool keep = true;
        for (uint32_t f = 0; f < nfilters && keep; ++f) {
            const uint32_t g = filters[f].group - 1;
            const int32_t b = caps[2 * g], e = caps[2 * g + 1];
            std::vector<int32_t> what(size_t(filters[f].re->marks + 1) * 2);
            const uint8_t* s = d_data + d_off[i] + (b >= 0 ? b : 0);
            keep = orx_fullmatch(filters[f].re->prog, s, b >= 0 ? uint32_t(e - b) : 0u, what.data()) == 1;
        }
        if (!keep) continue;
        if (survivors < packed_cap_rows) {
            int32_t* row = d_packed + size_t(survivors) * rowInts;
            row[0] = int32_t(i);
            row[1] = int32_t(d_off[i]);
            row[2] = int32_t(d_off[i + 1] - d_off[i] - sep_bytes);
            std::memcpy(row + 3, caps, size_t(ngroups) * 2 * sizeof(int32_t));
        }
        ++survivors;
    }
    d_counts[0] = n;
    d_counts[1] = survivors;
    d_counts[2] = failed;
    d_counts[3] = undecided;
    return LC_OK;
}

// ---------------------------------------------------------------------------------------------- the harness
extern "C" {
// fixture JSON in -> lc_pipeline_process -> fixture JSON out (malloc'ed; fd_free)
char* pd_process_json(lc_pipeline_t* p, const char* groupJson, char* err, size_t errcap) {
    // (lc_pipeline_process takes the fixture wrapper lc_event_group_t: a group built by lc_group_from_json is one)
    lc_event_group_t* g = lc_group_from_json(groupJson, err, errcap);
    if (!g) return nullptr;
    const int rc = lc_pipeline_process(p, g);
    if (rc != LC_OK) {
        std::snprintf(err, errcap, "lc_pipeline_process failed: %d (%s)", rc, lc_last_error());
        lc_group_free(g);
        return nullptr;
    }
    char* out = lc_group_to_json(g);
    lc_group_free(g);
    return out;
}
}  // extern "C"
