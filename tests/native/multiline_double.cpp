// tests/native/multiline_double.cpp -- TEST INFRASTRUCTURE ONLY: the HOST side of the two multiline processors on a box without a GPU.
//
// csrc/multiline_events.cpp (lc_multiline_process_group, lc_merge_multiline_process_group: which events come out, the in-place merge,
// HandleUnmatchLogs' event counting) and csrc/multiline_gpu.cpp (the Multiline options, the record scan's host model) reach the device
// through two internal calls only -- lcMultilineSplitTrip and lcMultilineViewsTrip (csrc/multiline_device.hip: upload, one status-only
// launch per pattern, the scan kernel).  This translation unit answers those two from the CPU oracle's regex (oracle/bt_regex.h,
// prefix search = regex_search + match_continuous) and the scan's own code run on the host (lc_multiline_bounds_model), and
// lc_regex_compile from the oracle's compiler, so that tests/test_multiline_host_double.py can build
//     multiline_events.cpp + multiline_gpu.cpp + event_model.cpp + this file  ->  tests/_build/libmultiline_double.so
// and run the PRODUCT's host translation units beside the reference's own processors (oracle/_ref/libref_processor.so) here.
// Second variant (-DLC_USE_REFERENCE_HEADERS): the same product sources on the reference's OWN event model (oracle/_ref/libref_models.so,
// fixtures through tests/native/ref_group_io.cpp) -> tests/_build/libmultiline_double_ref.so.
// What it does not cover is the device: the match kernels and the scan kernel are held to the same oracle by the -m gpu tests.
//
// It lives under tests/, is built only by the test that uses it and is never linked into loongcollector_amd/lib.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/lc_multiline.h"
#include "../../include/lc_regex_gpu.h"
#ifdef LC_USE_REFERENCE_HEADERS  // second variant: the same product sources on the reference's OWN event model (oracle/_ref/libref_models.so)
#include "models/LogEvent.h"
#include "models/PipelineEventGroup.h"
#include "models/RawEvent.h"
bool hdGroupFromJson(logtail::PipelineEventGroup& g, const std::string& json, std::string* error);  // tests/native/ref_group_io.cpp
std::string hdGroupToJson(const logtail::PipelineEventGroup& g);
#else
#include "../../loongcollector_amd/csrc/event_model.hpp"
#endif
#include "../../loongcollector_amd/csrc/multiline_gpu.hpp"
#include "../../loongcollector_amd/csrc/multiline_scan.hpp"
#include "../../oracle/bt_regex.h"

// ---------------------------------------------------------------------------------------------- the doubles
#ifndef MD_NO_REGEX_DOUBLES  // (tests/native/race_driver.cpp links this file next to filter_double.cpp, which has the regex doubles)
struct lc_regex {
    orx_prog* prog = nullptr;
    std::string pattern;
};
static thread_local std::string tLastError;

extern "C" int lc_regex_compile(const char* pattern, size_t n, uint32_t flags, int, lc_regex_t** out, char* err, size_t errcap) {
    if (!pattern || !out) return LC_ERR_ARG;
    *out = nullptr;
    unsigned oflags = 0;
    if (flags & LC_SYNTAX_ICASE) oflags |= ORX_ICASE;
    orx_prog* p = orx_compile(pattern, n, oflags, err, errcap);
    if (!p) return LC_ERR_SYNTAX;
    auto* re = new lc_regex;
    re->prog = p;
    re->pattern.assign(pattern, n);
    *out = re;
    return LC_OK;
}
extern "C" void lc_regex_free(lc_regex_t* re) {
    if (!re) return;
    orx_free(re->prog);
    delete re;
}
extern "C" const char* lc_last_error(void) { return tLastError.c_str(); }
#endif  // MD_NO_REGEX_DOUBLES

namespace {
uint8_t answers(const lc_multiline* m, const uint8_t* s, uint32_t n) {
    lc_regex_t* const res[3] = {m->start, m->cont, m->end};
    uint8_t f = 0;
    std::vector<int32_t> caps;
    for (int k = 0; k < 3; ++k) {
        if (!res[k]) continue;
        caps.assign(size_t(2) * size_t(orx_mark_count(res[k]->prog) + 1), -1);
        if (orx_prefixmatch(res[k]->prog, s, n, caps.data()) == 1) f |= uint8_t(1u << k);
    }
    return f;
}
uint32_t modeOf(const lc_multiline& m, bool flush, bool discard) {
    return (m.start ? ML_HAS_START : 0u) | (m.cont ? ML_HAS_CONT : 0u) | (m.end ? ML_HAS_END : 0u) | (discard ? ML_DISCARD : 0u) |
           (flush ? ML_FLUSH : 0u);
}
int scan(uint32_t mode, const std::vector<uint8_t>& flags, const uint32_t* off, uint32_t nbytes, std::vector<lc_ml_record_t>& out,
         uint32_t counts[ML_CNT_WORDS]) {
    const uint32_t n = uint32_t(flags.size());
    std::vector<lc_ml_record_t> recs(size_t(n) + 2);
    const int rc = lc_multiline_bounds_model(mode, flags.data(), n, off, nbytes, recs.data(), uint32_t(recs.size()), counts);
    if (rc != LC_OK) return rc;
    out.assign(recs.begin(), recs.begin() + counts[ML_CNT_RECORDS]);
    return LC_OK;
}
}  // namespace

int lcMultilineSplitTrip(lc_multiline* m, const uint8_t* data, uint32_t nbytes, std::vector<lc_ml_record_t>& out, uint32_t counts[ML_CNT_WORDS]) {
    out.clear();
    std::memset(counts, 0, ML_CNT_WORDS * 4);
    if (nbytes == 0) return LC_OK;
    // GetNextLine :382-392: '\n' separates; a trailing '\n' does not open an empty last line
    std::vector<uint32_t> off;
    std::vector<uint8_t> flags;
    uint32_t at = 0;
    while (at < nbytes) {
        const void* nl = std::memchr(data + at, '\n', nbytes - at);
        const uint32_t end = nl ? uint32_t(static_cast<const uint8_t*>(nl) - data) : nbytes;
        off.push_back(at);
        flags.push_back(uint8_t(answers(m, data + at, end - at) | (end == at ? 8u : 0u)));
        at = end + 1;
    }
    off.push_back(at);  // len[i] = off[i+1] - off[i] - 1, also for an unterminated last line
    return scan(modeOf(*m, true, m->discardUnmatched), flags, off.data(), nbytes, out, counts);
}

int lcMultilineViewsTrip(const lc_multiline* m, const uint8_t* const* ptrs, const uint32_t* lens, uint32_t n, bool flush, bool keepUnmatched,
                         std::vector<lc_ml_record_t>& out, uint32_t counts[ML_CNT_WORDS]) {
    out.clear();
    std::memset(counts, 0, ML_CNT_WORDS * 4);
    std::vector<uint8_t> flags(n);
    for (uint32_t i = 0; i < n; ++i) flags[i] = answers(m, ptrs[i], lens[i]);
    return scan(modeOf(*m, flush, m->discardUnmatched && !keepUnmatched), flags, nullptr, 0, out, counts);
}

// ---------------------------------------------------------------------------------------------- the harness
extern "C" {
// fixture JSON in -> lc_multiline_process_group / lc_merge_multiline_process_group -> fixture JSON out (malloc'ed)
static char* processJson(int (*fn)(void*, void*), void* h, const char* groupJson, char* err, size_t errcap) {
    logtail::PipelineEventGroup group(std::make_shared<logtail::SourceBuffer>());
    std::string error;
#ifdef LC_USE_REFERENCE_HEADERS
    if (!hdGroupFromJson(group, groupJson, &error)) {
#else
    if (!group.FromJsonString(groupJson, &error)) {
#endif
        std::snprintf(err, errcap, "%s", error.c_str());
        return nullptr;
    }
    const int rc = fn(h, &group);
    if (rc != LC_OK) {
        std::snprintf(err, errcap, "process_group failed: %d", rc);
        return nullptr;
    }
#ifdef LC_USE_REFERENCE_HEADERS
    return strdup(hdGroupToJson(group).c_str());
#else
    return strdup(group.ToJsonString().c_str());
#endif
}
char* md_split_json(lc_multiline_t* m, const char* groupJson, char* err, size_t errcap) {
    return processJson([](void* h, void* g) { return lc_multiline_process_group(static_cast<lc_multiline_t*>(h), g); }, m, groupJson, err, errcap);
}
char* md_merge_json(lc_merge_multiline_t* p, const char* groupJson, char* err, size_t errcap) {
    return processJson([](void* h, void* g) { return lc_merge_multiline_process_group(static_cast<lc_merge_multiline_t*>(h), g); }, p, groupJson, err,
                       errcap);
}
// What the line splitter leaves: ONE copy of `data` in the group's source buffer, one log event per line whose `key` content is a view
// of its line (timestamp = 1 + the line's index); an event WITHOUT contents in front of every line listed in emptyBefore (ascending;
// the line count = behind the last line); the lines listed in otherKey (ascending) carry the key "other" instead.  -> the group after lc_merge_multiline_process_group, as fixture JSON (malloc'ed).
char* md_merge_lines(lc_merge_multiline_t* p, const uint8_t* data, size_t nbytes, const char* key, const uint32_t* emptyBefore, uint32_t nEmpty,
                     const uint32_t* otherKey, uint32_t nOther, char* err, size_t errcap) {
    auto sb = std::make_shared<logtail::SourceBuffer>();
    logtail::PipelineEventGroup group(sb);
    const logtail::StringBuffer copy = sb->CopyString(reinterpret_cast<const char*>(data), nbytes);
    const logtail::StringBuffer k = sb->CopyString(key, std::strlen(key));
    const logtail::StringBuffer other = sb->CopyString("other", 5);
    uint32_t o = 0;
    uint32_t line = 0, e = 0, ts = 1000;
    auto empties = [&] {
        while (e < nEmpty && emptyBefore[e] == line) {
            group.AddLogEvent()->SetTimestamp(++ts);
            ++e;
        }
    };
    size_t at = 0;
    while (at < nbytes) {
        const void* nl = std::memchr(copy.data + at, '\n', nbytes - at);
        const size_t end = nl ? size_t(static_cast<const char*>(nl) - copy.data) : nbytes;
        empties();
        logtail::LogEvent* ev = group.AddLogEvent();
        const bool keyless = o < nOther && otherKey[o] == line;  // (an event that does not carry the source key)
        if (keyless) ++o;
        ev->SetContentNoCopy(keyless ? logtail::StringView(other.data, other.size) : logtail::StringView(k.data, k.size),
                             logtail::StringView(copy.data + at, end - at));
        ev->SetTimestamp(1 + line);
        ++line;
        at = end + 1;
    }
    empties();
    const int rc = lc_merge_multiline_process_group(p, &group);
    if (rc != LC_OK) {
        std::snprintf(err, errcap, "lc_merge_multiline_process_group failed: %d", rc);
        return nullptr;
    }
#ifdef LC_USE_REFERENCE_HEADERS
    return strdup(hdGroupToJson(group).c_str());
#else
    return strdup(group.ToJsonString().c_str());
#endif
}
#ifdef LC_USE_REFERENCE_HEADERS
int md_event_model_is_reference(void) { return 1; }
#else
int md_event_model_is_reference(void) { return 0; }
#endif
void md_free(void* p) { std::free(p); }
}  // extern "C"
