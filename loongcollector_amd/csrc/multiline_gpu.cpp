// multiline_gpu.cpp -- record boundaries of the multiline splitter over device-computed line flags (include/lc_multiline.h).
//
// Restates, for one source value:
//   MultilineOptions::ParseRegex / Init      core/file_server/MultilineOptions.cpp:100-262
//   ProcessorSplitMultilineLogStringNative::ProcessEvent / HandleUnmatchLogs / GetNextLine
//                                            core/plugin/processor/inner/ProcessorSplitMultilineLogStringNative.cpp:126-300,341-392
// The reference asks BoostRegexSearch (regex_search + match_continuous) line by line; here the answers for all lines come
// from one device launch per pattern and the state machine below only looks at flags.
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/lc_multiline.h"
#include "../../include/lc_regex_gpu.h"
#include "json_min.hpp"
#include "regex_handle.hpp"

namespace {

bool endsWith(const std::string& s, const char* suffix) {
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

}  // namespace

#include "multiline_gpu.hpp"

// MultilineOptions::ParseRegex :250-266 strips a trailing '$' and trailing ".*"s -- but only to decide validity and
// IsMultiline().  The PROCESSOR compiles the pattern strings as written (ProcessorSplitMultilineLogStringNative.cpp:66-76,
// mMultiline.mStartPattern is the original string, MultilineOptions.cpp:119) and a pattern counts as present when that
// string is not empty (Has*Pattern, .h:68-70): "END$" does need the line to end there, ".*" is a start pattern that
// matches every line, and ContinuePattern stays in use when all three are given.
static std::string trimmed(std::string pattern) {
    if (!pattern.empty() && endsWith(pattern, "$")) pattern.pop_back();
    while (!pattern.empty() && endsWith(pattern, ".*")) pattern.resize(pattern.size() - 2);
    return pattern;
}
// -> LC_OK (compiled, or nothing to compile), LC_ERR_SYNTAX (the reference ignores the pattern with a warning,
// MultilineOptions.cpp:109-118), LC_ERR_UNSUPPORTED (valid for Boost, not runnable on the device: Init must fail, there is
// no CPU path to fall back to)
static int parseRegex(const std::string& pattern, lc_regex_t** out, std::string& err) {
    if (pattern.empty()) return LC_OK;
    char buf[256];
    const std::string t = trimmed(pattern);  // ParseRegex judges validity on the stripped form
    if (!t.empty()) {
        lc_regex_t* probe = nullptr;
        const int rc = lc_regex_compile(t.data(), t.size(), LC_SYNTAX_PREFIX, LC_ENGINE_AUTO, &probe, buf, sizeof buf);
        lc_regex_free(probe);
        if (rc == LC_ERR_SYNTAX) {
            err = buf;
            return rc;
        }
    }
    const int rc = lc_regex_compile(pattern.data(), pattern.size(), LC_SYNTAX_PREFIX, LC_ENGINE_AUTO, out, buf, sizeof buf);
    if (rc != LC_OK) err = buf;
    return rc;
}

extern "C" int lc_multiline_create(const char* config_json, size_t config_len, lc_multiline_t** out, char* err,
                                   size_t errcap) {
    if (!config_json || !out) return LC_ERR_ARG;
    *out = nullptr;
    auto set = [&](const std::string& m) {
        if (err && errcap) std::snprintf(err, errcap, "%s", m.c_str());
    };
    auto m = std::make_unique<lc_multiline>();
    try {
        const lcjson::Value cfg = lcjson::parse(std::string(config_json, config_len));
        if (!cfg.isObject()) throw std::runtime_error("config must be a JSON object");
        auto str = [&](const char* key) {
            const lcjson::Value* v = cfg.find(key);
            return (v && v->isString()) ? v->str : std::string();
        };
        std::string kept[3];
        const char* names[3] = {"StartPattern", "ContinuePattern", "EndPattern"};
        lc_regex_t** regs[3] = {&m->start, &m->cont, &m->end};
        for (int i = 0; i < 3; ++i) {
            std::string why;
            const std::string pattern = str(names[i]);
            const int rc = parseRegex(pattern, regs[i], why);
            if (rc == LC_OK) {
                kept[i] = pattern;
            } else if (rc == LC_ERR_SYNTAX) {  // :109-118 -- a warning, the pattern counts as not given
                m->warnings += std::string("string param Multiline.") + names[i] + " is not a valid regex: " + why + "\n";
            } else {
                throw std::runtime_error(std::string("Multiline.") + names[i] + ": " + why);
            }
        }
        // The reference's state machine indexes an empty regex vector when it is handed ContinuePattern alone or no
        // pattern at all (GetStartPatternReg(), .cpp:176-184, :395) -- the input plugin never builds the processor for
        // such a config (InputFile.cpp:225, IsMultiline() is false).  Refused here.
        if (!m->start && !m->end)
            throw std::runtime_error("neither Multiline.StartPattern nor Multiline.EndPattern is usable: not a multiline config"
                                     + (m->warnings.empty() ? std::string() : " (" + m->warnings + ")"));
        // MultilineOptions::IsMultiline (:203-205) goes by what is left after the stripping
        m->isMultiline = !trimmed(kept[0]).empty() || !trimmed(kept[2]).empty();
        const std::string t = str("UnmatchedContentTreatment");               // :208-222
        m->discardUnmatched = t == "discard";
        // the processor's own parameters (ProcessorSplitMultilineLogStringNative::Init :41-65), used by lc_multiline_process_group
        if (const lcjson::Value* v = cfg.find("SourceKey"))
            if (v->isString()) m->sourceKey = v->str;
        if (const lcjson::Value* v = cfg.find("EnableRawContent"))
            if (v->isBool()) m->enableRawContent = v->b;
    } catch (const std::exception& e) {
        set(e.what());
        return LC_ERR_SYNTAX;
    }
    set("");
    *out = m.release();
    return LC_OK;
}
extern "C" void lc_multiline_free(lc_multiline_t* m) { delete m; }
extern "C" int lc_multiline_is_multiline(const lc_multiline_t* m) { return m && m->isMultiline; }
extern "C" const char* lc_multiline_warnings(const lc_multiline_t* m) { return m ? m->warnings.c_str() : ""; }
extern "C" int lc_multiline_patterns(const lc_multiline_t* m) {
    return m ? (m->start ? 1 : 0) | (m->cont ? 2 : 0) | (m->end ? 4 : 0) : 0;
}

extern "C" int lc_multiline_split_host(lc_multiline_t* m, const uint8_t* data, uint32_t nbytes, lc_ml_record_t** records,
                                       uint32_t* nrecords, uint32_t counters[3]) {
    if (!m || !records || !nrecords || (nbytes && !data)) return LC_ERR_ARG;
    *records = nullptr;
    *nrecords = 0;
    uint32_t inputLines = 0, unmatchLines = 0, matchedEvents = 0;
    // GetNextLine :382-392: lines are separated by '\n'; a trailing '\n' does not open an empty last line
    std::vector<uint32_t> off, len;
    for (uint32_t b = 0; b < nbytes;) {
        uint32_t e = b;
        while (e < nbytes && data[e] != '\n') ++e;
        off.push_back(b);
        len.push_back(e - b);
        b = e + 1;
    }
    const uint32_t n = uint32_t(off.size());
    std::vector<uint8_t> fStart(n, 0), fCont(n, 0), fEnd(n, 0);
    auto flags = [&](lc_regex_t* re, std::vector<uint8_t>& dst) -> int {
        if (!re || n == 0) return LC_OK;
        const int r = lc_regex_match_host(re, data, off.data(), len.data(), n, 0, nullptr, dst.data());
        if (r != LC_OK) return r;
        // "not decided" (decide pass switched off) must not drive the state machine as "no match"
        uint64_t gaveUp = 0;
        for (uint8_t st : dst) {
            if (st == LC_OVERFLOW) return LC_ERR_UNSUPPORTED;
            gaveUp += st == LC_GAVE_UP;  // BoostRegexSearch failed with an exception (StringTools.cpp:277-282): false, and counted
        }
        if (gaveUp) lcNoteGaveUp(gaveUp);
        return LC_OK;
    };
    int rc;
    if ((rc = flags(m->start, fStart)) != LC_OK || (rc = flags(m->cont, fCont)) != LC_OK ||
        (rc = flags(m->end, fEnd)) != LC_OK)
        return rc;

    std::vector<lc_ml_record_t> out;
    const bool hasStart = m->start, hasCont = m->cont, hasEnd = m->end;
    // `last`: the isLastLog argument the reference passes to CreateNewEvent -- that of the line BEING PROCESSED when the
    // record is emitted (it decides the record's position length, :327-329), true for the flush after the loop
    bool last = false;
    auto createNewEvent = [&](int64_t b, int64_t e) {  // [b, e) of the source value
        out.push_back({uint32_t(b), uint32_t(e > b ? e - b : 0), 1u | (last ? LC_ML_LAST : 0u)});
    };
    auto handleUnmatch = [&](int64_t b, int64_t e) {   // HandleUnmatchLogs :341-380: line by line
        for (int64_t p = b; p < e;) {
            int64_t q = p;
            while (q < e && data[q] != '\n') ++q;
            ++unmatchLines;
            if (!m->discardUnmatched) out.push_back({uint32_t(p), uint32_t(q - p), last ? LC_ML_LAST : 0u});
            p = q + 1;
        }
    };
    int64_t multiStart = -1;
    bool isPartialLog = false;
    if (!hasStart && !hasCont && hasEnd) {  // only an end pattern: it sticks to this state (:161-165)
        isPartialLog = true;
        multiStart = 0;
    }
    for (uint32_t i = 0; i < n; ++i) {
        const int64_t cb = off[i], ce = int64_t(off[i]) + len[i];
        last = ce == int64_t(nbytes);  // isLastLog :174
        ++inputLines;
        if (!isPartialLog) {
            const bool first = hasStart ? fStart[i] == LC_MATCH : fCont[i] == LC_MATCH;   // :176-184
            if (first) {
                multiStart = cb;
                isPartialLog = true;
            } else if (hasEnd && !hasStart && hasCont && fEnd[i] == LC_MATCH) {            // continue + end (:187-192)
                createNewEvent(cb, ce);
                multiStart = ce + 1;
                ++matchedEvents;
            } else {
                handleUnmatch(cb, ce);
            }
        } else {
            if (hasCont && fCont[i] == LC_MATCH) continue;                                // :199-203
            if (hasEnd) {
                if (hasCont) {                                                            // :206-228
                    if (fEnd[i] == LC_MATCH) {
                        createNewEvent(multiStart, ce);
                        ++matchedEvents;
                    } else {
                        handleUnmatch(multiStart, ce);
                    }
                    isPartialLog = false;
                } else if (fEnd[i] == LC_MATCH) {                                          // start + end, or end (:229-246)
                    createNewEvent(multiStart, ce);
                    if (hasStart) isPartialLog = false;
                    else multiStart = ce + 1;
                    ++matchedEvents;
                }
            } else if (!hasCont) {                                                         // start only (:250-260)
                if (fStart[i] == LC_MATCH) {
                    createNewEvent(multiStart, cb - 1);
                    multiStart = cb;
                    ++matchedEvents;
                }
            } else {                                                                       // start + continue (:261-282)
                createNewEvent(multiStart, cb - 1);
                ++matchedEvents;
                if (fStart[i] != LC_MATCH) {
                    handleUnmatch(cb, ce);
                    isPartialLog = false;
                } else {
                    multiStart = cb;
                }
            }
        }
    }
    last = true;
    if (isPartialLog && multiStart < int64_t(nbytes)) {                                    // :288-298
        if (!hasEnd) {
            createNewEvent(multiStart, nbytes);
            ++matchedEvents;
        } else {
            handleUnmatch(multiStart, nbytes);
        }
    }
    if (counters) {
        counters[0] = inputLines;
        counters[1] = unmatchLines;
        counters[2] = matchedEvents;
    }
    *nrecords = uint32_t(out.size());
    if (!out.empty()) {
        *records = static_cast<lc_ml_record_t*>(std::malloc(out.size() * sizeof(lc_ml_record_t)));
        if (!*records) return LC_ERR_ARG;
        std::memcpy(*records, out.data(), out.size() * sizeof(lc_ml_record_t));
    }
    return LC_OK;
}
extern "C" void lc_multiline_free_records(lc_ml_record_t* r) { std::free(r); }
