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
#include "multiline_scan.hpp"

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
// The MERGE processor does not compile the strings: it matches with the regexes MultilineOptions itself holds
// (mMultiline.Get*PatternReg(), ProcessorMergeMultilineLogNative.cpp:219-224,244-262) -- the STRIPPED forms of ParseRegex :250-266.
// A pattern that is empty after the stripping is not there ("END$" matches a line that merely begins with END, ".*" is no start
// pattern).  Behind a prefix search  X(.*)*  and  X  decide alike, so the string as written is compiled when only ".*"s went (the
// same automaton as the splitter's); the stripped form when a '$' went, or when only the stripped form is a valid regex.
static int parseRegexStripped(const std::string& pattern, lc_regex_t** out, std::string& err) {
    const std::string t = trimmed(pattern);
    if (t.empty()) return LC_OK;
    char buf[256];
    lc_regex_t* stripped = nullptr;  // validity is the stripped form's ("a\\.*" is ignored with a warning: "a\\" is no regex)
    const int rcT = lc_regex_compile(t.data(), t.size(), LC_SYNTAX_PREFIX, LC_ENGINE_AUTO, &stripped, buf, sizeof buf);
    if (rcT == LC_ERR_SYNTAX) {
        err = buf;
        return rcT;
    }
    if (!endsWith(pattern, "$")) {
        lc_regex_t* written = nullptr;
        if (lc_regex_compile(pattern.data(), pattern.size(), LC_SYNTAX_PREFIX, LC_ENGINE_AUTO, &written, buf, sizeof buf) == LC_OK) {
            lc_regex_free(stripped);
            *out = written;
            return LC_OK;
        }
    }
    if (rcT != LC_OK) err = buf;
    *out = stripped;
    return rcT;
}

// regPtrForm: the patterns as MultilineOptions' own regexes hold them (the merge processor); otherwise as the splitter compiles them
static int createMultiline(const char* config_json, size_t config_len, bool regPtrForm, lc_multiline_t** out, char* err, size_t errcap) {
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
            const int rc = regPtrForm ? parseRegexStripped(pattern, regs[i], why) : parseRegex(pattern, regs[i], why);
            if (rc == LC_OK) {
                kept[i] = pattern;
            } else if (rc == LC_ERR_SYNTAX) {  // :109-118 -- a warning, the pattern counts as not given
                m->warnings += std::string("string param Multiline.") + names[i] + " is not a valid regex: " + why + "\n";
            } else {
                throw std::runtime_error(std::string("Multiline.") + names[i] + ": " + why);
            }
        }
        if (regPtrForm) {  // MultilineOptions::Init :170-200, which the splitter's own regex vectors do not see
            if (m->start && m->cont && m->end) {
                lc_regex_free(m->cont);
                m->cont = nullptr;
                m->warnings += "none of param Multiline.StartPattern, Multiline.ContinuePattern and Multiline.EndPattern are empty: ignore param "
                               "Multiline.ContinuePattern\n";
            } else if (!m->start && !m->end && m->cont) {
                lc_regex_free(m->cont);
                m->cont = nullptr;
                m->warnings += "param Multiline.StartPattern and EndPattern are empty but ContinuePattern is not: ignore multiline config\n";
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
extern "C" int lc_multiline_create(const char* config_json, size_t config_len, lc_multiline_t** out, char* err, size_t errcap) {
    return createMultiline(config_json, config_len, false, out, err, errcap);
}
int lcMultilineCreateForMerge(const char* config_json, size_t config_len, lc_multiline_t** out, char* err, size_t errcap) {
    return createMultiline(config_json, config_len, true, out, err, errcap);
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
    // one device trip: upload, split, a status-only launch per pattern, the record scan (multiline_device.hip); what comes back
    // is what ProcessEvent :161-298 + HandleUnmatchLogs :341-380 would have emitted, in their order
    std::vector<lc_ml_record_t> out;
    uint32_t counts[ML_CNT_WORDS];
    const int rc = lcMultilineSplitTrip(m, data, nbytes, out, counts);
    if (rc != LC_OK) return rc;
    if (counters) {
        counters[0] = counts[ML_CNT_ITEMS];
        counters[1] = counts[ML_CNT_UNMATCHED];
        counters[2] = counts[ML_CNT_MATCHED_LOGS];
    }
    *nrecords = uint32_t(out.size());
    if (!out.empty()) {
        *records = static_cast<lc_ml_record_t*>(std::malloc(out.size() * sizeof(lc_ml_record_t)));
        if (!*records) return LC_ERR_ARG;
        std::memcpy(*records, out.data(), out.size() * sizeof(lc_ml_record_t));
    }
    return LC_OK;
}

// The kernel's four phases with the threads run one after the other (multiline_scan.hpp is the code of both).
namespace {
struct MlHostWriter {
    lc_ml_record_t* records;
    uint32_t recCap;
    const uint32_t* off;
    uint32_t n, nbytes, lastItemIsLast;
    std::vector<MlJob>* jobs;
    void record(uint32_t slot, uint32_t first, uint32_t last, uint32_t matched, uint32_t emitter) const {
        if (slot >= recCap) return;
        const uint32_t fl = matched | ((emitter >= n || (emitter + 1 == n && lastItemIsLast)) ? LC_ML_LAST : 0u);
        if (off) records[slot] = lc_ml_record_t{off[first], (matched & 1u) && emitter >= n ? nbytes - off[first] : off[last + 1] - 1u - off[first], fl};
        else records[slot] = lc_ml_record_t{first, last - first + 1u, fl};
    }
    void job(const MlJob& j) const { jobs->push_back(j); }
};
}  // namespace
extern "C" int lc_multiline_bounds_model(uint32_t mode, const uint8_t* flags, uint32_t n, const uint32_t* off, uint32_t nbytes,
                                         lc_ml_record_t* records, uint32_t record_cap, uint32_t counts[LC_ML_CNT_WORDS]) {
    if (!counts || (n && !flags) || (record_cap && !records)) return LC_ERR_ARG;
    static_assert(LC_ML_RUN == ML_REC_RUN, "header and scan agree");
    static_assert(int(LC_ML_CNT_WORDS) == int(ML_CNT_WORDS) && LC_ML_FLUSH == ML_FLUSH && LC_ML_DISCARD == ML_DISCARD, "header and scan agree");
    std::memset(counts, 0, ML_CNT_WORDS * 4);
    const uint32_t slices = mlSliceCount(n);
    std::vector<MlSummary> summaries(2 * size_t(slices) + 2);
    std::vector<MlEntry> entries(size_t(slices) + 1);
    for (uint32_t t = 0; t < slices; ++t) mlPhaseA(t, n, mode, flags, &summaries[2 * size_t(t)]);
    MlJob flush{};
    uint32_t flushMatchedFirst = kMlInherit;
    mlPhaseB(n, mode, summaries.data(), entries.data(), counts, flush, flushMatchedFirst);
    std::vector<MlJob> jobs;
    if (flush.first != kMlInherit) jobs.push_back(flush);
    MlHostWriter w{records, record_cap, off, n, nbytes, (off && n && off[n] == nbytes + 1u) ? 1u : 0u, &jobs};
    for (uint32_t t = 0; t < slices; ++t) mlPhaseC(t, n, mode, flags, entries[t], w);
    if (flushMatchedFirst != kMlInherit) w.record(flush.recBase, flushMatchedFirst, n - 1, 1u, n);
    for (const MlJob& job : jobs)
        for (uint32_t k = job.first; k <= job.last; ++k) w.record(job.recBase + (k - job.first), k, k, k > job.first ? ML_REC_RUN : 0u, job.emitter);
    return LC_OK;
}
extern "C" void lc_multiline_free_records(lc_ml_record_t* r) { std::free(r); }
