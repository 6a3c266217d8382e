// processor_go_regex_gpu.cpp -- the Go plugin processor_regex over the device matcher (include/lc_go_regex.h).
//
// Mirrors plugins/processor/regex/regex.go: fields keep their Go names (:32-41), Init = :50-66, ProcessLogs/ProcessLog =
// :72-101, shouldKeepSource = :103-105, processRegex = :107-129 -- with the per-log FindStringSubmatchIndex replaced by ONE
// batched device call over the SourceKey values of all logs.
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/lc_go_regex.h"
#include "json_min.hpp"
#include "regex_handle.hpp"
#include "processor_grok_gpu.hpp"  // lcgrok::Log / LogContent: the protocol.Log shape

namespace lcgrok {

class ProcessorRegexGpu {
public:
    std::string Regex;
    std::vector<std::string> Keys;
    bool FullMatch = false;
    bool NoKeyError = false;
    bool NoMatchError = true;
    bool KeepSource = false;
    bool KeepSourceIfParseError = true;
    std::string SourceKey;

    ~ProcessorRegexGpu() { lc_regex_free(re); }

    void Init() {
        if (Keys.empty()) throw GrokError("no regex key error");                      // :52-54
        char err[256];
        // regexp.Compile("(?s)" + Regex) :57 -- '.' matches '\n' (our default), single-line anchors, RE2 classes, search
        const uint32_t flags = LC_SYNTAX_SEARCH | LC_SYNTAX_NO_MULTILINE | LC_SYNTAX_REGEXP2;
        if (lc_regex_compile(Regex.data(), Regex.size(), flags, LC_ENGINE_AUTO, &re, err, sizeof err) != LC_OK)
            throw GrokError(std::string("init regex error: ") + err);
        groups = uint32_t(lc_regex_mark_count(re));  // group 1 = whole match
    }

    void ProcessLogs(std::vector<Log>& logs) {
        struct Ref {
            uint32_t log, content;
        };
        std::vector<Ref> refs;
        std::vector<uint8_t> data;
        std::vector<uint32_t> off, len;
        for (uint32_t l = 0; l < logs.size(); ++l)
            for (uint32_t c = 0; c < logs[l].Contents.size(); ++c) {
                const auto& cont = logs[l].Contents[c];
                if (!SourceKey.empty() && SourceKey != cont.Key) continue;               // :80
                refs.push_back({l, c});
                off.push_back(uint32_t(data.size()));
                len.push_back(uint32_t(cont.Value.size()));
                data.insert(data.end(), cont.Value.begin(), cont.Value.end());
                break;                                                                    // :86 only the first one
            }
        if (refs.empty()) return;
        data.resize(data.size() + 16);
        std::vector<int32_t> caps(refs.size() * 2 * size_t(groups));
        std::vector<uint8_t> status(refs.size());
        const int rc = lc_regex_match_host(re, data.data(), off.data(), len.data(), uint32_t(refs.size()), groups, caps.data(),
                                           status.data());
        if (rc != LC_OK) throw GrokError(std::string("device match failed: ") + lc_last_error());
        // LC_OVERFLOW = "not decided" (only possible with the decide pass switched off): never folded into "no match"
        uint64_t gaveUp = 0;
        for (size_t r = 0; r < refs.size(); ++r) {
            if (status[r] == LC_OVERFLOW) throw GrokError("device left a value undecided (LC_NFA_NO_DECIDE is set): logs untouched");
            gaveUp += status[r] == LC_GAVE_UP;  // (Go's regexp never gives up; here such a value is a parse error, and counted)
        }
        if (gaveUp) lcNoteGaveUp(gaveUp);
        for (size_t r = 0; r < refs.size(); ++r) {
            Log& log = logs[refs[r].log];
            const int32_t* c = &caps[r * 2 * groups];
            const char* val = reinterpret_cast<const char*>(data.data()) + off[r];
            bool ok = status[r] == LC_MATCH;                                              // processRegex :107-129
            if (ok && FullMatch && (c[0] != 0 || uint32_t(c[1]) != len[r])) ok = false;
            if (ok && groups - 1 < Keys.size()) ok = false;                               // :117 fewer groups than keys
            if (ok)
                for (size_t i = 0; i < Keys.size(); ++i) {
                    const int32_t b = c[2 * (i + 1)], e = c[2 * (i + 1) + 1];
                    if (b >= 0 && e >= b) log.Contents.push_back({Keys[i], std::string(val + b, size_t(e - b))});
                }
            if (!(KeepSource || (KeepSourceIfParseError && !ok)))                         // shouldKeepSource :103-105
                log.Contents.erase(log.Contents.begin() + refs[r].content);
        }
    }

    lc_regex_t* re = nullptr;
    uint32_t groups = 0;
};

}  // namespace lcgrok

struct lc_goregex {
    lcgrok::ProcessorRegexGpu p;
};

extern "C" int lc_goregex_create(const char* config_json, size_t config_len, lc_goregex_t** out, char* err, size_t errcap) {
    if (!config_json || !out) return LC_ERR_ARG;
    *out = nullptr;
    auto set = [&](const std::string& m) {
        if (err && errcap) std::snprintf(err, errcap, "%s", m.c_str());
    };
    auto g = std::make_unique<lc_goregex>();
    try {
        const lcjson::Value cfg = lcjson::parse(std::string(config_json, config_len));
        if (!cfg.isObject()) throw lcgrok::GrokError("config must be a JSON object");
        if (const lcjson::Value* v = cfg.find("Regex"))
            if (v->isString()) g->p.Regex = v->str;
        if (const lcjson::Value* v = cfg.find("SourceKey"))
            if (v->isString()) g->p.SourceKey = v->str;
        if (const lcjson::Value* v = cfg.find("Keys")) {
            if (!v->isArray()) throw lcgrok::GrokError("Keys must be an array of strings");
            for (const auto& e : v->arr) {
                if (!e.isString()) throw lcgrok::GrokError("Keys must be an array of strings");
                g->p.Keys.push_back(e.str);
            }
        }
        auto boolean = [&](const char* key, bool& dst) {
            if (const lcjson::Value* v = cfg.find(key))
                if (v->isBool()) dst = v->b;
        };
        boolean("FullMatch", g->p.FullMatch);
        boolean("NoKeyError", g->p.NoKeyError);
        boolean("NoMatchError", g->p.NoMatchError);
        boolean("KeepSource", g->p.KeepSource);
        boolean("KeepSourceIfParseError", g->p.KeepSourceIfParseError);
        g->p.Init();
    } catch (const std::exception& e) {
        set(e.what());
        return LC_ERR_UNSUPPORTED;
    }
    set("");
    *out = g.release();
    return LC_OK;
}
extern "C" void lc_goregex_free(lc_goregex_t* p) { delete p; }
extern "C" lc_regex_t* lc_goregex_regex(lc_goregex_t* p) { return p ? p->p.re : nullptr; }

extern "C" int lc_goregex_process_logs_json(lc_goregex_t* g, const char* logs_json, size_t len, char** out_json) {
    if (!g || !logs_json || !out_json) return LC_ERR_ARG;
    *out_json = nullptr;
    try {
        const lcjson::Value in = lcjson::parse(std::string(logs_json, len));
        if (!in.isArray()) return LC_ERR_ARG;
        std::vector<lcgrok::Log> logs;
        for (const auto& l : in.arr) {
            if (!l.isArray()) return LC_ERR_ARG;
            lcgrok::Log log;
            for (const auto& c : l.arr) {
                if (!c.isArray() || c.arr.size() != 2 || !c.arr[0].isString() || !c.arr[1].isString()) return LC_ERR_ARG;
                log.Contents.push_back({c.arr[0].str, c.arr[1].str});
            }
            logs.push_back(std::move(log));
        }
        g->p.ProcessLogs(logs);
        lcjson::Value out = lcjson::Value::makeArray();
        for (const auto& log : logs) {
            lcjson::Value l = lcjson::Value::makeArray();
            for (const auto& c : log.Contents) {
                lcjson::Value pair = lcjson::Value::makeArray();
                pair.arr.push_back(lcjson::Value::makeString(c.Key));
                pair.arr.push_back(lcjson::Value::makeString(c.Value));
                l.arr.push_back(std::move(pair));
            }
            out.arr.push_back(std::move(l));
        }
        const std::string text = lcjson::dump(out);
        *out_json = static_cast<char*>(std::malloc(text.size() + 1));
        if (!*out_json) return LC_ERR_ARG;
        std::memcpy(*out_json, text.c_str(), text.size() + 1);
        return LC_OK;
    } catch (const lcgrok::GrokError&) {
        return lc_device_count() <= 0 ? LC_ERR_NO_DEVICE : LC_ERR_HIP;
    } catch (const std::exception&) {
        return LC_ERR_ARG;
    }
}
extern "C" void lc_goregex_free_string(char* s) { std::free(s); }
