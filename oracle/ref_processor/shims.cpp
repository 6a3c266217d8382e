// oracle/ref_processor/shims.cpp -- TEST INFRASTRUCTURE.  What the reference sources compiled into oracle/_ref/libref_processor.so
// (Makefile: ProcessorParseRegexNative.cpp, CommonParserOptions.cpp, ParamExtractor.cpp, Processor.cpp, the constants) reference at link
// time from parts of the agent that are not compiled here, in the smallest form that lets the processor run:
//   * AppConfig: one process thread, parse alarms on;  ProcessorRunner::sThreadNo = 0
//   * AlarmManager: SendAlarm RECORDS (type, level, message) -- the harness below hands them to the test;  no low-level logging
//   * MetricsRecordRef::CreateCounter: a plain Counter, remembered by name for the harness
//   * CollectionPipelineContext::GetProjectName / GetLogstoreName / GetRegion: empty strings
//   * common/StringTools.cpp is NOT compiled (boost::split, boost::filesystem, regex_replace ...): the two functions of it the processor
//     calls are restated here -- SplitString (:118-123) and BoostRegexMatch (:183-211), the latter over stubs/boost/regex.hpp, i.e. the
//     oracle's matcher (oracle/bt_regex.c)
// and the C harness tests/test_processor_host_double.py drives (refp_*).
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "app_config/AppConfig.h"
#include "collection_pipeline/CollectionPipelineContext.h"
#include "common/StringTools.h"
#include "monitor/AlarmManager.h"
#include "monitor/MetricManager.h"
#include "plugin/processor/ProcessorFilterNative.h"
#include "collection_pipeline/plugin/creator/CProcessor.h"
#include "collection_pipeline/plugin/creator/DynamicCProcessorCreator.h"
#include "collection_pipeline/plugin/instance/ProcessorInstance.h"
#include "common/DynamicLibHelper.h"
#include "monitor/MetricManager.h"
#include "plugin/processor/ProcessorParseRegexNative.h"
#include "protobuf/sls/LogGroupSerializer.h"
#include "plugin/processor/inner/ProcessorMergeMultilineLogNative.h"
#include "plugin/processor/inner/ProcessorSplitLogStringNative.h"
#include "plugin/processor/inner/ProcessorSplitMultilineLogStringNative.h"
#include "runner/ProcessorRunner.h"

// tests/native/ref_group_io.cpp (fixture JSON <-> the reference's PipelineEventGroup), compiled into this library too
bool hdGroupFromJson(logtail::PipelineEventGroup& group, const std::string& json, std::string* error);
std::string hdGroupToJson(const logtail::PipelineEventGroup& group);

namespace {
struct Alarm {
    int type, level;
    std::string message;
};
std::mutex gMutex;
std::vector<Alarm> gAlarms;
std::map<std::string, logtail::CounterPtr> gCounters;  // by metric name: the counters of the processor created last
const std::string kEmpty;
}  // namespace

namespace logtail {
AppConfig::AppConfig() {
    mProcessThreadCount = 1;
    mLogParseAlarmFlag = true;
}
thread_local uint32_t ProcessorRunner::sThreadNo = 0;

AlarmManager::AlarmManager() {}
void AlarmManager::SendAlarm(const AlarmType& alarmType, const AlarmLevel& level, const std::string& message, const std::string&,
                             const std::string&, const std::string&, const std::string&) {
    std::lock_guard<std::mutex> g(gMutex);
    gAlarms.push_back({int(alarmType), int(level), message});
}
bool AlarmManager::IsLowLevelAlarmValid() { return false; }

MetricsRecordRef::~MetricsRecordRef() {}
CounterPtr MetricsRecordRef::CreateCounter(const std::string& name) {
    CounterPtr c = std::make_shared<Counter>(name);
    std::lock_guard<std::mutex> g(gMutex);
    gCounters[name] = c;
    return c;
}

// (ProcessorInstance's own counters, ProcessorInstance.cpp:29-44)
TimeCounterPtr MetricsRecordRef::CreateTimeCounter(const std::string& name) {
    TimeCounterPtr c = std::make_shared<TimeCounter>(name);
    std::lock_guard<std::mutex> g(gMutex);
    gCounters[name] = c;
    return c;
}
const std::string MetricCategory::METRIC_CATEGORY_PLUGIN = "plugin";
WriteMetrics::~WriteMetrics() {}
void WriteMetrics::CreateMetricsRecordRef(MetricsRecordRef&, const std::string&, MetricLabels&&, DynamicMetricLabels&&) {}
void WriteMetrics::CommitMetricsRecordRef(MetricsRecordRef&) {}

const std::string& CollectionPipelineContext::GetProjectName() const { return kEmpty; }
const std::string& CollectionPipelineContext::GetLogstoreName() const { return kEmpty; }
const std::string& CollectionPipelineContext::GetRegion() const { return kEmpty; }

// common/StringTools.cpp:118-123 (boost::split on any character of delim; empty tokens kept)
std::vector<std::string> SplitString(const std::string& str, const std::string& delim) {
    std::vector<std::string> tokens;
    size_t begin = 0;
    for (size_t i = 0; i <= str.size(); ++i)
        if (i == str.size() || delim.find(str[i]) != std::string::npos) {
            tokens.push_back(str.substr(begin, i - begin));
            begin = i + 1;
        }
    return tokens;
}
// common/StringTools.cpp:183-211
bool BoostRegexMatch(const char* buffer, size_t length, const boost::regex& reg, std::string& exception, boost::match_results<const char*>& what,
                     boost::match_flag_type flags) {
    try {
        if (boost::regex_match(buffer, buffer + length, what, reg, flags)) return true;
        return false;
    } catch (boost::regex_error& e) {
        exception.append("regex_error code is ");
        exception.append(e.what());
        return false;
    } catch (std::exception& e) {
        exception.append("exception message: ");
        exception.append(e.what());
        return false;
    } catch (...) {
        exception.append("unknown exception");
        return false;
    }
}
// common/StringTools.cpp:213-236 (the filter's leaves) and :263-288 (the multiline processor's start / continue / end patterns:
// regex_search with match_continuous -- the match must begin at the buffer's first byte); :30-34
bool BoostRegexMatch(const char* buffer, size_t size, const boost::regex& reg, std::string& exception) {
    try {
        return boost::regex_match(buffer, buffer + size, reg);
    } catch (std::exception& e) {
        exception.append("exception message: ");
        exception.append(e.what());
        return false;
    }
}
bool BoostRegexSearch(const char* buffer, size_t size, const boost::regex& reg, std::string& exception) {
    try {
        boost::match_results<const char*> what;
        return boost::regex_search(buffer, buffer + size, what, reg, boost::match_continuous);
    } catch (std::exception& e) {
        exception.append("exception message: ");
        exception.append(e.what());
        return false;
    }
}
// common/TimeUtil.cpp:440-448 (only the metric-event writer of LogGroupSerializer.cpp calls it: the last `length` digits, zero-padded)
std::string NumberToDigitString(uint32_t number, uint8_t length) {
    std::string digits = std::to_string(number);
    if (digits.size() < length) digits.insert(0, length - digits.size(), '0');
    return digits.substr(digits.size() - length);
}
std::string ToLowerCaseString(const std::string& orig) {
    std::string copy = orig;
    for (char& c : copy) c = char(::tolower(static_cast<unsigned char>(c)));
    return copy;
}
}  // namespace logtail

// ------------------------------------------------------------------------------------------------ harness
namespace {
struct RefProcessor {
    logtail::CollectionPipelineContext ctx;
    std::unique_ptr<logtail::Processor> proc;
    std::map<std::string, logtail::CounterPtr> counters;
};
char* dup(const std::string& s) {
    char* p = static_cast<char*>(std::malloc(s.size() + 1));
    if (p) std::memcpy(p, s.c_str(), s.size() + 1);
    return p;
}
}  // namespace

extern "C" {
// ProcessorParseRegexNative on a context of its own: SetContext + Init(config).  nullptr + err when Init returns false.
// kind: the reference plugin's name -- processor_parse_regex_native, processor_split_string_native,
// processor_split_multiline_log_string_native, processor_filter_regex_native, processor_merge_multiline_log_native
void* refp_create_kind(const char* kind, const char* config_json, char* err, size_t errcap) {
    auto p = std::make_unique<RefProcessor>();
    const std::string k = kind ? kind : "";
    if (k == logtail::ProcessorParseRegexNative::sName) p->proc = std::make_unique<logtail::ProcessorParseRegexNative>();
    else if (k == logtail::ProcessorSplitLogStringNative::sName) p->proc = std::make_unique<logtail::ProcessorSplitLogStringNative>();
    else if (k == logtail::ProcessorSplitMultilineLogStringNative::sName) p->proc = std::make_unique<logtail::ProcessorSplitMultilineLogStringNative>();
    else if (k == logtail::ProcessorFilterNative::sName) p->proc = std::make_unique<logtail::ProcessorFilterNative>();
    else if (k == logtail::ProcessorMergeMultilineLogNative::sName) p->proc = std::make_unique<logtail::ProcessorMergeMultilineLogNative>();
    else {
        if (err && errcap) snprintf(err, errcap, "unknown processor %s", k.c_str());
        return nullptr;
    }
    p->ctx.SetConfigName("test_config");
    p->proc->SetContext(p->ctx);
    Json::Value config;
    try {
        config = Json::Value::fromText(config_json);
    } catch (const std::exception& e) {
        if (err && errcap) snprintf(err, errcap, "%s", e.what());
        return nullptr;
    }
    {
        std::lock_guard<std::mutex> g(gMutex);
        gCounters.clear();
    }
    if (!p->proc->Init(config)) {
        if (err && errcap) {
            std::lock_guard<std::mutex> g(gMutex);
            snprintf(err, errcap, "%s", gAlarms.empty() ? "Init returned false" : gAlarms.back().message.c_str());
        }
        return nullptr;
    }
    std::lock_guard<std::mutex> g(gMutex);
    p->counters = gCounters;
    return p.release();
}
void* refp_create(const char* config_json, char* err, size_t errcap) {
    return refp_create_kind(logtail::ProcessorParseRegexNative::sName.c_str(), config_json, err, errcap);
}
void refp_destroy(void* h) { delete static_cast<RefProcessor*>(h); }
// Process(group) through the public interface (Processor::Process(std::vector<PipelineEventGroup>&)) -> the group as fixture JSON
char* refp_process_json(void* h, const char* group_json, char* err, size_t errcap) {
    auto* p = static_cast<RefProcessor*>(h);
    std::vector<logtail::PipelineEventGroup> groups;
    groups.emplace_back(std::make_shared<logtail::SourceBuffer>());
    std::string error;
    if (!hdGroupFromJson(groups[0], group_json, &error)) {
        if (err && errcap) snprintf(err, errcap, "%s", error.c_str());
        return nullptr;
    }
    p->proc->Process(groups);
    return dup(hdGroupToJson(groups[0]));
}
// the same group through two processors, one after the other (the line splitter, then the merge processor: the merge works on events
// that lie back to back in the group's buffer, which is what the splitter leaves)
char* refp_process_chain_json(void* h1, void* h2, const char* group_json, char* err, size_t errcap) {
    std::vector<logtail::PipelineEventGroup> groups;
    groups.emplace_back(std::make_shared<logtail::SourceBuffer>());
    std::string error;
    if (!hdGroupFromJson(groups[0], group_json, &error)) {
        if (err && errcap) snprintf(err, errcap, "%s", error.c_str());
        return nullptr;
    }
    static_cast<RefProcessor*>(h1)->proc->Process(groups);
    static_cast<RefProcessor*>(h2)->proc->Process(groups);
    return dup(hdGroupToJson(groups[0]));
}
// ... and through up to four, for the benchmark pipeline split -> parse -> filter (a null handle is skipped)
char* refp_process_chain4_json(void* h1, void* h2, void* h3, void* h4, const char* group_json, char* err, size_t errcap) {
    std::vector<logtail::PipelineEventGroup> groups;
    groups.emplace_back(std::make_shared<logtail::SourceBuffer>());
    std::string error;
    if (!hdGroupFromJson(groups[0], group_json, &error)) {
        if (err && errcap) snprintf(err, errcap, "%s", error.c_str());
        return nullptr;
    }
    for (void* h : {h1, h2, h3, h4})
        if (h) static_cast<RefProcessor*>(h)->proc->Process(groups);
    return dup(hdGroupToJson(groups[0]));
}
// What the line splitter leaves, built directly: ONE copy of `data` in the group's source buffer, one log event per line whose `key`
// content is a view of its line (timestamp = 1 + the line's index); an event WITHOUT contents in front of every line listed in
// emptyBefore (ascending; the line count = behind the last line), the key "other" on the lines listed in otherKey -- the same group tests/native/multiline_double.cpp builds for the
// product's merge processor.  -> the group after Process, as fixture JSON.
char* refp_process_lines(void* h, const uint8_t* data, size_t nbytes, const char* key, const uint32_t* emptyBefore, uint32_t nEmpty,
                         const uint32_t* otherKey, uint32_t nOther) {
    std::vector<logtail::PipelineEventGroup> groups;
    auto sb = std::make_shared<logtail::SourceBuffer>();
    groups.emplace_back(sb);
    logtail::PipelineEventGroup& group = groups[0];
    const logtail::StringBuffer copy = sb->CopyString(reinterpret_cast<const char*>(data), nbytes);
    const logtail::StringBuffer k = sb->CopyString(key, strlen(key));
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
        const void* nl = memchr(copy.data + at, '\n', nbytes - at);
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
    static_cast<RefProcessor*>(h)->proc->Process(groups);
    return dup(hdGroupToJson(groups[0]));
}
// ------------------------------------------------------------------------------------------------ the C-processor slot (section 8 b)
// A dynamic plugin loaded and driven by the reference's OWN code: the directory rule of PluginRegistry::LoadDynamicPlugins (:239:
// GetProcessExecutionDir() + "/plugins", handed to LoadDynLib as the PREFIX of "lib<type>.so") -> DynamicLibLoader::LoadDynLib / LoadMethod
// (common/DynamicLibHelper.cpp:68-98, compiled) -> the symbol and version check of PluginRegistry::LoadProcessorPlugin (:270-290,
// restated: PluginRegistry.cpp registers every plugin of the agent and is not compiled) -> DynamicCProcessorCreator::Create
// (compiled) -> ProcessorInstance::Init -> DynamicCProcessorProxy::Init hands the plugin &config (a Json::Value) and &context
// (compiled) -> ProcessorInstance::Process with its in / out counters (compiled).
namespace {
struct RefDynamic {
    logtail::CollectionPipelineContext ctx;
    std::unique_ptr<logtail::DynamicCProcessorCreator> creator;
    std::unique_ptr<logtail::PluginInstance> instance;
    std::map<std::string, logtail::CounterPtr> counters;
    std::string pathTried;
};
}  // namespace
void* refp_dyn_load(const char* process_execution_dir, const char* plugin_type, const char* config_json, char* err, size_t errcap) {
    auto fail = [&](const std::string& m) -> void* {
        if (err && errcap) snprintf(err, errcap, "%s", m.c_str());
        return nullptr;
    };
    auto d = std::make_unique<RefDynamic>();
    logtail::AppConfig::GetInstance()->SetProcessExecutionDir(process_execution_dir);
    const std::string pluginType = plugin_type;
    std::string error;
    const std::string pluginDir = logtail::AppConfig::GetInstance()->GetProcessExecutionDir() + "/plugins";  // PluginRegistry.cpp:239
    logtail::DynamicLibLoader loader;
    if (!loader.LoadDynLib(pluginType, error, pluginDir)) return fail("open plugin " + pluginType + ": " + error);  // :241-244
    auto* plugin = static_cast<processor_interface_t*>(loader.LoadMethod("processor_interface", error));          // :272
    if (!error.empty() || !plugin) return fail("load method plugin_interface: " + error);                          // :279-282
    if (plugin->version != PROCESSOR_INTERFACE_VERSION)                                                            // :283-288
        return fail("plugin interface version mismatch: expected " + std::to_string(PROCESSOR_INTERFACE_VERSION) + ", actual " +
                    std::to_string(plugin->version));
    d->creator = std::make_unique<logtail::DynamicCProcessorCreator>(plugin, loader.Release());                    // :289
    d->instance = d->creator->Create(logtail::PluginInstance::PluginMeta("1"));
    d->ctx.SetConfigName("test_config");
    Json::Value config;
    try {
        config = Json::Value::fromText(config_json);
    } catch (const std::exception& e) {
        return fail(e.what());
    }
    {
        std::lock_guard<std::mutex> g(gMutex);
        gCounters.clear();
    }
    if (!static_cast<logtail::ProcessorInstance*>(d->instance.get())->Init(config, d->ctx)) {
        // (d goes: ~DynamicCProcessorProxy calls finalize(plugin_state) also after a failed init, DynamicCProcessorProxy.cpp:25-28 -- the
        // plugin must have left a state that call can take)
        return fail("ProcessorInstance::Init returned false");
    }
    std::lock_guard<std::mutex> g(gMutex);
    d->counters = gCounters;
    return d.release();
}
// the reference's own processor_parse_regex_native in the same wrapper (ProcessorInstance), for the side-by-side: what the pipeline sees
// of a static plugin -- events and in / out counters -- is what it sees of the dynamic one.  Driven by the refp_dyn_* calls.
void* refp_static_instance(const char* config_json, char* err, size_t errcap) {
    auto d = std::make_unique<RefDynamic>();
    d->instance = std::make_unique<logtail::ProcessorInstance>(new logtail::ProcessorParseRegexNative, logtail::PluginInstance::PluginMeta("1"));
    d->ctx.SetConfigName("test_config");
    Json::Value config;
    try {
        config = Json::Value::fromText(config_json);
    } catch (const std::exception& e) {
        if (err && errcap) snprintf(err, errcap, "%s", e.what());
        return nullptr;
    }
    {
        std::lock_guard<std::mutex> g(gMutex);
        gCounters.clear();
    }
    if (!static_cast<logtail::ProcessorInstance*>(d->instance.get())->Init(config, d->ctx)) {
        if (err && errcap) snprintf(err, errcap, "ProcessorInstance::Init returned false");
        return nullptr;
    }
    std::lock_guard<std::mutex> g(gMutex);
    d->counters = gCounters;
    return d.release();
}
const char* refp_dyn_name(void* h) { return static_cast<RefDynamic*>(h)->instance->Name().c_str(); }
void refp_dyn_unload(void* h) {
    auto* d = static_cast<RefDynamic*>(h);
    d->instance.reset();  // ~DynamicCProcessorProxy: finalize(plugin_state)
    d->creator.reset();   // ~DynamicCProcessorCreator: CloseLib
    delete d;
}
char* refp_dyn_process_json(void* h, const char* group_json, char* err, size_t errcap) {
    auto* d = static_cast<RefDynamic*>(h);
    std::vector<logtail::PipelineEventGroup> groups;
    groups.emplace_back(std::make_shared<logtail::SourceBuffer>());
    std::string error;
    if (!hdGroupFromJson(groups[0], group_json, &error)) {
        if (err && errcap) snprintf(err, errcap, "%s", error.c_str());
        return nullptr;
    }
    static_cast<logtail::ProcessorInstance*>(d->instance.get())->Process(groups);
    return dup(hdGroupToJson(groups[0]));
}
// {"in_events_total": .., "out_events_total": .., "in_size_bytes": .., "out_size_bytes": .., "total_process_time_ms": ..}: ProcessorInstance's
char* refp_dyn_counters_json(void* h) {
    auto* d = static_cast<RefDynamic*>(h);
    std::string out = "{";
    for (const auto& kv : d->counters) {
        if (out.size() > 1) out += ",";
        out += "\"" + kv.first + "\":" + std::to_string(kv.second->GetValue());
    }
    return dup(out + "}");
}
// ------------------------------------------------------------------------------------------------ the serializer side (section 8 f, rank 4)
// The log events of a group on the SLS wire, by the reference's own writer (protobuf/sls/LogGroupSerializer.cpp, compiled): the two passes
// of SLSEventGroupSerializer -- CalculateLogEventSize (SLSSerializer.cpp:254-268) and SerializeLogEvent (:377-395), restated here because
// SLSSerializer.cpp itself needs the compressor, the batch types and the flusher's context -- over the events a processor left.
// h: a processor to run first, or null.  -> malloc'ed bytes, *out_len
char* refp_sls_serialize_group_json(void* h, const char* group_json, int enable_ns, size_t* out_len, char* err, size_t errcap) {
    std::vector<logtail::PipelineEventGroup> groups;
    groups.emplace_back(std::make_shared<logtail::SourceBuffer>());
    std::string error;
    if (!hdGroupFromJson(groups[0], group_json, &error)) {
        if (err && errcap) snprintf(err, errcap, "%s", error.c_str());
        return nullptr;
    }
    if (h) static_cast<RefProcessor*>(h)->proc->Process(groups);
    const auto& events = groups[0].GetEvents();
    size_t logGroupSZ = 0;
    std::vector<size_t> logSZ(events.size());
    for (size_t i = 0; i < events.size(); ++i) {  // CalculateLogEventSize
        const auto& e = events[i].Cast<logtail::LogEvent>();
        if (e.Empty()) continue;
        size_t contentSZ = 0;
        for (const auto& kv : e) contentSZ += logtail::GetLogContentSize(kv.first.size(), kv.second.size());
        logGroupSZ += logtail::GetLogSize(contentSZ, enable_ns && e.GetTimestampNanosecond(), logSZ[i]);
    }
    logtail::LogGroupSerializer serializer;
    serializer.Prepare(logGroupSZ);
    for (size_t i = 0; i < events.size(); ++i) {  // SerializeLogEvent
        const auto& e = events[i].Cast<logtail::LogEvent>();
        if (e.Empty()) continue;
        serializer.StartToAddLog(logSZ[i]);
        serializer.AddLogTime(e.GetTimestamp());
        for (const auto& kv : e) serializer.AddLogContent(kv.first, kv.second);
        if (enable_ns && e.GetTimestampNanosecond()) serializer.AddLogTimeNs(e.GetTimestampNanosecond().value());
    }
    const std::string& res = serializer.GetResult();
    char* out = static_cast<char*>(malloc(res.size() + 1));
    memcpy(out, res.data(), res.size());
    *out_len = res.size();
    return out;
}
// The same writer fed from a COLUMNAR table (the product's lc_columnar_t, include/lc_processor.h, passed as plain arrays): no LogEvent
// holds the fields; the size pass is content_bytes[i] as the product computed it, the write pass takes key k and the span
// [begin, end) of the event's source value.  state[i] == 1: parsed (others are not written).  ns[i] < 0: no nanosecond part.
char* refp_sls_serialize_columnar(uint32_t n_events, uint32_t n_keys, const char* const* keys, const uint32_t* key_len,
                                  const uint8_t* const* base, const int32_t* spans, const uint8_t* state, const uint64_t* content_bytes,
                                  const uint32_t* timestamps, const int64_t* ns, int enable_ns, size_t* out_len) {
    size_t logGroupSZ = 0;
    std::vector<size_t> logSZ(n_events);
    for (uint32_t i = 0; i < n_events; ++i)
        if (state[i] == 1) logGroupSZ += logtail::GetLogSize(size_t(content_bytes[i]), enable_ns && ns[i] >= 0, logSZ[i]);
    logtail::LogGroupSerializer serializer;
    serializer.Prepare(logGroupSZ);
    for (uint32_t i = 0; i < n_events; ++i) {
        if (state[i] != 1) continue;
        serializer.StartToAddLog(logSZ[i]);
        serializer.AddLogTime(timestamps[i]);
        for (uint32_t k = 0; k < n_keys; ++k) {
            const int32_t b = spans[(size_t(i) * n_keys + k) * 2], e = spans[(size_t(i) * n_keys + k) * 2 + 1];
            const logtail::StringView value = b < 0 ? logtail::StringView() : logtail::StringView(reinterpret_cast<const char*>(base[i]) + b, size_t(e - b));
            serializer.AddLogContent(logtail::StringView(keys[k], key_len[k]), value);
        }
        if (enable_ns && ns[i] >= 0) serializer.AddLogTimeNs(uint32_t(ns[i]));
    }
    const std::string& res = serializer.GetResult();
    char* out = static_cast<char*>(malloc(res.size() + 1));
    memcpy(out, res.data(), res.size());
    *out_len = res.size();
    return out;
}
// discarded, out_failed, out_key_not_found, out_successful
void refp_counters(void* h, uint64_t out[4]) {
    auto* p = static_cast<RefProcessor*>(h);
    const std::string* names[4] = {&logtail::METRIC_PLUGIN_DISCARDED_EVENTS_TOTAL, &logtail::METRIC_PLUGIN_OUT_FAILED_EVENTS_TOTAL,
                                   &logtail::METRIC_PLUGIN_OUT_KEY_NOT_FOUND_EVENTS_TOTAL, &logtail::METRIC_PLUGIN_OUT_SUCCESSFUL_EVENTS_TOTAL};
    for (int i = 0; i < 4; ++i) {
        auto it = p->counters.find(*names[i]);
        out[i] = it == p->counters.end() ? 0 : it->second->GetValue();
    }
}
// the alarms recorded since the last call: [[type, level, message], ...]
char* refp_take_alarms() {
    std::vector<Alarm> taken;
    {
        std::lock_guard<std::mutex> g(gMutex);
        taken.swap(gAlarms);
    }
    lcjson::Value arr = lcjson::Value::makeArray();
    for (const Alarm& a : taken) {
        lcjson::Value e = lcjson::Value::makeArray();
        e.arr.push_back(lcjson::Value::makeInt(a.type));
        e.arr.push_back(lcjson::Value::makeInt(a.level));
        e.arr.push_back(lcjson::Value::makeString(a.message));
        arr.arr.push_back(std::move(e));
    }
    return dup(lcjson::dump(arr));
}
// every counter the processor created, by metric name: {"name": value, ...}
char* refp_counters_json(void* h) {
    auto* p = static_cast<RefProcessor*>(h);
    lcjson::Value obj = lcjson::Value::makeObject();
    for (const auto& kv : p->counters) obj.set(kv.first, lcjson::Value::makeInt(int64_t(kv.second->GetValue())));
    return dup(lcjson::dump(obj));
}
int refp_regex_match_alarm_type() { return int(logtail::REGEX_MATCH_ALARM); }
void refp_free(char* p) { std::free(p); }
}
