// c_processor_slot.cpp -- C ABI of the processor layer (include/lc_processor.h) and the `processor_interface` data
// symbol LoongCollector's PluginRegistry looks up with dlsym
// (core/collection_pipeline/plugin/PluginRegistry.cpp:270-290, struct layout CProcessor.h:23-45).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "../../include/lc_processor.h"
#include "processor_parse_regex_gpu.hpp"
#include "processor_pipeline_gpu.hpp"
#ifdef LC_USE_REFERENCE_HEADERS
#include "json/json.h"  // the agent hands init() a Json::Value* (DynamicCProcessorProxy.cpp:30-32)
#endif

using logtail::PipelineEventGroup;
using logtail::ProcessorParseRegexGpu;

struct lc_processor {
    ProcessorParseRegexGpu impl;
    // the part ProcessorInstance adds around every plugin (ProcessorInstance.cpp:46-63)
    std::atomic<uint64_t> inEvents{0}, outEvents{0}, inBytes{0}, outBytes{0}, processUs{0};
};

#ifndef LC_USE_REFERENCE_HEADERS
struct lc_event_group {
    std::shared_ptr<logtail::SourceBuffer> buffer = std::make_shared<logtail::SourceBuffer>();
    PipelineEventGroup group{buffer};
};
#endif

static void setErr(char* err, size_t cap, const std::string& msg) {
    if (err && cap) std::snprintf(err, cap, "%s", msg.c_str());
}

static int processGroup(lc_processor* p, PipelineEventGroup& group) {
    p->inEvents += group.GetEvents().size();
#ifdef LC_USE_REFERENCE_HEADERS
    p->inBytes += group.DataSize();
    const auto t0 = std::chrono::steady_clock::now();
    p->impl.Process(group);
    p->processUs += uint64_t(std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count());
    p->outBytes += group.DataSize();
#else
    // group.DataSize() before and after, without two more walks over the events: the processor sums the events' sizes as it goes
    ProcessorParseRegexGpu::EventBytes bytes;
    const auto t0 = std::chrono::steady_clock::now();
    p->impl.Process(group, &bytes);
    p->processUs += uint64_t(std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count());
    const size_t rest = group.DataSizeWithoutEvents();  // (tags and the container: the processor does not touch them)
    p->inBytes += rest + bytes.in;
    p->outBytes += rest + bytes.out;
#endif
    p->outEvents += group.GetEvents().size();
    return 0;
}

extern "C" int lc_processor_create(const char* config_json, lc_processor_t** out, char* err, size_t errcap) {
    if (!config_json || !out) return LC_ERR_ARG;
    *out = nullptr;
    lcjson::Value cfg;
    try {
        cfg = lcjson::parse(config_json);
    } catch (const std::exception& e) {
        setErr(err, errcap, e.what());
        return LC_ERR_ARG;
    }
    auto p = std::make_unique<lc_processor>();
    if (const lcjson::Value* eng = cfg.find("_Engine")) {  // test hook: "tdfa" | "nfa"
        if (eng->str == "tdfa") p->impl.mEngineChoice = LC_ENGINE_TDFA;
        if (eng->str == "nfa") p->impl.mEngineChoice = LC_ENGINE_NFA;
    }
    std::string error;
    if (!p->impl.Init(cfg, error)) {
        setErr(err, errcap, error);
        return LC_ERR_SYNTAX;
    }
    setErr(err, errcap, "");
    *out = p.release();
    return LC_OK;
}

extern "C" void lc_processor_destroy(lc_processor_t* p) { delete p; }

extern "C" int lc_processor_key_count(const lc_processor_t* p) { return p ? int(p->impl.mKeys.size()) : -1; }
extern "C" const char* lc_processor_key(const lc_processor_t* p, int i) {
    if (!p || i < 0 || size_t(i) >= p->impl.mKeys.size()) return nullptr;
    return p->impl.mKeys[size_t(i)].c_str();
}

#ifndef LC_USE_REFERENCE_HEADERS
extern "C" int lc_processor_process(lc_processor_t* p, lc_event_group_t* g) {
    if (!p || !g) return LC_ERR_ARG;
    // a group that needs the device must not be half-processed when there is none: check first, fail loudly
    if (!p->impl.IsWholeLineMode() && lc_device_count() <= 0) {
        bool needsDevice = false;
        for (const auto& e : g->group.GetEvents())
            if (e.Is<logtail::LogEvent>() && e.Cast<logtail::LogEvent>().HasContent(p->impl.mSourceKey)) needsDevice = true;
        if (needsDevice) return LC_ERR_NO_DEVICE;
    }
    return processGroup(p, g->group);
}

#endif

extern "C" void lc_processor_set_alarm_sink(lc_processor_t* p, lc_alarm_sink_t sink, void* user) {
    if (p) p->impl.SetAlarmSink(sink, user);
}

extern "C" int lc_processor_counters(const lc_processor_t* p, uint64_t out[LC_CNT_COUNT]) {
    if (!p || !out) return LC_ERR_ARG;
    out[LC_CNT_DISCARDED_EVENTS] = p->impl.mDiscardedEventsTotal;
    out[LC_CNT_OUT_FAILED_EVENTS] = p->impl.mOutFailedEventsTotal;
    out[LC_CNT_OUT_KEY_NOT_FOUND] = p->impl.mOutKeyNotFoundEventsTotal;
    out[LC_CNT_OUT_SUCCESSFUL_EVENTS] = p->impl.mOutSuccessfulEventsTotal;
    out[LC_CNT_IN_EVENTS] = p->inEvents;
    out[LC_CNT_OUT_EVENTS] = p->outEvents;
    out[LC_CNT_IN_SIZE_BYTES] = p->inBytes;
    out[LC_CNT_OUT_SIZE_BYTES] = p->outBytes;
    out[LC_CNT_PROCESS_TIME_US] = p->processUs;
    out[LC_CNT_COMPLEXITY_EXCEEDED] = p->impl.mComplexityExceededEventsTotal;
    out[LC_CNT_UNDECIDED_EVENTS] = p->impl.mUndecidedEventsTotal;
    out[LC_CNT_DEVICE_FAILED_EVENTS] = p->impl.mDeviceFailedEventsTotal;
    return LC_OK;
}

#ifndef LC_USE_REFERENCE_HEADERS  // fixture helpers exist only in the standalone build
extern "C" lc_event_group_t* lc_group_from_json(const char* json, char* err, size_t errcap) {
    if (!json) return nullptr;
    auto g = std::make_unique<lc_event_group>();
    std::string error;
    if (!g->group.FromJsonString(json, &error)) {
        setErr(err, errcap, error);
        return nullptr;
    }
    setErr(err, errcap, "");
    return g.release();
}

// The group a file input hands over (LogFileReader read buffer -> ProcessorSplitLogStringNative,
// ProcessorSplitLogStringNative.cpp:130-160): ONE copy of the lines, back to back with a separator byte, in the group's
// SourceBuffer; one LogEvent per line whose `key` content is a view into that buffer.  bench.py and the tests use it to build
// in-agent shaped input without going through JSON.
extern "C" lc_event_group_t* lc_group_from_lines(const uint8_t* data, const uint32_t* off, const uint32_t* len, uint32_t n,
                                                 const char* key) {
    if (!data || !off || !len || !key) return nullptr;
    auto g = std::make_unique<lc_event_group>();
    size_t total = 0;
    for (uint32_t i = 0; i < n; ++i) total += size_t(len[i]) + 1;
    logtail::StringBuffer buf = g->group.GetSourceBuffer()->AllocateStringBuffer(total);
    const logtail::StringBuffer keyBuf = g->group.GetSourceBuffer()->CopyString(key, std::strlen(key));
    const logtail::StringView keyView(keyBuf.data, keyBuf.size);
    size_t at = 0;
    for (uint32_t i = 0; i < n; ++i) {
        std::memcpy(buf.data + at, data + off[i], len[i]);
        buf.data[at + len[i]] = '\n';
        logtail::LogEvent* ev = g->group.AddLogEvent();
        ev->SetTimestamp(1);
        ev->SetContentNoCopy(keyView, logtail::StringView(buf.data + at, len[i]));
        at += size_t(len[i]) + 1;
    }
    return g.release();
}

// a group as the file input hands it over: ONE log event holding a copy of the read buffer (LogFileReader: content + position)
extern "C" lc_event_group_t* lc_group_from_buffer(const uint8_t* data, size_t n, const char* key, uint64_t file_offset,
                                                  const char* file_offset_key) {
    if ((!data && n) || !key) return nullptr;
    auto g = std::make_unique<lc_event_group>();
    logtail::StringBuffer buf = g->group.GetSourceBuffer()->AllocateStringBuffer(n);
    if (n) std::memcpy(buf.data, data, n);
    const logtail::StringBuffer keyBuf = g->group.GetSourceBuffer()->CopyString(key, std::strlen(key));
    logtail::LogEvent* ev = g->group.AddLogEvent();
    ev->SetTimestamp(1);
    ev->SetContentNoCopy(logtail::StringView(keyBuf.data, keyBuf.size), logtail::StringView(buf.data, n));
    ev->SetPosition(file_offset, n);
    if (file_offset_key) g->group.SetMetadata(logtail::EventGroupMetaKey::LOG_FILE_OFFSET_KEY, file_offset_key);
    return g.release();
}

// ---- the fused split -> parse -> filter pipeline (processor_pipeline_gpu.hpp)
struct lc_pipeline {
    logtail::ProcessorPipelineGpu impl;
};
extern "C" int lc_pipeline_create(const char* config_json, lc_pipeline_t** out, char* err, size_t errcap) {
    if (!config_json || !out) return LC_ERR_ARG;
    *out = nullptr;
    lcjson::Value cfg;
    try {
        cfg = lcjson::parse(config_json);
    } catch (const std::exception& e) {
        setErr(err, errcap, e.what());
        return LC_ERR_ARG;
    }
    auto p = std::make_unique<lc_pipeline>();
    std::string error;
    if (!p->impl.Init(cfg, error)) {
        setErr(err, errcap, error);
        return LC_ERR_SYNTAX;
    }
    setErr(err, errcap, "");
    *out = p.release();
    return LC_OK;
}
extern "C" void lc_pipeline_destroy(lc_pipeline_t* p) { delete p; }
extern "C" int lc_pipeline_is_fused(const lc_pipeline_t* p) { return p && p->impl.IsFused(); }
extern "C" int lc_pipeline_process(lc_pipeline_t* p, lc_event_group_t* g) {
    if (!p || !g) return LC_ERR_ARG;
    if (lc_device_count() <= 0) return LC_ERR_NO_DEVICE;
    std::string error;
    if (!p->impl.Process(g->group, error)) {
        std::fprintf(stderr, "[%s] %s\n", logtail::ProcessorPipelineGpu::sName.c_str(), error.c_str());
        return lc_device_count() <= 0 ? LC_ERR_NO_DEVICE : LC_ERR_HIP;
    }
    return LC_OK;
}
extern "C" int lc_pipeline_counters(const lc_pipeline_t* p, uint64_t parse[LC_CNT_COUNT], uint64_t pipe[LC_PIPE_CNT_COUNT]) {
    if (!p || !parse || !pipe) return LC_ERR_ARG;
    for (int i = 0; i < LC_CNT_COUNT; ++i) parse[i] = 0;
    parse[LC_CNT_DISCARDED_EVENTS] = p->impl.mParse.mDiscardedEventsTotal;
    parse[LC_CNT_OUT_FAILED_EVENTS] = p->impl.mParse.mOutFailedEventsTotal;
    parse[LC_CNT_OUT_KEY_NOT_FOUND] = p->impl.mParse.mOutKeyNotFoundEventsTotal;
    parse[LC_CNT_OUT_SUCCESSFUL_EVENTS] = p->impl.mParse.mOutSuccessfulEventsTotal;
    parse[LC_CNT_COMPLEXITY_EXCEEDED] = p->impl.mParse.mComplexityExceededEventsTotal;
    parse[LC_CNT_UNDECIDED_EVENTS] = p->impl.mParse.mUndecidedEventsTotal;
    parse[LC_CNT_DEVICE_FAILED_EVENTS] = p->impl.mParse.mDeviceFailedEventsTotal;
    pipe[LC_PIPE_FILTER_IN_EVENTS] = p->impl.mFilter.mInEventsTotal;
    pipe[LC_PIPE_FILTER_OUT_EVENTS] = p->impl.mFilter.mOutEventsTotal;
    pipe[LC_PIPE_GROUPS_FUSED] = p->impl.mGroupsFused;
    pipe[LC_PIPE_GROUPS_CHAINED] = p->impl.mGroupsChained;
    pipe[LC_PIPE_LINES] = p->impl.mLinesTotal;
    pipe[LC_PIPE_SURVIVORS] = p->impl.mSurvivorsTotal;
    return LC_OK;
}

// ---- columnar hand-off (include/lc_processor.h): gather + device match, nothing stitched
namespace {
struct ColumnarStore {
    lc_columnar_t pub{};
    std::vector<const char*> keys;
    std::vector<uint32_t> keyLen, baseLen;
    std::vector<const uint8_t*> base;
    std::vector<int32_t> spans;
    std::vector<uint8_t> state;
    std::vector<uint64_t> contentBytes;
};
// protobuf sizes of LogGroupSerializer.cpp:227-252
inline uint64_t varintSize(uint64_t v) {
    uint64_t n = 1;
    while (v >= 128) {
        v >>= 7;
        ++n;
    }
    return n;
}
inline uint64_t stringSize(uint64_t len) { return 1 + varintSize(len) + len; }
inline uint64_t logContentSize(uint64_t keyLen, uint64_t valueLen) {
    const uint64_t inner = stringSize(keyLen) + stringSize(valueLen);
    return inner + 1 + varintSize(inner);
}
// (round 5) A table is ~100 KB of arrays per 1000-event group; taken from and given back to the allocator by 32 runner threads at once
// they met in it, and every group wrote its spans into memory nobody had touched (the leg lost throughput from 16 to 32 threads).  A
// freed table goes to a small pool of the thread that frees it -- the serializer's thread, in the bench the runner thread itself -- and
// the next group of that thread reuses its arrays.
struct ColumnarPool {
    std::vector<std::unique_ptr<ColumnarStore>> free;
    static constexpr size_t kKeep = 4;
};
thread_local ColumnarPool tlsColumnarPool;
}  // namespace

extern "C" int lc_processor_parse_columnar(lc_processor_t* p, lc_event_group_t* g, lc_columnar_t** out) {
    if (!p || !g || !out) return LC_ERR_ARG;
    *out = nullptr;
    if (p->impl.IsWholeLineMode()) return LC_ERR_UNSUPPORTED;  // "(.*)": there is nothing to hand over but the line itself
    const auto& events = g->group.GetEvents();
    const size_t n = events.size();
    const size_t K = p->impl.mKeys.size();
    const uint32_t G = uint32_t(p->impl.MarkCount());
    std::unique_ptr<ColumnarStore> st;
    if (!tlsColumnarPool.free.empty()) {
        st = std::move(tlsColumnarPool.free.back());
        tlsColumnarPool.free.pop_back();
        st->keys.clear();
        st->keyLen.clear();
    } else {
        st = std::make_unique<ColumnarStore>();
    }
    for (const std::string& k : p->impl.mKeys) {
        st->keys.push_back(k.c_str());
        st->keyLen.push_back(uint32_t(k.size()));
    }
    st->base.assign(n, nullptr);
    st->baseLen.assign(n, 0);
    st->spans.assign(n * 2 * K, -1);
    st->state.assign(n, LC_COL_SKIPPED);
    st->contentBytes.assign(n, 0);
    // (the call's scratch lives with the runner thread: five heap blocks per group, taken and given back by 32 threads at once, met
    // in the allocator -- the leg lost throughput from 16 to 32 threads)
    struct Scratch {
        std::vector<const uint8_t*> linePtr;
        std::vector<uint32_t> lineLen, lineEvent;
        std::vector<int32_t> caps;
        std::vector<uint8_t> status;
    };
    static thread_local Scratch tScratch;
    std::vector<const uint8_t*>& linePtr = tScratch.linePtr;
    std::vector<uint32_t>&lineLen = tScratch.lineLen, &lineEvent = tScratch.lineEvent;
    linePtr.clear();
    lineLen.clear();
    lineEvent.clear();
    for (size_t i = 0; i < n; ++i) {
        if (!events[i].Is<logtail::LogEvent>()) continue;
        const logtail::LogEvent& ev = events[i].Cast<logtail::LogEvent>();
        if (!ev.HasContent(p->impl.mSourceKey)) continue;
        const logtail::StringView raw = ev.GetContent(p->impl.mSourceKey);
        st->base[i] = reinterpret_cast<const uint8_t*>(raw.data());
        st->baseLen[i] = uint32_t(raw.size());
        linePtr.push_back(st->base[i]);
        lineLen.push_back(st->baseLen[i]);
        lineEvent.push_back(uint32_t(i));
    }
    const uint32_t nLines = uint32_t(linePtr.size());
    if (nLines) {
        std::vector<int32_t>& caps = tScratch.caps;
        std::vector<uint8_t>& status = tScratch.status;
        caps.resize(size_t(nLines) * 2 * G);
        status.resize(nLines);
        const int rc = lc_regex_match_host_views(const_cast<lc_regex_t*>(p->impl.Regex()), linePtr.data(), lineLen.data(), nLines, G,
                                                 caps.data(), status.data());
        if (rc != LC_OK) return rc;
        for (uint32_t li = 0; li < nLines; ++li) {
            const size_t i = lineEvent[li];
            // RegexLogLineParser :194-244: no match, or fewer groups than keys, is a parse failure
            if (status[li] != LC_MATCH || size_t(G) + 1 <= K) {
                st->state[i] = LC_COL_FAILED;
                continue;
            }
            st->state[i] = LC_COL_PARSED;
            uint64_t bytes = 0;
            for (size_t k = 0; k < K; ++k) {
                const int32_t b = caps[size_t(li) * 2 * G + 2 * k], e = caps[size_t(li) * 2 * G + 2 * k + 1];
                st->spans[(i * K + k) * 2] = b;
                st->spans[(i * K + k) * 2 + 1] = e;
                bytes += logContentSize(st->keyLen[k], b < 0 ? 0 : uint64_t(e - b));
            }
            st->contentBytes[i] = bytes;
        }
    }
    st->pub.n_events = uint32_t(n);
    st->pub.n_keys = uint32_t(K);
    st->pub.keys = st->keys.data();
    st->pub.key_len = st->keyLen.data();
    st->pub.base = st->base.data();
    st->pub.base_len = st->baseLen.data();
    st->pub.spans = st->spans.data();
    st->pub.state = st->state.data();
    st->pub.content_bytes = st->contentBytes.data();
    *out = &st.release()->pub;  // (pub is the first member: lc_columnar_free casts back)
    return LC_OK;
}
extern "C" void lc_columnar_free(lc_columnar_t* c) {
    if (!c) return;
    std::unique_ptr<ColumnarStore> st(reinterpret_cast<ColumnarStore*>(c));
    if (tlsColumnarPool.free.size() < ColumnarPool::kKeep) tlsColumnarPool.free.push_back(std::move(st));
}

extern "C" char* lc_group_to_json(const lc_event_group_t* g) {
    if (!g) return nullptr;
    const std::string s = g->group.ToJsonString();
    char* out = static_cast<char*>(std::malloc(s.size() + 1));
    std::memcpy(out, s.c_str(), s.size() + 1);
    return out;
}

extern "C" size_t lc_group_event_count(const lc_event_group_t* g) { return g ? g->group.GetEvents().size() : 0; }
extern "C" void* lc_group_native(lc_event_group_t* g) { return g ? &g->group : nullptr; }
extern "C" void lc_group_free(lc_event_group_t* g) { delete g; }
#endif
extern "C" void lc_free(void* p) { std::free(p); }

// ---------------------------------------------------------------------------------------------- the dlsym slot
// Call protocol (core/plugin/processor/DynamicCProcessorProxy.cpp:25-40): init(ins, &config, &context) must set
// ins->plugin_state and return 0; process(plugin_state, &group) mutates the group in place; finalize(plugin_state).
static int slotInit(processor_instance_t* ins, void* config, void* context) {
    if (!ins) return -1;
    // DynamicCProcessorProxy allocates the instance without initialising it and its destructor calls finalize(plugin_state) whether or
    // not init succeeded (DynamicCProcessorProxy.cpp:21-28): a refused config must not leave a wild pointer for that call
    ins->plugin_state = nullptr;
    if (!config) return -1;
    lc_processor_t* p = nullptr;
    char err[256];
#ifdef LC_USE_REFERENCE_HEADERS
    const std::string text = static_cast<const Json::Value*>(config)->toStyledString();
    const char* configText = text.c_str();
#else
    const char* configText = static_cast<const char*>(config);
#endif
    if (lc_processor_create(configText, &p, err, sizeof err) != LC_OK) {
        std::fprintf(stderr, "[processor_parse_regex_gpu] init failed: %s\n", err);
        return -1;
    }
#ifdef LC_USE_REFERENCE_HEADERS
    // the agent hands over its CollectionPipelineContext (DynamicCProcessorProxy.cpp:30-32): the REGEX_MATCH_ALARM paths of
    // RegexLogLineParser go to its AlarmManager and logger
    p->impl.SetContext(static_cast<logtail::CollectionPipelineContext*>(context));
#else
    (void)context;  // (the stand-in build has no context type: alarms go to the sink of lc_processor_set_alarm_sink, if any)
#endif
    ins->plugin_state = p;
    return 0;
}
static void slotFinalize(void* state) { lc_processor_destroy(static_cast<lc_processor_t*>(state)); }
static void slotProcess(void* state, void* logGroup) {
    if (!state || !logGroup) return;
    processGroup(static_cast<lc_processor_t*>(state), *static_cast<PipelineEventGroup*>(logGroup));
}

extern "C" {
processor_interface_t processor_interface = {LC_PROCESSOR_INTERFACE_VERSION, "processor_parse_regex_gpu", "c++/hip",
                                             slotInit, slotFinalize, slotProcess};
}
