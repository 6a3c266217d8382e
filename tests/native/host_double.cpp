// tests/native/host_double.cpp -- TEST INFRASTRUCTURE ONLY: the host side of processor_parse_regex_gpu on a box without a GPU.
//
// The processor's host work (gather -> ONE match call -> stitch + policy + compaction, csrc/processor_parse_regex_gpu.cpp, which
// follows ProcessorParseRegexNative.cpp:108-253) only talks to the device through five C-ABI calls of include/lc_regex_gpu.h.  This
// translation unit provides test doubles of exactly those five on top of the CPU oracle (oracle/bt_regex.h), so that
// tests/test_processor_host_double.py can build  processor_parse_regex_gpu.cpp + event_model.cpp + this file  into
// tests/_build/libhost_double.so and run every reference unit-test vector and the policy matrix through the PRODUCT's host
// translation units here, where there is no device -- and time the stitch (hd_bench_stitch) without a device in the way.
//
// Second variant (-DLC_USE_REFERENCE_HEADERS -DLC_REFERENCE_MODELS_ONLY): the same product translation unit compiled against the
// REFERENCE's own event-model headers and linked with oracle/_ref/libref_models.so -- core/models/LogEvent.cpp, PipelineEventGroup.cpp,
// SourceBuffer.h ... compiled from /root/reference by oracle/ref_models/Makefile.  There the stitch is the per-key form an agent build
// takes (K x LogEvent::SetContentNoCopy + DelContent, LogEvent.cpp:83-106) and runs against the real thing; fixtures go in and out
// through tests/native/ref_group_io.cpp (the reference's FromJsonString exists only in its own unit-test builds and needs jsoncpp).
//
// It lives under tests/, is built only by the test that uses it and is never linked into loongcollector_amd/lib: the product
// library has no such path and fails loudly without the HIP runtime (tests/test_processor_host.py).
#include <sys/resource.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/lc_regex_gpu.h"
#include "../../loongcollector_amd/csrc/processor_parse_regex_gpu.hpp"
#include "../../oracle/bt_regex.h"

// ---------------------------------------------------------------------------------------------- the five doubles
struct lc_regex {
    orx_prog* prog = nullptr;
    int marks = 0;
};
static thread_local std::string tLastError;
static int gForceStatus = -1;  // >= 0: every line gets this status byte (LC_GAVE_UP / LC_OVERFLOW policy paths)
static int gForceRc = 0;       // != 0: the match call fails with this code (the "device call failed" path)
// hd_bench_stitch: the answer of the next match calls, computed beforehand (the oracle's speed must not be in the stitch figure)
struct Precomputed {
    const int32_t* caps = nullptr;
    const uint8_t* status = nullptr;
    uint32_t n = 0, ngroups = 0;
};
static Precomputed gPre;

extern "C" int lc_regex_compile(const char* pattern, size_t n, uint32_t flags, int, lc_regex_t** out, char* err, size_t errcap) {
    if (!pattern || !out) return LC_ERR_ARG;
    *out = nullptr;
    unsigned oflags = 0;
    if (flags & LC_SYNTAX_ICASE) oflags |= ORX_ICASE;
    orx_prog* p = orx_compile(pattern, n, oflags, err, errcap);
    if (!p) return LC_ERR_SYNTAX;
    auto* re = new lc_regex;
    re->prog = p;
    re->marks = orx_mark_count(p);
    *out = re;
    return LC_OK;
}
extern "C" void lc_regex_free(lc_regex_t* re) {
    if (!re) return;
    orx_free(re->prog);
    delete re;
}
extern "C" int lc_regex_mark_count(const lc_regex_t* re) { return re ? re->marks : -1; }
extern "C" const char* lc_last_error(void) { return tLastError.c_str(); }
extern "C" int lc_regex_match_host_views(lc_regex_t* re, const uint8_t* const* lines, const uint32_t* len, uint32_t n, uint32_t ngroups,
                                         int32_t* caps, uint8_t* status) {
    if (!re || (n && (!lines || !len || !status))) return LC_ERR_ARG;
    if (gForceRc) {
        tLastError = "forced failure of the test double";
        return gForceRc;
    }
    if (gPre.caps && n == gPre.n && ngroups == gPre.ngroups) {
        std::memcpy(caps, gPre.caps, size_t(n) * ngroups * 2 * sizeof(int32_t));
        std::memcpy(status, gPre.status, n);
        return LC_OK;
    }
    std::vector<int32_t> what(size_t(re->marks + 1) * 2);
    for (uint32_t i = 0; i < n; ++i) {
        const int r = orx_fullmatch(re->prog, lines[i], len[i], what.data());
        status[i] = r == 1 ? LC_MATCH : r == 0 ? LC_NOMATCH : LC_GAVE_UP;
        if (gForceStatus >= 0) status[i] = uint8_t(gForceStatus);
        for (uint32_t g = 0; g < ngroups; ++g) {
            const bool have = r == 1 && int(g) < re->marks;
            caps[(size_t(i) * ngroups + g) * 2] = have ? what[(g + 1) * 2] : -1;
            caps[(size_t(i) * ngroups + g) * 2 + 1] = have ? what[(g + 1) * 2 + 1] : -1;
        }
    }
    return LC_OK;
}

// -DHD_DOUBLES_ONLY: nothing but the five doubles above -- tests/test_plugin_slot_reference.py links them under the product's dlsym slot
// (csrc/c_processor_slot.cpp in the agent's form) to get a plugin the REFERENCE's own loader and proxy can drive on a box without a GPU
#ifndef HD_DOUBLES_ONLY
#ifdef LC_USE_REFERENCE_HEADERS
bool hdGroupFromJson(logtail::PipelineEventGroup& group, const std::string& json, std::string* error);  // ref_group_io.cpp
std::string hdGroupToJson(const logtail::PipelineEventGroup& group);
#endif
// PipelineEventGroup::DataSize() minus its events' own sizes (tags + container), with either event model
static size_t groupBytesWithoutEvents(const logtail::PipelineEventGroup& group) {
    size_t n = group.DataSize();
    for (const auto& e : group.GetEvents()) n -= e->DataSize();
    return n;
}

// ---------------------------------------------------------------------------------------------- what the test drives
struct hd_processor {
    logtail::ProcessorParseRegexGpu impl;
    std::string alarms;  // kind \t message \n ...
    uint64_t sizes[4] = {0, 0, 0, 0};
};

extern "C" {
void hd_force(int status, int rc) {
    gForceStatus = status;
    gForceRc = rc;
}

hd_processor* hd_create(const char* configJson, char* err, size_t errcap) {
    auto p = std::make_unique<hd_processor>();
    std::string error;
    try {
        const lcjson::Value cfg = lcjson::parse(configJson);
        if (!p->impl.Init(cfg, error)) {
            std::snprintf(err, errcap, "%s", error.c_str());
            return nullptr;
        }
    } catch (const std::exception& e) {
        std::snprintf(err, errcap, "%s", e.what());
        return nullptr;
    }
    return p.release();
}
void hd_destroy(hd_processor* p) { delete p; }

void hd_want_alarms(hd_processor* p) {
    p->impl.SetAlarmSink(
        [](void* user, int kind, const char* m, size_t n) {
            auto* s = static_cast<std::string*>(user);
            *s += std::to_string(kind) + "\t" + std::string(m, n) + "\n";
        },
        &p->alarms);
}
char* hd_take_alarms(hd_processor* p) {
    char* out = strdup(p->alarms.c_str());
    p->alarms.clear();
    return out;
}

// fixture JSON in -> Process -> fixture JSON out (malloc'ed; hd_free).  NULL + err on a bad fixture.
char* hd_process_json(hd_processor* p, const char* groupJson, char* err, size_t errcap) {
    auto sb = std::make_shared<logtail::SourceBuffer>();
    logtail::PipelineEventGroup group(sb);
    std::string error;
#ifdef LC_USE_REFERENCE_HEADERS
    if (!hdGroupFromJson(group, groupJson, &error)) {
#else
    if (!group.FromJsonString(groupJson, &error)) {
#endif
        std::snprintf(err, errcap, "%s", error.c_str());
        return nullptr;
    }
    // (the sizes lc_processor_process reports as in/out_size_bytes come from the processor's own sums: checked against the walks)
    logtail::ProcessorParseRegexGpu::EventBytes bytes;
    p->sizes[0] = group.DataSize();
    p->impl.Process(group, &bytes);
    p->sizes[1] = group.DataSize();
    p->sizes[2] = groupBytesWithoutEvents(group) + bytes.in;
    p->sizes[3] = groupBytesWithoutEvents(group) + bytes.out;
#ifdef LC_USE_REFERENCE_HEADERS
    return strdup(hdGroupToJson(group).c_str());
#else
    return strdup(group.ToJsonString().c_str());
#endif
}
void hd_last_sizes(const hd_processor* p, uint64_t out[4]) {  // DataSize() before, after; the processor's sums for the same two
    for (int i = 0; i < 4; ++i) out[i] = p->sizes[i];
}
void hd_free(void* p) { std::free(p); }

void hd_counters(const hd_processor* p, uint64_t out[7]) {
    out[0] = p->impl.mDiscardedEventsTotal;
    out[1] = p->impl.mOutFailedEventsTotal;
    out[2] = p->impl.mOutKeyNotFoundEventsTotal;
    out[3] = p->impl.mOutSuccessfulEventsTotal;
    out[4] = p->impl.mComplexityExceededEventsTotal;
    out[5] = p->impl.mUndecidedEventsTotal;
    out[6] = p->impl.mDeviceFailedEventsTotal;
}

// The host share of one in-agent group, timed without a device: `groups` groups of `n` events whose `key` content is lines[i]
// (one copy in the group's SourceBuffer, as the file reader leaves them), Process()ed with the match call answered from a capture
// table computed ONCE beforehand (so the oracle's speed is not in the figure).  Returns microseconds per group: gather + stitch +
// policy + compaction + the in/out size sums -- everything lc_processor_process does except the device trip.  *dataSizeUs: one
// PipelineEventGroup::DataSize() walk over a processed group (what the size sums cost as passes of their own, twice per group).
}  // extern "C"

// The arena chunk pool of the event model (csrc/event_model.hpp ArenaChunkPool): full-size chunks of a dead SourceBuffer are handed
// to the next one, smaller chunks and big blocks are not, the pool is bounded.  Returns 0, or the number of the check that failed.
#ifdef LC_USE_REFERENCE_HEADERS
extern "C" int hd_arena_pool_check(void) { return -1; }          // (the stand-in's pool and container: not in this variant)
extern "C" int hd_contents_container_check(void) { return -1; }
extern "C" int hd_event_model_is_reference(void) { return 1; }
#else
#ifdef LC_REFERENCE_SHAPED_EVENT_MODEL
extern "C" int hd_event_model_is_reference(void) { return 2; }  // the stand-in built in the reference's shape
extern "C" int hd_arena_pool_check(void) { return -1; }
extern "C" int hd_contents_container_check(void) { return -1; }
#else
extern "C" int hd_event_model_is_reference(void) { return 0; }
extern "C" int hd_arena_pool_check(void) {
    using logtail::ArenaChunkPool;
    using logtail::SourceBuffer;
    ArenaChunkPool& pool = ArenaChunkPool::instance();
    while (char* p = pool.take()) delete[] p;  // start from an empty pool
    std::vector<char*> seen;
    {
        SourceBuffer sb;
        for (int i = 0; i < 3000; ++i) seen.push_back(sb.CopyString("0123456789012345678901234567890123456789", 40).data);  // 144 KB in small pieces
        sb.AllocateStringBuffer(512 * 1024);  // a big block of its own
        if (pool.pooled() != 0) return 1;
    }
    // 4 K + 8 K + 16 K + 32 K + 64 K chunks hold the first 124 KB; the rest went into ONE full-size chunk: that one is pooled
    if (pool.pooled() != 1) return 2;
    {
        SourceBuffer sb;
        char* last = nullptr;
        for (int i = 0; i < 3000; ++i) last = sb.CopyString("x", 1).data;  // 8 bytes each: 24 KB, the doubling chunks only
        (void)last;
        if (pool.pooled() != 1) return 3;
        for (int i = 0; i < 3000; ++i) last = sb.CopyString("0123456789012345678901234567890123456789", 40).data;
        if (pool.pooled() != 0) return 4;  // ... the full-size chunk came from the pool
        const char* inFirst = seen.back();  // the first buffer's last string lay in its full-size chunk
        if (!(last > inFirst - 128 * 1024 && last < inFirst + 128 * 1024)) return 5;  // same 128 KiB of memory
    }
    if (pool.pooled() != 1) return 6;
    {
        std::vector<std::unique_ptr<SourceBuffer>> many;
        for (size_t i = 0; i < ArenaChunkPool::kMaxPooled + 8; ++i) {
            many.push_back(std::make_unique<SourceBuffer>());
            for (int k = 0; k < 3000; ++k) many.back()->CopyString("0123456789012345678901234567890123456789", 40);
        }
    }
    if (pool.pooled() != ArenaChunkPool::kMaxPooled) return 7;  // bounded: the surplus went back to the allocator
    while (char* p = pool.take()) delete[] p;
    return 0;
}

// ArenaVector (the events' contents array): growth keeps the entries and their order, in the arena and -- for an event made outside
// a group -- on the heap; unzeroed appends are constructed by the caller.  Returns 0, or the number of the check that failed.
extern "C" int hd_contents_container_check(void) {
    using namespace logtail;
    static const char* kKeys[] = {"k0", "k1", "k2", "k3", "k4", "k5", "k6", "k7", "k8", "k9"};
    for (int where = 0; where < 2; ++where) {
        auto sb = std::make_shared<SourceBuffer>();
        PipelineEventGroup group(sb);
        std::unique_ptr<LogEvent> heapEvent = where ? std::make_unique<LogEvent>(nullptr) : nullptr;
        LogEvent* ev = where ? heapEvent.get() : group.AddLogEvent();
        std::vector<std::string> values;
        for (int i = 0; i < 100; ++i) values.push_back("value-" + std::to_string(i));
        for (int i = 0; i < 100; ++i) ev->SetContentNoCopy(StringView(kKeys[i % 10]), StringView(values[size_t(i)]));  // overwrites from i = 10 on
        if (ev->Size() != 10) return 10 * where + 1;
        for (int k = 0; k < 10; ++k)
            if (ev->GetContent(kKeys[k]) != StringView(values[size_t(90 + k)])) return 10 * where + 2;
        ev->DelContent("k3");
        const std::string raw = "0123456789abcdefghij";
        const int32_t caps[6] = {0, 4, -1, -1, 10, 20};
        const StringView fresh[3] = {StringView("f0"), StringView("f1"), StringView("f2")};
        const StringView drop("k5");
        ev->AppendCapturesNoCopy(fresh, 3, StringView(raw), caps, &drop);
        if (ev->Size() != 11 || ev->HasContent("k3") || ev->HasContent("k5")) return 10 * where + 3;
        if (ev->GetContent("f0") != StringView("0123") || !ev->HasContent("f1") || !ev->GetContent("f1").empty() ||
            ev->GetContent("f1").data() != raw.data() + raw.size() || ev->GetContent("f2") != StringView("abcdefghij"))
            return 10 * where + 4;
        std::string order;
        size_t bytes = 0;
        for (auto it = ev->begin(); it != ev->end(); ++it) {
            order += std::string(it->first) + ",";
            bytes += it->first.size() + it->second.size();
        }
        if (order != "k0,k1,k2,k4,k6,k7,k8,k9,f0,f1,f2,") return 10 * where + 5;
        if (ev->DataSize() != sizeof(time_t) + sizeof(std::optional<uint32_t>) + sizeof(std::vector<int>) + bytes) return 10 * where + 6;
    }
    return 0;
}

#endif  // LC_REFERENCE_SHAPED_EVENT_MODEL
#endif  // LC_USE_REFERENCE_HEADERS

static double gLastMinorFaultsPerGroup = 0;
extern "C" double hd_last_minor_faults_per_group(void) { return gLastMinorFaultsPerGroup; }  // of the last repeat

extern "C" double hd_bench_stitch(hd_processor* p, const uint8_t* data, const uint32_t* off, const uint32_t* len, uint32_t n,
                                  uint32_t groups, const char* key, uint32_t repeats, double* buildUs, double* dataSizeUs) {
    using clk = std::chrono::steady_clock;
    const uint32_t G = uint32_t(p->impl.MarkCount());
    // the capture table of the n lines, once
    std::vector<const uint8_t*> ptr(n);
    for (uint32_t i = 0; i < n; ++i) ptr[i] = data + off[i];
    std::vector<int32_t> caps(size_t(n) * G * 2);
    std::vector<uint8_t> status(n);
    lc_regex_match_host_views(const_cast<lc_regex_t*>(p->impl.Regex()), ptr.data(), len, n, G, caps.data(), status.data());
    double best = 1e30, bestBuild = 1e30, bestSize = 1e30;
    const size_t keyLen = std::strlen(key);
    for (uint32_t r = 0; r < repeats; ++r) {
        std::vector<std::unique_ptr<logtail::PipelineEventGroup>> gs;
        std::vector<std::shared_ptr<logtail::SourceBuffer>> sbs;
        const auto b0 = clk::now();
        for (uint32_t g = 0; g < groups; ++g) {
            auto sb = std::make_shared<logtail::SourceBuffer>();
            auto grp = std::make_unique<logtail::PipelineEventGroup>(sb);
            size_t total = 0;
            for (uint32_t i = 0; i < n; ++i) total += size_t(len[i]) + 1;
            logtail::StringBuffer buf = sb->AllocateStringBuffer(total);
            const logtail::StringBuffer kb = sb->CopyString(key, keyLen);
            size_t at = 0;
            for (uint32_t i = 0; i < n; ++i) {
                std::memcpy(buf.data + at, data + off[i], len[i]);
                buf.data[at + len[i]] = '\n';
                grp->AddLogEvent()->SetContentNoCopy(logtail::StringView(kb.data, kb.size), logtail::StringView(buf.data + at, len[i]));
                at += size_t(len[i]) + 1;
            }
            sbs.push_back(sb);
            gs.push_back(std::move(grp));
        }
        const auto b1 = clk::now();
        gPre.caps = caps.data();
        gPre.status = status.data();
        gPre.n = n;
        gPre.ngroups = G;
        rusage ru0, ru1;
        getrusage(RUSAGE_SELF, &ru0);
        const auto t0 = clk::now();
        size_t sizeSink = 0;
        for (auto& g : gs) {  // what lc_processor_process does around the device trip (csrc/c_processor_slot.cpp processGroup)
            logtail::ProcessorParseRegexGpu::EventBytes bytes;
            p->impl.Process(*g, &bytes);
            sizeSink += bytes.in + bytes.out;
        }
        const auto t1 = clk::now();
        if (sizeSink == 1) std::fprintf(stderr, " ");
        getrusage(RUSAGE_SELF, &ru1);
        gLastMinorFaultsPerGroup = double(ru1.ru_minflt - ru0.ru_minflt) / groups;
        gPre = Precomputed{};
        size_t sink = 0;
        const auto s0 = clk::now();
        for (auto& g : gs) sink += g->DataSize();
        const auto s1 = clk::now();
        if (sink == 1) std::fprintf(stderr, " ");
        best = std::min(best, std::chrono::duration<double, std::micro>(t1 - t0).count() / groups);
        bestBuild = std::min(bestBuild, std::chrono::duration<double, std::micro>(b1 - b0).count() / groups);
        bestSize = std::min(bestSize, std::chrono::duration<double, std::micro>(s1 - s0).count() / groups);
    }
    if (buildUs) *buildUs = bestBuild;
    if (dataSizeUs) *dataSizeUs = bestSize;
    return best;
}
#endif  // HD_DOUBLES_ONLY
