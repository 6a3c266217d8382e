// processor_filter_gpu.cpp -- see processor_filter_gpu.hpp.
#include "processor_filter_gpu.hpp"

#include <algorithm>
#include <cstring>
#include <memory>

#include "../../include/lc_processor.h"
#include "../../include/lc_regex_gpu.h"
#include "regex_handle.hpp"
#include <hip/hip_runtime_api.h>

#include "trip_buffers.hpp"

void lcFilterThreadRelease();  // (runtime_internal.hpp; that header is for the device translation units)

namespace {
// per runner thread: a stream, pinned staging, device buffers; grow-only
struct FilterThread {
    hipStream_t stream = nullptr;
    int device = -1;
    TripBuf hIn, hStatus;  // pinned: staging (values, offsets, lengths); status bytes + the trip's completion word
    TripBuf dIn, dStatus;  // device (dStatus: the kernels' capture scratch; dIn: only with LC_FILTER_COPY_TRIP)
    uint32_t tripSeq = 0;
    FilterThread() { hIn.pinned = hStatus.pinned = true; }
    ~FilterThread();
};
thread_local FilterThread tlsFilter;
constexpr size_t kFlagBytes = 64;  // hStatus: [completion word, one cache line][status bytes]
}  // namespace
void lcRegisterExitHook();
bool lcRuntimeUsable();  // gpu_runtime.hip
int lcHostEntryDevice(int* dev);  // gpu_runtime.hip: the calling thread's device binding
int lcQueueTripSignal(uint32_t* hFlag, uint32_t seq, hipStream_t stream);
int lcAwaitTripSignal(const uint32_t* hFlag, uint32_t seq, hipStream_t stream);
void lcSetJobTableInPlace(bool on);
void lcFilterThreadRelease() {
    FilterThread& T = tlsFilter;
    if (T.stream) {
        (void)hipStreamSynchronize(T.stream);
        (void)hipStreamDestroy(T.stream);
        T.stream = nullptr;
    }
    for (TripBuf* b : {&T.hIn, &T.hStatus, &T.dIn, &T.dStatus}) b->release();
    T.device = -1;
}
namespace {
FilterThread::~FilterThread() {
    if (lcRuntimeUsable() && (stream || hIn.p || dIn.p)) lcFilterThreadRelease();
}
}  // namespace

namespace logtail {

const std::string ProcessorFilterGpu::sName = "processor_filter_regex_gpu";

namespace {
std::string lower(std::string s) {
    for (auto& c : s) c = char(std::tolower(static_cast<unsigned char>(c)));
    return s;
}
std::string asString(const lcjson::Value* v) { return (v && v->isString()) ? v->str : std::string(); }  // Json::asString
}  // namespace

ProcessorFilterGpu::~ProcessorFilterGpu() {
    for (auto& l : mLeaves) lc_regex_free(l.reg);
}

int ProcessorFilterGpu::addLeaf(const std::string& key, const std::string& exp, std::string& error) {
    lc_regex_t* re = nullptr;
    char err[256];
    // IsRegexValid + boost::regex(exp) (:94-104, :133-145); a valid Perl regex no device engine can run also fails Init
    if (lc_regex_compile(exp.data(), exp.size(), 0, LC_ENGINE_AUTO, &re, err, sizeof err) != LC_OK) {
        error = std::string("regex `") + exp + "` : " + err;
        return -1;
    }
    mLeaves.push_back({key, re});
    return int(mLeaves.size()) - 1;
}

// ParseExpressionFromJSON :381-430
int ProcessorFilterGpu::parseExpression(const lcjson::Value& value, std::string& error) {
    if (!value.isObject()) return -1;
    const lcjson::Value* op = value.find("operator");
    const lcjson::Value* operands = value.find("operands");
    if (op && op->isString() && operands && operands->isArray()) {
        const std::string o = lower(op->str);
        if (o != "not" && o != "and" && o != "or") return -1;  // GetOperatorType :432-444
        if (o == "not" && operands->arr.size() == 1) {
            const int child = parseExpression(operands->arr[0], error);
            if (child < 0) return -1;
            mNodes.push_back({NOT, child, -1, -1});
            return int(mNodes.size()) - 1;
        }
        if ((o == "and" || o == "or") && operands->arr.size() == 2) {
            const int l = parseExpression(operands->arr[0], error);
            const int r = parseExpression(operands->arr[1], error);
            if (l < 0 || r < 0) return -1;
            mNodes.push_back({o == "and" ? AND : OR, l, r, -1});
            return int(mNodes.size()) - 1;
        }
        return -1;
    }
    const lcjson::Value *key = value.find("key"), *exp = value.find("exp"), *type = value.find("type");
    if ((key && key->isString() && exp && exp->isString()) || !(type && type->isString())) {
        if (lower(asString(type)) != "regex") return -1;  // GetNodeFuncType :446-454
        const int leaf = addLeaf(asString(key), asString(exp), error);
        if (leaf < 0) return -1;
        mNodes.push_back({LEAF, -1, -1, leaf});
        return int(mNodes.size()) - 1;
    }
    return -1;
}

bool ProcessorFilterGpu::Init(const lcjson::Value& config, std::string& error) {
    if (!config.isObject()) {
        error = "config is not an object";
        return false;
    }
    // for backward compatibility, ConditionExp prioritizes over FilterKey and FilterRegex (:33-61)
    if (const lcjson::Value* ce = config.find("ConditionExp")) {
        if (!ce->isObject()) {
            error = "object param ConditionExp is not of type object";
            return false;
        }
        std::string why;
        mRoot = parseExpression(*ce, why);
        if (mRoot < 0) {
            error = "object param ConditionExp is not valid" + (why.empty() ? std::string() : " (" + why + ")");
            return false;
        }
        mFilterMode = Mode::EXPRESSION_MODE;
    }
    auto stringList = [&](const char* name, std::vector<std::string>& out) {
        const lcjson::Value* v = config.find(name);
        if (!v) return true;
        if (!v->isArray()) {
            error = std::string("list param ") + name + " is not of type list";
            return false;
        }
        for (const auto& e : v->arr) {
            if (!e.isString()) {
                error = std::string("list param ") + name + " is not of type string list";
                return false;
            }
            out.push_back(e.str);
        }
        return true;
    };
    if (mFilterMode == Mode::BYPASS_MODE) {  // FilterKey + FilterRegex (:63-107)
        std::vector<std::string> keys, regs;
        if (!stringList("FilterKey", keys) || !stringList("FilterRegex", regs)) return false;
        if (keys.size() != regs.size()) {
            error = "param FilterKey and FilterRegex does not have the same size";
            return false;
        }
        for (size_t i = 0; i < keys.size(); ++i) {
            std::string why;
            const int leaf = addLeaf(keys[i], regs[i], why);
            if (leaf < 0) {
                error = "value in list param FilterRegex is not a valid regex: " + why;
                return false;
            }
            mRuleLeaves.push_back(leaf);
        }
        if (!keys.empty()) mFilterMode = Mode::RULE_MODE;
    }
    if (mFilterMode == Mode::BYPASS_MODE) {  // Include, deprecated (:109-143)
        if (const lcjson::Value* inc = config.find("Include")) {
            if (!inc->isObject()) {
                error = "map param Include is not of type map";
                return false;
            }
            for (const auto& kv : inc->obj) {
                if (!kv.second.isString()) {
                    error = "map param Include is not of type map<string, string>";
                    return false;
                }
                std::string why;
                const int leaf = addLeaf(kv.first, kv.second.str, why);
                if (leaf < 0) {
                    error = "value in map param Include is not a valid regex: " + why;
                    return false;
                }
                mRuleLeaves.push_back(leaf);
            }
            if (!inc->obj.empty()) mFilterMode = Mode::RULE_MODE;
        }
    }
    if (const lcjson::Value* d = config.find("DiscardingNonUTF8"))  // :145-155 (wrong type: warning, default kept)
        if (d->isBool()) mDiscardingNonUTF8 = d->b;
    return true;
}

bool ProcessorFilterGpu::eval(int node, const std::vector<std::vector<uint8_t>>& leafResult, size_t event) const {
    const Node& n = mNodes[size_t(node)];
    switch (n.op) {
        case LEAF: return leafResult[size_t(n.leaf)][event] != 0;          // RegexFilterValueNode::Match :456-478
        case NOT: return !eval(n.left, leafResult, event);                 // UnaryFilterOperatorNode::Match :480-485
        case AND: return eval(n.left, leafResult, event) && eval(n.right, leafResult, event);  // :432-442
        case OR: return eval(n.left, leafResult, event) || eval(n.right, leafResult, event);
    }
    return false;
}

// ProcessorFilterNative::noneUtf8 :297-379, restated: the same per-sequence checks in the same order
bool ProcessorFilterGpu::NoneUtf8(char* s, size_t n, bool modify) {
    auto cont = [&](size_t i) { return (static_cast<unsigned char>(s[i]) & 0xC0) == 0x80; };
    size_t i = 0;
    while (i < n) {
        const unsigned char c = static_cast<unsigned char>(s[i]);
        size_t need = 0;
        bool bad = false;
        if ((c & 0x80) == 0x00) {
            need = 1;
        } else if ((c & 0xE0) == 0xC0) {
            need = 2;
            if (i + 1 >= n || !cont(i + 1)) bad = true;
            else {
                const uint16_t u = uint16_t(((c & 0x1F) << 6) | (static_cast<unsigned char>(s[i + 1]) & 0x3F));
                bad = !(u >= 0x80 && u <= 0x7FF);
            }
        } else if ((c & 0xF0) == 0xE0) {
            need = 3;
            if (i + 2 >= n || !cont(i + 1) || !cont(i + 2)) bad = true;
            else {
                const uint16_t u = uint16_t(((c & 0x0F) << 12) | ((static_cast<unsigned char>(s[i + 1]) & 0x3F) << 6) |
                                            (static_cast<unsigned char>(s[i + 2]) & 0x3F));
                bad = !(u >= 0x800);
            }
        } else if ((c & 0xF8) == 0xF0) {
            need = 4;
            if (i + 3 >= n || !cont(i + 1) || !cont(i + 2) || !cont(i + 3)) bad = true;
            else {
                const uint32_t u = (uint32_t(c & 0x07) << 18) | (uint32_t(static_cast<unsigned char>(s[i + 1]) & 0x3F) << 12) |
                                   (uint32_t(static_cast<unsigned char>(s[i + 2]) & 0x3F) << 6) |
                                   uint32_t(static_cast<unsigned char>(s[i + 3]) & 0x3F);
                bad = !(u >= 0x10000 && u <= 0x10FFFF);
            }
        } else {
            bad = true;
        }
        if (bad) {  // FILL_BLUNK_AND_CONTINUE_IF_TRUE: only the offending byte is blanked, the scan resumes right after it
            if (!modify) return true;
            s[i] = ' ';
            ++i;
            continue;
        }
        i += need;
    }
    return false;
}

// DiscardingNonUTF8 :192-213
void ProcessorFilterGpu::sanitize(LogEvent& e) const {
    struct KV {
        StringView key, value;
    };
    std::vector<KV> live;
    for (auto it = e.begin(); it != e.end(); ++it) live.push_back({it->first, it->second});
    std::vector<KV> renamed;
    for (auto& kv : live) {
        StringView value = kv.value;
        if (NoneUtf8(const_cast<char*>(value.data()), value.size(), false)) {
            StringBuffer b = e.GetSourceBuffer()->CopyString(value.data(), value.size());
            NoneUtf8(b.data, b.size, true);
            value = StringView(b.data, b.size);
            e.SetContentNoCopy(kv.key, value);
        }
        if (NoneUtf8(const_cast<char*>(kv.key.data()), kv.key.size(), false)) {
            StringBuffer b = e.GetSourceBuffer()->CopyString(kv.key.data(), kv.key.size());
            NoneUtf8(b.data, b.size, true);
            renamed.push_back({StringView(b.data, b.size), value});
            e.DelContent(kv.key);
        }
    }
    for (auto& kv : renamed) e.SetContentNoCopy(kv.key, kv.value);
}

bool ProcessorFilterGpu::Process(PipelineEventGroup& logGroup, std::string& error) {
    if (logGroup.GetEvents().empty()) return true;
    EventsContainer& events = logGroup.MutableEvents();
    const size_t n = events.size();
    mInEventsTotal += n;

    // ONE device trip for all regex leaves: the values of every leaf's key (absent key: the leaf is false, :260-264 / :457-461)
    // are gathered into one pinned block and go up once; every leaf is a job of ONE lc_regex_match_device_multi call (leaves on
    // the tagged-DFA engine share a single launch, each workgroup staging its own leaf's tables); the status bytes of all values
    // come back with one copy, one synchronisation per group.  (Round 2: a gather, an upload, a launch and a download per leaf.)
    std::vector<std::vector<uint8_t>> leafResult(mLeaves.size());
    uint64_t gaveUp = 0;
    struct Val {
        const uint8_t* p;
        uint32_t len, owner;
    };
    std::vector<std::vector<Val>> vals(mLeaves.size());
    uint64_t totalBytes = 0;
    size_t totalVals = 0;
    for (size_t l = 0; l < mLeaves.size(); ++l) {
        leafResult[l].assign(n, 0);
        const StringView key(mLeaves[l].key);
        for (size_t i = 0; i < n; ++i) {
            if (!events[i].Is<LogEvent>()) continue;
            const LogEvent& e = events[i].Cast<LogEvent>();
            if (!e.HasContent(key)) continue;
            const StringView v = e.GetContent(key);
            vals[l].push_back({reinterpret_cast<const uint8_t*>(v.data()), uint32_t(v.size()), uint32_t(i)});
            totalBytes += v.size();
        }
        totalVals += vals[l].size();
    }
    auto fail = [&](const std::string& why) {
        error = why;
        mInEventsTotal -= n;
        return false;
    };
    if (totalVals) {
        if (lc_device_count() <= 0) return fail("no HIP device: the filter has no CPU path");
        if (totalBytes >= 0xFFFFFFF0ull) return fail("filter: more than 4 GiB of values in one group");
        FilterThread& T = tlsFilter;
        int dev = 0;
#define FILTER_TRY(expr)                                                                  \
    do {                                                                                  \
        const hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
        if (lcHostEntryDevice(&dev) != LC_OK) return fail(std::string("filter: ") + lc_last_error());  // the thread's binding
        if (T.stream && T.device != dev) lcFilterThreadRelease();  // (another device: old stream and buffers go, see PipeThread)
        if (!T.stream) {
        lcRegisterExitHook();
            FILTER_TRY(hipStreamCreateWithFlags(&T.stream, hipStreamNonBlocking));
            T.device = dev;
        }
        // staging: [value bytes, back to back][off: one word per value][len: one word per value]
        const size_t dataBytes = (size_t(totalBytes) + 31) & ~size_t(15);
        const size_t upBytes = dataBytes + totalVals * 8 + 16;
        FILTER_TRY(T.hIn.ensure(upBytes));
        if (getenv("LC_FILTER_COPY_TRIP")) FILTER_TRY(T.dIn.ensure(upBytes));
        FILTER_TRY(T.dStatus.ensure(totalVals + 64));
        FILTER_TRY(T.hStatus.ensure(kFlagBytes + totalVals + 192));
        uint8_t* h = static_cast<uint8_t*>(T.hIn.p);
        uint32_t* hOff = reinterpret_cast<uint32_t*>(h + dataBytes);
        uint32_t* hLen = hOff + totalVals;
        uint32_t at = 0;
        size_t k = 0;
        for (size_t l = 0; l < mLeaves.size(); ++l)
            for (const Val& v : vals[l]) {
                hOff[k] = at;
                hLen[k] = v.len;
                if (v.len) std::memcpy(h + at, v.p, v.len);
                at += v.len;
                ++k;
            }
        std::memset(h + at, 0, dataBytes - at);
        // Round 5: a ZERO-COPY trip, as the parse processor's (gpu_runtime.hip runHostPipeline): the kernels read the values and their
        // tables where the host wrote them, in pinned memory, and write the status bytes into pinned memory; a one-lane kernel behind
        // them stores the trip's number into a pinned word the thread spins on.  No copy command at all -- the three small copies of a
        // group (values up, job table up, status down) all went through the device's SDMA queue, where the groups of every runner
        // thread met: 11.3 GB/s with 16 threads, 8.4 with 32.  LC_FILTER_COPY_TRIP=1 keeps the copies (A/B measurements).
        static const bool copyTrip = getenv("LC_FILTER_COPY_TRIP") != nullptr;
        const uint8_t* dData = h;
        // The completion word has a home of its own, the FIRST 64 bytes of the block, and the status bytes lie behind it: through
        // round 5 it sat behind the status bytes, i.e. it moved with the group's size, and a shorter group found an earlier group's
        // status bytes (0..3 each: 0x00000100 = trip 256, 0x00000101 = 257 ...) where it expected its own trip number -- the wait
        // could return before the kernels had finished.  It is also cleared before every trip: a block that `ensure` has just
        // grown is fresh pinned memory with whatever it held.
        uint8_t* const hStatusBytes = static_cast<uint8_t*>(T.hStatus.p) + kFlagBytes;
        uint8_t* dStatus = hStatusBytes;
        volatile uint32_t* hFlag = reinterpret_cast<volatile uint32_t*>(T.hStatus.p);
        *hFlag = 0;
        if (copyTrip) {
            if (lc_upload_pinned(T.hIn.p, T.dIn.p, upBytes, T.stream) != LC_OK) return fail(lc_last_error());
            dData = static_cast<const uint8_t*>(T.dIn.p);
            dStatus = static_cast<uint8_t*>(T.dStatus.p);
        }
        const uint32_t* dOff = reinterpret_cast<const uint32_t*>(dData + dataBytes);
        const uint32_t* dLen = dOff + totalVals;
        int32_t* dCapsDummy = reinterpret_cast<int32_t*>(static_cast<uint8_t*>(T.dStatus.p) + ((totalVals + 15) & ~size_t(15)));  // (no group is asked for)
        std::vector<lc_match_job> jobs;
        size_t base = 0;
        for (size_t l = 0; l < mLeaves.size(); ++l) {
            if (!vals[l].empty())
                jobs.push_back({mLeaves[l].reg, dData, dOff + base, dLen + base, 0u, uint32_t(vals[l].size()), 0u, dCapsDummy, dStatus + base});
            base += vals[l].size();
        }
        lcSetJobTableInPlace(!copyTrip);
        const int rcMulti = lc_regex_match_device_multi(jobs.data(), uint32_t(jobs.size()), T.stream);
        lcSetJobTableInPlace(false);
        if (rcMulti != LC_OK) {
            (void)hipStreamSynchronize(T.stream);
            return fail(lc_last_error());
        }
        if (copyTrip) {
            FILTER_TRY(hipMemcpyAsync(hStatusBytes, dStatus, totalVals, hipMemcpyDeviceToHost, T.stream));
            FILTER_TRY(hipStreamSynchronize(T.stream));
        } else {
            const uint32_t seq = ++T.tripSeq ? T.tripSeq : ++T.tripSeq;  // (never 0: the word starts as 0)
            uint32_t* const flagWord = const_cast<uint32_t*>(hFlag);
            if (lcQueueTripSignal(flagWord, seq, T.stream) != LC_OK || lcAwaitTripSignal(flagWord, seq, T.stream) != LC_OK) {
                (void)hipStreamSynchronize(T.stream);  // (nothing queued here may still write the status block when the next trip reuses it)
                return fail(lc_last_error());
            }
        }
#undef FILTER_TRY
        const uint8_t* status = hStatusBytes;
        k = 0;
        for (size_t l = 0; l < mLeaves.size(); ++l)
            for (const Val& v : vals[l]) {
                // "not decided" (decide pass switched off) is neither true nor false: a NOT node would turn a guess into a keep
                if (status[k] == LC_OVERFLOW) return fail("device left a value undecided (LC_NFA_NO_DECIDE is set): group untouched");
                leafResult[l][v.owner] = status[k] == LC_MATCH;
                gaveUp += status[k] == LC_GAVE_UP;  // regex_match in Filter fail (:266-281): false, and counted
                ++k;
            }
    }

    if (gaveUp) {
        mComplexityExceededTotal += gaveUp;
        lcNoteGaveUp(gaveUp);
    }
    size_t wIdx = 0;
    for (size_t rIdx = 0; rIdx < n; ++rIdx) {  // Process :159-176 / ProcessEvent :178-216
        bool res = true;
        if (events[rIdx].Is<LogEvent>()) {
            LogEvent& e = events[rIdx].Cast<LogEvent>();
            if (mFilterMode == Mode::EXPRESSION_MODE) {
                res = !e.Empty() && eval(mRoot, leafResult, rIdx);              // FilterExpressionRoot :222-238
            } else if (mFilterMode == Mode::RULE_MODE) {
                res = !e.Empty();                                                 // FilterFilterRule :240-256
                for (size_t k = 0; k < mRuleLeaves.size() && res; ++k) res = leafResult[size_t(mRuleLeaves[k])][rIdx] != 0;
            }
            if (res && mDiscardingNonUTF8) sanitize(e);
        }
        if (res) {
            if (wIdx != rIdx) events[wIdx] = std::move(events[rIdx]);
            ++wIdx;
        }
    }
    events.resize(wIdx);
    mOutEventsTotal += wIdx;
    return true;
}

}  // namespace logtail

// ------------------------------------------------------------------------------------------------ C ABI
struct lc_filter {
    logtail::ProcessorFilterGpu impl;
};

extern "C" int lc_filter_create(const char* config_json, lc_filter_t** out, char* err, size_t errcap) {
    if (!config_json || !out) return LC_ERR_ARG;
    *out = nullptr;
    auto set = [&](const std::string& m) {
        if (err && errcap) std::snprintf(err, errcap, "%s", m.c_str());
    };
    lcjson::Value cfg;
    try {
        cfg = lcjson::parse(config_json);
    } catch (const std::exception& e) {
        set(e.what());
        return LC_ERR_ARG;
    }
    auto f = std::make_unique<lc_filter>();
    std::string error;
    if (!f->impl.Init(cfg, error)) {
        set(error);
        return LC_ERR_SYNTAX;
    }
    set("");
    *out = f.release();
    return LC_OK;
}
extern "C" void lc_filter_destroy(lc_filter_t* f) { delete f; }
extern "C" int lc_filter_mode(const lc_filter_t* f) { return f ? int(f->impl.mFilterMode) : -1; }
extern "C" int lc_filter_process(lc_filter_t* f, void* native_group) {
    if (!f || !native_group) return LC_ERR_ARG;
    std::string error;
    if (!f->impl.Process(*static_cast<logtail::PipelineEventGroup*>(native_group), error))
        return lc_device_count() <= 0 ? LC_ERR_NO_DEVICE : LC_ERR_HIP;
    return LC_OK;
}
extern "C" int lc_filter_none_utf8(char* buf, size_t n, int modify) {
    return (buf || n == 0) ? int(logtail::ProcessorFilterGpu::NoneUtf8(buf, n, modify != 0)) : 0;
}
extern "C" void lc_filter_counters(const lc_filter_t* f, uint64_t out[2]) {
    if (!f || !out) return;
    out[0] = f->impl.mInEventsTotal.load();
    out[1] = f->impl.mOutEventsTotal.load();
}
