// processor_pipeline_gpu.cpp -- see processor_pipeline_gpu.hpp and include/lc_processor.h (lc_pipeline_*).
#include "processor_pipeline_gpu.hpp"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstring>

#include "../../include/lc_regex_gpu.h"
#include "regex_handle.hpp"

namespace logtail {

namespace {
// a new event joins the container as the reference's splitter adds it (ProcessorSplitLogStringNative.cpp:157: pooled events)
void pushEvent(EventsContainer& out, std::unique_ptr<LogEvent>&& e) {
#ifdef LC_USE_REFERENCE_HEADERS
    out.emplace_back(std::move(e), true, nullptr);
#else
    out.emplace_back(std::move(e));
#endif
}
}  // namespace

const std::string ProcessorPipelineGpu::sName = "processor_split_parse_filter_gpu";

bool ProcessorPipelineGpu::Init(const lcjson::Value& config, std::string& error) {
    if (!config.isObject()) {
        error = "plugin config is not an object";
        return false;
    }
    if (const lcjson::Value* s = config.find("Split")) {
        if (!s->isObject()) {
            error = "param Split is not an object";
            return false;
        }
        if (const lcjson::Value* k = s->find("SourceKey"))
            if (k->isString() && !k->str.empty()) mSplitKey = k->str;
        if (const lcjson::Value* c = s->find("SplitChar"))
            if (c->isString() && c->str.size() == 1) mSplitChar = c->str[0];
    }
    if (const lcjson::Value* f = config.find("Fused"))
        if (f->isBool()) mFusedWanted = f->b;
    const lcjson::Value* parse = config.find("Parse");
    if (!parse) {
        error = "mandatory param Parse is missing";
        return false;
    }
    if (!mParse.Init(*parse, error)) return false;
    mHasFilter = false;
    if (const lcjson::Value* filter = config.find("Filter")) {
        if (!mFilter.Init(*filter, error)) return false;
        mHasFilter = true;
    }
    // ---- can the three steps travel together?
    mRules.clear();
    mFused = false;
    if (!mFusedWanted || mParse.mIsWholeLineMode || !mParse.mReg) return true;
    if (mParse.mSourceKey != mSplitKey) return true;                           // the parser reads what the splitter writes
    if (size_t(mParse.mMarkCount) + 1 <= mParse.mKeys.size()) return true;     // key-count mismatch: every event fails (:227-244)
    if (!mParse.mKeysDistinct) return true;                                    // a later key would overwrite an earlier one's value
    if (mHasFilter) {
        if (mFilter.mDiscardingNonUTF8) return true;
        if (mFilter.mFilterMode == ProcessorFilterGpu::Mode::EXPRESSION_MODE) return true;
        if (mFilter.mFilterMode == ProcessorFilterGpu::Mode::RULE_MODE) {
            if (mFilter.mRuleLeaves.size() > 8) return true;
            for (int leaf : mFilter.mRuleLeaves) {
                const auto& l = mFilter.mLeaves[size_t(leaf)];
                const auto it = std::find(mParse.mKeys.begin(), mParse.mKeys.end(), l.key);
                if (it == mParse.mKeys.end()) return true;  // (a rule on the kept / renamed source, or on a key nobody writes)
                // a line the parser cannot match must not be able to satisfy a rule: what the parser leaves in such an event is
                // the (renamed) source and the legacy raw-log copy, so no rule may name those.  (After a success they are added
                // with overwritten = false and never replace a parsed key: the key's value is the capture span.)
                if (l.key == mParse.mCommonParserOptions.mRenamedSourceKey || l.key == mParse.mSourceKey ||
                    l.key == GpuCommonParserOptions::legacyUnmatchedRawLogKey)
                    return true;
                if (lc_regex_prepare_span_filter(l.reg) != LC_OK) return true;
                mRules.push_back({l.reg, uint32_t(it - mParse.mKeys.begin()) + 1});
            }
        }
    }
    // without a rule an event the parser keeps after a failure (KeepingSourceWhenParseFail) would have to come back as well:
    // the fused trip is for pipelines whose filter names at least one parsed key
    if (mRules.empty()) return true;
    mFused = true;
    return true;
}

// ProcessorSplitLogStringNative.cpp:130-160, one line of one read buffer
std::unique_ptr<LogEvent> ProcessorPipelineGpu::NewLineEvent(PipelineEventGroup& logGroup, const LogEvent& sourceEvent,
                                                             StringView sourceVal, StringView sourceKey, uint32_t off,
                                                             uint32_t len) const {
    std::unique_ptr<LogEvent> targetEvent = logGroup.CreateLogEvent(true);
    const StringView content(sourceVal.data() + off, len);
    targetEvent->SetContentNoCopy(sourceKey, content);
    if (sourceEvent.GetTimestampNanosecond())
        targetEvent->SetTimestamp(sourceEvent.GetTimestamp(), *sourceEvent.GetTimestampNanosecond());
    else
        targetEvent->SetTimestamp(sourceEvent.GetTimestamp());
    const uint64_t offset = sourceEvent.GetPosition().first + off;
    const uint64_t length = size_t(off) + len == sourceVal.size() ? sourceEvent.GetPosition().second - off : uint64_t(len) + 1;
    targetEvent->SetPosition(offset, length);
    if (logGroup.HasMetadata(EventGroupMetaKey::LOG_FILE_OFFSET_KEY)) {
        StringBuffer offsetStr = logGroup.GetSourceBuffer()->CopyString(std::to_string(offset));
        targetEvent->SetContentNoCopy(logGroup.GetMetadata(EventGroupMetaKey::LOG_FILE_OFFSET_KEY),
                                      StringView(offsetStr.data, offsetStr.size));
    }
    return targetEvent;
}

void ProcessorPipelineGpu::SplitEvents(PipelineEventGroup& logGroup) const {
    EventsContainer newEvents;
    for (PipelineEventPtr& e : logGroup.MutableEvents()) {
        if (!e.Is<LogEvent>()) {  // :103-106
            newEvents.emplace_back(std::move(e));
            continue;
        }
        LogEvent& sourceEvent = e.Cast<LogEvent>();
        if (sourceEvent.Size() != 1 || !sourceEvent.HasContent(mSplitKey)) {  // :110-127 (the alarm is the agent's)
            newEvents.emplace_back(std::move(e));
            continue;
        }
        const StringView sourceVal = sourceEvent.GetContent(mSplitKey);
        StringBuffer sourceKey = logGroup.GetSourceBuffer()->CopyString(mSplitKey);
        size_t begin = 0;
        while (begin < sourceVal.size()) {  // :133-159 with GetNextLine :162-173
            const void* hit = std::memchr(sourceVal.data() + begin, mSplitChar, sourceVal.size() - begin);
            const size_t end = hit ? size_t(static_cast<const char*>(hit) - sourceVal.data()) : sourceVal.size();
            pushEvent(newEvents, NewLineEvent(logGroup, sourceEvent, sourceVal, StringView(sourceKey.data, sourceKey.size),
                                              uint32_t(begin), uint32_t(end - begin)));
            begin = end + 1;
        }
    }
    logGroup.SwapEvents(newEvents);
}

namespace {
// per runner thread: a stream, pinned staging, device buffers; grow-only
struct PipeBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    void release() {
        if (p) (void)(pinned ? hipHostFree(p) : hipFree(p));
        p = nullptr;
        cap = 0;
    }
    hipError_t ensure(size_t bytes) {
        if (p && cap >= bytes) return hipSuccess;
        release();
        const size_t want = bytes + (bytes >> 2) + 256;
        const hipError_t e = pinned ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
};
struct PipeThread {
    hipStream_t stream = nullptr;
    int device = -1;
    PipeBuf hIn, hOut;                                             // pinned
    PipeBuf dData, dOff, dCaps, dStatus, dPacked, dCounts, dScratch;  // device
    uint32_t survivorGuess = 64;
    PipeThread() { hIn.pinned = hOut.pinned = true; }
    ~PipeThread();  // a runner thread that ends without lc_thread_release() does not leak its stream and buffers
};
thread_local PipeThread tlsPipe;
}  // namespace
}  // namespace logtail
void lcRegisterExitHook();
bool lcRuntimeUsable();  // gpu_runtime.hip: false once the process is exiting (the HIP runtime may be gone)
int lcHostEntryDevice(int* dev);  // gpu_runtime.hip: the calling thread's device binding
// lc_thread_release(): the calling thread's staging and stream
void lcPipelineThreadRelease() {
    logtail::PipeThread& T = logtail::tlsPipe;
    if (T.stream) {
        (void)hipStreamSynchronize(T.stream);
        (void)hipStreamDestroy(T.stream);
        T.stream = nullptr;
    }
    for (logtail::PipeBuf* b : {&T.hIn, &T.hOut, &T.dData, &T.dOff, &T.dCaps, &T.dStatus, &T.dPacked, &T.dCounts, &T.dScratch}) b->release();
    T.device = -1;
}
namespace logtail {
namespace {
PipeThread::~PipeThread() {
    if (lcRuntimeUsable() && (stream || hIn.p || dData.p)) lcPipelineThreadRelease();
}

struct Trip {
    uint32_t lines = 0, survivors = 0, failed = 0, undecided = 0;
    std::vector<int32_t> rows;  // survivors sorted by line: [line, off, len, caps...]
};
}  // namespace

#define PIPE_TRY(expr)                                                              \
    do {                                                                            \
        const hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) {                                                     \
            error = std::string(#expr) + ": " + hipGetErrorString(e_);              \
            return false;                                                           \
        }                                                                           \
    } while (0)

bool ProcessorPipelineGpu::ProcessFused(PipelineEventGroup& logGroup, std::string& error, bool& fellBack) {
    fellBack = false;
    EventsContainer& events = logGroup.MutableEvents();
    for (const PipelineEventPtr& e : events) {
        if (!e.Is<LogEvent>()) {
            fellBack = true;
            return true;
        }
        const LogEvent& ev = e.Cast<LogEvent>();
        if (ev.Size() != 1 || !ev.HasContent(mSplitKey)) {
            fellBack = true;
            return true;
        }
    }
    if (mParse.AlarmsWanted()) {  // the alarm texts quote the failing lines: they have to come back, so nothing is saved
        fellBack = true;
        return true;
    }
    if (lc_device_count() <= 0) {
        error = "no HIP device: the pipeline has no CPU path";
        return false;
    }
    PipeThread& T = tlsPipe;
    int dev = 0;
    if (lcHostEntryDevice(&dev) != LC_OK) {  // the thread's binding
        error = std::string("pipeline: ") + lc_last_error();
        return false;
    }
    // (a thread that moved to another device starts over: its stream is destroyed, its grow-only buffers -- allocated on the old
    // device -- are released; they used to be kept and handed to kernels of the new device)
    if (T.stream && T.device != dev) lcPipelineThreadRelease();
    if (!T.stream) {
        lcRegisterExitHook();
        PIPE_TRY(hipStreamCreateWithFlags(&T.stream, hipStreamNonBlocking));
        T.device = dev;
    }
    const uint32_t G = uint32_t(mParse.mMarkCount);
    const uint32_t rowInts = 3 + 2 * G;
    std::vector<Trip> trips(events.size());
    std::vector<lc_span_filter_t> rules;
    for (const Rule& r : mRules) rules.push_back({r.re, r.group});
    // ---- one trip per read buffer: up, split, match, filter on the spans, survivors down
    for (size_t s = 0; s < events.size(); ++s) {
        const StringView sourceVal = events[s].Cast<LogEvent>().GetContent(mSplitKey);
        const uint64_t nbytes = sourceVal.size();
        if (nbytes == 0) continue;
        if (nbytes >= 0xFFFFFFF0ull) {
            fellBack = true;
            return true;
        }
        const uint32_t maxLines = uint32_t(nbytes);  // (a line has at least its separator, except the last)
        const size_t splitScratch = lc_split_scratch_bytes(nbytes);
        PIPE_TRY(T.hIn.ensure(nbytes + 16));
        PIPE_TRY(T.dData.ensure(nbytes + 16));
        PIPE_TRY(T.dOff.ensure((size_t(maxLines) + 2) * 4));
        PIPE_TRY(T.dCaps.ensure(size_t(maxLines) * 2 * G * 4));
        PIPE_TRY(T.dStatus.ensure(size_t(maxLines) + 16));
        PIPE_TRY(T.dCounts.ensure(128));  // [0] line count of the split | [8..11] the filter's counters
        PIPE_TRY(T.dScratch.ensure(splitScratch));
        std::memcpy(T.hIn.p, sourceVal.data(), nbytes);
        std::memset(static_cast<uint8_t*>(T.hIn.p) + nbytes, 0, 16);
        // no copy engine anywhere in the trip (its queue is shared by all runner threads): a kernel pulls
        // the buffer out of the pinned staging, and the filter kernel writes its survivors straight into pinned memory
        int rc = lc_upload_pinned(T.hIn.p, T.dData.p, nbytes + 16, T.stream);
        uint32_t* dNLines = static_cast<uint32_t*>(T.dCounts.p);
        if (rc == LC_OK)
            rc = lc_split_lines_device(static_cast<const uint8_t*>(T.dData.p), nbytes, uint8_t(mSplitChar), static_cast<uint32_t*>(T.dOff.p),
                                       maxLines + 2, dNLines, T.dScratch.p, splitScratch, T.stream);
        if (rc == LC_OK)
            rc = lc_regex_match_device_dyn(mParse.mReg, mParse.mEngineChoice, static_cast<const uint8_t*>(T.dData.p),
                                           static_cast<const uint32_t*>(T.dOff.p), 1, dNLines, maxLines, G,
                                           static_cast<int32_t*>(T.dCaps.p), static_cast<uint8_t*>(T.dStatus.p), T.stream);
        for (int attempt = 0; attempt < 2 && rc == LC_OK; ++attempt) {
            const uint32_t cap = attempt == 0 ? std::min(maxLines, T.survivorGuess) : trips[s].survivors;
            PIPE_TRY(T.hOut.ensure(64 + size_t(cap) * rowInts * 4));
            rc = lc_span_filter_device(rules.data(), uint32_t(rules.size()), static_cast<const uint8_t*>(T.dData.p),
                                       static_cast<const uint32_t*>(T.dOff.p), 1, dNLines, maxLines, G, static_cast<const int32_t*>(T.dCaps.p),
                                       static_cast<const uint8_t*>(T.dStatus.p),
                                       reinterpret_cast<int32_t*>(static_cast<uint8_t*>(T.hOut.p) + 64), cap,
                                       static_cast<uint32_t*>(T.dCounts.p) + 8, T.stream);
            if (rc != LC_OK) break;
            // (the kernel's counters are device atomics: they live in device memory and 16 bytes come down behind the kernel --
            // atomics on pinned host memory need PCIe atomic support the platform may not have; the survivor rows are plain stores)
            PIPE_TRY(hipMemcpyAsync(T.hOut.p, static_cast<uint32_t*>(T.dCounts.p) + 8, 16, hipMemcpyDeviceToHost, T.stream));
            PIPE_TRY(hipStreamSynchronize(T.stream));
            const uint32_t* c = static_cast<const uint32_t*>(T.hOut.p);
            trips[s].lines = c[0];
            trips[s].survivors = c[1];
            trips[s].failed = c[2];
            trips[s].undecided = c[3];
            if (c[1] <= cap) break;  // (else: more survivors than guessed -- once more with room for all of them)
        }
        if (rc != LC_OK) {
            error = rc == LC_ERR_NO_DEVICE ? "no HIP device: the pipeline has no CPU path" : lc_last_error();
            return false;
        }
        if (trips[s].undecided) {  // (only with the decide pass switched off / out of budget: the chained path books those exactly)
            fellBack = true;
            return true;
        }
        T.survivorGuess = std::max<uint32_t>(64, trips[s].survivors + trips[s].survivors / 4 + 16);
        const int32_t* packed = reinterpret_cast<const int32_t*>(static_cast<const uint8_t*>(T.hOut.p) + 64);
        std::vector<uint32_t> idx(trips[s].survivors);
        for (uint32_t k = 0; k < trips[s].survivors; ++k) idx[k] = k;
        std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return packed[size_t(a) * rowInts] < packed[size_t(b) * rowInts]; });
        trips[s].rows.resize(size_t(trips[s].survivors) * rowInts);
        for (uint32_t k = 0; k < trips[s].survivors; ++k)
            std::memcpy(&trips[s].rows[size_t(k) * rowInts], packed + size_t(idx[k]) * rowInts, rowInts * 4);
    }

    // ---- the events the three processors would have left, and their counters
    const GroupMetadata& metadata = logGroup.GetAllMetadata();
    ProcessorParseRegexGpu::Tally tally;
    uint64_t filterIn = 0, filterOut = 0, lines = 0;
    EventsContainer newEvents;
    // what the parser does with a line it cannot match depends on the configuration only (the events the splitter makes all look
    // alike): one representative event goes through the parser's own code, its tally is multiplied
    bool failedReachesFilter = false;
    ProcessorParseRegexGpu::Tally failedTally;
    bool haveFailedTally = false;
    for (size_t s = 0; s < events.size(); ++s) {
        const LogEvent& sourceEvent = events[s].Cast<LogEvent>();
        const StringView sourceVal = sourceEvent.GetContent(mSplitKey);
        const Trip& t = trips[s];
        if (!t.lines) continue;
        lines += t.lines;
        StringBuffer sourceKey = logGroup.GetSourceBuffer()->CopyString(mSplitKey);
        const StringView keyView(sourceKey.data, sourceKey.size);
        if (t.failed && !haveFailedTally) {
            std::unique_ptr<LogEvent> probe = NewLineEvent(logGroup, sourceEvent, sourceVal, keyView, 0, 0);
            failedTally.outFailed = 1;
            failedReachesFilter = mParse.FinishEvent(*probe, StringView(sourceVal.data(), 0), false, metadata, failedTally);
            haveFailedTally = true;
        }
        for (uint64_t k = 0; k < t.failed; ++k) {  // (sums of small integers: a loop keeps the struct private to the parser)
            tally.outFailed += failedTally.outFailed;
            tally.discarded += failedTally.discarded;
            tally.outSuccessful += failedTally.outSuccessful;
        }
        if (failedReachesFilter) filterIn += t.failed;  // no rule key in a failed event: IsMatched is false (:260-264) -- unless there
                                                        // is no rule at all (handled below)
        const uint64_t matched = uint64_t(t.lines) - t.failed;
        // every matched line leaves the parser as an event (ShouldEraseEvent is false after a success) and reaches the filter
        tally.outSuccessful += matched - t.survivors;
        filterIn += matched;
        const uint32_t rowInts2 = rowInts;
        for (uint32_t k = 0; k < t.survivors; ++k) {
            const int32_t* row = &t.rows[size_t(k) * rowInts2];
            std::unique_ptr<LogEvent> ev = NewLineEvent(logGroup, sourceEvent, sourceVal, keyView, uint32_t(row[1]), uint32_t(row[2]));
            const StringView raw(sourceVal.data() + row[1], size_t(row[2]));
            mParse.StitchMatched(*ev, raw, row + 3);
            if (mParse.FinishEvent(*ev, raw, true, metadata, tally)) {
                pushEvent(newEvents, std::move(ev));
                ++filterOut;
            }
        }
    }
    logGroup.SwapEvents(newEvents);
    mParse.AddTally(tally);
    if (mHasFilter) {
        mFilter.mInEventsTotal += filterIn;
        mFilter.mOutEventsTotal += filterOut;
    }
    mLinesTotal += lines;
    mSurvivorsTotal += filterOut;
    return true;
}

bool ProcessorPipelineGpu::Process(PipelineEventGroup& logGroup, std::string& error) {
    if (logGroup.GetEvents().empty()) return true;
    if (mFused) {
        bool fellBack = false;
        if (!ProcessFused(logGroup, error, fellBack)) return false;
        if (!fellBack) {
            ++mGroupsFused;
            return true;
        }
    }
    // the three steps one after the other
    ++mGroupsChained;
    SplitEvents(logGroup);
    mParse.Process(logGroup);
    if (mHasFilter) return mFilter.Process(logGroup, error);
    return true;
}

}  // namespace logtail
