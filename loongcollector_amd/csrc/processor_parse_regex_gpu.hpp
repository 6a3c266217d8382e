// processor_parse_regex_gpu.hpp -- MI355X drop-in for LoongCollector's processor_parse_regex_native.
//
// Mirrors, member for member, the reference class
//   core/plugin/processor/ProcessorParseRegexNative.h:28-72 / .cpp:27-257
// and its policy helper core/plugin/processor/CommonParserOptions.{h,cpp}; the only part that differs is
// where the regex arithmetic runs: instead of one boost::regex_match call per event
// (ProcessorParseRegexNative.cpp:194 -> core/common/StringTools.cpp:183-211) the whole event group is matched
// by ONE lc_regex_match_host_views() call on the GPU and the (offset,length) table is stitched back into the
// events as zero-copy views (LogEvent::SetContentNoCopy), exactly as :249-251 does.
#pragma once

#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/lc_regex_gpu.h"
#ifdef LC_USE_REFERENCE_HEADERS  // built inside the LoongCollector tree: the real event model
#include <memory>  // (core/app_config/AppConfig.h names std::set and std::shared_ptr without including their headers)
#include <set>

// LC_REFERENCE_MODELS_ONLY: the reference's event model without the agent around it (no AppConfig, pipeline context or alarm
// manager) -- what tests/test_processor_host_double.py links against oracle/_ref/libref_models.so, the reference's own
// core/models/*.cpp compiled here
#ifndef LC_REFERENCE_MODELS_ONLY
#define LC_HAVE_AGENT_CONTEXT 1
#include "app_config/AppConfig.h"
#include "collection_pipeline/CollectionPipelineContext.h"
#include "monitor/AlarmManager.h"
#endif
#include "models/LogEvent.h"
#include "models/PipelineEventGroup.h"
#else
#include "event_model.hpp"
#endif
#include "json_min.hpp"

namespace logtail {

// CommonParserOptions (core/plugin/processor/CommonParserOptions.h:28-41); renamed so that a build inside the
// reference tree does not collide with the original type
struct GpuCommonParserOptions {
    static const std::string legacyUnmatchedRawLogKey;  // "__raw_log__"
    bool mKeepingSourceWhenParseFail = false;
    bool mKeepingSourceWhenParseSucceed = false;
    std::string mRenamedSourceKey;
    bool mCopingRawLog = false;

    bool Init(const lcjson::Value& config, std::vector<std::string>& warnings);
    bool ShouldAddSourceContent(bool parseSuccess) const;
    bool ShouldAddLegacyUnmatchedRawLog(bool parseSuccess) const;
    bool ShouldEraseEvent(bool parseSuccess, const LogEvent& sourceEvent, const GroupMetadata& metadata) const;
};

class ProcessorPipelineGpu;

class ProcessorParseRegexGpu {
    friend class ProcessorPipelineGpu;  // the fused split -> parse -> filter trip stitches with this class's own code

public:
    static const std::string sName;  // "processor_parse_regex_gpu" (new Type name; static names win in PluginRegistry)
    ~ProcessorParseRegexGpu();

    const std::string& Name() const { return sName; }
    // config: the plugin's JSON object (SourceKey, Regex, Keys, KeepingSourceWhenParseFail, ...).
    // Returns false with `error` set exactly where the reference's Init returns false.
    bool Init(const lcjson::Value& config, std::string& error);
    void Process(PipelineEventGroup& logGroup);
    // The same, and the sum of the events' DataSize() before and after -- the part of ProcessorInstance's in/out_size_bytes
    // (ProcessorInstance.cpp:46-63) that costs a walk over every event; here each event is measured while the gather / the stitch
    // has it in hand (two extra passes over a 1000-event group are 9 us of its ~60 us of host work).
    struct EventBytes {
        size_t in = 0, out = 0;
    };
    void Process(PipelineEventGroup& logGroup, EventBytes* eventBytes);

    std::string mSourceKey;
    std::string mRegex;
    std::vector<std::string> mKeys;
    GpuCommonParserOptions mCommonParserOptions;

    // plugin counters (ProcessorParseRegexNative.cpp:100-103)
    std::atomic<uint64_t> mDiscardedEventsTotal{0}, mOutFailedEventsTotal{0}, mOutKeyNotFoundEventsTotal{0},
        mOutSuccessfulEventsTotal{0};
    // no reference counterpart: lines the decide kernel gave up on (boost would have thrown its complexity exception; they
    // are parse failures too) and lines left undecided because the decide pass was switched off (kept untouched)
    std::atomic<uint64_t> mComplexityExceededEventsTotal{0}, mUndecidedEventsTotal{0};
    std::atomic<uint64_t> mDeviceFailedEventsTotal{0};  // events passed on unparsed because the device call of their group failed
    std::vector<std::string> mInitWarnings;
    int mEngineChoice = LC_ENGINE_AUTO;  // test hook: force a device engine

    // The alarms of RegexLogLineParser (ProcessorParseRegexNative.cpp:196-244: REGEX_MATCH_ALARM with the texts below, raised per
    // failing event when AppConfig::IsLogParseAlarmValid()).  kind: 0 = no match ("errorlog:<line>"), 1 = the matcher gave up on
    // the line (boost: the complexity exception; "errorlog:<line> | exception:<text>"), 2 = key count mismatch ("parse key count
    // not match<what.size()>errorlog:<line>"), 3 = the device call of a group failed (no reference counterpart: the group goes on
    // unparsed).  In the agent build SetContext() routes them to the pipeline's AlarmManager and
    // logger exactly as the reference does; any build can also install a sink (tests, hosts without a context).
    using AlarmSink = void (*)(void* user, int kind, const char* message, size_t len);
    void SetAlarmSink(AlarmSink sink, void* user) {
        mAlarmSink = sink;
        mAlarmUser = user;
    }
#ifdef LC_HAVE_AGENT_CONTEXT
    void SetContext(CollectionPipelineContext* context) { mContext = context; }
#endif

    bool IsWholeLineMode() const { return mIsWholeLineMode; }
    const lc_regex_t* Regex() const { return mReg; }
    int MarkCount() const { return mMarkCount; }

private:
    bool IsSupportedEvent(const PipelineEventPtr& e) const { return e.Is<LogEvent>(); }
    // the per-event policy of ProcessorParseRegexNative::ProcessEvent (:132-168) given the match result
    // per-call tallies: the runner threads share this instance, and an atomic increment per EVENT on a shared cache line is
    // what stopped lc_processor_process from scaling with threads; Process() adds its tallies to the counters once, at the end
    struct Tally {
        uint64_t discarded = 0, outFailed = 0, keyNotFound = 0, outSuccessful = 0, complexityExceeded = 0, undecided = 0;
    };
    bool FinishEvent(LogEvent& sourceEvent, StringView rawContent, bool parseSuccess, const GroupMetadata& metadata, Tally& tally);
    // the same after the source key has been dealt with (:156-167)
    bool FinishSourceDropped(LogEvent& sourceEvent, StringView rawContent, bool parseSuccess, const GroupMetadata& metadata,
                             Tally& tally);
    void AddLog(const StringView& key, const StringView& value, LogEvent& targetEvent, bool overwritten = true);
    // the (key, view) pairs of one matched event: c = its 2 * mark_count capture offsets (:249-251)
    void StitchMatched(LogEvent& ev, StringView raw, const int32_t* c);
    void AddTally(const Tally& tally);
    bool AlarmsWanted() const;

    void RaiseAlarm(int kind, StringView buffer, StringView logPath) const;
    void ReportDeviceFailure(int rc, uint32_t nLines) const;  // alarm kind 3: "GPU match failed (rc=..: ..); N events left unparsed"
    AlarmSink mAlarmSink = nullptr;
    void* mAlarmUser = nullptr;
#ifdef LC_HAVE_AGENT_CONTEXT
    CollectionPipelineContext* mContext = nullptr;
#endif

    bool mSourceKeyOverwritten = false;
    bool mKeysDistinct = false;             // no key appears twice: the bulk stitch is allowed
    std::vector<StringView> mKeyViews;      // views of mKeys (what is stored in the events)
    bool mIsWholeLineMode = false;
    lc_regex_t* mReg = nullptr;
    int mMarkCount = 0;
};

}  // namespace logtail
