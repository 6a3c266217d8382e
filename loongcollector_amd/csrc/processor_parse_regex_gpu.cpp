// processor_parse_regex_gpu.cpp -- see processor_parse_regex_gpu.hpp.
#include "processor_parse_regex_gpu.hpp"

#include <cstdio>

namespace logtail {

const std::string ProcessorParseRegexGpu::sName = "processor_parse_regex_gpu";
const std::string GpuCommonParserOptions::legacyUnmatchedRawLogKey = "__raw_log__";

namespace {
const std::string kDefaultContentKey = "content";      // DEFAULT_CONTENT_KEY, core/constants/Constants.cpp:25
const std::string kContainerTimeKey = "_time_";        // ProcessorParseContainerLogNative.cpp:41
const std::string kContainerSourceKey = "_source_";    // ProcessorParseContainerLogNative.cpp:42

// GetMandatoryStringParam / GetOptional*Param (core/common/ParamExtractor.cpp:31-43,101-113,174-188)
bool mandatoryString(const lcjson::Value& cfg, const std::string& key, std::string& out, std::string& err) {
    const lcjson::Value* v = cfg.find(key);
    if (!v) {
        err = "mandatory param " + key + " is missing";
        return false;
    }
    if (!v->isString()) {
        err = "param " + key + " is not of type string";
        return false;
    }
    out = v->str;
    if (out.empty()) {
        err = "mandatory string param " + key + " is empty";
        return false;
    }
    return true;
}
bool optionalBool(const lcjson::Value& cfg, const std::string& key, bool& out, std::string& err) {
    const lcjson::Value* v = cfg.find(key);
    if (v) {
        if (!v->isBool()) {
            err = "param " + key + " is not of type bool";
            return false;
        }
        out = v->b;
    }
    return true;
}
bool optionalString(const lcjson::Value& cfg, const std::string& key, std::string& out, std::string& err) {
    const lcjson::Value* v = cfg.find(key);
    if (v) {
        if (!v->isString()) {
            err = "param " + key + " is not of type string";
            return false;
        }
        out = v->str;
    }
    return true;
}

// One runner thread's scratch for Process(), reached with ONE thread-local lookup per call (a function-scope thread_local of class
// type goes through an init-check wrapper at every use; inside the per-event loops that was measurable).
struct ProcessScratch {
    std::vector<uint8_t> kind, status;
    std::vector<const uint8_t*> linePtr;
    std::vector<uint32_t> lineLen;
    std::vector<int32_t> caps;
};

// LC_PER_KEY_STITCH: the K fields go in with K x LogEvent::SetContentNoCopy and the source goes with DelContent -- the only way the
// reference's LogEvent offers (its AppendContentNoCopy is private, LogEvent.h:112-118).  A build against the reference headers takes
// it, and so does the stand-in when it is built in the reference's shape (event_model.hpp LC_REFERENCE_SHAPED_EVENT_MODEL).
#if defined(LC_USE_REFERENCE_HEADERS) || defined(LC_REFERENCE_SHAPED_EVENT_MODEL)
#define LC_PER_KEY_STITCH 1
#endif

// HasContent + GetContent (:140-150) -- one scan of the event's contents where the event model offers it
inline bool sourceOf(const LogEvent& ev, const std::string& key, StringView& out) {
#ifdef LC_USE_REFERENCE_HEADERS
    // (the reference's public single-scan lookup, LogEvent.cpp:123-131 -- HasContent + GetContent are two scans)
    const LogEvent& cev = ev;
    const auto it = cev.FindContent(key);
    if (it == cev.end()) return false;
    out = it->second;
    return true;
#else
    const StringView* v = ev.FindContent(key);
    if (!v) return false;
    out = *v;
    return true;
#endif
}
}  // namespace

// ---------------------------------------------------------------------------------------------- GpuCommonParserOptions
// core/plugin/processor/CommonParserOptions.cpp:28-89: a wrongly typed optional only warns and keeps the default
bool GpuCommonParserOptions::Init(const lcjson::Value& config, std::vector<std::string>& warnings) {
    std::string err;
    if (!optionalBool(config, "KeepingSourceWhenParseFail", mKeepingSourceWhenParseFail, err)) warnings.push_back(err);
    if (!optionalBool(config, "KeepingSourceWhenParseSucceed", mKeepingSourceWhenParseSucceed, err)) warnings.push_back(err);
    if (!optionalString(config, "RenamedSourceKey", mRenamedSourceKey, err)) warnings.push_back(err);
    if (mRenamedSourceKey.empty()) {
        const lcjson::Value* sk = config.find("SourceKey");  // guaranteed to exist by the caller
        mRenamedSourceKey = sk ? sk->str : std::string();
    }
    if (!optionalBool(config, "CopingRawLog", mCopingRawLog, err)) warnings.push_back(err);
    return true;
}
// CommonParserOptions.cpp:91-97
bool GpuCommonParserOptions::ShouldAddLegacyUnmatchedRawLog(bool parseSuccess) const {
    return !parseSuccess && mKeepingSourceWhenParseFail && mCopingRawLog;
}
bool GpuCommonParserOptions::ShouldAddSourceContent(bool parseSuccess) const {
    return (parseSuccess && mKeepingSourceWhenParseSucceed) || (!parseSuccess && mKeepingSourceWhenParseFail);
}
// CommonParserOptions.cpp:99-117
bool GpuCommonParserOptions::ShouldEraseEvent(bool parseSuccess, const LogEvent& sourceEvent,
                                           const GroupMetadata& metadata) const {
    if (!parseSuccess && !mKeepingSourceWhenParseFail) {
        if (sourceEvent.Empty()) return true;
        const size_t size = sourceEvent.Size();
        auto offsetKey = metadata.find(EventGroupMetaKey::LOG_FILE_OFFSET_KEY);
        if (size == 1 && offsetKey != metadata.end() && sourceEvent.cbegin()->first == offsetKey->second) return true;
        if (size == 2 && sourceEvent.HasContent(kContainerTimeKey) && sourceEvent.HasContent(kContainerSourceKey))
            return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------- processor
ProcessorParseRegexGpu::~ProcessorParseRegexGpu() {
    if (mReg) lc_regex_free(mReg);
}

// ProcessorParseRegexNative::Init, core/plugin/processor/ProcessorParseRegexNative.cpp:29-106
bool ProcessorParseRegexGpu::Init(const lcjson::Value& config, std::string& error) {
    if (!config.isObject()) {
        error = "plugin config is not an object";
        return false;
    }
    if (!mandatoryString(config, "SourceKey", mSourceKey, error)) return false;  // :33-43
    if (!mandatoryString(config, "Regex", mRegex, error)) return false;          // :45-53
    mIsWholeLineMode = mRegex == "(.*)";                                         // :68
    {
        // :53-67: IsRegexValid + one compiled regex.  (The reference compiles one boost::regex per runner thread
        // because match_results are per thread; the device tables are immutable and shared by every thread.)
        char err[256];
        int rc = lc_regex_compile(mRegex.data(), mRegex.size(), 0, mEngineChoice, &mReg, err, sizeof err);
        if (rc == LC_ERR_SYNTAX) {
            error = "mandatory string param Regex is not a valid regex";
            return false;
        }
        if (rc != LC_OK) {
            if (!mIsWholeLineMode) {
                error = std::string("param Regex cannot be executed by the GPU engines: ") + err;
                return false;
            }
            mReg = nullptr;
        }
        mMarkCount = mReg ? lc_regex_mark_count(mReg) : 1;
    }
    // Keys :71-88 (mandatory, non-empty list of strings; legacy ["k1,k2"] form is split on ',')
    {
        const lcjson::Value* keys = config.find("Keys");
        if (!keys) {
            error = "mandatory param Keys is missing";
            return false;
        }
        if (!keys->isArray()) {
            error = "param Keys is not of type list";
            return false;
        }
        mKeys.clear();
        for (const auto& k : keys->arr) {
            if (!k.isString()) {
                error = "param Keys is not of type string list";
                return false;
            }
            mKeys.push_back(k.str);
        }
        if (mKeys.empty()) {
            error = "mandatory list param Keys is empty";
            return false;
        }
        if (mKeys.size() == 1 && mKeys[0].find(',') != std::string::npos) {
            std::vector<std::string> parts;
            size_t at = 0;
            const std::string joined = mKeys[0];
            for (;;) {
                size_t c = joined.find(',', at);
                parts.push_back(joined.substr(at, c == std::string::npos ? std::string::npos : c - at));
                if (c == std::string::npos) break;
                at = c + 1;
            }
            mKeys = parts;
        }
    }
    mKeyViews.assign(mKeys.begin(), mKeys.end());
    mKeysDistinct = true;
    for (size_t a = 0; a < mKeys.size(); ++a)
        for (size_t b = a + 1; b < mKeys.size(); ++b)
            if (mKeys[a] == mKeys[b]) mKeysDistinct = false;
    mSourceKeyOverwritten = false;  // :89-94
    for (const auto& k : mKeys)
        if (k == mSourceKey) {
            mSourceKeyOverwritten = true;
            break;
        }
    return mCommonParserOptions.Init(config, mInitWarnings);  // :96
}

// ProcessorParseRegexNative::AddLog :176-184
void ProcessorParseRegexGpu::AddLog(const StringView& key, const StringView& value, LogEvent& targetEvent,
                                    bool overwritten) {
    if (!overwritten && targetEvent.HasContent(key)) return;
    targetEvent.SetContentNoCopy(key, value);
}

// the tail of ProcessorParseRegexNative::ProcessEvent :153-167
bool ProcessorParseRegexGpu::FinishEvent(LogEvent& sourceEvent, StringView rawContent, bool parseSuccess,
                                         const GroupMetadata& metadata, Tally& tally) {
    if (!parseSuccess || !mSourceKeyOverwritten) sourceEvent.DelContent(mSourceKey);
    return FinishSourceDropped(sourceEvent, rawContent, parseSuccess, metadata, tally);
}

// :156-167, what follows the handling of the source key
bool ProcessorParseRegexGpu::FinishSourceDropped(LogEvent& sourceEvent, StringView rawContent, bool parseSuccess,
                                                 const GroupMetadata& metadata, Tally& tally) {
    if (mCommonParserOptions.ShouldAddSourceContent(parseSuccess))
        AddLog(mCommonParserOptions.mRenamedSourceKey, rawContent, sourceEvent, false);
    if (mCommonParserOptions.ShouldAddLegacyUnmatchedRawLog(parseSuccess))
        AddLog(GpuCommonParserOptions::legacyUnmatchedRawLogKey, rawContent, sourceEvent, false);
    if (mCommonParserOptions.ShouldEraseEvent(parseSuccess, sourceEvent, metadata)) {
        ++tally.discarded;
        return false;
    }
    ++tally.outSuccessful;
    return true;
}

// ProcessorParseRegexNative::Process :108-126 + ProcessEvent :132-168 + RegexLogLineParser :186-253, restructured
// as gather -> one device match for the whole group -> stitch.
// ProcessorParseRegexNative.cpp:196-244, the alarm side of RegexLogLineParser
void ProcessorParseRegexGpu::RaiseAlarm(int kind, StringView buffer, StringView logPath) const {
    bool wanted = mAlarmSink != nullptr;
#ifdef LC_HAVE_AGENT_CONTEXT
    const bool agent = mContext && AppConfig::GetInstance()->IsLogParseAlarmValid();
    wanted = wanted || agent;
#endif
    if (!wanted) return;
    static const char kGaveUp[] = "the matcher gave up: the line needs more work than the complexity budget allows";
    std::string message;
    if (kind == 2) message = "parse key count not match" + std::to_string(mMarkCount + 1);  // what.size() = mark_count + 1
    message += "errorlog:";
    message.append(buffer.data(), buffer.size());
    if (kind == 1) message += std::string(" | exception:") + kGaveUp;
    if (mAlarmSink) mAlarmSink(mAlarmUser, kind, message.data(), message.size());
#ifdef LC_HAVE_AGENT_CONTEXT
    if (!agent) return;
    if (mContext->GetAlarm().IsLowLevelAlarmValid()) {
        if (kind == 1) {
            LOG_ERROR(mContext->GetLogger(),
                      ("parse regex log fail", buffer)("exception", kGaveUp)("project", mContext->GetProjectName())(
                          "logstore", mContext->GetLogstoreName())("file", logPath));
        } else if (kind == 2) {
            LOG_WARNING(mContext->GetLogger(),
                        ("parse key count not match", mMarkCount + 1)("parse regex log fail", buffer)(
                            "project", mContext->GetProjectName())("logstore", mContext->GetLogstoreName())("file", logPath));
        } else {
            LOG_WARNING(mContext->GetLogger(),
                        ("parse regex log fail", buffer)("project", mContext->GetProjectName())(
                            "logstore", mContext->GetLogstoreName())("file", logPath));
        }
    }
    mContext->GetAlarm().SendAlarmWarning(REGEX_MATCH_ALARM, message, mContext->GetRegion(), mContext->GetProjectName(),
                                          mContext->GetConfigName(), mContext->GetLogstoreName());
#else
    (void)logPath;
#endif
}

// A failed device call: the group goes on unparsed (there is no CPU path) and the failure is reported where the parse alarms go --
// the installed sink (kind 3), in an agent build the pipeline's logger and alarm manager under the alarm's own rate limit.  Only a
// host with neither hears it on stderr, and there once per 1024 failures: a dead device fails every group of every runner thread.
void ProcessorParseRegexGpu::ReportDeviceFailure(int rc, uint32_t nLines) const {
    const std::string message = "GPU match failed (rc=" + std::to_string(rc) + ": " + lc_last_error() + "); " + std::to_string(nLines) +
                                " events left unparsed";
    bool heard = false;
    if (mAlarmSink) {
        mAlarmSink(mAlarmUser, 3, message.data(), message.size());
        heard = true;
    }
#ifdef LC_HAVE_AGENT_CONTEXT
    if (mContext) {
        if (mContext->GetAlarm().IsLowLevelAlarmValid())
            LOG_ERROR(mContext->GetLogger(), ("processor_parse_regex_gpu", message)("project", mContext->GetProjectName())(
                                                 "logstore", mContext->GetLogstoreName()));
        mContext->GetAlarm().SendAlarmWarning(REGEX_MATCH_ALARM, message, mContext->GetRegion(), mContext->GetProjectName(),
                                              mContext->GetConfigName(), mContext->GetLogstoreName());
        heard = true;
    }
#endif
    if (heard) return;
    static std::atomic<uint64_t> failures{0};
    const uint64_t k = failures.fetch_add(1, std::memory_order_relaxed);
    if ((k & 1023u) == 0)
        std::fprintf(stderr, "[%s] %s%s\n", sName.c_str(), message.c_str(), k ? " (and 1023 more such groups since the last line)" : "");
}

void ProcessorParseRegexGpu::Process(PipelineEventGroup& logGroup) { Process(logGroup, nullptr); }

void ProcessorParseRegexGpu::Process(PipelineEventGroup& logGroup, EventBytes* eventBytes) {
    if (logGroup.GetEvents().empty()) return;
    size_t bytesIn = 0, bytesOut = 0;  // sums of PipelineEvent::DataSize(), taken while each event is in hand anyway
    EventsContainer& events = logGroup.MutableEvents();
    const GroupMetadata& metadata = logGroup.GetAllMetadata();
    const StringView logPath = logGroup.GetMetadata(EventGroupMetaKey::LOG_FILE_PATH_RESOLVED);  // :110
    const size_t nEvents = events.size();

    enum Kind : uint8_t { Keep, Parse, WholeLine };
    static thread_local ProcessScratch tScratch;
    ProcessScratch& scratch = tScratch;
    scratch.kind.assign(nEvents, Keep);
    scratch.linePtr.resize(nEvents);
    scratch.lineLen.resize(nEvents);
    uint8_t* const kind = scratch.kind.data();
    const uint8_t** const linePtr = scratch.linePtr.data();
    uint32_t* const lineLen = scratch.lineLen.data();
    Tally tally;

    // gather: the source values are views into the group's SourceBuffer; nothing is copied here
    uint32_t nLines = 0;
    for (size_t i = 0; i < nEvents; ++i) {
        PipelineEventPtr& e = events[i];
        if (!IsSupportedEvent(e)) {  // :135-138
            ++tally.outFailed;
            if (eventBytes) bytesIn += e->DataSize();
            continue;
        }
        LogEvent& ev = e.Cast<LogEvent>();
        if (eventBytes) bytesIn += ev.LogEvent::DataSize();
        StringView raw;
        if (!sourceOf(ev, mSourceKey, raw)) {  // :140-143
            ++tally.keyNotFound;
            continue;
        }
        if (mIsWholeLineMode) {  // :147-148, no regex engine involved
            kind[i] = WholeLine;
            continue;
        }
        kind[i] = Parse;
        linePtr[nLines] = reinterpret_cast<const uint8_t*>(raw.data());
        lineLen[nLines] = uint32_t(raw.size());
        ++nLines;
    }

    const uint32_t G = uint32_t(mMarkCount);
    bool deviceOk = true;
    if (nLines) {
        scratch.caps.resize(size_t(nLines) * 2 * G);
        scratch.status.resize(nLines);
        int rc = lc_regex_match_host_views(mReg, linePtr, lineLen, nLines, G, scratch.caps.data(), scratch.status.data());
        if (rc != LC_OK) {
            // No CPU fallback exists.  Leave the events exactly as they came in (nothing is lost) and say so loudly.
            ReportDeviceFailure(rc, nLines);
            deviceOk = false;
            mDeviceFailedEventsTotal += nLines;
        }
    }

    // stitch + in-place compaction (:115-124)
    const int32_t* const caps = scratch.caps.data();
    const uint8_t* const status = scratch.status.data();
    const bool keyCountOk = size_t(G) + 1 > mKeys.size();  // what.size() > keys.size()  :227
#ifndef LC_PER_KEY_STITCH
    // no key equals the source key or another key: an event that holds the source content and nothing else takes its K fields and
    // loses its source in one call (same contents, order and size accounting as :249-251 followed by :153-155; the scan for the
    // source skips the K new entries, none of which can be it)
    const bool bulk = mKeysDistinct && !mSourceKeyOverwritten;
#else
    // Per-key form.  When no key is the source key and a successful parse drops the source (:153-155), the source is dropped BEFORE
    // the K fields go in: DelContent's scan from the back then meets one entry instead of K + 1, and the list that results is the
    // same -- the tombstone stays where the source was, the fields follow in Keys order (:249-251), the sizes add up alike.
    const bool dropSourceFirst = !mSourceKeyOverwritten;
#endif
    const StringView sourceKey(mSourceKey);
    size_t wIdx = 0, line = 0;
    for (size_t rIdx = 0; rIdx < nEvents; ++rIdx) {
        bool keep = true;
        if (kind[rIdx] == WholeLine) {
            LogEvent& ev = events[rIdx].Cast<LogEvent>();
            const StringView raw = ev.GetContent(mSourceKey);
            AddLog(StringView(mKeys.empty() ? kDefaultContentKey : mKeys[0]), raw, ev);  // :170-174
            keep = FinishEvent(ev, raw, true, metadata, tally);
        } else if (kind[rIdx] == Parse) {
            const size_t li = line++;
            if (deviceOk) {
                LogEvent& ev = events[rIdx].Cast<LogEvent>();
                const StringView raw(reinterpret_cast<const char*>(linePtr[li]), lineLen[li]);  // = ev.GetContent(mSourceKey)
#ifndef LC_PER_KEY_STITCH
                if (status[li] == LC_MATCH && keyCountOk && bulk && ev.Size() == 1) {
                    ev.AppendCapturesNoCopy(mKeyViews.data(), mKeyViews.size(), raw, &caps[li * 2 * G], &sourceKey);
#else
                if (status[li] == LC_MATCH && keyCountOk && dropSourceFirst) {
                    ev.DelContent(sourceKey);
                    StitchMatched(ev, raw, &caps[li * 2 * G]);
#endif
                    if (FinishSourceDropped(ev, raw, true, metadata, tally)) {
                        if (eventBytes) bytesOut += ev.LogEvent::DataSize();
                        if (wIdx != rIdx) events[wIdx] = std::move(events[rIdx]);
                        ++wIdx;
                    }
                    continue;
                }
                bool parseSuccess = true;
                if (status[li] == LC_OVERFLOW) {
                    // The line was NOT decided (only possible with the decide pass switched off, LC_NFA_NO_DECIDE): boost
                    // might match it, so it is neither a success nor a parse failure.  The event goes on untouched and is
                    // counted under its own counter.
                    ++tally.undecided;
                    if (eventBytes) bytesOut += ev.LogEvent::DataSize();
                    if (wIdx != rIdx) events[wIdx] = std::move(events[rIdx]);
                    ++wIdx;
                    continue;
                }
                if (status[li] == LC_GAVE_UP) ++tally.complexityExceeded;  // boost: complexity exception -> parse failure
                if (status[li] != LC_MATCH) {  // :194-226
                    RaiseAlarm(status[li] == LC_GAVE_UP ? 1 : 0, raw, logPath);
                    ++tally.outFailed;
                    parseSuccess = false;
                } else if (!keyCountOk) {  // what.size() <= keys.size()  :227-244, no counter
                    RaiseAlarm(2, raw, logPath);
                    parseSuccess = false;
                }
                if (parseSuccess) StitchMatched(ev, raw, &caps[li * 2 * G]);
                keep = FinishEvent(ev, raw, parseSuccess, metadata, tally);
            }
        }
        if (keep) {
            if (eventBytes) bytesOut += events[rIdx]->DataSize();
            if (wIdx != rIdx) events[wIdx] = std::move(events[rIdx]);
            ++wIdx;
        }
    }
    events.resize(wIdx);
    if (eventBytes) {
        eventBytes->in = bytesIn;
        eventBytes->out = bytesOut;
    }
    AddTally(tally);
}

void ProcessorParseRegexGpu::AddTally(const Tally& tally) {
    if (tally.discarded) mDiscardedEventsTotal += tally.discarded;
    if (tally.outFailed) mOutFailedEventsTotal += tally.outFailed;
    if (tally.keyNotFound) mOutKeyNotFoundEventsTotal += tally.keyNotFound;
    if (tally.outSuccessful) mOutSuccessfulEventsTotal += tally.outSuccessful;
    if (tally.complexityExceeded) mComplexityExceededEventsTotal += tally.complexityExceeded;
    if (tally.undecided) mUndecidedEventsTotal += tally.undecided;
}

bool ProcessorParseRegexGpu::AlarmsWanted() const {
#ifdef LC_HAVE_AGENT_CONTEXT
    if (mContext && AppConfig::GetInstance()->IsLogParseAlarmValid()) return true;
#endif
    return mAlarmSink != nullptr;
}

void ProcessorParseRegexGpu::StitchMatched(LogEvent& ev, StringView raw, const int32_t* c) {
#ifndef LC_PER_KEY_STITCH
    if (mKeysDistinct && !mSourceKeyOverwritten && ev.Size() == 1) {
        // the event holds only the source content and no key can collide: append all K views at once
        // instead of K reverse scans (same contents, same order as the loop below)
        ev.AppendCapturesNoCopy(mKeyViews.data(), mKeyViews.size(), raw, c);
        return;
    }
#endif
    for (size_t k = 0; k < mKeys.size(); ++k) {  // :249-251
        const int32_t b = c[2 * k], en = c[2 * k + 1];
        // an unmatched group is boost's {last,last,matched=false}: empty value at end of input
        const StringView val = b < 0 ? StringView(raw.data() + raw.size(), 0) : StringView(raw.data() + b, size_t(en - b));
        AddLog(StringView(mKeys[k]), val, ev);
    }
}

}  // namespace logtail
