// multiline_events.cpp -- the two multiline processors on whole event groups (include/lc_multiline.h).
//
//   lc_multiline_process_group        ProcessorSplitMultilineLogStringNative::Process / ProcessEvent / CreateNewEvent
//                                     core/plugin/processor/inner/ProcessorSplitMultilineLogStringNative.cpp:95-112,126-160,302-339
//   lc_merge_multiline_process_group  ProcessorMergeMultilineLogNative::Process / MergeLogsByFlag / MergeLogsByRegex /
//                                     MergeEvents / HandleUnmatchLogs
//                                     core/plugin/processor/inner/ProcessorMergeMultilineLogNative.cpp:80-92,113-159,161-330,332-392
//
// Which line starts / continues / ends a log is decided on the device (one status-only launch per configured pattern over
// all lines / all events of the group, LC_SYNTAX_PREFIX = regex_search + match_continuous, StringTools.cpp:263-289); what
// is restated here is what the reference does with those answers: which events come out, what their contents, timestamps
// and positions are.  New contents are VIEWS into the source value (or, for merged events, the same in-place memmove the
// reference does): no line is copied.  Alarms and log lines are the agent's business and are not emitted.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/lc_multiline.h"
#include "../../include/lc_regex_gpu.h"
#include "json_min.hpp"
#include "regex_handle.hpp"
#include "multiline_gpu.hpp"
#ifdef LC_USE_REFERENCE_HEADERS
#include "models/LogEvent.h"
#include "models/PipelineEventGroup.h"
#include "models/RawEvent.h"
#else
#include "event_model.hpp"
#endif

using namespace logtail;

namespace {

// a new event of the group (the reference: logGroup.CreateLogEvent(true) + newEvents.emplace_back(std::move(e), true, nullptr),
// i.e. events drawn from the group's pool; the stand-in model has no pool)
#ifdef LC_USE_REFERENCE_HEADERS
void pushLog(PipelineEventGroup& g, EventsContainer& out, std::unique_ptr<LogEvent>&& e) { out.emplace_back(std::move(e), true, nullptr); }
void pushRaw(PipelineEventGroup& g, EventsContainer& out, std::unique_ptr<RawEvent>&& e) { out.emplace_back(std::move(e), true, nullptr); }
std::unique_ptr<LogEvent> newLog(PipelineEventGroup& g) { return g.CreateLogEvent(true); }
std::unique_ptr<RawEvent> newRaw(PipelineEventGroup& g) { return g.CreateRawEvent(true); }
#else
void pushLog(PipelineEventGroup&, EventsContainer& out, std::unique_ptr<LogEvent>&& e) { out.emplace_back(std::move(e)); }
void pushRaw(PipelineEventGroup&, EventsContainer& out, std::unique_ptr<RawEvent>&& e) { out.emplace_back(std::move(e)); }
std::unique_ptr<LogEvent> newLog(PipelineEventGroup& g) { return std::make_unique<LogEvent>(&g); }
std::unique_ptr<RawEvent> newRaw(PipelineEventGroup& g) { return std::make_unique<RawEvent>(&g); }
#endif

void setTimestamp(PipelineEvent& dst, const PipelineEvent& src) {
    if (src.GetTimestampNanosecond()) dst.SetTimestamp(src.GetTimestamp(), *src.GetTimestampNanosecond());
    else dst.SetTimestamp(src.GetTimestamp());
}

}  // namespace

extern "C" int lc_multiline_process_group(lc_multiline_t* m, void* pipelineEventGroup) {
    if (!m || !pipelineEventGroup) return LC_ERR_ARG;
    PipelineEventGroup& logGroup = *static_cast<PipelineEventGroup*>(pipelineEventGroup);
    if (logGroup.GetEvents().empty()) return LC_OK;  // :96-98
    uint64_t inputLines = 0, unmatchLines = 0, matchedEvents = 0;
    EventsContainer newEvents;
    const StringView srcKeyView(m->sourceKey.data(), m->sourceKey.size());
    for (PipelineEventPtr& e : logGroup.MutableEvents()) {
        // ProcessEvent :133-156: anything that is not a log event holding exactly the source content passes through
        if (!e.Is<LogEvent>()) {
            newEvents.emplace_back(std::move(e));
            continue;
        }
        LogEvent& sourceEvent = e.Cast<LogEvent>();
        if (sourceEvent.Size() != 1 || !sourceEvent.HasContent(srcKeyView)) {
            newEvents.emplace_back(std::move(e));
            continue;
        }
        const StringView sourceVal = sourceEvent.GetContent(srcKeyView);
        const StringBuffer sourceKey = logGroup.GetSourceBuffer()->CopyString(m->sourceKey);  // :160
        lc_ml_record_t* recs = nullptr;
        uint32_t nrecs = 0, counters[3] = {0, 0, 0};
        const int rc = lc_multiline_split_host(m, reinterpret_cast<const uint8_t*>(sourceVal.data()), uint32_t(sourceVal.size()), &recs,
                                               &nrecs, counters);
        if (rc != LC_OK) {  // no CPU path: the source event goes on untouched, loudly
            std::fprintf(stderr, "[processor_split_multiline_log_string_gpu] device pass failed (rc=%d: %s); event left unsplit\n", rc,
                         lc_last_error());
            newEvents.emplace_back(std::move(e));
            continue;
        }
        inputLines += counters[0];
        unmatchLines += counters[1];
        matchedEvents += counters[2];
        const auto pos = sourceEvent.GetPosition();
        for (uint32_t r = 0; r < nrecs; ++r) {  // CreateNewEvent :302-339
            const StringView content(sourceVal.data() + recs[r].begin, recs[r].length);
            const bool isLastLog = (recs[r].matched & LC_ML_LAST) != 0;
            if (m->enableRawContent) {
                std::unique_ptr<RawEvent> target = newRaw(logGroup);
                target->SetContentNoCopy(content);
                setTimestamp(*target, sourceEvent);
                pushRaw(logGroup, newEvents, std::move(target));
                continue;
            }
            std::unique_ptr<LogEvent> target = newLog(logGroup);
            target->SetContentNoCopy(StringView(sourceKey.data, sourceKey.size), content);
            setTimestamp(*target, sourceEvent);
            const uint64_t delta = recs[r].begin;
            const uint64_t offset = pos.first + delta;
            const uint64_t length = isLastLog ? pos.second - delta : uint64_t(content.size()) + 1;
            target->SetPosition(offset, length);
            if (logGroup.HasMetadata(EventGroupMetaKey::LOG_FILE_OFFSET_KEY)) {
                const StringBuffer offsetStr = logGroup.GetSourceBuffer()->CopyString(std::to_string(offset));
                target->SetContentNoCopy(logGroup.GetMetadata(EventGroupMetaKey::LOG_FILE_OFFSET_KEY),
                                         StringView(offsetStr.data, offsetStr.size));
            }
            pushLog(logGroup, newEvents, std::move(target));
        }
        lc_multiline_free_records(recs);
    }
    m->matchedLinesTotal += inputLines - unmatchLines;  // :108-109
    m->unmatchedLinesTotal += unmatchLines;
    m->matchedEventsTotal += matchedEvents;
    logGroup.MutableEvents().swap(newEvents);  // SwapEvents :110
    return LC_OK;
}

extern "C" int lc_multiline_counters(const lc_multiline_t* m, uint64_t counters[3]) {
    if (!m || !counters) return LC_ERR_ARG;
    counters[0] = m->matchedLinesTotal;
    counters[1] = m->unmatchedLinesTotal;
    counters[2] = m->matchedEventsTotal;
    return LC_OK;
}

// ------------------------------------------------------------------------------------------------ processor_merge_multiline_log_native
struct lc_merge_multiline {
    std::string sourceKey = "content";
    bool byFlag = false;
    lc_multiline_t* ml = nullptr;  // MergeType "regex": the Multiline options (patterns, UnmatchedContentTreatment)
    bool ignoringUnmatchWarning = false;
    std::atomic<uint64_t> mergedEventsTotal{0}, unmatchedEventsTotal{0};
    ~lc_merge_multiline() { lc_multiline_free(ml); }
};

extern "C" int lc_merge_multiline_create(const char* config_json, size_t config_len, lc_merge_multiline_t** out, char* err,
                                         size_t errcap) {
    if (!config_json || !out) return LC_ERR_ARG;
    *out = nullptr;
    auto set = [&](const std::string& s) {
        if (err && errcap) std::snprintf(err, errcap, "%s", s.c_str());
    };
    auto p = std::make_unique<lc_merge_multiline>();
    try {
        const lcjson::Value cfg = lcjson::parse(std::string(config_json, config_len));
        if (!cfg.isObject()) throw std::runtime_error("config must be a JSON object");
        if (const lcjson::Value* v = cfg.find("SourceKey"))
            if (v->isString()) p->sourceKey = v->str;  // (a wrong type is a warning + the default, :36-47)
        const lcjson::Value* mt = cfg.find("MergeType");  // mandatory, :49-76
        if (!mt) throw std::runtime_error("mandatory param MergeType is missing");
        if (!mt->isString()) throw std::runtime_error("mandatory string param MergeType is not of type string");
        if (mt->str == "flag") p->byFlag = true;
        else if (mt->str == "regex") {
            char buf[512];
            const int rc = lc_multiline_create(config_json, config_len, &p->ml, buf, sizeof buf);
            if (rc != LC_OK) throw std::runtime_error(buf);
            if (const lcjson::Value* v = cfg.find("IgnoringUnmatchWarning"))
                if (v->isBool()) p->ignoringUnmatchWarning = v->b;
        } else throw std::runtime_error("string param MergeType is not valid");
    } catch (const std::exception& e) {
        set(e.what());
        return LC_ERR_SYNTAX;
    }
    set("");
    *out = p.release();
    return LC_OK;
}
extern "C" void lc_merge_multiline_free(lc_merge_multiline_t* p) { delete p; }
extern "C" int lc_merge_multiline_counters(const lc_merge_multiline_t* p, uint64_t counters[2]) {
    if (!p || !counters) return LC_ERR_ARG;
    counters[0] = p->mergedEventsTotal;
    counters[1] = p->unmatchedEventsTotal;
    return LC_OK;
}

namespace {

const char kPartLogFlag[] = "P";  // ProcessorMergeMultilineLogNative::PartLogFlag :31

// MergeEvents :332-358: the first event's value is extended IN PLACE over the following ones (the events of one read buffer
// lie back to back in the group's source buffer, each followed by the byte its line feed occupied)
void mergeEvents(lc_merge_multiline& p, std::vector<LogEvent*>& logEvents, bool insertLineBreak) {
    if (logEvents.empty()) return;
    p.mergedEventsTotal += logEvents.size();
    if (logEvents.size() == 1) {
        logEvents.clear();
        return;
    }
    const StringView key(p.sourceKey.data(), p.sourceKey.size());
    LogEvent* target = logEvents[0];
    const StringView targetValue = target->GetContent(key);
    char* begin = const_cast<char*>(targetValue.data());
    char* end = begin + targetValue.size();
    for (size_t i = 1; i < logEvents.size(); ++i) {
        if (insertLineBreak) *end++ = '\n';
        const StringView cur = logEvents[i]->GetContent(key);
        std::memmove(end, cur.data(), cur.size());
        end += cur.size();
    }
    target->SetContentNoCopy(key, StringView(begin, size_t(end - begin)));
    logEvents.clear();
}

// HandleUnmatchLogs :360-392 (without the alarms)
void handleUnmatch(lc_merge_multiline& p, std::vector<PipelineEventPtr>& logEvents, size_t& newSize, size_t begin, size_t end) {
    p.unmatchedEventsTotal += end - begin + 1;
    if (p.ml->discardUnmatched) return;
    for (size_t i = begin; i <= end; ++i) logEvents[newSize++] = std::move(logEvents[i]);
}

void mergeLogsByFlag(lc_merge_multiline& p, PipelineEventGroup& logGroup) {  // :113-159
    auto& sourceEvents = logGroup.MutableEvents();
    size_t size = 0;
    std::vector<LogEvent*> events;
    bool isPartialLog = false;
    size_t begin = 0;
    const StringView flag(kPartLogFlag, 1);
    for (size_t cur = 0; cur < sourceEvents.size(); ++cur) {
        if (!sourceEvents[cur].Is<LogEvent>()) {
            if (events.empty()) begin = cur;
            for (size_t i = begin; i < sourceEvents.size(); ++i) sourceEvents[size++] = std::move(sourceEvents[i]);
            sourceEvents.resize(size);
            return;
        }
        LogEvent* sourceEvent = &sourceEvents[cur].Cast<LogEvent>();
        if (sourceEvent->Empty()) continue;
        events.emplace_back(sourceEvent);
        if (isPartialLog) {
            if (!sourceEvent->HasContent(flag)) {  // p p p ... p(last) notP(cur)
                mergeEvents(p, events, false);
                sourceEvents[size++] = std::move(sourceEvents[begin]);
                begin = cur + 1;
                isPartialLog = false;
            }
        } else if (sourceEvent->HasContent(flag)) {
            sourceEvent->DelContent(flag);
            isPartialLog = true;
        } else {
            mergeEvents(p, events, false);
            sourceEvents[size++] = std::move(sourceEvents[begin]);
            begin = cur + 1;
        }
    }
    if (isPartialLog) {
        mergeEvents(p, events, false);
        sourceEvents[size++] = std::move(sourceEvents[begin]);
    }
    sourceEvents.resize(size);
}

int mergeLogsByRegex(lc_merge_multiline& p, PipelineEventGroup& logGroup) {  // :161-330
    auto& sourceEvents = logGroup.MutableEvents();
    const lc_multiline& ml = *p.ml;
    const bool hasStart = ml.start, hasCont = ml.cont, hasEnd = ml.end;
    const StringView key(p.sourceKey.data(), p.sourceKey.size());
    // the device pass: the flags of every event the loop below can reach (it stops for good at the first event that is not a
    // log event or lacks the source key)
    std::vector<const uint8_t*> ptrs;
    std::vector<uint32_t> lens;
    std::vector<int32_t> flagIndex(sourceEvents.size(), -1);
    for (size_t i = 0; i < sourceEvents.size(); ++i) {
        if (!sourceEvents[i].Is<LogEvent>()) break;
        const LogEvent& ev = sourceEvents[i].Cast<LogEvent>();
        if (ev.Empty()) continue;
        if (!ev.HasContent(key)) break;
        const StringView v = ev.GetContent(key);
        flagIndex[i] = int32_t(ptrs.size());
        ptrs.push_back(reinterpret_cast<const uint8_t*>(v.data()));
        lens.push_back(uint32_t(v.size()));
    }
    const uint32_t n = uint32_t(ptrs.size());
    std::vector<uint8_t> fStart(n, 0), fCont(n, 0), fEnd(n, 0);
    auto flags = [&](lc_regex_t* re, std::vector<uint8_t>& dst) -> int {
        if (!re || n == 0) return LC_OK;
        const int r = lc_regex_match_host_views(re, ptrs.data(), lens.data(), n, 0, nullptr, dst.data());
        if (r != LC_OK) return r;
        uint64_t gaveUp = 0;
        for (uint8_t st : dst) {
            if (st == LC_OVERFLOW) return LC_ERR_UNSUPPORTED;
            gaveUp += st == LC_GAVE_UP;  // BoostRegexSearch failed with an exception (StringTools.cpp:277-282): false, and counted
        }
        if (gaveUp) lcNoteGaveUp(gaveUp);  // "not decided" must not drive the state machine as "no match"
        return LC_OK;
    };
    int rc;
    if ((rc = flags(ml.start, fStart)) != LC_OK || (rc = flags(ml.cont, fCont)) != LC_OK || (rc = flags(ml.end, fEnd)) != LC_OK)
        return rc;
    auto isStart = [&](size_t cur) { return fStart[size_t(flagIndex[cur])] == LC_MATCH; };
    auto isCont = [&](size_t cur) { return fCont[size_t(flagIndex[cur])] == LC_MATCH; };
    auto isEnd = [&](size_t cur) { return fEnd[size_t(flagIndex[cur])] == LC_MATCH; };

    size_t begin = 0, newSize = 0;
    std::vector<LogEvent*> events;
    bool isPartialLog = false;
    if (!hasStart && !hasCont && hasEnd) isPartialLog = true;  // only an end pattern: it sticks to this state (:174-178)
    for (size_t cur = 0; cur < sourceEvents.size(); ++cur) {
        if (!sourceEvents[cur].Is<LogEvent>()) {  // :180-188
            if (events.empty()) begin = cur;
            for (size_t i = begin; i < sourceEvents.size(); ++i) sourceEvents[newSize++] = std::move(sourceEvents[i]);
            sourceEvents.resize(newSize);
            return LC_OK;
        }
        LogEvent* sourceEvent = &sourceEvents[cur].Cast<LogEvent>();
        if (sourceEvent->Empty()) continue;
        if (!sourceEvent->HasContent(key)) {  // :193-216
            if (events.empty()) begin = cur;
            for (size_t i = begin; i < sourceEvents.size(); ++i) sourceEvents[newSize++] = std::move(sourceEvents[i]);
            sourceEvents.resize(newSize);
            return LC_OK;
        }
        if (!isPartialLog) {
            if (hasStart ? isStart(cur) : isCont(cur)) {  // :219-230
                events.emplace_back(sourceEvent);
                begin = cur;
                isPartialLog = true;
            } else if (hasEnd && !hasStart && hasCont && isEnd(cur)) {  // continue + end: matched against the end pattern (:231-239)
                begin = cur;
                p.mergedEventsTotal += 1;
                sourceEvents[newSize++] = std::move(sourceEvents[begin]);
            } else {
                handleUnmatch(p, sourceEvents, newSize, cur, cur);
            }
        } else {
            if (hasCont && isCont(cur)) {  // :244-249
                events.emplace_back(sourceEvent);
                continue;
            }
            if (hasEnd) {
                events.emplace_back(sourceEvent);  // start + end, continue + end, or end (:250-252)
                if (hasCont) {
                    if (isEnd(cur)) {
                        mergeEvents(p, events, true);
                        sourceEvents[newSize++] = std::move(sourceEvents[begin]);
                    } else {
                        handleUnmatch(p, sourceEvents, newSize, begin, cur);
                        events.clear();
                    }
                    isPartialLog = false;
                } else if (isEnd(cur)) {  // :266-280
                    mergeEvents(p, events, true);
                    sourceEvents[newSize++] = std::move(sourceEvents[begin]);
                    if (hasStart) isPartialLog = false;
                    else begin = cur + 1;  // only an end pattern: the next log starts by itself
                }
            } else if (!hasCont) {  // start only (:283-294)
                if (!isStart(cur)) {
                    events.emplace_back(sourceEvent);
                } else {
                    mergeEvents(p, events, true);
                    sourceEvents[newSize++] = std::move(sourceEvents[begin]);
                    begin = cur;
                    events.emplace_back(sourceEvent);
                }
            } else {  // start + continue, and the line is no continuation (:295-311)
                mergeEvents(p, events, true);
                sourceEvents[newSize++] = std::move(sourceEvents[begin]);
                if (!isStart(cur)) {
                    handleUnmatch(p, sourceEvents, newSize, cur, cur);
                    isPartialLog = false;
                } else {
                    begin = cur;
                    events.emplace_back(sourceEvent);
                }
            }
        }
    }
    if (isPartialLog && begin < sourceEvents.size()) {  // :316-323
        if (!hasEnd) {
            mergeEvents(p, events, true);
            sourceEvents[newSize++] = std::move(sourceEvents[begin]);
        } else {
            handleUnmatch(p, sourceEvents, newSize, begin, sourceEvents.size() - 1);
        }
    }
    sourceEvents.resize(newSize);
    return LC_OK;
}

}  // namespace

extern "C" int lc_merge_multiline_process_group(lc_merge_multiline_t* p, void* pipelineEventGroup) {
    if (!p || !pipelineEventGroup) return LC_ERR_ARG;
    PipelineEventGroup& logGroup = *static_cast<PipelineEventGroup*>(pipelineEventGroup);
    if (logGroup.GetEvents().empty()) return LC_OK;  // :81-83
    if (!p->byFlag) return mergeLogsByRegex(*p, logGroup);
    if (logGroup.HasMetadata(EventGroupMetaKey::HAS_PART_LOG)) {  // :86-90
        mergeLogsByFlag(*p, logGroup);
        logGroup.DelMetadata(EventGroupMetaKey::HAS_PART_LOG);
    }
    return LC_OK;
}
