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
#include "multiline_scan.hpp"
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
            // the merge processor matches with MultilineOptions' own regexes, not with the strings as written (:219-224)
            const int rc = lcMultilineCreateForMerge(config_json, config_len, &p->ml, buf, sizeof buf);
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
extern "C" int lc_merge_multiline_patterns(const lc_merge_multiline_t* p) { return p && p->ml ? lc_multiline_patterns(p->ml) : 0; }
extern "C" const char* lc_merge_multiline_warnings(const lc_merge_multiline_t* p) { return p && p->ml ? lc_multiline_warnings(p->ml) : ""; }
extern "C" int lc_merge_multiline_counters(const lc_merge_multiline_t* p, uint64_t counters[2]) {
    if (!p || !counters) return LC_ERR_ARG;
    counters[0] = p->mergedEventsTotal;
    counters[1] = p->unmatchedEventsTotal;
    return LC_OK;
}

namespace {

const char kPartLogFlag[] = "P";  // ProcessorMergeMultilineLogNative::PartLogFlag :31

// MergeEvents :332-358: the first event's value is extended IN PLACE over the following ones (the events of one read buffer
// lie back to back in the group's source buffer, each followed by the byte its line feed occupied).  That is the layout the line
// splitter leaves, and the reference relies on it: its memmove runs over whatever lies between two values.  Here the in-place join is
// taken when the values DO lie that way -- value, one byte, value -- and the same bytes are put together in a fresh block of the
// group's source buffer when they do not (values a fixture reader or another processor copied one by one: between them lie keys,
// and, in the stand-in event model, the events' own contents arrays).  The merged value is the same either way.
void mergeEvents(lc_merge_multiline& p, PipelineEventGroup& logGroup, std::vector<LogEvent*>& logEvents, bool insertLineBreak) {
    if (logEvents.empty()) return;
    p.mergedEventsTotal += logEvents.size();
    if (logEvents.size() == 1) {
        logEvents.clear();
        return;
    }
    const StringView key(p.sourceKey.data(), p.sourceKey.size());
    LogEvent* target = logEvents[0];
    const StringView targetValue = target->GetContent(key);
    // In place only when the first value has bytes of its own to grow from: an empty value may carry a null pointer (`nullptr + 1`
    // below, memcpy from null: undefined whatever the length), and "one byte behind an empty value" says nothing about where the next
    // value lies.  (The in-place branch writes only inside [first value's begin, last value's end): the bytes of the values and the
    // one-byte gaps between neighbours, which cur.data() == prevEnd + 1 has just shown to exist in the same block.)
    bool backToBack = targetValue.data() != nullptr && targetValue.size() != 0;
    size_t total = targetValue.size();
    const char* prevEnd = targetValue.data() + targetValue.size();
    for (size_t i = 1; i < logEvents.size(); ++i) {
        const StringView cur = logEvents[i]->GetContent(key);
        backToBack = backToBack && cur.data() == prevEnd + 1;
        prevEnd = cur.data() + cur.size();
        total += cur.size() + (insertLineBreak ? 1 : 0);
    }
    char* begin = const_cast<char*>(targetValue.data());
    if (!backToBack) {
        const StringBuffer block = logGroup.GetSourceBuffer()->AllocateStringBuffer(total);
        begin = block.data;
        if (targetValue.size()) std::memcpy(begin, targetValue.data(), targetValue.size());
    }
    char* end = begin + targetValue.size();
    for (size_t i = 1; i < logEvents.size(); ++i) {
        if (insertLineBreak) *end++ = '\n';
        const StringView cur = logEvents[i]->GetContent(key);
        if (cur.size()) std::memmove(end, cur.data(), cur.size());
        end += cur.size();
    }
    target->SetContentNoCopy(key, StringView(begin, size_t(end - begin)));
    logEvents.clear();
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
                mergeEvents(p, logGroup, events, false);
                sourceEvents[size++] = std::move(sourceEvents[begin]);
                begin = cur + 1;
                isPartialLog = false;
            }
        } else if (sourceEvent->HasContent(flag)) {
            sourceEvent->DelContent(flag);
            isPartialLog = true;
        } else {
            mergeEvents(p, logGroup, events, false);
            sourceEvents[size++] = std::move(sourceEvents[begin]);
            begin = cur + 1;
        }
    }
    if (isPartialLog) {
        mergeEvents(p, logGroup, events, false);
        sourceEvents[size++] = std::move(sourceEvents[begin]);
    }
    sourceEvents.resize(size);
}

int mergeLogsByRegex(lc_merge_multiline& p, PipelineEventGroup& logGroup) {  // :161-330
    auto& sourceEvents = logGroup.MutableEvents();
    lc_multiline& ml = *p.ml;
    const StringView key(p.sourceKey.data(), p.sourceKey.size());
    // the ITEMS of the walk: the non-empty log events that carry the source key, up to the first event the reference's loop stops
    // at for good (not a log event :180-188, or without the key :193-216)
    std::vector<const uint8_t*> ptrs;
    std::vector<uint32_t> lens, itemEvent;
    size_t stopAt = sourceEvents.size();
    for (size_t i = 0; i < sourceEvents.size(); ++i) {
        if (!sourceEvents[i].Is<LogEvent>()) {
            stopAt = i;
            break;
        }
        const LogEvent& ev = sourceEvents[i].Cast<LogEvent>();
        if (ev.Empty()) continue;
        if (!ev.HasContent(key)) {
            stopAt = i;
            break;
        }
        const StringView v = ev.GetContent(key);
        itemEvent.push_back(uint32_t(i));
        ptrs.push_back(reinterpret_cast<const uint8_t*>(v.data()));
        lens.push_back(uint32_t(v.size()));
    }
    const bool truncated = stopAt < sourceEvents.size();
    const uint32_t n = uint32_t(ptrs.size());
    // ONE device trip: the values go up once, a status-only launch per pattern, the walk as a scan over the flags
    // (multiline_scan.hpp); records = item ranges, in the order the reference emits them.  UnmatchedContentTreatment is applied
    // here: HandleUnmatchLogs counts EVENTS (:360-392), the empty ones between two items included, and those are no items.
    std::vector<lc_ml_record_t> recs;
    uint32_t counts[ML_CNT_WORDS];
    const bool discard = ml.discardUnmatched;
    const int rc = lcMultilineViewsTrip(&ml, ptrs.data(), lens.data(), n, !truncated, /*keepUnmatched=*/true, recs, counts);
    if (rc != LC_OK) return rc;

    size_t newSize = 0, prevEvent = 0;
    // Only an end pattern: the walk never leaves the partial state and a closed log restarts `begin` at cur + 1 (:283), which is the
    // next EVENT, not the next item -- when that event has no contents (:190 skips it without touching `begin`), the log that follows
    // is merged into its first item but the event moved to the output is sourceEvents[begin], the one without contents (:279).  Such
    // events do not come out of the line splitter; the reference's behaviour is kept as it is.
    const bool endOnly = !ml.start && !ml.cont && ml.end;
    size_t endOnlyBegin = 0;
    std::vector<LogEvent*> events;
    auto handleUnmatch = [&](size_t b, size_t e) {  // HandleUnmatchLogs :360-392 (without the alarms)
        p.unmatchedEventsTotal += e - b + 1;
        if (!discard)
            for (size_t i = b; i <= e; ++i) sourceEvents[newSize++] = std::move(sourceEvents[i]);
    };
    for (size_t r = 0; r < recs.size(); ++r) {
        const uint32_t first = recs[r].begin, cnt = recs[r].length;
        if (recs[r].matched & 1u) {  // MergeEvents :332-358 + the move of the log's first event
            events.clear();
            for (uint32_t k = first; k < first + cnt; ++k) events.push_back(&sourceEvents[itemEvent[k]].Cast<LogEvent>());
            mergeEvents(p, logGroup, events, true);
            sourceEvents[newSize++] = std::move(sourceEvents[endOnly ? endOnlyBegin : size_t(itemEvent[first])]);
            endOnlyBegin = size_t(itemEvent[first + cnt - 1]) + 1;
            continue;
        }
        size_t b = itemEvent[first], e = b;
        if (recs[r].matched & LC_ML_RUN) b = prevEvent + 1;  // one [begin, cur] call: the empty events in between go with it
        else if (endOnly) b = endOnlyBegin;                  // (the flush runs from `begin`, :321)
        if ((recs[r].matched & LC_ML_LAST) && r + 1 == recs.size()) e = sourceEvents.size() - 1;  // the flush runs to the group's end (:321)
        handleUnmatch(b, e);
        prevEvent = e;
    }
    if (truncated) {  // the rest passes through, from the log under construction on (`if (events.empty()) begin = cur`)
        const bool open = counts[ML_CNT_FINAL_PARTIAL] && counts[ML_CNT_FINAL_START] < n;
        for (size_t i = open ? (endOnly ? endOnlyBegin : size_t(itemEvent[counts[ML_CNT_FINAL_START]])) : stopAt; i < sourceEvents.size(); ++i)
            sourceEvents[newSize++] = std::move(sourceEvents[i]);
    } else if (counts[ML_CNT_FINAL_PARTIAL] && counts[ML_CNT_FINAL_START] >= n) {
        // only an end pattern, and the last item closed a log (begin = cur + 1): events behind it -- empty ones -- are what
        // `begin < sourceEvents.size()` (:316) still hands to HandleUnmatchLogs
        const size_t b = n ? size_t(itemEvent[n - 1]) + 1 : 0;
        if (b < sourceEvents.size()) handleUnmatch(b, sourceEvents.size() - 1);
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
