// tests/native/ref_group_io.cpp -- TEST INFRASTRUCTURE: fixture JSON <-> the REFERENCE's PipelineEventGroup, through its public interface
// only (core/models/PipelineEventGroup.h:80-158, LogEvent.h:64-132, RawEvent.h).  The reference's own FromJsonString / ToJsonString
// exist only under APSARA_UNIT_TEST_MAIN and need jsoncpp (core/models/LogEvent.cpp:169-209); this file reads and writes the same
// fixture format with the repo's small JSON reader, so that tests/native/host_double.cpp can drive the product's processor against
// oracle/_ref/libref_models.so.  Same conventions as csrc/event_model.cpp: object contents are applied in key order (jsoncpp's
// iteration order), the [[key, value], ...] form keeps the given order; live contents are written in list order.
#include <algorithm>
#include <string>
#include <vector>

#include "../../loongcollector_amd/csrc/json_min.hpp"
#include "models/LogEvent.h"
#include "models/PipelineEventGroup.h"
#include "models/RawEvent.h"

namespace {
using logtail::EventGroupMetaKey;
const std::pair<const char*, EventGroupMetaKey> kMetaNames[] = {
    {"log.file.path_resolved", EventGroupMetaKey::LOG_FILE_PATH_RESOLVED},
    {"log.file.offset", EventGroupMetaKey::LOG_FILE_OFFSET_KEY},
    {"has.part.log", EventGroupMetaKey::HAS_PART_LOG},
    {"source.id", EventGroupMetaKey::SOURCE_ID},
};
}  // namespace

bool hdGroupFromJson(logtail::PipelineEventGroup& group, const std::string& json, std::string* error) {
    using namespace logtail;
    lcjson::Value root;
    try {
        root = lcjson::parse(json);
    } catch (const std::exception& e) {
        if (error) *error = e.what();
        return false;
    }
    if (const lcjson::Value* md = root.find("metadata"))
        for (const auto& kv : md->obj)
            for (const auto& name : kMetaNames)
                if (kv.first == name.first) group.SetMetadata(name.second, kv.second.str);
    if (const lcjson::Value* tags = root.find("tags"))
        for (const auto& kv : tags->obj) group.SetTag(kv.first, kv.second.str);
    const lcjson::Value* events = root.find("events");
    if (!events) return true;
    for (const lcjson::Value& ev : events->arr) {
        const lcjson::Value* type = ev.find("type");
        const int t = type ? int(type->inum) : 1;
        PipelineEvent* base = nullptr;
        if (t == int(PipelineEvent::Type::LOG)) {
            LogEvent* le = group.AddLogEvent();
            base = le;
            if (const lcjson::Value* contents = ev.find("contents")) {
                if (contents->isObject()) {
                    std::vector<const std::pair<std::string, lcjson::Value>*> members;
                    for (const auto& kv : contents->obj) members.push_back(&kv);
                    std::stable_sort(members.begin(), members.end(), [](const auto* a, const auto* b) { return a->first < b->first; });
                    for (const auto* kv : members) le->SetContent(kv->first, kv->second.str);
                } else {
                    for (const auto& pair : contents->arr)
                        if (pair.arr.size() == 2) le->SetContent(pair.arr[0].str, pair.arr[1].str);
                }
            }
            const lcjson::Value* fo = ev.find("fileOffset");
            const lcjson::Value* rs = ev.find("rawSize");
            if (fo && rs) le->SetPosition(uint64_t(fo->inum), uint64_t(rs->inum));
        } else {
            RawEvent* re = group.AddRawEvent();
            base = re;
            if (const lcjson::Value* c = ev.find("content")) re->SetContent(c->str);
        }
        const lcjson::Value* ts = ev.find("timestamp");
        const lcjson::Value* ns = ev.find("timestampNanosecond");
        if (ts && ns) base->SetTimestamp(time_t(ts->inum), uint32_t(ns->inum));
        else if (ts) base->SetTimestamp(time_t(ts->inum));
    }
    return true;
}

std::string hdGroupToJson(const logtail::PipelineEventGroup& group) {
    using namespace logtail;
    lcjson::Value root = lcjson::Value::makeObject();
    if (!group.GetAllMetadata().empty()) {
        lcjson::Value md = lcjson::Value::makeObject();
        for (const auto& kv : group.GetAllMetadata())
            for (const auto& name : kMetaNames)
                if (kv.first == name.second) md.set(name.first, lcjson::Value::makeString(kv.second.to_string()));
        root.set("metadata", std::move(md));
    }
    if (!group.GetTags().empty()) {
        lcjson::Value tags = lcjson::Value::makeObject();
        for (const auto& kv : group.GetTags()) tags.set(kv.first.to_string(), lcjson::Value::makeString(kv.second.to_string()));
        root.set("tags", std::move(tags));
    }
    if (!group.GetEvents().empty()) {
        lcjson::Value events = lcjson::Value::makeArray();
        for (const auto& e : group.GetEvents()) {
            lcjson::Value ev = lcjson::Value::makeObject();
            if (e.Is<LogEvent>()) {
                const LogEvent& le = e.Cast<LogEvent>();
                if (!le.Empty()) {
                    lcjson::Value contents = lcjson::Value::makeObject();
                    for (auto it = le.cbegin(); it != le.cend(); ++it)
                        contents.obj.emplace_back(it->first.to_string(), lcjson::Value::makeString(it->second.to_string()));
                    ev.set("contents", std::move(contents));
                }
                if (le.GetPosition().second) {
                    ev.set("fileOffset", lcjson::Value::makeInt(int64_t(le.GetPosition().first)));
                    ev.set("rawSize", lcjson::Value::makeInt(int64_t(le.GetPosition().second)));
                }
            } else if (e.Is<RawEvent>()) {
                ev.set("content", lcjson::Value::makeString(e.Cast<RawEvent>().GetContent().to_string()));
            }
            ev.set("timestamp", lcjson::Value::makeInt(int64_t(e->GetTimestamp())));
            if (e->GetTimestampNanosecond()) ev.set("timestampNanosecond", lcjson::Value::makeInt(int64_t(*e->GetTimestampNanosecond())));
            ev.set("type", lcjson::Value::makeInt(int64_t(e->GetType())));
            events.arr.push_back(std::move(ev));
        }
        root.set("events", std::move(events));
    }
    return lcjson::dump(root);
}

#ifndef LC_REF_GROUP_IO_ONLY  // (oracle/ref_processor builds this file for the fixture reader / writer alone)
// ---- the parser plugins' source-key / erase policy: the REFERENCE's own CommonParserOptions (plugin/processor/CommonParserOptions.cpp,
// compiled into oracle/_ref/libref_models.so) against the product's restatement (csrc/processor_parse_regex_gpu.cpp GpuCommonParserOptions),
// on the same events of the reference's event model.  -> number of (option set, outcome, event shape, function) cases compared;
// *mismatches = how many differed, `first` (optional) describes the first one.
#include "../../loongcollector_amd/csrc/processor_parse_regex_gpu.hpp"
#include "plugin/processor/CommonParserOptions.h"
#include "plugin/processor/inner/ProcessorParseContainerLogNative.h"

extern "C" int hd_policy_matrix_vs_reference(int* mismatches, char* first, size_t firstCap) {
    using namespace logtail;
    int cases = 0, bad = 0;
    if (first && firstCap) first[0] = 0;
    const std::string offsetKeyName = "__file_offset__";
    const std::string& tKey = ProcessorParseContainerLogNative::containerTimeKey;
    const std::string& sKey = ProcessorParseContainerLogNative::containerSourceKey;
    const std::vector<std::vector<std::pair<std::string, std::string>>> shapes = {
        {},
        {{offsetKeyName, "123"}},
        {{tKey, "2024-01-01T00:00:00Z"}, {sKey, "stdout"}},
        {{sKey, "stderr"}, {tKey, "t"}},
        {{"other", "x"}},
        {{tKey, "t"}, {"other", "x"}},
        {{offsetKeyName, "1"}, {"other", "x"}},
        {{tKey, "t"}, {sKey, "stdout"}, {"third", "3"}},
        {{"a", "1"}, {"b", "2"}},
        {{"__file_offset__x", "1"}},
    };
    for (int bits = 0; bits < 16; ++bits) {
        const bool keepFail = bits & 1, keepSucceed = bits & 2, copingRaw = bits & 4, success = bits & 8;
        CommonParserOptions ref;
        GpuCommonParserOptions mine;
        ref.mKeepingSourceWhenParseFail = mine.mKeepingSourceWhenParseFail = keepFail;
        ref.mKeepingSourceWhenParseSucceed = mine.mKeepingSourceWhenParseSucceed = keepSucceed;
        ref.mCopingRawLog = mine.mCopingRawLog = copingRaw;
        for (int withMeta = 0; withMeta < 2; ++withMeta)
            for (size_t sh = 0; sh < shapes.size(); ++sh) {
                PipelineEventGroup group(std::make_shared<SourceBuffer>());
                if (withMeta) group.SetMetadata(EventGroupMetaKey::LOG_FILE_OFFSET_KEY, offsetKeyName);
                LogEvent* ev = group.AddLogEvent();
                for (const auto& kv : shapes[sh]) ev->SetContent(kv.first, kv.second);
                const bool r[3] = {ref.ShouldAddSourceContent(success), ref.ShouldAddLegacyUnmatchedRawLog(success),
                                   ref.ShouldEraseEvent(success, *ev, group.GetAllMetadata())};
                const bool m[3] = {mine.ShouldAddSourceContent(success), mine.ShouldAddLegacyUnmatchedRawLog(success),
                                   mine.ShouldEraseEvent(success, *ev, group.GetAllMetadata())};
                for (int f = 0; f < 3; ++f) {
                    ++cases;
                    if (r[f] != m[f]) {
                        if (!bad && first && firstCap)
                            snprintf(first, firstCap, "options %d metadata %d shape %zu function %d: reference %d, product %d", bits, withMeta, sh, f,
                                     int(r[f]), int(m[f]));
                        ++bad;
                    }
                }
            }
    }
    // the legacy key's name is the same constant
    ++cases;
    if (CommonParserOptions::legacyUnmatchedRawLogKey != GpuCommonParserOptions::legacyUnmatchedRawLogKey) {
        if (!bad && first && firstCap)
            snprintf(first, firstCap, "legacyUnmatchedRawLogKey differs: reference '%s', product '%s'", CommonParserOptions::legacyUnmatchedRawLogKey.c_str(),
                     GpuCommonParserOptions::legacyUnmatchedRawLogKey.c_str());
        ++bad;
    }
    if (mismatches) *mismatches = bad;
    return cases;
}
#endif  // LC_REF_GROUP_IO_ONLY
