// event_model.cpp -- JSON fixture I/O for the stand-in event model (the format the reference's unit tests use:
// PipelineEventGroup::FromJsonString / ToJsonString, core/models/LogEvent.cpp:169-209).
#include "event_model.hpp"

#include <algorithm>

#include "json_min.hpp"

namespace logtail {

namespace {
const std::pair<const char*, EventGroupMetaKey> kMetaNames[] = {
    {"log.file.path", EventGroupMetaKey::LOG_FILE_PATH},
    {"log.file.path_resolved", EventGroupMetaKey::LOG_FILE_PATH_RESOLVED},
    {"log.file.inode", EventGroupMetaKey::LOG_FILE_INODE},
    {"log.file.offset", EventGroupMetaKey::LOG_FILE_OFFSET_KEY},
    {"has.part.log", EventGroupMetaKey::HAS_PART_LOG},
    {"source.id", EventGroupMetaKey::SOURCE_ID},
};
}

bool PipelineEventGroup::FromJsonString(const std::string& json, std::string* error) {
    lcjson::Value root;
    try {
        root = lcjson::parse(json);
    } catch (const std::exception& e) {
        if (error) *error = e.what();
        return false;
    }
    if (const lcjson::Value* md = root.find("metadata")) {
        for (const auto& kv : md->obj)
            for (const auto& name : kMetaNames)
                if (kv.first == name.first) SetMetadata(name.second, kv.second.str);
    }
    if (const lcjson::Value* tags = root.find("tags"))
        for (const auto& kv : tags->obj) SetTag(kv.first, kv.second.str);
    const lcjson::Value* events = root.find("events");
    if (!events) return true;
    for (const lcjson::Value& ev : events->arr) {
        const lcjson::Value* type = ev.find("type");
        const int t = type ? int(type->inum) : 1;
        PipelineEvent* base = nullptr;
        if (t == int(PipelineEvent::Type::LOG)) {
            LogEvent* le = AddLogEvent();
            base = le;
            if (const lcjson::Value* contents = ev.find("contents")) {
                if (contents->isObject()) {
                    // jsoncpp iterates object members in key order; the reference fixture loader inherits that
                    std::vector<const std::pair<std::string, lcjson::Value>*> members;
                    for (const auto& kv : contents->obj) members.push_back(&kv);
                    std::stable_sort(members.begin(), members.end(),
                                     [](const auto* a, const auto* b) { return a->first < b->first; });
                    for (const auto* kv : members) le->SetContent(kv->first, kv->second.str);
                } else {  // extension: [[key, value], ...] keeps the given order
                    for (const auto& pair : contents->arr)
                        if (pair.arr.size() == 2) le->SetContent(pair.arr[0].str, pair.arr[1].str);
                }
            }
            const lcjson::Value* fo = ev.find("fileOffset");
            const lcjson::Value* rs = ev.find("rawSize");
            if (fo && rs) le->SetPosition(uint64_t(fo->inum), uint64_t(rs->inum));
        } else {
            RawEvent* re = AddRawEvent();
            base = re;
            if (const lcjson::Value* c = ev.find("content")) {
                StringBuffer b = mSourceBuffer->CopyString(c->str);
                re->SetContentNoCopy(StringView(b.data, b.size));
            }
        }
        const lcjson::Value* ts = ev.find("timestamp");
        const lcjson::Value* ns = ev.find("timestampNanosecond");
        if (ts && ns) base->SetTimestamp(time_t(ts->inum), uint32_t(ns->inum));
        else if (ts) base->SetTimestamp(time_t(ts->inum));
    }
    return true;
}

std::string PipelineEventGroup::ToJsonString() const {
    lcjson::Value root = lcjson::Value::makeObject();
    if (!mMetadata.empty()) {
        lcjson::Value md = lcjson::Value::makeObject();
        for (const auto& kv : mMetadata)
            for (const auto& name : kMetaNames)
                if (kv.first == name.second) md.set(name.first, lcjson::Value::makeString(kv.second.to_string()));
        root.set("metadata", std::move(md));
    }
    if (!mTags.empty()) {
        lcjson::Value tags = lcjson::Value::makeObject();
        for (const auto& kv : mTags) tags.set(kv.first.to_string(), lcjson::Value::makeString(kv.second.to_string()));
        root.set("tags", std::move(tags));
    }
    if (!mEvents.empty()) {
        lcjson::Value events = lcjson::Value::makeArray();
        for (const auto& e : mEvents) {
            lcjson::Value ev = lcjson::Value::makeObject();
            if (e.Is<LogEvent>()) {
                const LogEvent& le = e.Cast<LogEvent>();
                if (!le.Empty()) {
                    lcjson::Value contents = lcjson::Value::makeObject();  // live contents, in list order
                    for (auto it = le.cbegin(); it != le.cend(); ++it)
                        contents.obj.emplace_back(it->first.to_string(), lcjson::Value::makeString(it->second.to_string()));
                    ev.set("contents", std::move(contents));
                }
                if (le.GetPosition().second) {  // (LogEvent::ToJson prints the position with enableEventMeta; here: whenever one is set)
                    ev.set("fileOffset", lcjson::Value::makeInt(int64_t(le.GetPosition().first)));
                    ev.set("rawSize", lcjson::Value::makeInt(int64_t(le.GetPosition().second)));
                }
            } else {
                ev.set("content", lcjson::Value::makeString(e.Cast<RawEvent>().GetContent().to_string()));
            }
            ev.set("timestamp", lcjson::Value::makeInt(int64_t(e->GetTimestamp())));
            if (e->GetTimestampNanosecond())
                ev.set("timestampNanosecond", lcjson::Value::makeInt(int64_t(*e->GetTimestampNanosecond())));
            ev.set("type", lcjson::Value::makeInt(int64_t(e->GetType())));
            events.arr.push_back(std::move(ev));
        }
        root.set("events", std::move(events));
    }
    return lcjson::dump(root);
}

}  // namespace logtail
