// oracle/ref_models/link_stubs.cpp -- TEST INFRASTRUCTURE.  What the reference's core/models/*.cpp reference at link time from parts
// of the agent that are not compiled here (logger, jsoncpp, protobuf, trace-tag constants, the hash helper).  None of it is on
// the path the tests exercise (LogEvent / PipelineEventGroup / SourceBuffer): the logger never logs, the JSON value is inert.
#include <cstddef>
#include <string>

#include "json/json.h"
#include "protobuf/sls/checkpoint.pb.h"
#include "spdlog/spdlog.h"

namespace Json {
const Value& Value::operator[](const char*) const { return *this; }   // (CommonParserOptions::Init reads config["SourceKey"]: compiled, never called)
std::string Value::asString() const { return std::string(); }
Value::Value(ValueType) {}
Value::Value(Int64) {}
Value::Value(const std::string&) {}
Value& Value::operator=(const Value&) { return *this; }
Value& Value::operator[](const std::string&) { return *this; }
Value::~Value() {}
}  // namespace Json

namespace spdlog {
bool logger::should_log(level::level_enum) const { return false; }
}  // namespace spdlog

// core/logger/Logger.h: `extern std::shared_ptr<spdlog::logger> sLogger` is what LOG_* macros of the model sources name
#include <memory>
std::shared_ptr<spdlog::logger> sLogger = std::make_shared<spdlog::logger>();

namespace logtail {
bool RangeCheckpointPB::has_hash_key() const { return false; }
void HashCombine(size_t& seed, size_t value) { seed ^= value + 0x9e3779b97f4a7c15ull + (seed << 6) + (seed >> 2); }
extern const std::string DEFAULT_TRACE_TAG_TRACE_ID = "traceId";
extern const std::string DEFAULT_TRACE_TAG_SPAN_ID = "spanId";
extern const std::string DEFAULT_TRACE_TAG_TRACE_STATE = "traceState";
extern const std::string DEFAULT_TRACE_TAG_TIMESTAMP = "timestamp";
extern const std::string DEFAULT_TRACE_TAG_ATTRIBUTES = "attributes";
extern const std::string DEFAULT_TRACE_TAG_SPAN_EVENT_NAME = "name";
}  // namespace logtail

// core/plugin/processor/inner/ProcessorParseContainerLogNative.cpp:41-42 -- the values come from that file at build time (Makefile)
#include "plugin/processor/inner/ProcessorParseContainerLogNative.h"
namespace logtail {
const std::string ProcessorParseContainerLogNative::containerTimeKey = LC_REF_CONTAINER_TIME_KEY;
const std::string ProcessorParseContainerLogNative::containerSourceKey = LC_REF_CONTAINER_SOURCE_KEY;
}  // namespace logtail
