// tests/refhdr -- TEST INFRASTRUCTURE: a declarations-only stand-in for JsonCpp's <json/json.h> (not installed in this image), just
// enough for the reference's headers and for the shim's LC_USE_REFERENCE_HEADERS variant (which receives its plugin config as
// a `const Json::Value*`, core/plugin/processor/DynamicCProcessorProxy.cpp:30-32) to be type-checked.  Never linked or run.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace Json {
using Int = int;
using UInt = unsigned;
using Int64 = int64_t;
using UInt64 = uint64_t;
using ArrayIndex = unsigned;
using String = std::string;
enum ValueType { nullValue = 0, intValue, uintValue, realValue, stringValue, booleanValue, arrayValue, objectValue };

class Value {
public:
    using Members = std::vector<std::string>;
    class const_iterator {
    public:
        const Value& operator*() const;
        const Value* operator->() const;
        const_iterator& operator++();
        bool operator!=(const const_iterator&) const;
        bool operator==(const const_iterator&) const;
        std::string name() const;
        Value key() const;
    };
    using iterator = const_iterator;
    Value(ValueType = nullValue);
    Value(Int);
    Value(UInt);
    Value(Int64);
    Value(UInt64);
    Value(double);
    Value(const char*);
    Value(const std::string&);
    Value(bool);
    Value(const Value&);
    Value& operator=(const Value&);
    ~Value();
    ValueType type() const;
    bool isNull() const;
    bool isBool() const;
    bool isInt() const;
    bool isUInt() const;
    bool isInt64() const;
    bool isUInt64() const;
    bool isIntegral() const;
    bool isDouble() const;
    bool isNumeric() const;
    bool isString() const;
    bool isArray() const;
    bool isObject() const;
    bool asBool() const;
    Int asInt() const;
    UInt asUInt() const;
    Int64 asInt64() const;
    UInt64 asUInt64() const;
    double asDouble() const;
    std::string asString() const;
    const char* asCString() const;
    ArrayIndex size() const;
    bool empty() const;
    void clear();
    Value& operator[](ArrayIndex);
    const Value& operator[](ArrayIndex) const;
    Value& operator[](int);
    const Value& operator[](int) const;
    Value& operator[](const char*);
    const Value& operator[](const char*) const;
    Value& operator[](const std::string&);
    const Value& operator[](const std::string&) const;
    Value& append(const Value&);
    bool isMember(const char*) const;
    bool isMember(const std::string&) const;
    const Value* find(const char* begin, const char* end) const;
    Value get(const char*, const Value& defaultValue) const;
    Value get(const std::string&, const Value& defaultValue) const;
    Value removeMember(const char*);
    Value removeMember(const std::string&);
    Members getMemberNames() const;
    std::string toStyledString() const;
    const_iterator begin() const;
    const_iterator end() const;
    bool operator==(const Value&) const;
    bool operator!=(const Value&) const;
    bool operator<(const Value&) const;
    static const Value null;
    static const Value& nullSingleton();
};

class StreamWriterBuilder {
public:
    Value& operator[](const std::string&);
};
std::string writeString(const StreamWriterBuilder&, const Value&);
class CharReader {
public:
    virtual ~CharReader();
    virtual bool parse(const char* begin, const char* end, Value* root, std::string* errs) = 0;
};
class CharReaderBuilder {
public:
    CharReader* newCharReader() const;
    Value& operator[](const std::string&);
};
}  // namespace Json
