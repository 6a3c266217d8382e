// oracle/ref_processor/stubs -- TEST INFRASTRUCTURE.  <json/json.h> (JsonCpp is not in this image) for the build of the reference's
// ProcessorParseRegexNative.cpp, CommonParserOptions.cpp and ParamExtractor.cpp: a Json::Value that WORKS for what a plugin config
// needs -- a tree of objects, arrays, strings, booleans and numbers, filled from the repo's small JSON reader (csrc/json_min.hpp) --
// under JsonCpp's names for the calls those files make (isMember, find, begin/end, is*/as*, operator[]).  The model headers only name
// the type (ToJson / FromJson declarations).
#pragma once
#include <cstdio>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../../../loongcollector_amd/csrc/json_min.hpp"

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
    Value(ValueType t = nullValue) : mType(t) {}
    Value(const std::string& s) : mType(stringValue), mStr(s) {}
    Value(const char* s) : mType(stringValue), mStr(s ? s : "") {}
    Value(bool b) : mType(booleanValue), mBool(b) {}
    Value(Int i) : mType(intValue), mInt(i) {}
    Value(Int64 i) : mType(intValue), mInt(i) {}
    Value(double d) : mType(realValue), mDouble(d) {}
    static Value fromText(const std::string& json) { return fromLc(lcjson::parse(json)); }
    static Value fromLc(const lcjson::Value& v) {
        Value out;
        switch (v.type) {
            case lcjson::Value::Null: break;
            case lcjson::Value::Bool: out = Value(v.b); break;
            case lcjson::Value::String: out = Value(v.str); break;
            case lcjson::Value::Number:
                if (v.isInt) out = Value(Int64(v.inum));
                else out = Value(v.num);
                break;
            case lcjson::Value::Array:
                out.mType = arrayValue;
                for (const auto& c : v.arr) out.mArr.push_back(fromLc(c));
                break;
            case lcjson::Value::Object:
                out.mType = objectValue;
                for (const auto& kv : v.obj) out.mObj.emplace_back(kv.first, fromLc(kv.second));
                break;
        }
        return out;
    }
    ValueType type() const { return mType; }
    bool isNull() const { return mType == nullValue; }
    bool isBool() const { return mType == booleanValue; }
    bool isString() const { return mType == stringValue; }
    bool isArray() const { return mType == arrayValue; }
    bool isObject() const { return mType == objectValue; }
    bool isInt64() const { return mType == intValue; }
    bool isInt() const { return mType == intValue && mInt >= INT32_MIN && mInt <= INT32_MAX; }
    bool isUInt64() const { return mType == intValue && mInt >= 0; }
    bool isUInt() const { return mType == intValue && mInt >= 0 && mInt <= int64_t(UINT32_MAX); }
    bool isIntegral() const { return mType == intValue; }
    bool isDouble() const { return mType == realValue || mType == intValue; }
    bool isNumeric() const { return isDouble(); }
    std::string asString() const { return mType == stringValue ? mStr : std::string(); }
    bool asBool() const { return mType == booleanValue && mBool; }
    Int asInt() const { return Int(mInt); }
    UInt asUInt() const { return UInt(mInt); }
    Int64 asInt64() const { return mInt; }
    UInt64 asUInt64() const { return UInt64(mInt); }
    double asDouble() const { return mType == realValue ? mDouble : double(mInt); }
    ArrayIndex size() const { return mType == arrayValue ? ArrayIndex(mArr.size()) : mType == objectValue ? ArrayIndex(mObj.size()) : 0u; }
    bool empty() const { return size() == 0; }
    const Value* find(const char* begin, const char* end) const {
        if (mType != objectValue) return nullptr;
        const std::string key(begin, end);
        for (const auto& kv : mObj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool isMember(const std::string& key) const { return find(key.data(), key.data() + key.size()) != nullptr; }
    bool isMember(const char* key) const { return isMember(std::string(key)); }
    const Value& operator[](const std::string& key) const {
        const Value* c = find(key.data(), key.data() + key.size());
        return c ? *c : null();
    }
    const Value& operator[](const char* key) const { return (*this)[std::string(key)]; }
    const Value& operator[](ArrayIndex i) const { return mType == arrayValue && i < mArr.size() ? mArr[i] : null(); }
    const Value& operator[](int i) const { return (*this)[ArrayIndex(i)]; }
    Members getMemberNames() const {
        Members m;
        for (const auto& kv : mObj) m.push_back(kv.first);
        return m;
    }
    // JSON text of the tree (JsonCpp's writer formats with indentation; a reader cannot tell)
    std::string toStyledString() const {
        std::string out;
        write(out);
        out += '\n';
        return out;
    }
    class const_iterator {
    public:
        const_iterator(const Value* owner, size_t i) : mOwner(owner), mI(i) {}
        const Value& operator*() const { return mOwner->mType == objectValue ? mOwner->mObj[mI].second : mOwner->mArr[mI]; }
        const Value* operator->() const { return &**this; }
        const_iterator& operator++() {
            ++mI;
            return *this;
        }
        bool operator!=(const const_iterator& o) const { return mI != o.mI || mOwner != o.mOwner; }
        bool operator==(const const_iterator& o) const { return !(*this != o); }
        std::string name() const { return mOwner->mType == objectValue ? mOwner->mObj[mI].first : std::string(); }
        Value key() const { return Value(name()); }

    private:
        const Value* mOwner;
        size_t mI;
    };
    using iterator = const_iterator;
    const_iterator begin() const { return const_iterator(this, 0); }
    const_iterator end() const { return const_iterator(this, size()); }

private:
    static void quote(std::string& out, const std::string& s) {
        out += '"';
        for (unsigned char c : s) {
            if (c == '"' || c == '\\') {
                out += '\\';
                out += char(c);
            } else if (c < 0x20) {
                char buf[8];
                std::snprintf(buf, sizeof buf, "\\u%04x", unsigned(c));
                out += buf;
            } else {
                out += char(c);
            }
        }
        out += '"';
    }
    void write(std::string& out) const {
        switch (mType) {
            case nullValue: out += "null"; break;
            case booleanValue: out += mBool ? "true" : "false"; break;
            case intValue:
            case uintValue: out += std::to_string(mInt); break;
            case realValue: {
                char buf[40];
                std::snprintf(buf, sizeof buf, "%.17g", mDouble);
                out += buf;
                break;
            }
            case stringValue: quote(out, mStr); break;
            case arrayValue:
                out += '[';
                for (size_t i = 0; i < mArr.size(); ++i) {
                    if (i) out += ',';
                    mArr[i].write(out);
                }
                out += ']';
                break;
            case objectValue:
                out += '{';
                for (size_t i = 0; i < mObj.size(); ++i) {
                    if (i) out += ',';
                    quote(out, mObj[i].first);
                    out += ':';
                    mObj[i].second.write(out);
                }
                out += '}';
                break;
        }
    }
    static const Value& null() {
        static const Value v;
        return v;
    }
    ValueType mType;
    std::string mStr;
    bool mBool = false;
    int64_t mInt = 0;
    double mDouble = 0;
    std::vector<Value> mArr;
    std::vector<std::pair<std::string, Value>> mObj;
};
}  // namespace Json
