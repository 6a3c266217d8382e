// json_min.hpp -- a small JSON reader/writer for plugin configs and test fixtures (header only).
// Stands in for jsoncpp's Json::Value on this side of the C ABI: the reference hands plugin Init a Json::Value
// (core/plugin/processor/ProcessorParseRegexNative.cpp:29); the shim serialises it to text, this parses the text.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace lcjson {

struct Value {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    double num = 0;
    int64_t inum = 0;
    bool isInt = false;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;  // insertion order preserved

    bool isNull() const { return type == Null; }
    bool isBool() const { return type == Bool; }
    bool isString() const { return type == String; }
    bool isArray() const { return type == Array; }
    bool isObject() const { return type == Object; }
    bool isNumber() const { return type == Number; }
    const Value* find(const std::string& key) const {
        if (type != Object) return nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    Value& set(const std::string& key, Value v) {
        type = Object;
        for (auto& kv : obj)
            if (kv.first == key) {
                kv.second = std::move(v);
                return kv.second;
            }
        obj.emplace_back(key, std::move(v));
        return obj.back().second;
    }
    static Value makeString(std::string s) {
        Value v;
        v.type = String;
        v.str = std::move(s);
        return v;
    }
    static Value makeInt(int64_t i) {
        Value v;
        v.type = Number;
        v.isInt = true;
        v.inum = i;
        v.num = double(i);
        return v;
    }
    static Value makeObject() {
        Value v;
        v.type = Object;
        return v;
    }
    static Value makeArray() {
        Value v;
        v.type = Array;
        return v;
    }
};

struct ParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class Parser {
public:
    explicit Parser(const std::string& text) : s(text) {}
    Value parse() {
        Value v = value();
        ws();
        if (i != s.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string& s;
    size_t i = 0;
    [[noreturn]] void fail(const char* what) const {
        throw ParseError(std::string("json: ") + what + " at offset " + std::to_string(i));
    }
    void ws() {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) ++i;
    }
    static void appendUtf8(std::string& out, uint32_t cp) {
        if (cp < 0x80) {
            out.push_back(char(cp));
        } else if (cp < 0x800) {
            out.push_back(char(0xC0 | (cp >> 6)));
            out.push_back(char(0x80 | (cp & 0x3F)));
        } else if (cp < 0x10000) {
            out.push_back(char(0xE0 | (cp >> 12)));
            out.push_back(char(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back(char(0x80 | (cp & 0x3F)));
        } else {
            out.push_back(char(0xF0 | (cp >> 18)));
            out.push_back(char(0x80 | ((cp >> 12) & 0x3F)));
            out.push_back(char(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back(char(0x80 | (cp & 0x3F)));
        }
    }
    uint32_t hex4() {
        if (i + 4 > s.size()) fail("bad \\u escape");
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k) {
            char c = s[i++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= uint32_t(c - '0');
            else if (c >= 'a' && c <= 'f') v |= uint32_t(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= uint32_t(c - 'A' + 10);
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string string() {
        if (s[i] != '"') fail("expected string");
        ++i;
        std::string out;
        for (;;) {
            if (i >= s.size()) fail("unterminated string");
            char c = s[i++];
            if (c == '"') break;
            if (c != '\\') {
                out.push_back(c);
                continue;
            }
            if (i >= s.size()) fail("unterminated escape");
            char e = s[i++];
            switch (e) {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    uint32_t cp = hex4();
                    if (cp >= 0xD800 && cp < 0xDC00 && i + 1 < s.size() && s[i] == '\\' && s[i + 1] == 'u') {
                        i += 2;
                        uint32_t lo = hex4();
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    appendUtf8(out, cp);
                    break;
                }
                default: fail("bad escape");
            }
        }
        return out;
    }
    // nesting guard (jsoncpp, the reference's parser, stops at stackLimit = 1000 as well): value() recurses
    struct Depth {
        int& d;
        explicit Depth(int& x) : d(x) { ++d; }
        ~Depth() { --d; }
    };
    int depth = 0;
    Value value() {
        Depth guard(depth);
        if (depth > 1000) fail("nesting too deep");
        ws();
        if (i >= s.size()) fail("unexpected end");
        char c = s[i];
        Value v;
        if (c == '{') {
            ++i;
            v.type = Value::Object;
            ws();
            if (i < s.size() && s[i] == '}') {
                ++i;
                return v;
            }
            for (;;) {
                ws();
                std::string key = string();
                ws();
                if (i >= s.size() || s[i] != ':') fail("expected ':'");
                ++i;
                v.obj.emplace_back(std::move(key), value());
                ws();
                if (i < s.size() && s[i] == ',') {
                    ++i;
                    continue;
                }
                if (i < s.size() && s[i] == '}') {
                    ++i;
                    break;
                }
                fail("expected ',' or '}'");
            }
            return v;
        }
        if (c == '[') {
            ++i;
            v.type = Value::Array;
            ws();
            if (i < s.size() && s[i] == ']') {
                ++i;
                return v;
            }
            for (;;) {
                v.arr.push_back(value());
                ws();
                if (i < s.size() && s[i] == ',') {
                    ++i;
                    continue;
                }
                if (i < s.size() && s[i] == ']') {
                    ++i;
                    break;
                }
                fail("expected ',' or ']'");
            }
            return v;
        }
        if (c == '"') {
            v.type = Value::String;
            v.str = string();
            return v;
        }
        if (s.compare(i, 4, "true") == 0) {
            i += 4;
            v.type = Value::Bool;
            v.b = true;
            return v;
        }
        if (s.compare(i, 5, "false") == 0) {
            i += 5;
            v.type = Value::Bool;
            return v;
        }
        if (s.compare(i, 4, "null") == 0) {
            i += 4;
            return v;
        }
        size_t st = i;
        if (i < s.size() && (s[i] == '-' || s[i] == '+')) ++i;
        bool frac = false;
        while (i < s.size() && ((s[i] >= '0' && s[i] <= '9') || s[i] == '.' || s[i] == 'e' || s[i] == 'E' || s[i] == '-' ||
                                s[i] == '+')) {
            if (s[i] == '.' || s[i] == 'e' || s[i] == 'E') frac = true;
            ++i;
        }
        if (i == st) fail("unexpected character");
        v.type = Value::Number;
        const std::string tok = s.substr(st, i - st);
        v.num = std::strtod(tok.c_str(), nullptr);
        if (!frac) {
            v.isInt = true;
            v.inum = std::strtoll(tok.c_str(), nullptr, 10);
        }
        return v;
    }
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }

inline void escapeTo(std::string& out, const std::string& s) {
    out.push_back('"');
    for (unsigned char c : s) {
        switch (c) {
            case '"': out += "\\\""; break;
            case '\\': out += "\\\\"; break;
            case '\n': out += "\\n"; break;
            case '\r': out += "\\r"; break;
            case '\t': out += "\\t"; break;
            case '\b': out += "\\b"; break;
            case '\f': out += "\\f"; break;
            default:
                if (c < 0x20) {
                    char buf[8];
                    std::snprintf(buf, sizeof buf, "\\u%04x", c);
                    out += buf;
                } else {
                    out.push_back(char(c));
                }
        }
    }
    out.push_back('"');
}

inline void dumpTo(std::string& out, const Value& v) {
    switch (v.type) {
        case Value::Null: out += "null"; break;
        case Value::Bool: out += v.b ? "true" : "false"; break;
        case Value::Number:
            if (v.isInt) {
                out += std::to_string(v.inum);
            } else {
                char buf[40];
                std::snprintf(buf, sizeof buf, "%.17g", v.num);
                out += buf;
            }
            break;
        case Value::String: escapeTo(out, v.str); break;
        case Value::Array:
            out.push_back('[');
            for (size_t k = 0; k < v.arr.size(); ++k) {
                if (k) out.push_back(',');
                dumpTo(out, v.arr[k]);
            }
            out.push_back(']');
            break;
        case Value::Object:
            out.push_back('{');
            for (size_t k = 0; k < v.obj.size(); ++k) {
                if (k) out.push_back(',');
                escapeTo(out, v.obj[k].first);
                out.push_back(':');
                dumpTo(out, v.obj[k].second);
            }
            out.push_back('}');
            break;
    }
}

inline std::string dump(const Value& v) {
    std::string out;
    dumpTo(out, v);
    return out;
}

}  // namespace lcjson
