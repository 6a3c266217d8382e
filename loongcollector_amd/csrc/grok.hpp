// grok.hpp -- Grok pattern library: %{SYNTAX[:alias[:type]]} expansion into plain regexes.
//
// Restates the pattern compiler of the reference's Go plugin:
//   ProcessorGrok.Init              plugins/processor/grok/processor_grok.go:62-102   (defaults, dirs, map, build, compile)
//   addPatternsFromPath / FromMap   :197-236
//   buildPatterns                   :239-279   (reference graph, cycle check, expansion in dependency order)
//   denormalizePattern              :282-316   (%{X} -> "(" stored ")", %{X:alias} -> "(?P<alias>" stored ")")
//   aliasizePatternName/nameToAlias :319-332   (\W -> '_' in group names; emitted key is the original alias)
//   valid / normal / symbolic       :378-382
// The expanded strings are pinned byte for byte by the reference's own test (processor_grok_test.go:36-41), which
// tests/test_grok_host.py replays from tests/golden/grok_expansions.json.
#pragma once

#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace lcgrok {

struct GrokError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// one %{...} token found by the `normal` expression
struct Token {
    size_t begin, end;   // [begin, end) of "%{...}" in the pattern
    std::string body;    // text between the braces: SYNTAX[:alias[:type]]
};

std::vector<Token> findTokens(const std::string& pattern);  // `normal.FindAllStringSubmatch`  (:380)
bool validToken(const std::string& body);                    // `valid.MatchString`              (:379)
std::string aliasize(const std::string& name);               // `symbolic.ReplaceAllString(name, "_")` (:320)

class PatternLibrary {
public:
    void addDefaults();                                            // :68
    void addFromPath(const std::string& path);                     // :197-229; throws GrokError("invalid path :...")
    void addFromText(const std::string& text);                     // the line format of :217-223
    void add(const std::string& name, const std::string& pattern); // :232-236 (later definitions override)

    void build();                                                  // :239-279; throws GrokError
    // :282-316 on an arbitrary pattern (a Match entry); records aliases
    std::string denormalize(const std::string& pattern);

    const std::map<std::string, std::string>& original() const { return mOriginal; }
    const std::map<std::string, std::string>& processed() const { return mProcessed; }
    // :326-332 -- regex group name -> key that is emitted
    std::string nameToAlias(const std::string& groupName) const;

private:
    std::map<std::string, std::string> mOriginal, mProcessed, mAliases;
};

}  // namespace lcgrok
