// processor_filter_gpu.hpp -- processor_filter_regex_native on the device (the step AFTER the
// parser in the reference's benchmark pipeline).
//
// Mirrors core/plugin/processor/ProcessorFilterNative.{h,cpp} of the reference:
//   Init                 :30-157   ConditionExp (expression tree) > FilterKey+FilterRegex > Include (deprecated); DiscardingNonUTF8
//   Process/ProcessEvent :159-216  keep/drop per event with in-place compaction; non-UTF-8 bytes blanked when asked
//   IsMatched            :258-286  rule mode: every key present and regex_match(value)
//   expression nodes     :381-486  and / or / not over {key, exp, type: "regex"} leaves
// Every regex leaf is a boolean boost::regex_match (StringTools.cpp:183-211) -- the same arithmetic as the parser, without
// captures.  Here each leaf is ONE device launch over the values of its key across the whole event group (status bytes
// only); the boolean tree is then evaluated per event on the host.  && / || short-circuiting in the reference only
// decides which matches are run, never the result, so evaluating every leaf is equivalent.
#pragma once

#include <atomic>
#include <string>
#include <vector>

#ifdef LC_USE_REFERENCE_HEADERS  // built inside the LoongCollector tree: the real event model
#include "models/LogEvent.h"
#include "models/PipelineEventGroup.h"
#else
#include "event_model.hpp"
#endif
#include "json_min.hpp"

struct lc_regex;

namespace logtail {

class ProcessorPipelineGpu;

class ProcessorFilterGpu {
    friend class ProcessorPipelineGpu;  // the fused split -> parse -> filter trip runs the rule leaves on the capture spans

public:
    static const std::string sName;  // "processor_filter_regex_gpu"
    enum class Mode { BYPASS_MODE, EXPRESSION_MODE, RULE_MODE };

    ~ProcessorFilterGpu();
    bool Init(const lcjson::Value& config, std::string& error);
    // returns false (group untouched) when regex leaves exist and there is no HIP device: no CPU path
    bool Process(PipelineEventGroup& logGroup, std::string& error);

    Mode mFilterMode = Mode::BYPASS_MODE;
    bool mDiscardingNonUTF8 = false;
    // one instance is shared by the runner threads (like ProcessorParseRegexGpu's counters)
    std::atomic<uint64_t> mInEventsTotal{0}, mOutEventsTotal{0};
    std::atomic<uint64_t> mComplexityExceededTotal{0};  // leaf values the matcher gave up on: taken as false (regex_match failed)

    // ProcessorFilterNative::noneUtf8 (:297-379): true if `s` holds a byte sequence that is not UTF-8 as that routine
    // defines it; with modify, every offending byte is overwritten with ' '
    static bool NoneUtf8(char* s, size_t n, bool modify);

private:
    enum Op { LEAF, NOT, AND, OR };
    struct Node {
        Op op;
        int left, right;  // node indices (NOT: left only)
        int leaf;         // LEAF: index into mLeaves
    };
    struct Leaf {
        std::string key;
        lc_regex* reg;
    };
    std::vector<Node> mNodes;
    std::vector<Leaf> mLeaves;
    int mRoot = -1;                 // EXPRESSION_MODE
    std::vector<int> mRuleLeaves;   // RULE_MODE: all must hold

    int parseExpression(const lcjson::Value& v, std::string& error);  // ParseExpressionFromJSON :381-430; -1 = invalid
    int addLeaf(const std::string& key, const std::string& exp, std::string& error);
    bool eval(int node, const std::vector<std::vector<uint8_t>>& leafResult, size_t event) const;
    void sanitize(LogEvent& e) const;
};

}  // namespace logtail
