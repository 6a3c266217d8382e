// processor_pipeline_gpu.hpp -- the reference's benchmark pipeline as ONE device trip per read buffer.
//
//   inputs:  input_file  (one LogEvent per read buffer, <= 512 KB, key "content": LogFileReader)
//   inner:   ProcessorSplitLogStringNative       core/plugin/processor/inner/ProcessorSplitLogStringNative.cpp:101-174
//   1:       processor_parse_regex_native        core/plugin/processor/ProcessorParseRegexNative.cpp:108-253
//   2:       processor_filter_regex_native       core/plugin/processor/ProcessorFilterNative.cpp:159-286 (FilterKey / FilterRegex)
//   (test/benchmark/local/test_cases/performance_file_to_blackhole_loongcollector/loongcollector.yaml:1-27 -- the pipeline
//   behind the reference's 68 MB/s figure)
//
// Each of the three processors does its own pass over the events in the reference; here the raw buffer goes up once, the device
// finds the lines (split_*_kernel), matches them (the parse regex's kernel, line count read on the device), runs the filter's
// rules on the capture spans of the keys they name (span_filter_pack_kernel) and only the SURVIVORS come back, as
// [line, offset, length, capture offsets].  The host creates events for the survivors only -- exactly the events the three
// processors would have left, with the same contents in the same order, the same positions and timestamps -- and books the
// counters of all three as if every line had gone through them.
//
// What does not fit the fused trip runs the three steps one after the other (same classes, same results): groups with events that
// are not plain read buffers, rules on keys the parser does not produce, ConditionExp / Include filters, DiscardingNonUTF8,
// whole-line mode, a key-count mismatch, alarms wanted per failing line, lines the device left undecided.
#pragma once

#include <atomic>
#include <string>
#include <vector>

#include "processor_filter_gpu.hpp"
#include "processor_parse_regex_gpu.hpp"

namespace logtail {

class ProcessorPipelineGpu {
public:
    static const std::string sName;  // "processor_split_parse_filter_gpu"
    // config: {"Split": {"SourceKey": "content", "SplitChar": "\n"}, "Parse": {<processor_parse_regex_native keys>},
    //          "Filter": {<processor_filter_regex_native keys>}, "Fused": true}
    bool Init(const lcjson::Value& config, std::string& error);
    // logGroup: what the file input hands over (events holding read buffers).  false + error: device failure, group untouched.
    bool Process(PipelineEventGroup& logGroup, std::string& error);

    std::string mSplitKey = "content";  // ProcessorSplitLogStringNative::mSourceKey (DEFAULT_CONTENT_KEY)
    char mSplitChar = '\n';
    bool mFusedWanted = true;
    ProcessorParseRegexGpu mParse;
    ProcessorFilterGpu mFilter;
    bool mHasFilter = false;

    bool IsFused() const { return mFused; }
    std::atomic<uint64_t> mGroupsFused{0}, mGroupsChained{0}, mLinesTotal{0}, mSurvivorsTotal{0};

    // ProcessorSplitLogStringNative::ProcessEvent for every event of the group (:101-160), on the host: the chained path
    void SplitEvents(PipelineEventGroup& logGroup) const;

private:
    struct Rule {
        lc_regex* re;
        uint32_t group;  // capture group of the parse regex (1-based) that produces the rule's key
    };
    std::vector<Rule> mRules;
    bool mFused = false;
    bool ProcessFused(PipelineEventGroup& logGroup, std::string& error, bool& fellBack);
    std::unique_ptr<LogEvent> NewLineEvent(PipelineEventGroup& logGroup, const LogEvent& sourceEvent, StringView sourceVal,
                                           StringView sourceKey, uint32_t off, uint32_t len) const;
};

}  // namespace logtail
