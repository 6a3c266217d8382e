/*
 * lc_processor.h -- C ABI of the processor layer: the MI355X replacement for processor_parse_regex_native, plus the
 * dynamic C processor slot LoongCollector already loads with dlopen.
 *
 *   reference                                                                 this ABI
 *   ------------------------------------------------------------------------  -----------------------------
 *   ProcessorParseRegexNative::Init(const Json::Value&)                       lc_processor_create
 *     core/plugin/processor/ProcessorParseRegexNative.cpp:29-106
 *   ProcessorInstance::Process -> ProcessorParseRegexNative::Process(group)   lc_processor_process
 *     core/collection_pipeline/plugin/instance/ProcessorInstance.cpp:46-63
 *     core/plugin/processor/ProcessorParseRegexNative.cpp:108-168,186-253
 *   plugin + instance counters (cpp:100-103, ProcessorInstance.cpp:37-42)      lc_processor_counters
 *   PipelineEventGroup::FromJsonString / ToJsonString (unit-test fixtures)     lc_group_from_json / lc_group_to_json
 *     core/models/PipelineEventGroup.h:140-146
 *   processor_interface_t + dlsym("processor_interface")                      processor_interface (data symbol)
 *     core/collection_pipeline/plugin/creator/CProcessor.h:23-45
 *     core/collection_pipeline/plugin/PluginRegistry.cpp:233-290
 *
 * The event group handed to lc_processor_process / processor_interface.process is a logtail::PipelineEventGroup.
 * In this repository that is the header-compatible stand-in (loongcollector_amd/csrc/event_model.hpp); built inside
 * the reference tree with -DLC_USE_REFERENCE_HEADERS it is the reference's own class.
 */
#ifndef LC_PROCESSOR_H
#define LC_PROCESSOR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lc_processor lc_processor_t;
typedef struct lc_event_group lc_event_group_t;

/* counters, in the order lc_processor_counters fills them */
enum {
    LC_CNT_DISCARDED_EVENTS = 0,      /* discarded_events_total */
    LC_CNT_OUT_FAILED_EVENTS = 1,     /* out_failed_events_total */
    LC_CNT_OUT_KEY_NOT_FOUND = 2,     /* out_key_not_found_events_total */
    LC_CNT_OUT_SUCCESSFUL_EVENTS = 3, /* out_successful_events_total */
    LC_CNT_IN_EVENTS = 4,             /* instance: in_events_total */
    LC_CNT_OUT_EVENTS = 5,            /* instance: out_events_total */
    LC_CNT_IN_SIZE_BYTES = 6,         /* instance: in_size_bytes */
    LC_CNT_OUT_SIZE_BYTES = 7,        /* instance: out_size_bytes */
    LC_CNT_PROCESS_TIME_US = 8,       /* instance: total_process_time (microseconds) */
    LC_CNT_COMPLEXITY_EXCEEDED = 9,   /* no reference counterpart: lines the depth-first decide kernel gave up on (LC_GAVE_UP);
                                         counted in out_failed_events_total too, like boost's complexity exception */
    LC_CNT_UNDECIDED_EVENTS = 10,     /* no reference counterpart: lines left LC_OVERFLOW because the decide pass was switched
                                         off; such events pass through untouched and are in NO other plugin counter */
    LC_CNT_DEVICE_FAILED_EVENTS = 11, /* no reference counterpart: events passed on UNPARSED because the device call of their group
                                         failed in mid-run (there is no CPU path; lc_processor_process also returns the error) */
    LC_CNT_COUNT = 12
};

/* config_json: the plugin's JSON object, e.g.
 *   {"SourceKey":"content","Regex":"(\\w+)\\t(\\w+).*","Keys":["key1","key2"],"KeepingSourceWhenParseFail":true}
 * Returns 0 on success; non-zero (and a message in err) wherever the reference's Init returns false. */
int lc_processor_create(const char* config_json, lc_processor_t** out, char* err, size_t errcap);
void lc_processor_destroy(lc_processor_t* p);
/* number of keys after the legacy ["k1,k2"] split; key i (NULL when out of range) */
int lc_processor_key_count(const lc_processor_t* p);
const char* lc_processor_key(const lc_processor_t* p, int i);

/* Runs the processor over one event group, in place.  Returns 0, or an LC_ERR_* code when the GPU could not be
 * used (the group is then left untouched: there is no CPU fallback). */
int lc_processor_process(lc_processor_t* p, lc_event_group_t* group);
/* The alarms of RegexLogLineParser (core/plugin/processor/ProcessorParseRegexNative.cpp:196-244): one call per failing event with
 * the text the reference hands to AlarmManager::SendAlarmWarning(REGEX_MATCH_ALARM, ...): kind 0 "errorlog:<line>" (no match),
 * kind 1 "errorlog:<line> | exception:<why>" (the matcher gave up: boost's complexity exception), kind 2
 * "parse key count not match<mark_count + 1>errorlog:<line>".  Built inside the agent (LC_USE_REFERENCE_HEADERS) the slot's init
 * keeps the CollectionPipelineContext it is given and raises the alarms there, gated by AppConfig::IsLogParseAlarmValid() as in
 * the reference; this sink works in every build (called from the thread that runs lc_processor_process). */
typedef void (*lc_alarm_sink_t)(void* user, int kind, const char* message, size_t len);
void lc_processor_set_alarm_sink(lc_processor_t* p, lc_alarm_sink_t sink, void* user);
int lc_processor_counters(const lc_processor_t* p, uint64_t out[LC_CNT_COUNT]);

/* test fixtures in the reference unit tests' JSON format */
lc_event_group_t* lc_group_from_json(const char* json, char* err, size_t errcap);
/* malloc'd NUL-terminated JSON; release with lc_free */
/* ---- columnar hand-off (the serializer side of the bulk stitch).
 * The reference materialises K (key, view) pairs in every LogEvent (RegexLogLineParser :249-251) only for the serializer to
 * walk them twice more (SLSEventGroupSerializer::CalculateLogEventSize + SerializeLogEvent, SLSSerializer.cpp:254-264,377-388:
 * per event, per content, GetLogContentSize(key.size(), value.size()) and AddLogContent(key, value)).  What the device returns
 * is already columnar -- one (begin, end) pair per event and key -- so a serializer can take it as is: lc_processor_parse_columnar
 * runs the gather + device match of lc_processor_process and hands back the capture table next to the values' base pointers,
 * WITHOUT touching the group (no stitching, no source-key policy: both stay with the caller).  content_bytes[i] is the sum the
 * serializer's first pass computes for event i's K parsed fields (protobuf sizes: 1 + varint(len) + len per string, 1 + varint
 * per content, LogGroupSerializer.cpp:227-252). */
enum { LC_COL_SKIPPED = 0, LC_COL_PARSED = 1, LC_COL_FAILED = 2 };
typedef struct lc_columnar {
    uint32_t n_events;             /* events of the group, in order */
    uint32_t n_keys;               /* K = number of Keys */
    const char* const* keys;       /* [K], NUL-terminated, owned by the processor */
    const uint32_t* key_len;       /* [K] */
    const uint8_t* const* base;    /* [n_events] start of the event's source value (a view into the group's SourceBuffer); NULL
                                      for an event the processor would not parse (not a log event, no SourceKey content) */
    const uint32_t* base_len;      /* [n_events] */
    const int32_t* spans;          /* [n_events][2K]: begin, end of field k relative to base; (-1, -1): empty value (a group
                                      that did not take part: boost's {last, last}) */
    const uint8_t* state;          /* [n_events] LC_COL_* */
    const uint64_t* content_bytes; /* [n_events] sum over k of GetLogContentSize(key_len[k], end - begin); 0 unless PARSED */
} lc_columnar_t;
int lc_processor_parse_columnar(lc_processor_t* p, lc_event_group_t* group, lc_columnar_t** out);
void lc_columnar_free(lc_columnar_t* c);

/* the group a file input hands over: one copy of the n lines (data + off[i], len[i]) back to back in the group's SourceBuffer,
 * one log event per line whose `key` content is a view into it (ProcessorSplitLogStringNative.cpp:130-160) */
lc_event_group_t* lc_group_from_lines(const uint8_t* data, const uint32_t* off, const uint32_t* len, uint32_t n, const char* key);
/* the group the file input hands over BEFORE the line splitter: ONE log event whose `key` content is a copy of the read buffer
 * (position = file_offset, n); file_offset_key != NULL sets the group's LOG_FILE_OFFSET_KEY metadata (the splitter then adds the
 * offset content to every line, ProcessorSplitLogStringNative.cpp:151-156) */
lc_event_group_t* lc_group_from_buffer(const uint8_t* data, size_t n, const char* key, uint64_t file_offset, const char* file_offset_key);
char* lc_group_to_json(const lc_event_group_t* g);
size_t lc_group_event_count(const lc_event_group_t* g);
/* the logtail::PipelineEventGroup* inside the fixture wrapper (what processor_interface.process expects) */
void* lc_group_native(lc_event_group_t* g);
void lc_group_free(lc_event_group_t* g);
void lc_free(void* p);

/* ---- the reference's benchmark pipeline in ONE device trip per read buffer: split -> processor_parse_regex_native ->
 * processor_filter_regex_native (test/benchmark/local/test_cases/performance_file_to_blackhole_loongcollector/loongcollector.yaml;
 * ProcessorSplitLogStringNative.cpp:101-174, ProcessorParseRegexNative.cpp:108-253, ProcessorFilterNative.cpp:159-286).
 * config_json: {"Split": {"SourceKey": "content", "SplitChar": "\n"}, "Parse": {<the parser's keys>}, "Filter": {<the filter's
 * keys>}, "Fused": true}.  The raw buffer goes up once; lines are found, matched and filtered (the filter's FilterKey / FilterRegex
 * rules run on the capture spans of the keys they name) on the device; only the survivors come back and only they become events --
 * the same events, contents, positions and counters the three processors leave one after the other.  What the fused trip cannot
 * express (rules on keys the parser does not produce, ConditionExp, DiscardingNonUTF8, alarms wanted, ...) runs the three steps
 * one after the other with the same classes: lc_pipeline_is_fused says which.  lc_pipeline_process takes a group as made by
 * lc_group_from_buffer (or any group: events that are not plain read buffers send the group down the chained path). */
typedef struct lc_pipeline lc_pipeline_t;
enum { LC_PIPE_FILTER_IN_EVENTS = 0, LC_PIPE_FILTER_OUT_EVENTS, LC_PIPE_GROUPS_FUSED, LC_PIPE_GROUPS_CHAINED, LC_PIPE_LINES,
       LC_PIPE_SURVIVORS, LC_PIPE_CNT_COUNT };
int lc_pipeline_create(const char* config_json, lc_pipeline_t** out, char* err, size_t errcap);
void lc_pipeline_destroy(lc_pipeline_t* p);
int lc_pipeline_is_fused(const lc_pipeline_t* p);
int lc_pipeline_process(lc_pipeline_t* p, lc_event_group_t* group);
/* parse[]: the parser's counters in LC_CNT_* order (instance-level entries are 0); pipe[]: LC_PIPE_* */
int lc_pipeline_counters(const lc_pipeline_t* p, uint64_t parse[LC_CNT_COUNT], uint64_t pipe[LC_PIPE_CNT_COUNT]);

/* ---- the dynamic C processor slot (layout identical to CProcessor.h:23-45) ---- */
#define LC_PROCESSOR_INTERFACE_VERSION 100

struct processor_instance_t;
/* ---- processor_filter_regex_native (core/plugin/processor/ProcessorFilterNative.cpp) on the device: the step after the
 * parser in the reference's benchmark pipeline.  config_json: ConditionExp | FilterKey+FilterRegex | Include, and
 * DiscardingNonUTF8 (same keys, same precedence as ProcessorFilterNative::Init :30-157).  Each regex leaf is one device
 * launch over the values of its key across the group; events that fail are removed in place (:159-176).
 * lc_filter_process takes the logtail::PipelineEventGroup* (lc_group_native() of a fixture group). */
typedef struct lc_filter lc_filter_t;
int lc_filter_create(const char* config_json, lc_filter_t** out, char* err, size_t errcap);
void lc_filter_destroy(lc_filter_t* f);
int lc_filter_mode(const lc_filter_t* f);          /* 0 bypass, 1 expression, 2 rule (ProcessorFilterNative::Mode) */
int lc_filter_process(lc_filter_t* f, void* native_group);
void lc_filter_counters(const lc_filter_t* f, uint64_t out[2]);   /* in_events_total, out_events_total */
/* ProcessorFilterNative::noneUtf8 (:297-379) on a buffer: returns 1 if it holds bytes that routine rejects; with modify
 * != 0 they are overwritten with ' ' in place (what DiscardingNonUTF8 does to keys and values) */
int lc_filter_none_utf8(char* buf, size_t n, int modify);

typedef int (*processor_init_func_t)(struct processor_instance_t* ins, void* config, void* context);
typedef void (*processor_finialize_func_t)(void* plugin_state);
typedef void (*processor_process_func_t)(void* plugin_state, void* logGroup);

typedef struct processor_interface_t {
    int version;
    const char* name;
    const char* language;
    processor_init_func_t init;
    processor_finialize_func_t finalize;
    processor_process_func_t process;
} processor_interface_t;

typedef struct processor_instance_t {
    const processor_interface_t* plugin;
    void* plugin_state;
} processor_instance_t;

/* Looked up by PluginRegistry::LoadProcessorPlugin with dlsym.  init(): `config` is the plugin's JSON config as
 * NUL-terminated text in this build (a Json::Value* when built with LC_USE_REFERENCE_HEADERS); `context` is unused.
 * process(): `logGroup` is a logtail::PipelineEventGroup*. */
extern processor_interface_t processor_interface;

#ifdef __cplusplus
}
#endif
#endif
