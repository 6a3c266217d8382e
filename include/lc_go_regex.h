/*
 * lc_go_regex.h -- C ABI of the Go plugin processor_regex on the device.
 *
 * Replaces plugins/processor/regex/regex.go of the reference:
 *   lc_goregex_create            <- ProcessorRegex.Init        :50-66   regexp.Compile("(?s)" + Regex); Keys must not be empty
 *   lc_goregex_process_logs_json <- ProcessLogs / ProcessLog / processRegex / shouldKeepSource   :72-129
 *   lc_goregex_regex             -> the compiled handle for callers that stitch themselves (cgo): one
 *                                   lc_regex_match_host / lc_regex_match_device call per batch of SourceKey values
 * The pattern is compiled with LC_SYNTAX_SEARCH | LC_SYNTAX_NO_MULTILINE | LC_SYNTAX_REGEXP2 ('.' matches '\n' because of
 * the (?s) the plugin prepends; '^' '$' only at the ends; RE2's \s): leftmost-first search like Go's regexp.Compile.
 * Capture row of a value: group 1 = the whole match (FindStringSubmatchIndex[0:2]), group i+2 = Keys[i].
 * Bytes, not runes; RE2-only syntax errors of Go (e.g. look-arounds are errors there, accepted here when the
 * device engines can run them) are not reproduced.
 */
#ifndef LC_GO_REGEX_H
#define LC_GO_REGEX_H

#include <stddef.h>
#include <stdint.h>

#include "lc_regex_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lc_goregex lc_goregex_t;

/* config_json: {"Regex": "...", "Keys": [...], "FullMatch": false, "NoKeyError": false, "NoMatchError": true,
 *               "KeepSource": false, "KeepSourceIfParseError": true, "SourceKey": ""}  (defaults of regex.go:131-139) */
int lc_goregex_create(const char* config_json, size_t config_len, lc_goregex_t** out, char* err, size_t errcap);
void lc_goregex_free(lc_goregex_t* p);
lc_regex_t* lc_goregex_regex(lc_goregex_t* p);
/* logs as [[["key","value"],...],...] in, same shape out; *out_json is malloc'ed, release with lc_goregex_free_string */
int lc_goregex_process_logs_json(lc_goregex_t* p, const char* logs_json, size_t len, char** out_json);
void lc_goregex_free_string(char* s);

#ifdef __cplusplus
}
#endif
#endif /* LC_GO_REGEX_H */
