/*
 * oracle/processor_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU baseline leg of bench.py).
 *
 * The per-event work of the reference processor with its default options, restated in C so that it can be timed:
 *   ProcessorParseRegexNative::ProcessEvent + RegexLogLineParser + AddLog
 *       core/plugin/processor/ProcessorParseRegexNative.cpp:132-253
 *   LogEvent::{HasContent, GetContent, SetContentNoCopy, DelContent}   core/models/LogEvent.cpp:50-106
 * i.e. per line: source-key lookup, regex_match with sub-matches (oracle/bt_regex.c), one SetContentNoCopy per key
 * (reverse linear scan of the content list, overwrite or append), tombstone the source key, counters.
 * Defaults as in the reference benchmark pipeline: KeepingSourceWhenParseFail/Succeed = false, no raw-log copy.
 * The behavioural oracle used by the parity tests is oracle/processor_oracle.py; this file only reproduces the same
 * work at C speed for the reported baseline.
 */
#include <stdlib.h>
#include <string.h>

#include "bt_regex.h"

typedef struct { const char* k; size_t kl; const char* v; size_t vl; int alive; } content;
typedef struct { content* c; size_t n, cap; size_t allocated; size_t cnt; } log_event;

static content* find_live(log_event* e, const char* k, size_t kl) { /* reverse scan, LogEvent.cpp:50-58 */
    for (size_t i = e->n; i-- > 0;) {
        content* c = &e->c[i];
        if (c->alive && c->kl == kl && memcmp(c->k, k, kl) == 0) return c;
    }
    return NULL;
}
static void set_nocopy(log_event* e, const char* k, size_t kl, const char* v, size_t vl) { /* LogEvent.cpp:83-95 */
    content* c = find_live(e, k, kl);
    if (c) {
        e->allocated += kl + vl - c->kl - c->vl;
        c->k = k; c->kl = kl; c->v = v; c->vl = vl;
    } else {
        if (e->n == e->cap) { e->cap = e->cap ? e->cap * 2 : 16; e->c = (content*)realloc(e->c, e->cap * sizeof(content)); }
        content* d = &e->c[e->n++];
        d->k = k; d->kl = kl; d->v = v; d->vl = vl; d->alive = 1;
        e->cnt++; e->allocated += kl + vl;
    }
}
static void del_content(log_event* e, const char* k, size_t kl) { /* LogEvent.cpp:97-106 */
    content* c = find_live(e, k, kl);
    if (c) { c->alive = 0; e->cnt--; e->allocated -= c->kl + c->vl; }
}

/* Processes n single-content events ("content" -> line i).  keys: nkeys NUL-terminated strings back to back.
 * counters[4] = discarded, out_failed, out_key_not_found, out_successful.  Returns a checksum of the stitched
 * views so the work cannot be optimised away. */
unsigned long orx_process_batch(const orx_prog* p, const uint8_t* data, const uint32_t* off, const uint32_t* len,
                                size_t nlines, const char* keys, int nkeys, unsigned long counters[4]) {
    static const char source_key[] = "content";
    const size_t skl = sizeof source_key - 1;
    const char** kp = (const char**)malloc(sizeof(char*) * (size_t)(nkeys ? nkeys : 1));
    size_t* kl = (size_t*)malloc(sizeof(size_t) * (size_t)(nkeys ? nkeys : 1));
    int overwritten = 0;
    for (int i = 0; i < nkeys; ++i) {
        kp[i] = keys; kl[i] = strlen(keys); keys += kl[i] + 1;
        if (kl[i] == skl && memcmp(kp[i], source_key, skl) == 0) overwritten = 1;
    }
    const int ngroups = orx_mark_count(p);
    int32_t* caps = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(ngroups + 1));
    log_event ev; memset(&ev, 0, sizeof ev);
    unsigned long sum = 0;
    for (size_t i = 0; i < nlines; ++i) {
        ev.n = 0; ev.cnt = 0; ev.allocated = 0;                      /* a fresh event holding only the source content */
        const char* raw = (const char*)data + off[i];
        set_nocopy(&ev, source_key, skl, raw, len[i]);
        if (!find_live(&ev, source_key, skl)) { counters[2]++; continue; }            /* :140-143 */
        int ok = orx_fullmatch(p, (const uint8_t*)raw, len[i], caps) == 1;            /* :194 */
        if (!ok) counters[1]++;
        else if (ngroups + 1 <= nkeys) ok = 0;                                        /* :227 */
        if (ok)
            for (int k = 0; k < nkeys; ++k) {                                         /* :249-251 */
                const int32_t b = caps[2 * (k + 1)], e = caps[2 * (k + 1) + 1];
                if (b < 0) set_nocopy(&ev, kp[k], kl[k], raw + len[i], 0);
                else set_nocopy(&ev, kp[k], kl[k], raw + b, (size_t)(e - b));
            }
        if (!ok || !overwritten) del_content(&ev, source_key, skl);                   /* :153-155 */
        if (!ok && ev.cnt == 0) { counters[0]++; continue; }                          /* ShouldEraseEvent */
        counters[3]++;
        sum += ev.allocated + ev.cnt;
    }
    free(ev.c); free(caps); free(kp); free(kl);
    return sum;
}
