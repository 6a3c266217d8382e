/* oracle/grok_baseline.c -- TEST INFRASTRUCTURE ONLY (bench.py / tools/grok_bench.py cpu_baseline leg).
 *
 * ProcessorGrok.processGrok (plugins/processor/grok/processor_grok.go:148-194) for a batch of values, driven from C so that the
 * CPU baseline times the regex engine and not a Python loop: for every value, the Match entries in order; per entry
 * FindStringMatch, then FindNextMatch from the end of the previous match (:150-183); the first entry whose matches yield a
 * NON-EMPTY named capture wins (:185-191).  The engine is oracle/bt_regex.c (the backtracking restatement the parity tests use;
 * regexp2 is a backtracking engine too).  Results are only the winning entry per value: the fields are the oracle's business
 * (oracle/grok_oracle.py, which the parity gates compare against), this file exists to be TIMED.
 */
#include <stdint.h>
#include <stdlib.h>

#include "bt_regex.h"

/* progs[k]: compiled entry k; named[namedOff[k] .. namedOff[k+1]): its named groups (1-based group numbers).
 * pattern[i] = index of the first entry that yields a non-empty named capture for value i, -1: none, -3: an entry gave up
 * (complexity budget: regexp2's match time-out ends the walk, :156-160).  Returns the number of values some entry won. */
long orx_grok_first_match(const orx_prog* const* progs, int nProgs, const int32_t* named, const int32_t* namedOff,
                          const uint8_t* data, const uint32_t* off, const uint32_t* len, size_t n, int32_t* pattern) {
    int maxGroups = 0;
    for (int k = 0; k < nProgs; ++k) {
        const int g = orx_mark_count(progs[k]);
        if (g > maxGroups) maxGroups = g;
    }
    int32_t* caps = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(maxGroups + 1));
    if (!caps) return -1;
    long won = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* s = data + off[i];
        const size_t L = len[i];
        int32_t winner = -1;
        for (int k = 0; k < nProgs && winner == -1; ++k) {
            size_t start = 0;
            int got = 0;
            while (start <= L) {
                const int r = orx_search(progs[k], s, L, start, caps);
                if (r < 0) {
                    winner = -3;
                    break;
                }
                if (r == 0) break;
                for (int32_t j = namedOff[k]; j < namedOff[k + 1]; ++j) {
                    const int g = named[j];
                    if (caps[2 * g] >= 0 && caps[2 * g + 1] > caps[2 * g]) got = 1;
                }
                const int32_t b0 = caps[0], e0 = caps[1];
                start = (size_t)(e0 > b0 ? e0 : e0 + 1);
            }
            if (winner == -1 && got) winner = k;
        }
        pattern[i] = winner;
        won += winner >= 0;
    }
    free(caps);
    return won;
}
