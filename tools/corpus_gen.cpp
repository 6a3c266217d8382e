// tools/corpus_gen.cpp -- BENCH / TEST TOOLING (not product): the headline corpus of SURVEY.md section 8(d), generated line by line.
//
//   "Lines: synthetic Apache/nginx-combined following remote_file_benchmark.py:27 field recipe (ipv4 - - [dd/Mon/YYYY:HH:MM:SS +0000]
//    "METHOD url HTTP/1.1" status bytes "referer" "user-agent"; for R_A add the request_time/length numeric fields in its order), URL and
//    UA padded with [A-Za-z0-9/%._-] so each line is exactly 512 B excluding \n; every line must fully match.  PRNG: std::mt19937_64,
//    seed 20260921; IPv4 octets U[1,254]; method in {GET 70 %, POST 20 %, PUT 5 %, DELETE 5 %}; status in {200 80 %, 304, 404, 500, 502};
//    bytes U[1,10000]."
//
// Through round 4 the bench drew its 1 Mi lines with replacement from a pool of 8 192 lines made with numpy's MT19937 -- not the stated
// recipe.  Here every line is generated on its own from ONE std::mt19937_64 stream; the engine's raw 64-bit
// outputs are reduced with % (std::uniform_int_distribution is implementation-defined: the corpus must not depend on the C++ library).
// Built by loongcollector_amd/build.py into loongcollector_amd/lib/libcorpus_gen.so; loongcollector_amd/corpus.py apache_lines().
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

namespace {
const char kPad[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789/%._-";
const char* const kMonths[12] = {"Jan", "Feb", "Mar", "Apr", "May", "Jun", "Jul", "Aug", "Sep", "Oct", "Nov", "Dec"};

struct Gen {
    std::mt19937_64 rng;
    explicit Gen(uint64_t seed) : rng(seed) {}
    uint64_t below(uint64_t n) { return rng() % n; }               // [0, n)
    uint64_t range(uint64_t lo, uint64_t hi) { return lo + below(hi - lo + 1); }  // [lo, hi]
    size_t pad(char* out, size_t n) {
        for (size_t i = 0; i < n; ++i) out[i] = kPad[below(sizeof kPad - 1)];
        return n;
    }
};

// one line of exactly lineBytes bytes (no separator) into out; kind 'A': the 10-group doc regex, 'B': the 11-group benchmark regex
bool oneLine(Gen& g, char kind, size_t lineBytes, char* out) {
    char head[128], mid[160];
    const unsigned m = unsigned(g.below(100));
    const char* method = m < 70 ? "GET" : m < 90 ? "POST" : m < 95 ? "PUT" : "DELETE";
    const unsigned st = unsigned(g.below(100));
    const char* status = st < 80 ? "200" : st < 85 ? "304" : st < 90 ? "404" : st < 95 ? "500" : "502";
    const int nHead = std::snprintf(head, sizeof head, "%u.%u.%u.%u - - [%02u/%s/%04u:%02u:%02u:%02u %s] \"%s /", unsigned(g.range(1, 254)),
                                    unsigned(g.range(1, 254)), unsigned(g.range(1, 254)), unsigned(g.range(1, 254)), unsigned(g.range(1, 28)),
                                    kMonths[g.below(12)], unsigned(g.range(2020, 2026)), unsigned(g.below(24)), unsigned(g.below(60)),
                                    unsigned(g.below(60)), kind == 'A' ? "+0800" : "+0000", method);
    const unsigned bytes = unsigned(g.range(1, 10000));
    int nMid;
    if (kind == 'A') {  // ... "METHOD url" request_time request_length status length "ref" "ua"   (length: a number, or "-" one time in ten)
        char len[16];
        if (g.below(10)) std::snprintf(len, sizeof len, "%u", bytes);
        else std::snprintf(len, sizeof len, "-");
        nMid = std::snprintf(mid, sizeof mid, "\" %u.%03u %u %s %s \"https://example.com/", unsigned(g.below(10)), unsigned(g.below(1000)),
                             unsigned(g.range(1, 99999)), status, len);
    } else {
        nMid = std::snprintf(mid, sizeof mid, " HTTP/1.1\" %s %u \"https://example.com/", status, bytes);
    }
    static const char tail[] = "\" \"Mozilla/5.0 ";
    const size_t fixed = size_t(nHead) + size_t(nMid) + (sizeof tail - 1) + 1;
    if (lineBytes < fixed + 3) return false;
    const size_t free_ = lineBytes - fixed;
    const size_t ref = size_t(g.below(free_ - 2 < 40 ? free_ - 2 : 40));
    const size_t rest = free_ - ref;
    const size_t url = 1 + size_t(g.below(rest - 1));
    const size_t ua = rest - url;
    char* p = out;
    std::memcpy(p, head, size_t(nHead));
    p += nHead;
    p += g.pad(p, url);
    std::memcpy(p, mid, size_t(nMid));
    p += nMid;
    p += g.pad(p, ref);
    std::memcpy(p, tail, sizeof tail - 1);
    p += sizeof tail - 1;
    p += g.pad(p, ua);
    *p++ = '"';
    return size_t(p - out) == lineBytes;
}
}  // namespace

// n lines of lineBytes bytes, each followed by '\n', into out[n * (lineBytes + 1)]; returns 0, or -1 if lineBytes is too small
extern "C" int lc_corpus_apache_lines(char kind, uint64_t nLines, uint64_t lineBytes, uint64_t seed, char* out) {
    Gen g(seed);
    for (uint64_t i = 0; i < nLines; ++i) {
        char* line = out + i * (lineBytes + 1);
        if (!oneLine(g, kind, size_t(lineBytes), line)) return -1;
        line[lineBytes] = '\n';
    }
    return 0;
}
