// tests/native/race_driver.cpp -- TEST INFRASTRUCTURE ONLY: the product's HOST code under runner threads, for ThreadSanitizer.
//
// Runner threads share a processor instance (core/collection_pipeline/queue/ProcessQueueManager.cpp:167-205: queues are not pinned to
// threads; the reference keeps one boost::regex per thread for that reason, ProcessorParseRegexNative.cpp:255-257).  The product shares
// the instance and its device tables and keeps staging per thread; its counters are atomics.  This driver runs N threads on ONE instance
// of every processor -- parse (stitched and columnar), filter, the fused pipeline, the multiline splitter, the merge processor -- with
// the device calls answered by the doubles (pipeline_double.cpp / filter_double.cpp / multiline_double.cpp over the CPU oracle), checks
// every group against the answer the same code gave single-threaded, and the counters against their sums.  tests/test_host_races.py
// builds it with -fsanitize=thread and fails on any report.
#define MD_NO_REGEX_DOUBLES
#include "pipeline_double.cpp"
#include "multiline_double.cpp"

#include <atomic>
#include <functional>
#include <thread>

namespace {
const char* kLines[] = {"GET 200 curl/8.1", "POST 404 Mozilla/5.0", "GET 301 Googlebot/2.1", "HEAD 204 bot", "garbage", "", "GET 2000 x", "PUT 500 ",
                        "2024-01-04 boom", "  at com.example.A.b(A.java:1)", "[ERROR] x", "END", "DELETE 200 cafe/1.0"};
constexpr int kNLines = int(sizeof kLines / sizeof kLines[0]);
struct Rng {
    uint64_t s;
    uint32_t next() {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return uint32_t(s >> 33);
    }
};
std::string quote(const std::string& v) {
    std::string o = "\"";
    for (char c : v) {
        if (c == '"' || c == '\\') o += '\\';
        if (c == '\n') {
            o += "\\n";
            continue;
        }
        o += c;
    }
    return o + "\"";
}
// the merge processor joins events IN PLACE: its input is what the line splitter leaves -- views lying back to back in one buffer
// (md_merge_lines builds that group), not values a fixture reader copied one by one
std::string rawLines(Rng& r) {
    std::string v;
    const int m = 1 + int(r.next() % 30);
    for (int k = 0; k < m; ++k) v += std::string(k ? "\n" : "") + kLines[r.next() % kNLines];
    return v;
}
// one event per line, or one read buffer per event (buffers = true)
std::string fixture(Rng& r, bool buffers) {
    std::string o = "{\"events\":[";
    const int n = 1 + int(r.next() % (buffers ? 3 : 24));
    for (int i = 0; i < n; ++i) {
        std::string v;
        if (buffers) {
            const int m = int(r.next() % 20);
            for (int k = 0; k < m; ++k) v += std::string(kLines[r.next() % kNLines]) + "\n";
        } else {
            v = kLines[r.next() % kNLines];
        }
        o += std::string(i ? "," : "") + "{\"contents\":{\"content\":" + quote(v) + "},\"timestamp\":" + std::to_string(1700000000 + i) + ",\"type\":1}";
    }
    return o + "]}";
}
std::string take(char* p) {
    std::string s = p ? p : "<null>";
    std::free(p);
    return s;
}
struct Job {
    const char* name;
    int input;  // 0: one event per line (fixture JSON), 1: read buffers (fixture JSON), 2: raw lines for md_merge_lines
    std::function<std::string(const std::string&)> run;  // fixture JSON -> result JSON, on the SHARED instance
};
}  // namespace

int main(int argc, char** argv) {
    const int nThreads = argc > 1 ? std::atoi(argv[1]) : 8, perThread = argc > 2 ? std::atoi(argv[2]) : 60;
    char err[512];
    const char* parseCfg = "{\"SourceKey\":\"content\",\"Regex\":\"(\\\\w+) (\\\\d{3}) (.*)\",\"Keys\":[\"method\",\"status\",\"ua\"],\"KeepingSourceWhenParseFail\":true}";
    const std::string pipeCfg = std::string("{\"Split\":{\"SourceKey\":\"content\",\"SplitChar\":\"\\n\"},\"Parse\":") + parseCfg +
                                ",\"Filter\":{\"FilterKey\":[\"status\",\"ua\"],\"FilterRegex\":[\"2\\\\d\\\\d|30[14]\",\".*(?:bot|curl|x).*\"]},\"Fused\":true}";
    lc_processor_t* parse = nullptr;
    lc_filter_t* filter = nullptr;
    lc_pipeline_t* pipe = nullptr;
    lc_multiline_t* ml = nullptr;
    lc_merge_multiline_t* merge = nullptr;
    if (lc_processor_create(parseCfg, &parse, err, sizeof err) != LC_OK ||
        lc_filter_create("{\"ConditionExp\":{\"operator\":\"or\",\"operands\":[{\"type\":\"regex\",\"key\":\"content\",\"exp\":\"GET.*\"},{\"operator\":\"not\",\"operands\":[{\"type\":\"regex\",\"key\":\"content\",\"exp\":\".*\\\\d.*\"}]}]},\"DiscardingNonUTF8\":true}",
                         &filter, err, sizeof err) != LC_OK ||
        lc_pipeline_create(pipeCfg.c_str(), &pipe, err, sizeof err) != LC_OK) {
        std::fprintf(stderr, "create: %s\n", err);
        return 2;
    }
    const char* mlCfg = "{\"StartPattern\":\"\\\\d{4}-\\\\d{2}-\\\\d{2} .*\",\"ContinuePattern\":\"\\\\s+at\\\\s.*\",\"UnmatchedContentTreatment\":\"single_line\"}";
    const char* mergeCfg = "{\"MergeType\":\"regex\",\"StartPattern\":\"\\\\[\\\\w+\\\\].*\",\"EndPattern\":\"END$\",\"UnmatchedContentTreatment\":\"discard\"}";
    if (lc_multiline_create(mlCfg, std::strlen(mlCfg), &ml, err, sizeof err) != LC_OK ||
        lc_merge_multiline_create(mergeCfg, std::strlen(mergeCfg), &merge, err, sizeof err) != LC_OK) {
        std::fprintf(stderr, "create: %s\n", err);
        return 2;
    }
    auto viaGroup = [](const std::string& fx, const std::function<int(lc_event_group_t*)>& fn) {
        char e[256];
        lc_event_group_t* g = lc_group_from_json(fx.c_str(), e, sizeof e);
        if (!g) return std::string("<bad fixture>");
        const int rc = fn(g);
        std::string out = rc == LC_OK ? take(lc_group_to_json(g)) : "<rc " + std::to_string(rc) + ">";
        lc_group_free(g);
        return out;
    };
    std::vector<Job> jobs = {
        {"parse", 0, [&](const std::string& fx) { return viaGroup(fx, [&](lc_event_group_t* g) { return lc_processor_process(parse, g); }); }},
        {"columnar", 0, [&](const std::string& fx) {
             return viaGroup(fx, [&](lc_event_group_t* g) {
                 lc_columnar_t* c = nullptr;
                 const int rc = lc_processor_parse_columnar(parse, g, &c);
                 uint64_t sum = 0;
                 if (rc == LC_OK)
                     for (uint32_t i = 0; i < c->n_events; ++i) sum += c->content_bytes[i] + c->state[i];
                 lc_columnar_free(c);
                 return rc == LC_OK && sum != ~0ull ? LC_OK : LC_ERR_ARG;
             });
         }},
        {"filter", 0, [&](const std::string& fx) { return viaGroup(fx, [&](lc_event_group_t* g) { return lc_filter_process(filter, lc_group_native(g)); }); }},
        {"pipeline", 1, [&](const std::string& fx) { return viaGroup(fx, [&](lc_event_group_t* g) { return lc_pipeline_process(pipe, g); }); }},
        {"multiline", 1, [&](const std::string& fx) { return viaGroup(fx, [&](lc_event_group_t* g) { return lc_multiline_process_group(ml, lc_group_native(g)); }); }},
        {"merge", 2, [&](const std::string& raw) {
             char e[256];
             return take(md_merge_lines(merge, reinterpret_cast<const uint8_t*>(raw.data()), raw.size(), "content", nullptr, 0, nullptr, 0, e, sizeof e));
         }},
    };
    // the fixtures of every thread and their answers, single-threaded
    std::vector<std::vector<std::string>> fixtures(size_t(nThreads) * jobs.size()), answers(fixtures.size());
    for (int t = 0; t < nThreads; ++t)
        for (size_t j = 0; j < jobs.size(); ++j) {
            Rng r{uint64_t(1000 * t + 7 * j + 1)};
            auto& fx = fixtures[size_t(t) * jobs.size() + j];
            auto& an = answers[size_t(t) * jobs.size() + j];
            for (int k = 0; k < perThread; ++k) {
                fx.push_back(jobs[j].input == 2 ? rawLines(r) : fixture(r, jobs[j].input == 1));
                an.push_back(jobs[j].run(fx.back()));
            }
        }
    uint64_t parseBefore[LC_CNT_COUNT], pipeParse[LC_CNT_COUNT], pipeBefore[LC_PIPE_CNT_COUNT], filterBefore[2], mlBefore[3], mergeBefore[2];
    lc_processor_counters(parse, parseBefore);
    lc_pipeline_counters(pipe, pipeParse, pipeBefore);
    lc_filter_counters(filter, filterBefore);
    lc_multiline_counters(ml, mlBefore);
    lc_merge_multiline_counters(merge, mergeBefore);
    // ... and now all threads at once on the same instances
    std::atomic<uint64_t> mismatches{0};
    std::vector<std::thread> threads;
    for (int t = 0; t < nThreads; ++t)
        threads.emplace_back([&, t] {
            for (int k = 0; k < perThread; ++k)
                for (size_t j = 0; j < jobs.size(); ++j) {
                    const size_t at = size_t(t) * jobs.size() + j;
                    if (jobs[j].run(fixtures[at][size_t(k)]) != answers[at][size_t(k)]) ++mismatches;
                }
        });
    for (auto& th : threads) th.join();
    uint64_t parseAfter[LC_CNT_COUNT], pipeAfter[LC_PIPE_CNT_COUNT], filterAfter[2], mlAfter[3], mergeAfter[2];
    lc_processor_counters(parse, parseAfter);
    lc_pipeline_counters(pipe, pipeParse, pipeAfter);
    lc_filter_counters(filter, filterAfter);
    lc_multiline_counters(ml, mlAfter);
    lc_merge_multiline_counters(merge, mergeAfter);
    // the second pass saw exactly the first pass's groups: every counter doubled
    uint64_t bad = 0;
    for (int i : {int(LC_CNT_DISCARDED_EVENTS), int(LC_CNT_OUT_FAILED_EVENTS), int(LC_CNT_OUT_SUCCESSFUL_EVENTS)}) bad += parseAfter[i] != 2 * parseBefore[i];
    for (int i : {int(LC_PIPE_FILTER_IN_EVENTS), int(LC_PIPE_FILTER_OUT_EVENTS), int(LC_PIPE_LINES), int(LC_PIPE_SURVIVORS)}) bad += pipeAfter[i] != 2 * pipeBefore[i];
    for (int i = 0; i < 2; ++i) bad += filterAfter[i] != 2 * filterBefore[i];
    for (int i = 0; i < 3; ++i) bad += mlAfter[i] != 2 * mlBefore[i];
    for (int i = 0; i < 2; ++i) bad += mergeAfter[i] != 2 * mergeBefore[i];
    std::printf("threads %d, groups per thread and processor %d: %llu mismatching groups, %llu counters off; parse ok %llu, pipeline lines %llu survivors %llu, merged %llu\n",
                nThreads, perThread, (unsigned long long)mismatches.load(), (unsigned long long)bad, (unsigned long long)parseAfter[LC_CNT_OUT_SUCCESSFUL_EVENTS],
                (unsigned long long)pipeAfter[LC_PIPE_LINES], (unsigned long long)pipeAfter[LC_PIPE_SURVIVORS], (unsigned long long)mergeAfter[0]);
    lc_processor_destroy(parse);
    lc_filter_destroy(filter);
    lc_pipeline_destroy(pipe);
    lc_multiline_free(ml);
    lc_merge_multiline_free(merge);
    return mismatches.load() || bad ? 1 : 0;
}
