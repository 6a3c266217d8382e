// tools/inagent_bench.cpp -- the in-agent calling shape, natively: N runner threads (core/runner/ProcessorRunner.cpp:138-142)
// each handing ~1000-line event groups to ONE shared plugin instance through the C ABI (lc_processor_process), plus the
// match entry point alone (lc_regex_match_host_views) on the same groups, to see what the host side costs.
//   g++ -O2 -std=c++17 -I include tools/inagent_bench.cpp -o scratch/inagent_bench -L loongcollector_amd/lib -llc_regex_gpu -lpthread
//   LD_LIBRARY_PATH=loongcollector_amd/lib scratch/inagent_bench [lines] [group_lines] [threads...]
#include <sys/resource.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "lc_processor.h"
#include "lc_regex_gpu.h"

static const char* kRegexA =
    "([\\d\\.]+) \\S+ \\S+ \\[(\\S+) \\S+\\] \\\"(\\w+) ([^\\\\\"]*)\\\" ([\\d\\.]+) (\\d+) (\\d+) (\\d+|-) \\\"([^\\\\\"]*)\\\" \\\"([^\\\\\"]*)\\\"";

static std::string makeLine(unsigned seed) {  // a 512-byte Apache-combined line regex A matches
    char head[256];
    snprintf(head, sizeof head, "10.%u.%u.%u - - [25/Jun/2024:23:59:%02u +0800] \"GET /api/v%u/items/%u?", seed & 255, (seed >> 8) & 255,
             (seed >> 16) & 255, seed % 60, seed % 7, seed);
    std::string s = head;
    std::string tail = "\" 0.123 " + std::to_string(100 + seed % 900) + " 200 " + std::to_string(1000 + seed % 9000) +
                       " \"http://example.com/ref\" \"Mozilla/5.0 (X11; Linux x86_64)\"";
    while (s.size() + tail.size() < 512) s.push_back(char('a' + (s.size() * 7 + seed) % 26));
    return s + tail;
}

// all runner threads warm up (their first call allocates pinned staging, a stream ...), then start the timed part together
struct StartGate {
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    std::atomic<long long> lastEndNs{0};  // latest moment a thread finished its groups (thread teardown is not part of the figure)
    void finished() {
        const long long t = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        long long cur = lastEndNs.load();
        while (t > cur && !lastEndNs.compare_exchange_weak(cur, t)) {}
    }
    double secondsSince(std::chrono::steady_clock::time_point t0) const {
        return double(lastEndNs.load() - std::chrono::duration_cast<std::chrono::nanoseconds>(t0.time_since_epoch()).count()) * 1e-9;
    }
    void arrive() {
        ready.fetch_add(1);
        while (!go.load()) std::this_thread::yield();
    }
    // waits until all n threads have warmed up, then starts the clock and lets them go
    std::chrono::steady_clock::time_point open(int n) {
        while (ready.load() < n) std::this_thread::yield();
        const auto t0 = std::chrono::steady_clock::now();
        go.store(true);
        return t0;
    }
};

int main(int argc, char** argv) {
    // INAGENT_REGEX: another pattern with ten groups over the same lines (round 6: one that runs on the device backtracking engine)
    if (const char* alt = getenv("INAGENT_REGEX")) kRegexA = alt;
    const unsigned nLines = argc > 1 ? unsigned(atoi(argv[1])) : 256000;
    const unsigned groupLines = argc > 2 ? unsigned(atoi(argv[2])) : 1000;
    std::vector<int> threadCounts;
    for (int i = 3; i < argc; ++i) threadCounts.push_back(atoi(argv[i]));
    if (threadCounts.empty()) threadCounts = {1, 2, 4, 8, 16, 32};
    std::vector<uint8_t> data;
    std::vector<uint32_t> off, len;
    for (unsigned i = 0; i < nLines; ++i) {
        const std::string l = makeLine(i * 2654435761u);
        off.push_back(uint32_t(data.size()));
        len.push_back(uint32_t(l.size()));
        data.insert(data.end(), l.begin(), l.end());
        data.push_back('\n');
    }
    const std::string cfg = std::string("{\"SourceKey\":\"content\",\"Regex\":\"") + [&] {
        std::string e;
        for (const char* p = kRegexA; *p; ++p) {
            if (*p == '\\' || *p == '"') e.push_back('\\');
            e.push_back(*p);
        }
        return e;
    }() + "\",\"Keys\":[\"ip\",\"time\",\"method\",\"url\",\"request_time\",\"request_length\",\"status\",\"length\",\"ref_url\",\"browser\"]}";
    char err[256];
    lc_processor_t* proc = nullptr;
    if (lc_processor_create(cfg.c_str(), &proc, err, sizeof err) != 0) {
        fprintf(stderr, "create: %s\n", err);
        return 2;
    }
    lc_regex_t* re = nullptr;
    if (lc_regex_compile(kRegexA, strlen(kRegexA), 0, LC_ENGINE_AUTO, &re, err, sizeof err) != 0) {
        fprintf(stderr, "compile: %s\n", err);
        return 2;
    }
    const unsigned nGroups = nLines / groupLines;
    const double payload = double(nGroups) * groupLines * 512.0;
    for (int T : threadCounts) {
        // (a) the match entry point alone: views of the lines of each group
        {
            std::vector<std::thread> th;
            StartGate gate;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    std::vector<const uint8_t*> ptr(groupLines);
                    std::vector<int32_t> caps(size_t(groupLines) * 20);
                    std::vector<uint8_t> st(groupLines);
                    for (unsigned i = 0; i < groupLines; ++i) ptr[i] = data.data() + off[i];
                    lc_regex_match_host_views(re, ptr.data(), &len[0], groupLines, 10, caps.data(), st.data());
                    gate.arrive();
                    for (unsigned g = unsigned(t); g < nGroups; g += unsigned(T)) {
                        for (unsigned i = 0; i < groupLines; ++i) ptr[i] = data.data() + off[g * groupLines + i];
                        if (lc_regex_match_host_views(re, ptr.data(), &len[g * groupLines], groupLines, 10, caps.data(), st.data()) != 0) {
                            fprintf(stderr, "match: %s\n", lc_last_error());
                            exit(3);
                        }
                    }
                    gate.finished();
                });
            const auto t0 = gate.open(T);
            for (auto& x : th) x.join();
            const double dt = gate.secondsSince(t0);
            printf("threads %2d  match_host_views only : %8.1f MB/s  (%.1f us per %u-line group per thread)\n", T, payload / dt / 1e6,
                   dt * 1e6 * T / nGroups, groupLines);
        }
        // (b) the plugin: groups built beforehand (the reader's job), processed by T threads
        {
            std::vector<lc_event_group_t*> groups(nGroups);
            for (unsigned g = 0; g < nGroups; ++g)
                groups[g] = lc_group_from_lines(data.data(), &off[g * groupLines], &len[g * groupLines], groupLines, "content");
            std::vector<lc_event_group_t*> warm{size_t(T), nullptr};
            for (int t = 0; t < T; ++t) warm[size_t(t)] = lc_group_from_lines(data.data(), &off[0], &len[0], groupLines, "content");
            std::vector<std::thread> th;
            StartGate gate;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    lc_processor_process(proc, warm[size_t(t)]);
                    gate.arrive();
                    for (unsigned g = unsigned(t); g < nGroups; g += unsigned(T))
                        if (lc_processor_process(proc, groups[g]) != 0) {
                            fprintf(stderr, "process failed\n");
                            exit(3);
                        }
                    gate.finished();
                });
            struct rusage ru0, ru1;
            getrusage(RUSAGE_SELF, &ru0);
            const auto t0 = gate.open(T);
            for (auto& x : th) x.join();
            getrusage(RUSAGE_SELF, &ru1);
            const double dt = gate.secondsSince(t0);
            for (auto* g : warm) lc_group_free(g);
            uint64_t c[LC_CNT_COUNT];
            lc_processor_counters(proc, c);
            printf("threads %2d  lc_processor_process   : %8.1f MB/s  (%.1f us per group per thread; %.0f minor faults per group; %llu events parsed so far)\n", T,
                   payload / dt / 1e6, dt * 1e6 * T / nGroups, double(ru1.ru_minflt - ru0.ru_minflt) / nGroups,
                   (unsigned long long)c[LC_CNT_OUT_SUCCESSFUL_EVENTS]);
            for (auto* g : groups) lc_group_free(g);
        }
        // (c) the same with a bounded number of groups alive -- what an agent's queues allow: WINDOW groups per thread are built
        // (the reader's job, untimed), processed by the T runner threads (timed), dropped (the flusher's job, untimed), round after
        // round.  Part (b) keeps all its groups alive, so every arena chunk a stitch takes is memory nobody has touched yet: 108
        // first-touch page faults per 1000-event group.  Here the chunks of the groups dropped a round ago come back through the
        // event model's pool (csrc/event_model.hpp ArenaChunkPool).  INAGENT_WINDOW=0 skips this part.
        const unsigned window = getenv("INAGENT_WINDOW") ? unsigned(atoi(getenv("INAGENT_WINDOW"))) : 16u;
        if (window) {
            const unsigned perRound = window * unsigned(T);
            const unsigned rounds = std::max(2u, nGroups / perRound);
            std::vector<lc_event_group_t*> cur(perRound, nullptr);
            std::atomic<unsigned> roundNo{0}, done{0};
            std::atomic<bool> stop{false};
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    lc_event_group_t* w = lc_group_from_lines(data.data(), &off[0], &len[0], groupLines, "content");
                    lc_processor_process(proc, w);  // this thread's first call: staging, stream
                    lc_group_free(w);
                    done.fetch_add(1);
                    for (unsigned r = 1;; ++r) {
                        while (roundNo.load(std::memory_order_acquire) < r && !stop.load()) std::this_thread::yield();
                        if (stop.load()) return;
                        for (unsigned k = 0; k < window; ++k)
                            if (lc_processor_process(proc, cur[size_t(t) * window + k]) != 0) {
                                fprintf(stderr, "process failed\n");
                                exit(3);
                            }
                        done.fetch_add(1, std::memory_order_release);
                    }
                });
            while (done.load() < unsigned(T)) std::this_thread::yield();
            double timed = 0;
            long faults = 0;
            unsigned next = 0;
            for (unsigned r = 1; r <= rounds; ++r) {
                for (unsigned k = 0; k < perRound; ++k) {
                    const unsigned g = next++ % nGroups;
                    cur[k] = lc_group_from_lines(data.data(), &off[g * groupLines], &len[g * groupLines], groupLines, "content");
                }
                done.store(0);
                struct rusage ru0, ru1;
                getrusage(RUSAGE_SELF, &ru0);
                const auto t0 = std::chrono::steady_clock::now();
                roundNo.store(r, std::memory_order_release);
                while (done.load(std::memory_order_acquire) < unsigned(T)) std::this_thread::yield();
                const auto t1 = std::chrono::steady_clock::now();
                getrusage(RUSAGE_SELF, &ru1);
                if (r > 1) {  // the first round fills the pool
                    timed += std::chrono::duration<double>(t1 - t0).count();
                    faults += ru1.ru_minflt - ru0.ru_minflt;
                }
                for (auto*& g : cur) {
                    lc_group_free(g);
                    g = nullptr;
                }
            }
            stop.store(true);
            for (auto& x : th) x.join();
            const double groupsTimed = double(rounds - 1) * perRound;
            printf("threads %2d  lc_processor_process, %u groups alive per thread: %8.1f MB/s  (%.1f us per group per thread; %.0f minor faults per group)\n",
                   T, window, groupsTimed * groupLines * 512.0 / timed / 1e6, timed * 1e6 * T / groupsTimed, double(faults) / groupsTimed);
        }
        fflush(stdout);
    }
    lc_regex_free(re);
    lc_processor_destroy(proc);
    return 0;
}
