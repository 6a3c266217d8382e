// tests/native/combiner_race.cpp -- TEST INFRASTRUCTURE ONLY: csrc/group_combiner.hpp (the group commit of lc_grok_match_host: concurrent
// runner threads share ONE device batch, core/runner/ProcessorRunner.cpp:138-142) under ThreadSanitizer, with a "device" that is a
// function: every job's values go into one shared staging block (each caller copies its own, side by side), the batch computes a checksum
// row per value, every caller takes its own rows out.  Checked: every job got exactly the rows of ITS values, whatever batch it travelled
// in; one thread never lingers; sixteen threads converge on batches of sixteen; a failing batch fails its jobs and nobody else's; stop()
// with callers in flight.
//      combiner_race THREADS ROUNDS
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "gather_pool.hpp"
#include "group_combiner.hpp"

namespace {
struct Job {
    std::vector<uint32_t> values;   // the caller's group
    std::vector<uint64_t> rows;     // what comes back: one row per value
    uint32_t* staged = nullptr;     // place(): where the values go
    const uint64_t* rowsSrc = nullptr;
    bool poison = false;            // the batch that carries this job fails
    bool throws = false;            // ... its gather throws (a hook that throws fails the batch, it never strands it)
    uint32_t lines() const { return uint32_t(values.size()); }
};
uint64_t rowOf(uint32_t v) { return uint64_t(v) * 0x9E3779B97F4A7C15ull ^ (v >> 3); }

struct Device {
    std::vector<uint32_t> stagingIn;
    std::vector<uint64_t> stagingOut;
    std::atomic<int> inFlight{0};
    std::atomic<int> overlaps{0};
    unsigned batchUs = 300;
};
}  // namespace

int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 16;
    const int rounds = argc > 2 ? atoi(argv[2]) : 200;
    Device dev;
    using Combiner = lccombine::GroupCombiner<Job>;
    std::atomic<bool> started{false}, ended{false};
    Combiner::Hooks hooks;
    hooks.threadStart = [&] { started = true; };
    hooks.threadEnd = [&] { ended = true; };
    hooks.place = [&](std::vector<Job*>& jobs) {
        size_t n = 0;
        for (Job* j : jobs) n += j->values.size();
        dev.stagingIn.assign(n, 0xDEADBEEFu);
        dev.stagingOut.assign(n, 0);
        size_t at = 0;
        for (Job* j : jobs) {
            j->staged = dev.stagingIn.data() + at;
            j->rowsSrc = dev.stagingOut.data() + at;
            at += j->values.size();
        }
        return 0;
    };
    hooks.gather = [&](Job& j) {
        if (j.throws) throw std::bad_alloc();
        std::memcpy(j.staged, j.values.data(), j.values.size() * 4);
    };
    hooks.run = [&](std::vector<Job*>& jobs) {
        if (dev.inFlight.fetch_add(1) != 0) dev.overlaps.fetch_add(1);   // batches never overlap on the device
        std::this_thread::sleep_for(std::chrono::microseconds(dev.batchUs));
        int rc = 0;
        for (Job* j : jobs) rc |= j->poison ? 7 : 0;
        for (size_t i = 0; i < dev.stagingIn.size(); ++i) dev.stagingOut[i] = rowOf(dev.stagingIn[i]);
        dev.inFlight.fetch_sub(1);
        return rc;
    };
    auto takeOut = [](Job& j, int rc) {
        if (rc == 0) j.rows.assign(j.rowsSrc, j.rowsSrc + j.values.size());
    };
    int bad = 0;

    // (a) one thread: never lingers (nobody else is expected)
    {
        lccombine::CombinerOptions o;
        o.gapUs = 20000;  // a linger would be seen: 20 ms per call
        o.lingerUs = 50000;
        Combiner c(hooks, o);
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 20; ++r) {
            Job j;
            j.values = {uint32_t(r), 7u, 9u};
            if (c.submit(j, takeOut) != 0 || j.rows.size() != 3 || j.rows[0] != rowOf(uint32_t(r))) ++bad;
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > 200) {
            printf("one thread lingered: %.1f ms for 20 calls\n", ms);
            ++bad;
        }
        if (c.stats().batches != 20 || c.stats().lingerExpired != 0) ++bad;
    }
    // (b) THREADS threads, ROUNDS groups each, on one combiner
    {
        Combiner c(hooks);
        std::vector<std::thread> pool;
        std::atomic<int> mismatches{0}, failedAsAsked{0}, failedUnasked{0};
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&, t] {
                uint32_t seed = 1000u * uint32_t(t) + 17u;
                for (int r = 0; r < rounds; ++r) {
                    Job j;
                    const uint32_t n = 1 + (seed = seed * 1664525u + 1013904223u) % 64;
                    for (uint32_t i = 0; i < n; ++i) j.values.push_back((uint32_t(t) << 24) | (uint32_t(r) << 8) | i);
                    j.poison = (t == 3 && r % 50 == 49);
                    const int rc = c.submit(j, takeOut);
                    if (rc != 0) {
                        (j.poison ? failedAsAsked : failedUnasked).fetch_add(1);  // (a batch that carries a poisoned job fails whole)
                        continue;
                    }
                    if (j.rows.size() != j.values.size()) {
                        mismatches.fetch_add(1);
                        continue;
                    }
                    for (uint32_t i = 0; i < n; ++i)
                        if (j.rows[i] != rowOf(j.values[i])) {
                            mismatches.fetch_add(1);
                            break;
                        }
                }
            });
        for (auto& th : pool) th.join();
        const lccombine::CombinerStats s = c.stats();
        printf("%d threads x %d groups: %llu batches, %.1f jobs per batch, largest %llu, linger expired %llu times, %d overlapping batches, %d mismatching jobs, %d failed as asked, %d failed with them\n",
               threads, rounds, (unsigned long long)s.batches, double(s.jobs) / double(s.batches ? s.batches : 1), (unsigned long long)s.largestBatchJobs,
               (unsigned long long)s.lingerExpired, dev.overlaps.load(), mismatches.load(), failedAsAsked.load(), failedUnasked.load());
        if (mismatches.load() || dev.overlaps.load() || s.jobs != uint64_t(threads) * uint64_t(rounds)) ++bad;
        if (failedAsAsked.load() != rounds / 50) ++bad;
        if (threads >= 4 && s.largestBatchJobs < uint64_t(threads) - 1) ++bad;                       // the threads find each other ...
        if (threads >= 4 && double(s.jobs) / double(s.batches) < 0.6 * threads) ++bad;              // ... and stay together
    }
    // (c) stop() while callers are in flight: everyone who was accepted gets an answer, later callers are refused
    {
        Combiner c(hooks);
        std::atomic<int> answered{0}, refused{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&] {
                for (int r = 0; r < 30; ++r) {
                    Job j;
                    j.values = {1u, 2u, 3u};
                    const int rc = c.submit(j, takeOut);
                    (rc == 0 ? answered : refused).fetch_add(1);
                    if (rc == 0 && (j.rows.size() != 3 || j.rows[2] != rowOf(3u))) ++bad;
                }
            });
        std::this_thread::sleep_for(std::chrono::milliseconds(3));
        c.stop();
        for (auto& th : pool) th.join();
        printf("stop in flight: %d answered, %d refused\n", answered.load(), refused.load());
        if (answered.load() + refused.load() != threads * 30 || refused.load() == 0) ++bad;
    }
    // (c2) a hook that throws on a caller's thread: the batch fails with -2 for everyone in it, the next batch runs
    {
        Combiner c(hooks);
        std::atomic<int> failed{0}, fine{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < 8; ++t)
            pool.emplace_back([&, t] {
                for (int r = 0; r < 40; ++r) {
                    Job j;
                    j.values = {uint32_t(t), uint32_t(r)};
                    j.throws = (t == 2 && r == 7);
                    const int rc = c.submit(j, takeOut);
                    if (rc == -2) failed.fetch_add(1);
                    else if (rc == 0 && j.rows.size() == 2 && j.rows[1] == rowOf(uint32_t(r))) fine.fetch_add(1);
                    else ++bad;
                }
            });
        for (auto& th : pool) th.join();
        printf("throwing gather: %d jobs failed with it, %d fine\n", failed.load(), fine.load());
        if (failed.load() < 1 || failed.load() + fine.load() != 8 * 40) ++bad;
    }
    // (d) csrc/gather_pool.hpp: four runner threads at once split their copies over the shared helpers
    {
        std::vector<std::thread> pool;
        std::atomic<int> wrong{0};
        for (int t = 0; t < 4; ++t)
            pool.emplace_back([&, t] {
                std::vector<uint8_t> src(size_t(3) << 20), dst(src.size() + 64);
                for (int r = 0; r < 12; ++r) {
                    for (size_t i = 0; i < src.size(); i += 4096) src[i] = uint8_t(i >> 12) ^ uint8_t(t * 16 + r);
                    std::fill(dst.begin(), dst.end(), 0xEE);
                    lcgather::parallelCopy(dst.data(), src.data(), src.size() - size_t(r), 4);
                    if (std::memcmp(dst.data(), src.data(), src.size() - size_t(r)) != 0 || dst[src.size() - size_t(r)] != 0xEE) wrong.fetch_add(1);
                }
            });
        for (auto& th : pool) th.join();
        printf("gather pool: %u wide, %d wrong copies\n", lcgather::GatherPool::instance().width(), wrong.load());
        if (wrong.load()) ++bad;
    }
    if (!started.load() || !ended.load()) ++bad;
    printf("%d checks failed\n", bad);
    return bad ? 1 : 0;
}
