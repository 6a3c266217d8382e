// group_combiner.hpp -- concurrent runner threads share ONE device batch (round 6).
//
// The reference calls a processor with one event group per call, synchronously, from process_thread_count runner threads that share
// the plugin instance (core/runner/ProcessorRunner.cpp:138-142).  A Grok batch costs the device about the same from 1 000 to 16 000
// values -- it waits for its longest value's chain of single-wave kernels, not for the chip -- so sixteen threads that each bring
// 1 000 values should cost ONE 3.4 ms batch, not sixteen.  Round 5 had a leader / follower group commit inside lcGrokMatchHost:
// whichever thread found nothing in flight ran what had gathered, up to two batches side by side.  Measured on the MI355X
// (profiles/round6_grok_inagent_before.txt): batches of 1 000 / 2 000 / 11 000 / 3 000 values in turn -- the first thread back
// always left alone --, two batches of 17 streams each over the runtime's 16 hardware queues slowing each other from 3.4 to 5-9 ms,
// every one of the sixteen threads growing a plan of its own (17 streams, events, a dozen device buffers) because any of them
// could lead, and the leader copying 12 MB of values alone: 0.99 M lines/s from sixteen threads against 0.30 M from one.
//
// Here:
//   * ONE worker thread per (processor, device) runs the batches: plan state, streams, pools and pinned staging exist once, not per
//     runner thread, and batches never overlap on the device;
//   * a batch is started when the device is free AND the callers that can be expected have arrived: the threads seen in the last
//     three batches (those of the batch that has just ended are all on their way back), bounded by a linger that ends `gapUs` after
//     the last arrival and `lingerUs` after the device became free.  One runner thread never waits: nobody else is expected;
//   * every caller copies its OWN values into the batch's pinned staging (the worker hands out the offsets), side by side;
//   * callers block on a condition variable; the contract stays ProcessorRunner's: synchronous, one group per call.
//
// No HIP in this file: the batch itself is a callback.  tests/native/combiner_race.cpp runs it under ThreadSanitizer.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace lccombine {

struct CombinerOptions {
    size_t maxLines = 65536;       // a batch beyond this is throughput-bound anyway: the rest waits for the next one
    unsigned gapUs = 100;          // the linger ends this long after the last arrival (or after the device became free) ...
    unsigned lingerUs = 500;       // ... and this long after the device became free, whatever arrives
};

struct CombinerStats {
    uint64_t batches = 0, jobs = 0, lines = 0, largestBatchJobs = 0, lingerExpired = 0;
    // where the worker's time went, microseconds: waiting for the first caller, lingering for the others, laying out the staging,
    // the callers' gather, the device trip, the callers taking their rows
    uint64_t usIdle = 0, usLinger = 0, usPlace = 0, usGather = 0, usRun = 0, usTakeOut = 0;
};

// Job: anything with  uint32_t lines() const.  The combiner owns nothing of a job; a job lives on its caller's stack.
template <class Job>
class GroupCombiner {
   public:
    // Runs on the worker thread.  `place(jobs)`: the batch is fixed -- lay out the staging, after which every caller runs
    // `gather(job)` on its own thread; `run(jobs)`: everything has been gathered -- the device trip; fills the jobs' results.
    struct Hooks {
        std::function<int(std::vector<Job*>&)> place;   // != 0: the batch fails with this code, run is not called
        std::function<void(Job&)> gather;               // on the CALLER's thread, between place and run
        std::function<int(std::vector<Job*>&)> run;     // the batch's return code, for every job
        std::function<void()> threadStart, threadEnd;   // on the worker thread: bind the device / release what it accumulated
    };

    GroupCombiner(Hooks hooks, CombinerOptions opts = CombinerOptions()) : mHooks(std::move(hooks)), mOpts(opts) {}
    ~GroupCombiner() { stop(); }
    GroupCombiner(const GroupCombiner&) = delete;
    GroupCombiner& operator=(const GroupCombiner&) = delete;

    // Blocks until the job's batch has run and `takeOut(job, rc)` has copied the job's results -- or, rc != 0, the batch's error text --
    // out of the batch's staging (on the caller's thread, while the staging is still the batch's: the worker does not touch it before
    // every caller of the batch is through).  Returns the batch's code.  A hook that throws fails the batch (-2), it never strands it.
    int submit(Job& job, const std::function<void(Job&, int)>& takeOut) {
        Slot slot;
        slot.job = &job;
        std::unique_lock<std::mutex> lk(mMutex);
        if (mStopping) return -1;
        if (!mWorker.joinable()) mWorker = std::thread([this] { workerLoop(); });
        noteCaller(callerId());
        mPending.push_back(&slot);
        mLastArrival = Clock::now();
        mWorkerCv.notify_one();
        // 1. the batch is laid out: copy my values in
        mCallerCv.wait(lk, [&] { return slot.phase != Phase::QUEUED; });
        if (slot.phase == Phase::GATHER) {
            lk.unlock();
            bool gathered = true;
            try {
                mHooks.gather(job);
            } catch (...) {
                gathered = false;
            }
            lk.lock();
            if (!gathered) mGatherFailed = true;
            slot.phase = Phase::GATHERED;
            if (--mGathering == 0) mWorkerCv.notify_one();
            // 2. the batch has run
            mCallerCv.wait(lk, [&] { return slot.phase == Phase::DONE; });
        }
        int rc = slot.rc;
        if (takeOut) {
            lk.unlock();
            try {
                takeOut(job, rc);
            } catch (...) {
                if (rc == 0) rc = -2;
            }
            lk.lock();
        }
        if (--mTakingOut == 0) mWorkerCv.notify_one();
        return rc;
    }

    void stop() {
        {
            std::lock_guard<std::mutex> lk(mMutex);
            mStopping = true;
        }
        mWorkerCv.notify_all();
        if (mWorker.joinable()) mWorker.join();
    }

    CombinerStats stats() {
        std::lock_guard<std::mutex> lk(mMutex);
        return mStats;
    }

   private:
    using Clock = std::chrono::steady_clock;
    enum class Phase { QUEUED, GATHER, GATHERED, DONE };
    struct Slot {
        Job* job = nullptr;
        Phase phase = Phase::QUEUED;
        int rc = 0;
    };
    // who calls: a process-wide number per thread; the combiner remembers in which batch it last saw each
    static uint64_t callerId() {
        static std::atomic<uint64_t> next{1};
        thread_local uint64_t mine = next.fetch_add(1, std::memory_order_relaxed);
        return mine;
    }
    void noteCaller(uint64_t id) {  // (mMutex held)
        for (auto& c : mCallers)
            if (c.first == id) {
                c.second = mBatchNo;
                return;
            }
        if (mCallers.size() >= 256)  // threads that came and went
            mCallers.erase(std::remove_if(mCallers.begin(), mCallers.end(), [&](const std::pair<uint64_t, uint64_t>& c) { return c.second + 3 <= mBatchNo; }),
                           mCallers.end());
        mCallers.emplace_back(id, mBatchNo);
    }
    size_t expectedCallers() const {  // threads seen while one of the last three batches was being formed or run
        size_t n = 0;
        for (const auto& c : mCallers) n += c.second + 3 > mBatchNo ? 1 : 0;
        return n;
    }

    // (gcc 11's ThreadSanitizer runtime does not intercept pthread_cond_clockwait, which is what a wait on the steady clock becomes:
    // it then loses track of the mutex and reports double locks.  Under the sanitizer the wait goes by the system clock.)
    void timedWait(std::unique_lock<std::mutex>& lk, Clock::duration d) {
#if defined(__SANITIZE_THREAD__)
        mWorkerCv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::duration_cast<std::chrono::system_clock::duration>(d));
#else
        mWorkerCv.wait_for(lk, d);
#endif
    }

    void workerLoop() {
        if (mHooks.threadStart) mHooks.threadStart();
        std::unique_lock<std::mutex> lk(mMutex);
        std::vector<Slot*> batch;
        std::vector<Job*> jobs;
        auto us = [](Clock::time_point a, Clock::time_point b) { return uint64_t(std::chrono::duration_cast<std::chrono::microseconds>(b - a).count()); };
        for (;;) {
            const auto tIdle = Clock::now();
            mWorkerCv.wait(lk, [&] { return mStopping || !mPending.empty(); });
            if (mPending.empty()) break;  // (stopping, and nobody is waiting)
            if (mStats.batches) mStats.usIdle += us(tIdle, Clock::now());
            // the linger: the callers of the previous batch are on their way back
            const auto first = Clock::now();
            bool expired = false;
            while (!mStopping && mPending.size() < expectedCallers()) {
                const auto now = Clock::now();
                const auto hard = first + std::chrono::microseconds(mOpts.lingerUs);
                const auto soft = std::max(mLastArrival, first) + std::chrono::microseconds(mOpts.gapUs);
                const auto until = std::min(hard, soft);
                if (now >= until) {
                    expired = true;
                    break;
                }
                timedWait(lk, until - now);
            }
            if (expired) ++mStats.lingerExpired;
            // the batch: in arrival order, up to maxLines (always at least one job)
            batch.clear();
            jobs.clear();
            size_t lines = 0;
            while (!mPending.empty() && (batch.empty() || lines + mPending.front()->job->lines() <= mOpts.maxLines)) {
                lines += mPending.front()->job->lines();
                batch.push_back(mPending.front());
                jobs.push_back(mPending.front()->job);
                mPending.erase(mPending.begin());
            }
            ++mBatchNo;
            lk.unlock();
            const auto t0 = Clock::now();
            auto t1 = t0, t2 = t0;
            int rc = 0;
            try {
                rc = mHooks.place ? mHooks.place(jobs) : 0;
            } catch (...) {
                rc = -2;
            }
            t1 = t2 = Clock::now();
            if (rc == 0) {
                lk.lock();
                mGathering = batch.size();
                mGatherFailed = false;
                for (Slot* s : batch) s->phase = Phase::GATHER;
                mCallerCv.notify_all();
                mWorkerCv.wait(lk, [&] { return mGathering == 0; });
                const bool gatherFailed = mGatherFailed;
                lk.unlock();
                t2 = Clock::now();
                try {
                    rc = gatherFailed ? -2 : mHooks.run(jobs);
                } catch (...) {
                    rc = -2;
                }
            }
            const auto t3 = Clock::now();
            lk.lock();
            mStats.usLinger += us(first, t0);
            mStats.usPlace += us(t0, t1);
            mStats.usGather += us(t1, t2);
            mStats.usRun += us(t2, t3);
            ++mStats.batches;
            mStats.jobs += batch.size();
            mStats.lines += lines;
            mStats.largestBatchJobs = std::max<uint64_t>(mStats.largestBatchJobs, batch.size());
            mTakingOut = batch.size();
            for (Slot* s : batch) {
                s->rc = rc;
                s->phase = Phase::DONE;
            }
            mCallerCv.notify_all();
            // the staging is the batch's until every caller has taken its rows (the slots live on the callers' stacks: not touched
            // after this wait)
            mWorkerCv.wait(lk, [&] { return mTakingOut == 0; });
            mStats.usTakeOut += us(t3, Clock::now());
        }
        lk.unlock();
        if (mHooks.threadEnd) mHooks.threadEnd();
    }

    Hooks mHooks;
    CombinerOptions mOpts;
    std::mutex mMutex;
    std::condition_variable mWorkerCv, mCallerCv;
    std::vector<Slot*> mPending;
    std::thread mWorker;
    bool mStopping = false;
    std::vector<std::pair<uint64_t, uint64_t>> mCallers;  // (caller id, number of the batch being formed when it last came in)
    uint64_t mBatchNo = 3;
    size_t mGathering = 0, mTakingOut = 0;
    bool mGatherFailed = false;
    Clock::time_point mLastArrival{};
    CombinerStats mStats;
};

}  // namespace lccombine
