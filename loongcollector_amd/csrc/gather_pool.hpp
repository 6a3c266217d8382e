// gather_pool.hpp -- a few helper threads for the host side of a LARGE host-fed batch (round 6).
//
// lc_regex_match_host on a batch of several chunks is a pipeline: gather the chunk's lines into pinned staging, copy up, match,
// copy down, copy the capture table out.  The device stages overlap; the two host copies do not, and one thread moves ~10 GB/s: the
// path stayed at 16-20 GB/s of payload (0.3 of PCIe Gen5) for three rounds, gather-bound.  A chunk is 32 MB: split four ways the
// copies keep up with the bus.  The pool is process-wide, started on first use, and used only by calls that have more than one chunk
// (what ProcessorRunner hands over -- one ~1000-line group -- never gets here: core/runner/ProcessorRunner.cpp:138-142); helper threads
// touch host memory only, never the HIP runtime.  LC_HOST_GATHER_THREADS: helpers + the caller (default 4, 1 = the caller alone).
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace lcgather {

class GatherPool {
   public:
    static GatherPool& instance() {
        static GatherPool* pool = new GatherPool();  // (never destroyed: helper threads may outlive static destruction order)
        return *pool;
    }
    unsigned width() const { return unsigned(mHelpers.size()) + 1; }
    // runs part(0) .. part(parts - 1), part 0 on the calling thread; returns when all are done.  One batch at a time.
    void run(unsigned parts, const std::function<void(unsigned)>& part) {
        if (parts <= 1 || mHelpers.empty()) {
            for (unsigned p = 0; p < parts; ++p) part(p);
            return;
        }
        std::unique_lock<std::mutex> turn(mTurn);  // (callers of different runner threads take turns: the helpers are shared)
        {
            std::lock_guard<std::mutex> g(mMutex);
            mPart = &part;
            mParts = parts;
            mNext = 1;
            mPending = parts - 1;
            ++mGeneration;
        }
        mWake.notify_all();
        part(0);
        std::unique_lock<std::mutex> lk(mMutex);
        // (the caller helps with what is left instead of waiting for a helper that is slow to wake)
        while (mNext < mParts) {
            const unsigned p = mNext++;
            lk.unlock();
            part(p);
            lk.lock();
            --mPending;
        }
        mDone.wait(lk, [&] { return mPending == 0; });
        mPart = nullptr;
    }

   private:
    GatherPool() {
        unsigned want = 4;
        if (const char* e = getenv("LC_HOST_GATHER_THREADS")) want = unsigned(atoi(e));
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && want > hw) want = hw;
        if (want < 1) want = 1;
        if (want > 16) want = 16;
        for (unsigned i = 1; i < want; ++i) mHelpers.emplace_back([this] { helperLoop(); });
        for (auto& t : mHelpers) t.detach();
    }
    void helperLoop() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mMutex);
        for (;;) {
            mWake.wait(lk, [&] { return mGeneration != seen; });
            seen = mGeneration;
            while (mPart && mNext < mParts) {
                const unsigned p = mNext++;
                const std::function<void(unsigned)>* fn = mPart;
                lk.unlock();
                (*fn)(p);
                lk.lock();
                if (--mPending == 0) mDone.notify_all();
            }
        }
    }
    std::vector<std::thread> mHelpers;
    std::mutex mTurn, mMutex;
    std::condition_variable mWake, mDone;
    const std::function<void(unsigned)>* mPart = nullptr;
    unsigned mParts = 0, mNext = 0, mPending = 0;
    uint64_t mGeneration = 0;
};

// dst[0, bytes) = src[0, bytes) in `ways` pieces (64-byte aligned cuts)
inline void parallelCopy(void* dst, const void* src, size_t bytes, unsigned ways) {
    if (ways <= 1 || bytes < (size_t(1) << 20)) {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t piece = ((bytes + ways - 1) / ways + 63) & ~size_t(63);
    GatherPool::instance().run(ways, [&](unsigned p) {
        const size_t lo = size_t(p) * piece;
        if (lo >= bytes) return;
        const size_t n = lo + piece < bytes ? piece : bytes - lo;
        std::memcpy(static_cast<char*>(dst) + lo, static_cast<const char*>(src) + lo, n);
    });
}

}  // namespace lcgather
