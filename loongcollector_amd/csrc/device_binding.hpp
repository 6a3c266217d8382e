// device_binding.hpp -- which GPU a runner thread uses: the POLICY, without the HIP runtime (round 6: extracted from gpu_runtime.hip so that
// it runs -- against a two-device double, under ThreadSanitizer -- on a box without any GPU: tests/native/binding_race.cpp).
//
// "In-agent: map runner thread -> GPU (threadNo % nGPU)".  The reference calls Process from
// process_thread_count runner threads (core/runner/ProcessorRunner.cpp:138-142; the index is ProcessorRunner::GetThreadNo,
// ProcessorRunner.h:40, and selects the thread's regex copy, ProcessorParseRegexNative.cpp:255-257).  The index does not cross the
// C slot, and an agent never calls hipSetDevice: a fresh thread's current HIP device is 0, so through round 4 a plugin on an
// 8-GPU node ran on GPU 0.  Every HOST entry point (processors, lc_*_match_host, multiline, filter, pipeline) asks
// hostEntryDevice: the first call of a thread binds it -- by the process-wide policy -- and makes that device current for
// the thread; the thread's staging, streams and table uploads follow (they are per device already).
//
// Api: a type with static  int count();  bool get(int* dev);  bool set(int dev);  (the HIP runtime in gpu_runtime.hip; a double in tests).
// The error text of a failed call goes to *err.  Policy values: LC_BIND_* of include/lc_regex_gpu.h.
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

namespace lcbind {

constexpr int kInherit = 0, kRoundRobin = 1, kFixed = 2;   // == LC_BIND_INHERIT / LC_BIND_ROUND_ROBIN / LC_BIND_FIXED
constexpr int kOk = 0, kErrArg = -1, kErrNoDevice = -2, kErrRuntime = -3;

template <class Api, int MaxDevices>
class Binder {
   public:
    // Ordinals are dealt lowest-free-first and come BACK: when a thread releases its resources (lc_thread_release) or ends.  Through
    // round 5 an ordinal was the order of first entry, for good -- a short-lived helper thread that entered once consumed one and
    // skewed the deal for every runner thread behind it.  With runner threads that live as long as the process the deal is still
    // ProcessorRunner's threadNo % nGPU.
    struct Thread {
        Binder* owner = nullptr;
        int device = -1;           // bound device, -1 = not bound
        int ordinal = -1;          // this thread's ordinal (lowest free one at its first host entry), -1 = none taken
        int inherited = -1;        // kInherit: the current device as last asked from the runtime
        bool inheritOnly = false;  // this thread asked for kInherit itself: the process-wide policy does not bind it
        void unbind() {            // (a device the host chose itself -- setThreadDevice, kFixed -- holds no ordinal and stays)
            if (ordinal < 0) return;
            owner->returnOrdinal(uint32_t(ordinal));
            ordinal = -1;
            device = -1;
        }
        ~Thread() {
            if (ordinal >= 0 && owner) owner->returnOrdinal(uint32_t(ordinal));  // (plain host state: safe at any point of a process's life)
        }
    };

    static int deviceForOrdinal(uint32_t ordinal, int ndevices) { return ndevices > 0 ? int(ordinal % uint32_t(ndevices)) : -1; }

    int setPolicy(int policy, int device) {
        if (policy != kInherit && policy != kRoundRobin && policy != kFixed) return kErrArg;
        if (policy == kFixed && device < 0) return kErrArg;
        if (policy == kFixed) mFixedDevice.store(device);
        mPolicy.store(policy);
        return kOk;
    }
    int policy() {
        int p = mPolicy.load(std::memory_order_relaxed);
        if (p >= 0) return p;
        p = kRoundRobin;
        int fixedDev = 0;
        if (const char* e = getenv("LC_BIND_POLICY")) {  // inherit | rr | fixed:<d>
            if (!strcmp(e, "inherit")) p = kInherit;
            else if (!strncmp(e, "fixed:", 6)) {
                p = kFixed;
                fixedDev = atoi(e + 6);
            }
        }
        int expected = -1;
        if (mPolicy.compare_exchange_strong(expected, p)) {
            if (p == kFixed) mFixedDevice.store(fixedDev);
            return p;
        }
        return expected;
    }

    int setThreadDevice(Thread& t, int device, std::string* err) { return apply(t, device, err); }

    // -> the device (>= 0) or a negative error
    int bindThread(Thread& b, int pol, std::string* err) {
        b.owner = this;
        if (pol < 0) pol = policy();
        if (pol == kInherit) {
            b.device = -1;
            b.inheritOnly = true;
            int cur = 0;
            if (Api::count() <= 0) {
                *err = "no HIP device";
                return kErrNoDevice;
            }
            if (!Api::get(&cur)) return kErrRuntime;
            b.inherited = cur;
            return cur;
        }
        int want = 0;
        if (pol == kFixed) {
            want = mFixedDevice.load();
        } else if (pol == kRoundRobin) {
            const int n = Api::count();
            if (n <= 0) {
                *err = "no HIP device";
                return kErrNoDevice;
            }
            // a thread whose current device is not the runtime's default has been placed by its host (hipSetDevice, torch.cuda.set_device):
            // that is kept.  Device 0 is what a thread gets without asking -- those threads are dealt out by their ordinal.
            int cur = 0;
            if (!Api::get(&cur)) return kErrRuntime;
            if (cur != 0) want = cur;
            else {
                if (b.ordinal < 0) b.ordinal = int(takeOrdinal());
                want = deviceForOrdinal(uint32_t(b.ordinal), n);
            }
        } else {
            return kErrArg;
        }
        const int rc = apply(b, want, err);
        return rc == kOk ? want : rc;
    }

    // the device of a HOST entry point for thread `b` (bound on first use); kOk or an error
    int hostEntryDevice(Thread& b, int* dev, std::string* err) {
        b.owner = this;
        if (b.device >= 0) {
            // A host library (torch, another plugin) may have moved the thread's current device since the last group: asked PER CALL --
            // the runtime reads a thread-local, no lock -- because up to round 5 it was asked every 256th call, and the groups in between
            // were launched on the thread's cached streams of device A while device B was current.  The bound device is
            // made current again and STAYS current behind the call: INTEGRATION.md section 11 says so.
            int cur = -1;
            if (!Api::get(&cur)) return kErrRuntime;
            if (cur != b.device && !Api::set(b.device)) return kErrRuntime;
            *dev = b.device;
            return kOk;
        }
        if (b.inheritOnly || policy() == kInherit) {
            if (!Api::get(&b.inherited)) return kErrRuntime;
            *dev = b.inherited;
            return *dev < MaxDevices ? kOk : kErrArg;
        }
        const int d = bindThread(b, -1, err);
        if (d < 0) return d;
        *dev = d;
        return kOk;
    }

    // (tests) ordinals handed out and not returned
    size_t ordinalsInUse() {
        std::lock_guard<std::mutex> g(mOrdinalMutex);
        return size_t(mNextOrdinal) - mFreeOrdinals.size();
    }

   private:
    int apply(Thread& t, int device, std::string* err) {
        t.owner = this;
        const int n = Api::count();
        if (n <= 0) {
            *err = "no HIP device";
            return kErrNoDevice;
        }
        if (device < 0 || device >= n || device >= MaxDevices) {
            *err = "thread binding: device " + std::to_string(device) + " of " + std::to_string(n) + " visible";
            return kErrArg;
        }
        if (!Api::set(device)) return kErrRuntime;
        t.device = device;
        t.inheritOnly = false;
        return kOk;
    }
    uint32_t takeOrdinal() {
        std::lock_guard<std::mutex> g(mOrdinalMutex);
        if (!mFreeOrdinals.empty()) {
            const uint32_t o = mFreeOrdinals.back();
            mFreeOrdinals.pop_back();
            return o;
        }
        return mNextOrdinal++ & 0x7FFFFFFFu;
    }
    void returnOrdinal(uint32_t o) {
        std::lock_guard<std::mutex> g(mOrdinalMutex);
        mFreeOrdinals.insert(std::upper_bound(mFreeOrdinals.begin(), mFreeOrdinals.end(), o, std::greater<uint32_t>()), o);
    }

    std::atomic<int> mPolicy{-1};  // -1: not decided yet (LC_BIND_POLICY is read at first use)
    std::atomic<int> mFixedDevice{0};
    std::mutex mOrdinalMutex;
    uint32_t mNextOrdinal = 0;
    std::vector<uint32_t> mFreeOrdinals;  // kept sorted descending: back() is the lowest free one
};

}  // namespace lcbind
