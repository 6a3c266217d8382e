// tests/native/binding_race.cpp -- TEST INFRASTRUCTURE ONLY: csrc/device_binding.hpp (which GPU a runner thread uses: SURVEY.md section 8(e),
// ProcessorRunner.h:40 "threadNo % nGPU") against a HIP double that reports TWO devices, under ThreadSanitizer, on a box without any GPU --
// the CPU twin of the multi-device branches of tests/test_gpu_binding.py, which no box of this project has ever been able to run.
//      binding_race
#include <atomic>
#include <cstdio>
#include <set>
#include <thread>
#include <vector>

#include "device_binding.hpp"

namespace {
std::atomic<int> gDevices{2};
thread_local int tCurrent = 0;      // a fresh thread's current device is 0, as in the HIP runtime
std::atomic<uint64_t> gSets{0};
struct FakeApi {
    static int count() { return gDevices.load(); }
    static bool get(int* d) {
        *d = tCurrent;
        return true;
    }
    static bool set(int d) {
        if (d < 0 || d >= gDevices.load()) return false;
        tCurrent = d;
        gSets.fetch_add(1);
        return true;
    }
};
typedef lcbind::Binder<FakeApi, 16> Binder;
int bad = 0;
#define CHECK(c)                                                     \
    do {                                                             \
        if (!(c)) {                                                  \
            printf("FAILED line %d: %s\n", __LINE__, #c);            \
            ++bad;                                                   \
        }                                                            \
    } while (0)
}  // namespace

int main() {
    // (a) the default deal: the k-th thread to enter gets device k mod 2; long-lived threads keep it; the bound device stays current
    {
        Binder B;
        std::vector<int> got(8, -1);
        std::vector<std::thread> pool;
        std::atomic<int> turn{0};
        for (int t = 0; t < 8; ++t)
            pool.emplace_back([&, t] {
                while (turn.load() != t) std::this_thread::yield();  // (enter in order: the deal is by order of first entry)
                Binder::Thread me;
                std::string err;
                int dev = -1;
                CHECK(B.hostEntryDevice(me, &dev, &err) == lcbind::kOk);
                got[size_t(t)] = dev;
                turn.fetch_add(1);
                CHECK(tCurrent == dev);
                for (int k = 0; k < 1000; ++k) {
                    int d2 = -1;
                    if (k % 97 == 5) tCurrent = 1 - dev;  // another library on the thread moves its current device ...
                    CHECK(B.hostEntryDevice(me, &d2, &err) == lcbind::kOk && d2 == dev && tCurrent == dev);  // ... the next call moves it back
                }
                while (turn.load() < 8) std::this_thread::yield();  // (nobody leaves before everyone has entered: ordinals 0..7)
            });
        for (auto& th : pool) th.join();
        for (int t = 0; t < 8; ++t) CHECK(got[size_t(t)] == t % 2);
        CHECK(B.ordinalsInUse() == 0);  // every thread has ended: its ordinal came back
    }
    // (b) helper threads that come and go do not skew the deal: 200 of them, then two runner threads still get devices 0 and 1
    {
        Binder B;
        for (int i = 0; i < 200; ++i) {
            std::thread([&] {
                Binder::Thread me;
                std::string err;
                int dev = -1;
                CHECK(B.hostEntryDevice(me, &dev, &err) == lcbind::kOk && dev == 0);  // (alone: always the lowest free ordinal)
            }).join();
        }
        CHECK(B.ordinalsInUse() == 0);
        Binder::Thread a, b;  // (two "threads" on this one: the state is per Thread object)
        std::string err;
        int da = -1, db = -1;
        tCurrent = 0;
        CHECK(B.hostEntryDevice(a, &da, &err) == lcbind::kOk && da == 0);
        tCurrent = 0;
        CHECK(B.hostEntryDevice(b, &db, &err) == lcbind::kOk && db == 1);
        a.unbind();  // lc_thread_release: the ordinal goes back, the next entry takes it again
        CHECK(B.ordinalsInUse() == 1);
        tCurrent = 0;
        CHECK(B.hostEntryDevice(a, &da, &err) == lcbind::kOk && da == 0);
    }
    // (c) the other policies; a device the host chose holds no ordinal; errors
    {
        Binder B;
        std::string err;
        Binder::Thread t1, t2, t3;
        int dev = -1;
        CHECK(B.setPolicy(lcbind::kFixed, 1) == lcbind::kOk);
        tCurrent = 0;
        CHECK(B.hostEntryDevice(t1, &dev, &err) == lcbind::kOk && dev == 1 && tCurrent == 1 && t1.ordinal < 0);
        CHECK(B.setPolicy(lcbind::kInherit, 0) == lcbind::kOk);
        tCurrent = 1;
        CHECK(B.hostEntryDevice(t2, &dev, &err) == lcbind::kOk && dev == 1);
        tCurrent = 0;
        CHECK(B.hostEntryDevice(t2, &dev, &err) == lcbind::kOk && dev == 0);  // (inherit: whatever is current, asked per call)
        CHECK(B.setThreadDevice(t3, 1, &err) == lcbind::kOk && t3.device == 1 && t3.ordinal < 0);
        t3.unbind();
        CHECK(t3.device == 1);  // (an explicit choice survives lc_thread_release)
        CHECK(B.setThreadDevice(t3, 2, &err) == lcbind::kErrArg && err.find("device 2 of 2") != std::string::npos);
        CHECK(B.setPolicy(7, 0) == lcbind::kErrArg && B.setPolicy(lcbind::kFixed, -1) == lcbind::kErrArg);
        CHECK(B.setPolicy(lcbind::kRoundRobin, 0) == lcbind::kOk);
        Binder::Thread placed;
        tCurrent = 1;  // a thread its host placed (hipSetDevice, torch.cuda.set_device) keeps its device and takes no ordinal
        CHECK(B.hostEntryDevice(placed, &dev, &err) == lcbind::kOk && dev == 1 && placed.ordinal < 0);
        gDevices.store(0);
        Binder::Thread none;
        CHECK(B.hostEntryDevice(none, &dev, &err) == lcbind::kErrNoDevice);
        gDevices.store(2);
        CHECK(Binder::deviceForOrdinal(5, 2) == 1 && Binder::deviceForOrdinal(5, 0) == -1);
    }
    // (d) sixteen threads enter at once, release and re-enter, 2 000 times each: every call lands on one of the two devices, the
    //     thread's current device is the one returned, ordinals never leak
    {
        Binder B;
        std::atomic<int> onDev[2] = {{0}, {0}};
        std::atomic<int> entered{0}, finished{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < 16; ++t)
            pool.emplace_back([&] {
                Binder::Thread me;
                std::string err;
                {
                    int dev = -1;  // (everyone holds an ordinal before anyone starts to release and re-enter)
                    if (B.hostEntryDevice(me, &dev, &err) != lcbind::kOk) ++bad;
                    entered.fetch_add(1);
                    while (entered.load() < 16) std::this_thread::yield();
                }
                for (int k = 0; k < 2000; ++k) {
                    int dev = -1;
                    if (B.hostEntryDevice(me, &dev, &err) != lcbind::kOk || dev < 0 || dev > 1 || tCurrent != dev) {
                        ++bad;
                        break;
                    }
                    onDev[dev].fetch_add(1);
                    if (k % 50 == 49) {
                        me.unbind();
                        tCurrent = 0;  // (what a thread without a binding has)
                    }
                }
                finished.fetch_add(1);
                while (finished.load() < 16) std::this_thread::yield();  // (runner threads live as long as the process)
            });
        for (auto& th : pool) th.join();
        CHECK(onDev[0].load() + onDev[1].load() == 16 * 2000);
        CHECK(onDev[0].load() > 16 * 2000 / 4 && onDev[1].load() > 16 * 2000 / 4);  // both devices carry their share
        CHECK(B.ordinalsInUse() == 0);
    }
    printf("%d checks failed (%llu device switches)\n", bad, (unsigned long long)gSets.load());
    return bad ? 1 : 0;
}
