// processor_grok_gpu.hpp -- host side of the Grok processor: the reference's Go plugin restated over the device matcher.
//
// Mirrors plugins/processor/grok/processor_grok.go: the exported fields keep their names (:42-53), Init follows :62-102,
// ProcessLogs / processLog / processGrok follow :108-194 -- with the per-log regexp2 loop replaced by ONE batched device
// call over the SourceKey values of all logs (lcGrokMatchHost, grok_device.hip).  Logs are the protocol.Log shape: an
// ordered list of (Key, Value) contents (duplicates allowed; fields are appended, processor_grok.go:183-185).
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <thread>
#include <map>
#include <mutex>
#include <memory>
#include <string>
#include <vector>

#include "grok.hpp"
#include "grok_runtime.hpp"

struct lc_regex;

namespace lcgrok {

struct LogContent {
    std::string Key, Value;
};
struct Log {
    std::vector<LogContent> Contents;
};

class ProcessorGrokGpu {
public:
    // exported fields, names as in the Go struct
    std::vector<std::string> CustomPatternDir;
    std::map<std::string, std::string> CustomPatterns;
    std::string SourceKey = "content";
    std::vector<std::string> Match;
    int64_t TimeoutMilliSeconds = 0;
    bool IgnoreParseFailure = true;
    bool KeepSource = true;
    // not a reference key: compile, for every entry that runs on the NFA engine, the ANCHORED search as well (the match must start
    // at the first byte: a tagged DFA of 2 000 - 30 000 states with its tables in L2) and try it first -- log formats match from the
    // first byte.  Those automata cost seconds each, and the entries whose automaton turns out too large cost as much: they are
    // compiled BEHIND Init, on a warm-up thread, and join the device loop as they arrive (results do not depend on them, only speed;
    // WaitReady() blocks until the thread is done).  Instances that only probe a pattern switch it off.
    bool AnchoredFirst = true;
    int64_t AnchoredBudgetMB = 256;   // host + device bytes all anchored automata of this instance may take together
    // not reference keys either (GrokOptions, grok_runtime.hpp; results never depend on them): evaluate all (entry, value) pairs that
    // pass the screens at the same time (default) or walk the list entry by entry; the sequential path's prefix-screen threshold;
    // worker streams of the speculative path
    bool Speculative = true;
    int64_t PrefixScreenAbove = 65536;
    int64_t Streams = 16;  // worker streams of a batch's entries (1..16; see gpu_runtime.hip on GPU_MAX_HW_QUEUES)
    // not a reference key: the entries whose automaton does not determinise get a LAZY one (regex_handle.hpp LcLazyTdfa), built by a
    // background thread along the values the processor is handed -- the first batches at once, later ones now and then (results never
    // depend on it; "LazyTdfa": false, or LC_LAZY_TDFA=0 in the environment, leaves those entries to the thread-list kernels alone)
    bool LazyTdfa = true;
    bool NoKeyError = false;
    bool NoMatchError = true;
    bool TimeoutError = true;

    ~ProcessorGrokGpu();
    void Init();                          // throws GrokError
    void ProcessLogs(std::vector<Log>& logs);

    // one field of one value: key index (keys()), byte range inside the value
    struct Field {
        uint32_t key, begin, end;
    };
    // processGrok over a batch of values: winning Match index per value (-1 matchFail, -2 undecidable) and its fields in
    // emission order.  fieldOff has n+1 entries.
    void MatchValues(const uint8_t* data, const uint32_t* off, const uint32_t* len, uint32_t n, int32_t* pattern,
                     std::vector<uint32_t>& fieldOff, std::vector<Field>& fields);

    const PatternLibrary& library() const { return mLibrary; }
    PatternLibrary& library() { return mLibrary; }
    const std::vector<std::string>& expanded() const { return mExpanded; }
    const std::vector<std::string>& keys() const { return mKeys; }
    const std::vector<std::vector<uint32_t>>& columnKeys() const { return mColumnKey; }
    std::vector<GrokDevicePattern> devicePatterns();  // (a snapshot: the anchored handles arrive from the warm-up thread, which
                                                      // the first call starts -- an instance that never matches never pays)
    const std::vector<GrokDevicePattern>& compiledPatterns() const { return mDevice; }  // (without the anchored searches)
    void WaitReady();                                       // returns when the warm-up thread has compiled what it can
    uint32_t rowInts() const { return mRowInts; }
    uint64_t anchoredBytes() const { return mAnchoredBytes.load(); }
    GrokDeviceState* deviceState() { return mState; }
    int CombinerStats(uint64_t out[11]) { return lcGrokCombinerStats(mState, out); }  // lc_grok_combiner_stats
    // the lazy automata's trainer: a batch in host memory is offered (copied, at most kLazyOfferValues of its values, when the trainer is
    // idle and the last offer is old enough); OfferDeviceBatch fetches them from the device first.  LazySettle waits until the trainer
    // has nothing to do; LazyStats: {handles with a lazy automaton in use, builds, values offered, values kept in samples, batches taken}
    void OfferBatch(const uint8_t* data, const uint32_t* off, const uint32_t* len, uint32_t n);
    void OfferDeviceBatch(const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len, uint32_t n, void* stream);
    bool LazySettle(uint32_t timeoutMs);
    void LazyStats(uint64_t out[5]);
    GrokOptions options() const;
    int engine(size_t i) const;

private:
    PatternLibrary mLibrary;
    std::vector<std::string> mExpanded;
    std::vector<lc_regex*> mCompiled;
    std::vector<lc_regex*> mScreens;                   // the screens of the Match entries (prefix, relaxed; nullptr = none), owned
    std::unique_ptr<std::atomic<lc_regex*>[]> mAnchored;  // [Match entry] the anchored search, once compiled (nullptr: none), owned
    std::thread mWarmup;
    std::atomic<bool> mStopWarmup{false};
    std::mutex mWarmupMutex;
    bool mWarmupStarted = false;
    std::atomic<uint64_t> mAnchoredBytes{0};
    std::vector<size_t> mWarmupWant;                   // the entries the warm-up thread compiles an anchored search for
    void startWarmup();
    void stopWarmup();
    // ---- lazy-automaton trainer
    static constexpr uint32_t kLazyOfferValues = 4096;
    struct LazyBatch {
        std::vector<uint8_t> data;
        std::vector<uint32_t> off, len;
    };
    std::thread mTrainer;
    std::mutex mTrainerMutex;
    std::condition_variable mTrainerCv;
    std::unique_ptr<LazyBatch> mTrainerMail;   // the batch waiting for the trainer (at most one)
    bool mTrainerBusy = false, mTrainerStop = false;
    uint64_t mTrainerTaken = 0;                // batches the trainer has worked through
    std::chrono::steady_clock::time_point mLastOffer{};
    std::atomic<uint32_t> mOfferRotor{0};
    uint32_t offerWindow(uint32_t n, uint32_t take);
    bool wantsOffer();                         // cheap test on the caller's thread
    void postOffer(std::unique_ptr<LazyBatch> b);
    void stopTrainer();
    void trainerLoop();
    std::vector<GrokDevicePattern> mDevice;
    GrokDeviceState* mState = nullptr;                 // literal index + screen table on the device(s), built on first use
    std::vector<std::string> mKeys;                    // distinct emitted keys
    std::vector<std::vector<uint32_t>> mColumnKey;     // [pattern][column] -> key index
    // [pattern] fields in Groups() order: key index + the columns that share the name
    struct MergedField {
        uint32_t key;
        std::vector<uint32_t> columns;
    };
    std::vector<std::vector<MergedField>> mFields;
    uint32_t mRowInts = 2;

    void emitRow(size_t p, const int32_t* row, std::vector<Field>& out) const;
};

}  // namespace lcgrok
