// processor_grok_gpu.cpp -- see processor_grok_gpu.hpp and include/lc_grok.h.
#include "processor_grok_gpu.hpp"

#include <atomic>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <chrono>
#include <cstring>

#include "../../include/lc_grok.h"
#include "../../include/lc_regex_gpu.h"
#include "json_min.hpp"
#include "regex_handle.hpp"


namespace lcgrok {

namespace {
// regexp2.Compile(pattern, regexp2.RE2) (processor_grok.go:343) as the device compilers understand it, searched
// leftmost-first, only named groups capturing (the numbered ones are never emitted, :169)
constexpr uint32_t kGrokSyntax =
    LC_SYNTAX_SEARCH | LC_SYNTAX_NAMED_ONLY | LC_SYNTAX_NO_DOTALL | LC_SYNTAX_NO_MULTILINE | LC_SYNTAX_REGEXP2;

bool allDigits(const std::string& s) {  // strconv.ParseInt(name, 10, 32) succeeds (:169) -> numbered group, skipped
    if (s.empty()) return false;
    size_t i = (s[0] == '-' || s[0] == '+') ? 1 : 0;
    if (i == s.size()) return false;
    for (; i < s.size(); ++i)
        if (s[i] < '0' || s[i] > '9') return false;
    return true;
}
constexpr uint32_t kNoKey = 0xFFFFFFFFu;
}  // namespace

void ProcessorGrokGpu::stopWarmup() {
    mStopWarmup = true;
    {
        std::lock_guard<std::mutex> g(mWarmupMutex);
        if (mWarmup.joinable()) mWarmup.join();
        mWarmupStarted = false;
    }
    mStopWarmup = false;
    if (mAnchored)
        for (size_t i = 0; i < mExpanded.size(); ++i)
            if (lc_regex* re = mAnchored[i].exchange(nullptr)) lc_regex_free(re);
    mAnchored.reset();
}

void ProcessorGrokGpu::WaitReady() {
    startWarmup();
    std::lock_guard<std::mutex> g(mWarmupMutex);
    if (mWarmup.joinable()) mWarmup.join();
}

std::vector<GrokDevicePattern> ProcessorGrokGpu::devicePatterns() {
    startWarmup();
    std::vector<GrokDevicePattern> out = mDevice;
    if (mAnchored)
        for (size_t i = 0; i < out.size(); ++i) out[i].anchored = mAnchored[i].load(std::memory_order_acquire);
    return out;
}

GrokOptions ProcessorGrokGpu::options() const {
    GrokOptions o;
    o.speculative = Speculative;
    o.prefixScreenAbove = PrefixScreenAbove < 0 ? 0u : uint32_t(std::min<int64_t>(PrefixScreenAbove, 0xFFFFFFFFll));
    o.streams = uint32_t(std::max<int64_t>(1, std::min<int64_t>(Streams, 16)));
    return o;
}

ProcessorGrokGpu::~ProcessorGrokGpu() {
    stopTrainer();
    stopWarmup();
    lcGrokStateFree(mState);
    for (lc_regex* re : mCompiled) lc_regex_free(re);
    for (lc_regex* re : mScreens) lc_regex_free(re);
}

int ProcessorGrokGpu::engine(size_t i) const { return i < mCompiled.size() ? mCompiled[i]->engine : 0; }

void ProcessorGrokGpu::Init() {
    stopTrainer();
    stopWarmup();
    lcGrokStateFree(mState);
    mState = lcGrokStateCreate();
    for (lc_regex* re : mCompiled) lc_regex_free(re);
    for (lc_regex* re : mScreens) lc_regex_free(re);
    mCompiled.clear();
    mScreens.clear();
    mDevice.clear();
    mExpanded.clear();
    mKeys.clear();
    mColumnKey.clear();
    mFields.clear();
    mLibrary = PatternLibrary();
    mLibrary.addDefaults();                                               // :68
    for (const auto& path : CustomPatternDir) mLibrary.addFromPath(path); // :70-78
    for (const auto& kv : CustomPatterns) mLibrary.add(kv.first, kv.second);  // :80-82
    mLibrary.build();                                                     // :84
    if (TimeoutMilliSeconds <= 0) TimeoutMilliSeconds = 100;              // :91-93 (kept for config parity: nothing here
                                                                          // backtracks, so nothing can time out)
    std::map<std::string, uint32_t> keyIndex;
    uint32_t maxColumns = 0;
    // compileMatchs :335-359.  The entries are independent and each costs up to a second (the automata of a log format and
    // of its two screens): they are compiled on the host's cores side by side, then booked in order.
    struct Compiled {
        lc_regex_t* re = nullptr;
        lc_regex* screen = nullptr;
        lc_regex* relaxed = nullptr;
        int rc = LC_OK;
        std::string err;
    };
    for (size_t i = 0; i < Match.size(); ++i) mExpanded.push_back(mLibrary.denormalize(Match[i]));
    static const bool noRelaxed = getenv("LC_GROK_NO_RELAXED") != nullptr;    // (A/B measurements)
    static const bool noAnchored = getenv("LC_GROK_NO_ANCHORED") != nullptr;
    std::vector<Compiled> compiled(Match.size());
    auto inParallel = [](size_t nTasks, const std::function<void(size_t)>& task) {
        const size_t workers = std::max<size_t>(1, std::min<size_t>(nTasks, std::min(16u, std::thread::hardware_concurrency())));
        std::atomic<size_t> next{0};
        std::vector<std::thread> pool;
        auto work = [&] {
            for (size_t i = next.fetch_add(1); i < nTasks; i = next.fetch_add(1)) task(i);
        };
        for (size_t w = 1; w < workers; ++w) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
    };
    // 1. the patterns themselves
    inParallel(Match.size(), [&](size_t i) {
        Compiled& c = compiled[i];
        char err[512] = "";
        c.rc = lc_regex_compile(mExpanded[i].data(), mExpanded[i].size(), kGrokSyntax, LC_ENGINE_AUTO, &c.re, err, sizeof err);
        if (c.rc != LC_OK) c.err = err;
    });
    // 2. what stands in front of the slow NFA kernel, for the entries that run on it: a TDFA screen for the pattern's prefix when
    //    one is small enough; one for the whole pattern, relaxed until it is small (the anchored searches follow on the warm-up
    //    thread, at the end of Init)
    std::vector<std::pair<size_t, int>> extras;
    for (size_t i = 0; i < Match.size(); ++i)
        if (compiled[i].re && compiled[i].re->engine == LC_ENGINE_NFA)
            for (int kind = 0; kind < 2; ++kind) extras.emplace_back(i, kind);
    inParallel(extras.size(), [&](size_t t) {
        const size_t i = extras[t].first;
        Compiled& c = compiled[i];
        const char* pat = mExpanded[i].data();
        const size_t len = mExpanded[i].size();
        if (extras[t].second == 0) {
            c.screen = lcCompilePrefixScreen(pat, len, kGrokSyntax, 1024, 32 * 1024);
        } else if (!noRelaxed) {
            // (round 6) ... and preferably until its table can be staged into LDS (grok_device.hip kGrokScreenStageMax = 44 KiB);
            // LC_GROK_RELAX_PREFER_BYTES: 0 = the first automaton that fits at all, as in round 5.  Measured (profiles/round6_grok_steps.txt):
            // the same (entry, value) pairs reach round 0 on the bench corpus, 16 Ki values 2.31 -> 2.02 ms -- the four screens that were
            // walked through L2 were the longest walks of the screen launch and of the remainder screens
            static const size_t preferBytes = [] {
                const char* v = getenv("LC_GROK_RELAX_PREFER_BYTES");
                return v ? size_t(atol(v)) : size_t(44 * 1024);
            }();
            c.relaxed = lcCompileRelaxedScreenPreferring(pat, len, kGrokSyntax, 20000, 2u << 20, preferBytes);
        }
    });
    for (size_t i = 0; i < Match.size(); ++i) {  // everything that was compiled is owned from here on, whatever happens next
        if (compiled[i].re) mCompiled.push_back(compiled[i].re);
        if (compiled[i].re) lcPreferWaveTdfa(compiled[i].re);  // an entry's batch is a few hundred long values: one value per wavefront
        mScreens.push_back(compiled[i].screen);
        mScreens.push_back(compiled[i].relaxed);
    }
    for (size_t i = 0; i < Match.size(); ++i)
        if (compiled[i].rc != LC_OK) throw GrokError("Match[" + std::to_string(i) + "] " + Match[i] + ": " + compiled[i].err);
    for (size_t i = 0; i < Match.size(); ++i) {
        lc_regex_t* re = compiled[i].re;
        const uint32_t columns = uint32_t(lc_regex_mark_count(re)) - 1;   // group 1 is the whole match
        maxColumns = std::max(maxColumns, columns);
        mDevice.push_back({re, columns, compiled[i].screen, compiled[i].relaxed, nullptr});
        std::vector<uint32_t> colKey(columns, kNoKey);
        std::vector<MergedField> fields;
        std::map<std::string, size_t> byName;
        for (uint32_t c = 0; c < columns; ++c) {
            const char* nm = lc_regex_group_name(re, int(c) + 2);
            const std::string name = nm ? nm : "";
            if (name.empty() || allDigits(name)) continue;
            const std::string key = mLibrary.nameToAlias(name);           // :171
            auto k = keyIndex.find(key);
            if (k == keyIndex.end()) {
                k = keyIndex.emplace(key, uint32_t(mKeys.size())).first;
                mKeys.push_back(key);
            }
            colKey[c] = k->second;
            auto f = byName.find(name);
            if (f == byName.end()) {
                byName.emplace(name, fields.size());
                fields.push_back({k->second, {c}});
            } else {
                fields[f->second].columns.push_back(c);                   // same-named groups are one regexp2 group
            }
        }
        mColumnKey.push_back(std::move(colKey));
        mFields.push_back(std::move(fields));
    }
    mRowInts = 2 * (1 + maxColumns);
    // (round 6) an entry no automaton runs -- a back-reference by name, a general look-around: regexp2 backtracks through them -- is on
    // the device backtracking engine (bt_vm.hpp).  The speculative plan is built from automata (screens, literal index, lazy tables): a
    // handle with such an entry walks its list entry by entry instead (grok_device.hip grokMatchSequential: engine-agnostic).
    for (lc_regex_t* re : mCompiled)
        if (re->engine == LC_ENGINE_BT) Speculative = false;
    // 3. the anchored searches, behind Init
    mAnchored.reset(new std::atomic<lc_regex*>[Match.size() ? Match.size() : 1]);
    for (size_t i = 0; i < Match.size(); ++i) mAnchored[i].store(nullptr);
    mWarmupWant.clear();
    mAnchoredBytes = 0;
    if (AnchoredFirst && !noAnchored)
        for (size_t i = 0; i < Match.size(); ++i)
            if (mCompiled[i]->engine == LC_ENGINE_NFA) mWarmupWant.push_back(i);
}

void ProcessorGrokGpu::startWarmup() {
    {
        std::lock_guard<std::mutex> g(mWarmupMutex);
        if (mWarmupStarted || mWarmupWant.empty()) return;
        mWarmupStarted = true;
        mWarmup = std::thread([this] {
            const std::vector<size_t>& want = mWarmupWant;
            // a log agent has other uses for its cores and memory: a few workers, and a budget for the tables (an anchored automaton
            // is kept as the L2 blob on the host plus its copy on the device; the entries are taken in list order until it is spent)
            const size_t workers = std::max<size_t>(1, std::min<size_t>(want.size(), std::min(8u, std::max(1u, std::thread::hardware_concurrency() / 2))));
            const int64_t budget = std::max<int64_t>(0, AnchoredBudgetMB) << 20;
            std::atomic<int64_t> spent{0};
            std::atomic<size_t> next{0};
            auto work = [&] {
                for (size_t t = next.fetch_add(1); t < want.size() && !mStopWarmup.load(); t = next.fetch_add(1)) {
                    if (spent.load() >= budget) break;
                    const size_t i = want[t];
                    lc_regex_t* re = nullptr;
                    char err[64];
                    // The automaton may be too large even anchored (a format with two or three IPv6 alternations).  The anchored search
                    // still pays on the thread-list engine: without the wrapper's lazy prefix no new attempt is started at every byte
                    // that can begin the format -- a log format matches FROM the first byte, and then its thread list is one or two
                    // threads whose steady runs the kernel skips (nfa_kernel.hpp), instead of a list that changes on every byte.
                    if (lc_regex_compile(mExpanded[i].data(), mExpanded[i].size(), kGrokSyntax | LC_SYNTAX_PREFIX, LC_ENGINE_TDFA, &re,
                                         err, sizeof err) != LC_OK &&
                        lc_regex_compile(mExpanded[i].data(), mExpanded[i].size(), kGrokSyntax | LC_SYNTAX_PREFIX, LC_ENGINE_NFA, &re,
                                         err, sizeof err) != LC_OK)
                        continue;
                    lcPreferWaveTdfa(re);
                    const int64_t bytes = 2 * int64_t((re->tdfaL2Blob.size() + re->tdfaBlob.size() + re->tdfaWideBlob.size()) * 4);
                    if (spent.fetch_add(bytes) + bytes > budget) {
                        lc_regex_free(re);  // over the budget: this entry keeps searching on the NFA engine
                        continue;
                    }
                    mAnchoredBytes += uint64_t(bytes);
                    mAnchored[i].store(re, std::memory_order_release);
                }
            };
            std::vector<std::thread> pool;
            for (size_t w = 1; w < workers; ++w) pool.emplace_back(work);
            work();
            for (auto& t : pool) t.join();
        });
    }
}

// ------------------------------------------------------------------------------------------------ the lazy automata's trainer
// Entries whose automaton does not determinise run the thread-list kernels (12 000 cycles per byte step) -- unless a LAZY automaton
// stands in front of them (regex_handle.hpp LcLazyTdfa: built along sample values; a value that steps on a transition nobody has
// computed falls through to the thread-list kernels).  The sample is the processor's own traffic: a batch is OFFERED -- a copy of at
// most kLazyOfferValues of its values, a window that moves through the batch from offer to offer -- when the trainer is idle: the first
// 64 batches at once, later ones every 200 ms at most
// (a log source drifts: new hosts, new message kinds).  The trainer walks the copy through every such entry's current tables on the
// host, keeps what misses and rebuilds (lcRegexLazyTrain): tens of milliseconds per entry, off the data path.
bool ProcessorGrokGpu::wantsOffer() {
    if (!LazyTdfa) return false;
    static const bool envOff = [] {
        const char* e = getenv("LC_LAZY_TDFA");
        return e && e[0] == '0';
    }();
    if (envOff) return false;
    std::lock_guard<std::mutex> g(mTrainerMutex);
    if (mTrainerStop || mTrainerBusy || mTrainerMail) return false;
    if (mTrainerTaken < 64) return true;
    return std::chrono::steady_clock::now() - mLastOffer >= std::chrono::milliseconds(200);
}

// the window of a batch an offer copies: it moves on with every offer, so a source that sends the same kinds of lines in the same places
// of its batches is seen whole
uint32_t ProcessorGrokGpu::offerWindow(uint32_t n, uint32_t take) {
    if (n <= take) return 0;
    const uint32_t windows = (n + take - 1) / take;
    const uint32_t w = mOfferRotor.fetch_add(1, std::memory_order_relaxed) % windows;
    return std::min(w * take, n - take);
}

void ProcessorGrokGpu::postOffer(std::unique_ptr<LazyBatch> b) {
    std::lock_guard<std::mutex> g(mTrainerMutex);
    if (mTrainerStop || mTrainerMail) return;
    mTrainerMail = std::move(b);
    mLastOffer = std::chrono::steady_clock::now();
    if (!mTrainer.joinable()) mTrainer = std::thread([this] { trainerLoop(); });
    mTrainerCv.notify_all();
}

void ProcessorGrokGpu::OfferBatch(const uint8_t* data, const uint32_t* off, const uint32_t* len, uint32_t n) {
    if (!n || !wantsOffer()) return;
    auto b = std::make_unique<LazyBatch>();
    const uint32_t take = std::min(n, kLazyOfferValues);
    const uint32_t first = offerWindow(n, take);
    size_t bytes = 0;
    for (uint32_t i = first; i < first + take; ++i) bytes += len[i];
    b->data.reserve(bytes + 16);
    for (uint32_t i = first; i < first + take; ++i) {
        b->off.push_back(uint32_t(b->data.size()));
        b->len.push_back(len[i]);
        b->data.insert(b->data.end(), data + off[i], data + off[i] + len[i]);
    }
    postOffer(std::move(b));
}

void ProcessorGrokGpu::OfferDeviceBatch(const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len, uint32_t n, void* stream) {
    if (!n || !wantsOffer()) return;
    auto b = std::make_unique<LazyBatch>();
    const uint32_t take = std::min(n, kLazyOfferValues);
    const uint32_t first = offerWindow(n, take);
    if (lcGrokSampleDevice(d_data, d_off + first, d_len + first, take, take, stream, b->data, b->off, b->len) != LC_OK || b->len.empty()) return;
    postOffer(std::move(b));
}

void ProcessorGrokGpu::trainerLoop() {
    for (;;) {
        std::unique_ptr<LazyBatch> b;
        {
            std::unique_lock<std::mutex> lk(mTrainerMutex);
            mTrainerCv.wait(lk, [&] { return mTrainerStop || mTrainerMail; });
            if (mTrainerStop) return;
            b = std::move(mTrainerMail);
            mTrainerBusy = true;
        }
        const uint32_t n = uint32_t(b->len.size());
        for (size_t i = 0; i < mCompiled.size(); ++i) {
            {
                std::lock_guard<std::mutex> lk(mTrainerMutex);
                if (mTrainerStop) break;
            }
            // the search itself, and its anchored form when the warm-up thread has delivered one (round 0 runs that)
            (void)lcRegexLazyTrain(mCompiled[i], b->data.data(), b->off.data(), b->len.data(), n, nullptr);
            if (mAnchored)
                if (lc_regex* a = mAnchored[i].load(std::memory_order_acquire))
                    (void)lcRegexLazyTrain(a, b->data.data(), b->off.data(), b->len.data(), n, nullptr);
        }
        {
            std::lock_guard<std::mutex> lk(mTrainerMutex);
            mTrainerBusy = false;
            ++mTrainerTaken;
        }
        mTrainerCv.notify_all();
    }
}

void ProcessorGrokGpu::stopTrainer() {
    {
        std::lock_guard<std::mutex> g(mTrainerMutex);
        mTrainerStop = true;
    }
    mTrainerCv.notify_all();
    if (mTrainer.joinable()) mTrainer.join();
    std::lock_guard<std::mutex> g(mTrainerMutex);
    mTrainerStop = false;
    mTrainerBusy = false;
    mTrainerMail.reset();
    mTrainerTaken = 0;
}

bool ProcessorGrokGpu::LazySettle(uint32_t timeoutMs) {
    std::unique_lock<std::mutex> lk(mTrainerMutex);
    return mTrainerCv.wait_for(lk, std::chrono::milliseconds(timeoutMs), [&] { return !mTrainerBusy && !mTrainerMail; });
}

void ProcessorGrokGpu::LazyStats(uint64_t out[5]) {
    for (int i = 0; i < 5; ++i) out[i] = 0;
    auto add = [&](lc_regex* re) {
        if (!re) return;
        out[0] += re->lazyReady.load() ? 1 : 0;
        out[1] += re->lazy.builds.load();
        out[2] += re->lazy.offered.load();
        out[3] += re->lazy.kept.load();
    };
    for (size_t i = 0; i < mCompiled.size(); ++i) {
        add(mCompiled[i]);
        if (mAnchored) add(mAnchored[i].load(std::memory_order_acquire));
    }
    std::lock_guard<std::mutex> g(mTrainerMutex);
    out[4] = mTrainerTaken;
}

// named non-empty groups of one match, in Groups() order (:167-175)
void ProcessorGrokGpu::emitRow(size_t p, const int32_t* row, std::vector<Field>& out) const {
    for (const auto& f : mFields[p]) {
        int32_t b = -1, e = -1;
        for (uint32_t c : f.columns) {
            const int32_t cb = row[2 + 2 * c], ce = row[3 + 2 * c];
            if (cb >= 0 && cb >= b) {  // merged group: the capture furthest along is its last capture
                b = cb;
                e = ce;
            }
        }
        if (b >= 0 && e > b) out.push_back({f.key, uint32_t(b), uint32_t(e)});
    }
}

void ProcessorGrokGpu::MatchValues(const uint8_t* data, const uint32_t* off, const uint32_t* len, uint32_t n,
                                   int32_t* pattern, std::vector<uint32_t>& fieldOff, std::vector<Field>& fields) {
    fieldOff.assign(size_t(n) + 1, 0);
    fields.clear();
    if (n == 0) return;
    if (mDevice.empty()) {  // no Match entries: every value is matchFail
        for (uint32_t i = 0; i < n; ++i) pattern[i] = -1;
        return;
    }
    std::vector<int32_t> extra;
    const int32_t* first = nullptr;
    OfferBatch(data, off, len, n);  // (the lazy automata's trainer: a copy now and then, see above)
    int rc = lcGrokMatchHost(devicePatterns(), mState, options(), mRowInts, data, off, len, n, pattern, &first, extra);
    if (rc != LC_OK) throw GrokError(std::string("grok device match failed: ") + lc_last_error());
    const size_t w = mRowInts + 2;
    size_t x = 0;
    const size_t nx = extra.size() / w;
    for (uint32_t i = 0; i < n; ++i) {
        fieldOff[i] = uint32_t(fields.size());
        if (pattern[i] >= 0) {
            emitRow(size_t(pattern[i]), &first[size_t(i) * mRowInts], fields);
            while (x < nx && uint32_t(extra[x * w]) == i) {
                emitRow(size_t(pattern[i]), &extra[x * w + 2], fields);
                ++x;
            }
        }
        while (x < nx && uint32_t(extra[x * w]) == i) ++x;  // rows of a pattern that did not win in the end: none exist
    }
    fieldOff[n] = uint32_t(fields.size());
}

// ProcessLogs :108-113 + processLog :115-146, batched: every content that the Go loop would hand to processGrok is
// matched in one device call, then the logs are edited in the Go loop's order.
void ProcessorGrokGpu::ProcessLogs(std::vector<Log>& logs) {
    struct Ref {
        uint32_t log, content;
    };
    std::vector<Ref> refs;
    std::vector<uint8_t> data;
    std::vector<uint32_t> off, len;
    for (uint32_t l = 0; l < logs.size(); ++l)
        for (uint32_t c = 0; c < logs[l].Contents.size(); ++c) {
            const auto& cont = logs[l].Contents[c];
            if (!SourceKey.empty() && SourceKey != cont.Key) continue;    // :118
            refs.push_back({l, c});
            off.push_back(uint32_t(data.size()));
            len.push_back(uint32_t(cont.Value.size()));
            data.insert(data.end(), cont.Value.begin(), cont.Value.end());
        }
    if (refs.empty()) return;
    data.resize(data.size() + 16);
    std::vector<int32_t> pattern(refs.size());
    std::vector<uint32_t> fieldOff;
    std::vector<Field> fields;
    MatchValues(data.data(), off.data(), len.data(), uint32_t(refs.size()), pattern.data(), fieldOff, fields);
    for (size_t r = 0; r < refs.size(); ++r) {
        Log& log = logs[refs[r].log];
        const bool success = pattern[r] >= 0;
        for (uint32_t f = fieldOff[r]; f < fieldOff[r + 1]; ++f)          // :182-186
            log.Contents.push_back({mKeys[fields[f].key],
                                    std::string(reinterpret_cast<const char*>(data.data()) + off[r] + fields[f].begin,
                                                fields[f].end - fields[f].begin)});
        if ((success && !KeepSource) || (!success && !IgnoreParseFailure)) {  // :135-137, by the index the loop holds
            const uint32_t i = refs[r].content;
            if (i < log.Contents.size()) log.Contents.erase(log.Contents.begin() + i);
        }
    }
}

}  // namespace lcgrok

// ------------------------------------------------------------------------------------------------ C ABI
using lcgrok::ProcessorGrokGpu;

struct lc_grok {
    ProcessorGrokGpu p;
    std::vector<uint32_t> literalIndex;  // lc_grok_literal_index: built on first request
};
struct lc_grok_result {
    std::vector<uint32_t> fieldOff, key, begin, end;
};

static void setErr(char* err, size_t cap, const std::string& msg) {
    if (err && cap) std::snprintf(err, cap, "%s", msg.c_str());
}

extern "C" int lc_grok_create(const char* config_json, size_t config_len, lc_grok_t** out, char* err, size_t errcap) {
    if (!config_json || !out) return LC_ERR_ARG;
    *out = nullptr;
    auto g = std::make_unique<lc_grok>();
    try {
        const lcjson::Value cfg = lcjson::parse(std::string(config_json, config_len));
        if (!cfg.isObject()) throw lcgrok::GrokError("config must be a JSON object");
        auto strings = [&](const char* key, std::vector<std::string>& dst) {
            if (const lcjson::Value* v = cfg.find(key)) {
                if (!v->isArray()) throw lcgrok::GrokError(std::string(key) + " must be an array of strings");
                for (const auto& e : v->arr) {
                    if (!e.isString()) throw lcgrok::GrokError(std::string(key) + " must be an array of strings");
                    dst.push_back(e.str);
                }
            }
        };
        auto boolean = [&](const char* key, bool& dst) {
            if (const lcjson::Value* v = cfg.find(key)) {
                if (!v->isBool()) throw lcgrok::GrokError(std::string(key) + " must be a boolean");
                dst = v->b;
            }
        };
        strings("CustomPatternDir", g->p.CustomPatternDir);
        strings("Match", g->p.Match);
        if (const lcjson::Value* v = cfg.find("CustomPatterns")) {
            if (!v->isObject()) throw lcgrok::GrokError("CustomPatterns must be an object");
            for (const auto& kv : v->obj) {
                if (!kv.second.isString()) throw lcgrok::GrokError("CustomPatterns values must be strings");
                g->p.CustomPatterns[kv.first] = kv.second.str;
            }
        }
        if (const lcjson::Value* v = cfg.find("SourceKey")) {
            if (!v->isString()) throw lcgrok::GrokError("SourceKey must be a string");
            g->p.SourceKey = v->str;
        }
        if (const lcjson::Value* v = cfg.find("TimeoutMilliSeconds"))
            if (v->isNumber()) g->p.TimeoutMilliSeconds = v->isInt ? v->inum : int64_t(v->num);
        boolean("IgnoreParseFailure", g->p.IgnoreParseFailure);
        boolean("KeepSource", g->p.KeepSource);
        boolean("AnchoredFirst", g->p.AnchoredFirst);
        boolean("Speculative", g->p.Speculative);
        boolean("LazyTdfa", g->p.LazyTdfa);
        if (const lcjson::Value* v = cfg.find("AnchoredBudgetMB"))
            if (v->isNumber()) g->p.AnchoredBudgetMB = v->isInt ? v->inum : int64_t(v->num);
        if (const lcjson::Value* v = cfg.find("PrefixScreenAbove"))
            if (v->isNumber()) g->p.PrefixScreenAbove = v->isInt ? v->inum : int64_t(v->num);
        if (const lcjson::Value* v = cfg.find("Streams"))
            if (v->isNumber()) g->p.Streams = v->isInt ? v->inum : int64_t(v->num);
        // CacheDir: compiled automata (and the verdicts of constructions that ran into their limits) are kept there across process
        // restarts -- include/lc_regex_gpu.h lc_runtime_set_table_cache_dir.  Process-wide: the last processor to name one wins.
        if (const lcjson::Value* v = cfg.find("CacheDir")) {
            if (!v->isString()) throw lcgrok::GrokError("CacheDir must be a string");
            lc_runtime_set_table_cache_dir(v->str.c_str());
        }
        boolean("NoKeyError", g->p.NoKeyError);
        boolean("NoMatchError", g->p.NoMatchError);
        boolean("TimeoutError", g->p.TimeoutError);
        g->p.Init();
    } catch (const std::exception& e) {
        setErr(err, errcap, e.what());
        return LC_ERR_UNSUPPORTED;
    }
    setErr(err, errcap, "");
    *out = g.release();
    return LC_OK;
}

extern "C" void lc_grok_free(lc_grok_t* g) { delete g; }
extern "C" int lc_grok_literal_index(lc_grok_t* g, const uint32_t** words, size_t* nwords) {
    if (!g || !words || !nwords) return LC_ERR_ARG;
    *words = nullptr;
    *nwords = 0;
    if (g->literalIndex.empty()) {
        const auto& patterns = g->p.compiledPatterns();
        std::vector<std::string> lits;
        size_t withLiteral = 0;
        for (const auto& gp : patterns) {
            lits.push_back(lcGrokLiteralOf(gp.re));
            withLiteral += !lits.back().empty();
        }
        if (patterns.size() <= 64 && withLiteral >= 2) g->literalIndex = lcBuildGrokLiteralBlob(lits);
    }
    if (!g->literalIndex.empty()) {
        *words = g->literalIndex.data();
        *nwords = g->literalIndex.size();
    }
    return LC_OK;
}

extern "C" void lc_grok_wait_ready(lc_grok_t* g) {
    if (g) g->p.WaitReady();
}

extern "C" int lc_grok_match_count(const lc_grok_t* g) { return g ? int(g->p.expanded().size()) : 0; }
extern "C" const char* lc_grok_expanded(const lc_grok_t* g, int i) {
    return (g && i >= 0 && size_t(i) < g->p.expanded().size()) ? g->p.expanded()[size_t(i)].c_str() : nullptr;
}
extern "C" const char* lc_grok_processed(const lc_grok_t* g, const char* name) {
    if (!g || !name) return nullptr;
    auto it = g->p.library().processed().find(name);
    return it == g->p.library().processed().end() ? nullptr : it->second.c_str();
}
extern "C" char* lc_grok_denormalize(lc_grok_t* g, const char* pattern, char* err, size_t errcap) {
    if (!g || !pattern) return nullptr;
    try {
        const std::string text = g->p.library().denormalize(pattern);
        char* out = static_cast<char*>(std::malloc(text.size() + 1));
        if (out) std::memcpy(out, text.c_str(), text.size() + 1);
        setErr(err, errcap, "");
        return out;
    } catch (const std::exception& e) {
        setErr(err, errcap, e.what());
        return nullptr;
    }
}
extern "C" int lc_grok_engine(const lc_grok_t* g, int i) { return (g && i >= 0) ? g->p.engine(size_t(i)) : 0; }
// diagnostics (tools/grok_entries.py): how Match[i] is run.  out[0] engine of the search form, [1] its tagged DFA's states (0: none),
// [2] 1 = that DFA's tables fit LDS, 2 = they live in global memory; [3] prefix screen states, [4] relaxed screen states;
// [5] anchored search present, [6] its states, [7] 1 = LDS / 2 = global tables, [8] bytes of its global-memory tables, [9] its registers,
// [10] its byte classes, [11] bytes of the search form's global-memory tables
extern "C" int lc_grok_entry_info(lc_grok_t* g, int i, uint32_t out[12]) {
    if (!g || !out || i < 0) return LC_ERR_ARG;
    const std::vector<GrokDevicePattern> dp = g->p.devicePatterns();
    if (size_t(i) >= dp.size()) return LC_ERR_ARG;
    auto where = [](const lc_regex* re) { return !re ? 0u : re->hasTdfa ? 1u : !re->tdfaL2Blob.empty() ? 2u : 0u; };
    std::memset(out, 0, 12 * sizeof(uint32_t));
    const GrokDevicePattern& e = dp[size_t(i)];
    out[0] = uint32_t(e.re->engine);
    out[1] = where(e.re) ? e.re->tdfa.nStates : 0;
    out[2] = where(e.re);
    out[3] = e.screen ? e.screen->tdfa.nStates : 0;
    out[4] = e.relaxed ? (e.relaxed->screenBlob.empty() ? e.relaxed->tdfa.nStates : e.relaxed->screenBlob[1]) : 0;
    out[5] = e.anchored != nullptr;
    if (e.anchored) {
        out[6] = e.anchored->tdfa.nStates;
        out[7] = e.anchored->engine == LC_ENGINE_NFA ? 3u : where(e.anchored);  // (3: anchored search on the thread-list engine)
        out[8] = uint32_t(e.anchored->tdfaL2Blob.size() * 4);
        out[9] = e.anchored->tdfa.nRegs;
        out[10] = e.anchored->tdfa.nClasses;
    }
    out[11] = uint32_t(e.re->tdfaL2Blob.size() * 4);
    return LC_OK;
}
extern "C" int lc_grok_key_count(const lc_grok_t* g) { return g ? int(g->p.keys().size()) : 0; }
extern "C" const char* lc_grok_key(const lc_grok_t* g, int key) {
    return (g && key >= 0 && size_t(key) < g->p.keys().size()) ? g->p.keys()[size_t(key)].c_str() : nullptr;
}
extern "C" int lc_grok_column_count(const lc_grok_t* g, int i) {
    return (g && i >= 0 && size_t(i) < g->p.columnKeys().size()) ? int(g->p.columnKeys()[size_t(i)].size()) : 0;
}
extern "C" int lc_grok_column_key(const lc_grok_t* g, int i, int column) {
    if (!g || i < 0 || size_t(i) >= g->p.columnKeys().size()) return -1;
    const auto& ck = g->p.columnKeys()[size_t(i)];
    if (column < 0 || size_t(column) >= ck.size() || ck[size_t(column)] == 0xFFFFFFFFu) return -1;
    return int(ck[size_t(column)]);
}
extern "C" int lc_grok_row_ints(const lc_grok_t* g) { return g ? int(g->p.rowInts()) : 0; }

extern "C" size_t lc_grok_scratch_bytes(const lc_grok_t* g, uint32_t n) {
    return g ? lcGrokScratchBytes(n, g->p.rowInts()) : 0;
}

extern "C" int lc_grok_match_device(lc_grok_t* g, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len,
                                    uint32_t n, int32_t* d_pattern, int32_t* d_first, int32_t* d_extra, uint32_t extra_cap,
                                    uint32_t* d_nextra, void* d_scratch, size_t scratch_bytes, void* stream) {
    if (!g) return LC_ERR_ARG;
    if (!g->p.deviceState()) return LC_ERR_ARG;
    g->p.OfferDeviceBatch(d_data, d_off, d_len, n, stream);  // (the lazy automata's trainer: a copy now and then)
    return lcGrokMatchDevice(g->p.devicePatterns(), g->p.deviceState(), g->p.options(), g->p.rowInts(), d_data, d_off, d_len, n,
                             d_pattern, d_first, d_extra, extra_cap, d_nextra, d_scratch, scratch_bytes, stream);
}

extern "C" void lc_grok_last_batch_stats(uint32_t out[5]) {
    if (!out) return;
    const GrokBatchStats s = lcGrokLastBatchStats();
    out[0] = s.hostSyncs;
    out[1] = s.activeEntries;
    out[2] = s.pairs;
    out[3] = s.deferredEntries;
    out[4] = s.speculative;
}

extern "C" int lc_grok_match_host(lc_grok_t* g, const uint8_t* data, const uint32_t* off, const uint32_t* len, uint32_t n,
                                  int32_t* pattern, lc_grok_result_t** result) {
    if (!g || !result || (n && (!data || !off || !len || !pattern))) return LC_ERR_ARG;
    *result = nullptr;
    auto r = std::make_unique<lc_grok_result>();
    try {
        static const bool traceHost = getenv("LC_GROK_TRACE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<ProcessorGrokGpu::Field> fields;
        g->p.MatchValues(data, off, len, n, pattern, r->fieldOff, fields);
        const auto t1 = std::chrono::steady_clock::now();
        r->key.reserve(fields.size());
        r->begin.reserve(fields.size());
        r->end.reserve(fields.size());
        for (const auto& f : fields) {
            r->key.push_back(f.key);
            r->begin.push_back(f.begin);
            r->end.push_back(f.end);
        }
        if (traceHost)
            fprintf(stderr, "grok host call: MatchValues %.3f ms, result arrays %.3f ms (%zu fields)\n",
                    std::chrono::duration<double, std::milli>(t1 - t0).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count(), fields.size());
    } catch (const std::exception&) {
        return lc_device_count() <= 0 ? LC_ERR_NO_DEVICE : LC_ERR_HIP;
    }
    *result = r.release();
    return LC_OK;
}

extern "C" int lc_grok_lazy_settle(lc_grok_t* g, uint32_t timeout_ms) {
    if (!g) return LC_ERR_ARG;
    return g->p.LazySettle(timeout_ms) ? LC_OK : LC_ERR_ARG;
}
extern "C" int lc_grok_lazy_stats(lc_grok_t* g, uint64_t out[5]) {
    if (!g || !out) return LC_ERR_ARG;
    g->p.LazyStats(out);
    return LC_OK;
}

extern "C" int lc_grok_combiner_stats(lc_grok_t* g, uint64_t out[11]) {
    if (!g || !out) return LC_ERR_ARG;
    return g->p.CombinerStats(out);
}

extern "C" void lc_grok_result_arrays(const lc_grok_result_t* r, const uint32_t** field_off, const uint32_t** key,
                                      const uint32_t** begin, const uint32_t** end) {
    if (!r) return;
    if (field_off) *field_off = r->fieldOff.data();
    if (key) *key = r->key.data();
    if (begin) *begin = r->begin.data();
    if (end) *end = r->end.data();
}
extern "C" void lc_grok_result_free(lc_grok_result_t* r) { delete r; }

extern "C" int lc_grok_process_logs_json(lc_grok_t* g, const char* logs_json, size_t len, char** out_json) {
    if (!g || !logs_json || !out_json) return LC_ERR_ARG;
    *out_json = nullptr;
    try {
        const lcjson::Value in = lcjson::parse(std::string(logs_json, len));
        if (!in.isArray()) return LC_ERR_ARG;
        std::vector<lcgrok::Log> logs;
        for (const auto& l : in.arr) {
            if (!l.isArray()) return LC_ERR_ARG;
            lcgrok::Log log;
            for (const auto& c : l.arr) {
                if (!c.isArray() || c.arr.size() != 2 || !c.arr[0].isString() || !c.arr[1].isString()) return LC_ERR_ARG;
                log.Contents.push_back({c.arr[0].str, c.arr[1].str});
            }
            logs.push_back(std::move(log));
        }
        g->p.ProcessLogs(logs);
        lcjson::Value out = lcjson::Value::makeArray();
        for (const auto& log : logs) {
            lcjson::Value l = lcjson::Value::makeArray();
            for (const auto& c : log.Contents) {
                lcjson::Value pair = lcjson::Value::makeArray();
                pair.arr.push_back(lcjson::Value::makeString(c.Key));
                pair.arr.push_back(lcjson::Value::makeString(c.Value));
                l.arr.push_back(std::move(pair));
            }
            out.arr.push_back(std::move(l));
        }
        const std::string text = lcjson::dump(out);
        *out_json = static_cast<char*>(std::malloc(text.size() + 1));
        if (!*out_json) return LC_ERR_ARG;
        std::memcpy(*out_json, text.c_str(), text.size() + 1);
        return LC_OK;
    } catch (const lcgrok::GrokError&) {
        return lc_device_count() <= 0 ? LC_ERR_NO_DEVICE : LC_ERR_HIP;
    } catch (const std::exception&) {
        return LC_ERR_ARG;
    }
}
extern "C" void lc_grok_free_string(char* s) { std::free(s); }
