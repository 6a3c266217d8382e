// table_cache.cpp -- see tdfa.hpp: buildTdfa / buildScreenDfa behind an on-disk cache keyed by the construction's input.
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "tdfa.hpp"

namespace lcregex {
namespace {

constexpr uint32_t kMagic = 0x4C435443u;  // "LCTC"
constexpr uint32_t kFormat = 1;
// any rebuild of the library invalidates the cache: the construction is deterministic for one binary, nothing more is promised
const char kBuildStamp[] = __DATE__ " " __TIME__;

std::mutex gDirMutex;
std::string gDir;
bool gDirFromEnvRead = false;
std::atomic<uint64_t> gHits{0}, gMisses{0}, gStored{0}, gFailuresRecalled{0};

std::string cacheDir() {
    std::lock_guard<std::mutex> g(gDirMutex);
    if (!gDirFromEnvRead) {
        gDirFromEnvRead = true;
        if (gDir.empty())
            if (const char* e = getenv("LC_TABLE_CACHE_DIR")) gDir = e;
    }
    return gDir;
}

// 128-bit FNV-1a style hash in two independent 64-bit lanes (collisions would hand a pattern another pattern's tables: two lanes)
struct Hasher {
    uint64_t a = 0xcbf29ce484222325ull, b = 0x84222325cbf29ce4ull;
    void bytes(const void* p, size_t n) {
        const uint8_t* s = static_cast<const uint8_t*>(p);
        for (size_t i = 0; i < n; ++i) {
            a = (a ^ s[i]) * 0x100000001b3ull;
            b = (b ^ (s[i] + 0x9eu)) * 0x100000001b3ull;
            b ^= b >> 29;
        }
    }
    template <class T>
    void pod(const T& v) {
        bytes(&v, sizeof v);
    }
};

void hashNfa(Hasher& h, const FollowNfa& nfa) {
    h.pod(int32_t(nfa.groupCount));
    h.pod(int32_t(nfa.atomicCount));
    h.pod(int32_t(nfa.searchPrefix));
    h.pod(int32_t(nfa.searchSuffix));
    h.pod(uint32_t(nfa.positions.size()));
    for (const ByteSet& s : nfa.positions) h.pod(s.w);
    h.pod(uint32_t(nfa.asserts.size()));
    for (const LookAssert& a : nfa.asserts) {
        h.pod(uint8_t(a.behind));
        h.pod(uint8_t(a.edgeOk));
        h.pod(a.set.w);
    }
    h.pod(uint32_t(nfa.follow.size()));
    for (const auto& lst : nfa.follow) {
        h.pod(uint32_t(lst.size()));
        for (const FollowPath& p : lst) {
            h.pod(int32_t(p.target));
            h.pod(p.tags.w);
            h.pod(p.cond);
            h.pod(uint32_t(p.atoms.size()));
            for (const auto& ev : p.atoms) {
                h.pod(ev.code);
                h.pod(ev.visit);
            }
        }
    }
    h.pod(uint32_t(nfa.runGroups.size()));
    for (const auto& rg : nfa.runGroups) {
        h.pod(int32_t(rg.first));
        h.pod(rg.second.w);
    }
}

std::string keyOf(char kind, const FollowNfa& nfa, const TdfaLimits& lim) {
    Hasher h;
    h.bytes(kBuildStamp, sizeof kBuildStamp);
    h.pod(kFormat);
    h.pod(kind);
    h.pod(lim.maxStates);
    h.pod(lim.maxPathWork);
    h.pod(lim.maxCommitWork);
    h.pod(uint8_t(lim.ldsWindow));
    hashNfa(h, nfa);
    char buf[40];
    snprintf(buf, sizeof buf, "%016llx%016llx", (unsigned long long)h.a, (unsigned long long)h.b);
    return buf;
}

template <class T>
void putVec(std::vector<uint8_t>& out, const std::vector<T>& v) {
    const uint64_t n = v.size();
    const uint8_t* p = reinterpret_cast<const uint8_t*>(&n);
    out.insert(out.end(), p, p + 8);
    const uint8_t* d = reinterpret_cast<const uint8_t*>(v.data());
    out.insert(out.end(), d, d + n * sizeof(T));
}
template <class T>
bool getVec(const std::vector<uint8_t>& in, size_t& at, std::vector<T>& v) {
    if (at + 8 > in.size()) return false;
    uint64_t n;
    std::memcpy(&n, in.data() + at, 8);
    at += 8;
    if (n > (uint64_t(1) << 31) || at + n * sizeof(T) > in.size()) return false;
    v.resize(size_t(n));
    if (n) std::memcpy(v.data(), in.data() + at, size_t(n) * sizeof(T));
    at += size_t(n) * sizeof(T);
    return true;
}

// file: magic, format, ok (1 = tables, 0 = failure message), then the payload
bool load(const std::string& path, TdfaTables& t, std::string& failure, bool& ok) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> in;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) in.insert(in.end(), buf, buf + n);
    fclose(f);
    if (in.size() < 12) return false;
    uint32_t hdr[3];
    std::memcpy(hdr, in.data(), 12);
    if (hdr[0] != kMagic || hdr[1] != kFormat) return false;
    size_t at = 12;
    ok = hdr[2] == 1;
    if (!ok) {
        std::vector<char> msg;
        if (!getVec(in, at, msg)) return false;
        failure.assign(msg.begin(), msg.end());
        return at == in.size();
    }
    uint32_t scalars[5];
    if (at + sizeof scalars > in.size()) return false;
    std::memcpy(scalars, in.data() + at, sizeof scalars);
    at += sizeof scalars;
    t.nStates = scalars[0];
    t.nClasses = scalars[1];
    t.nRegs = scalars[2];
    t.nSlots = scalars[3];
    t.startState = scalars[4];
    if (!getVec(in, at, t.classMap) || !getVec(in, at, t.trans) || !getVec(in, at, t.opsStart) || !getVec(in, at, t.ops) ||
        !getVec(in, at, t.finalId) || !getVec(in, at, t.finalMap) || !getVec(in, at, t.startAfter))
        return false;
    // (a truncated or foreign file must not become tables: the sizes have to fit together)
    return at == in.size() && t.classMap.size() == 256 && t.trans.size() == size_t(t.nStates) * t.nClasses && t.finalId.size() == t.nStates;
}

void store(const std::string& dir, const std::string& key, const TdfaTables* t, const std::string& failure) {
    std::vector<uint8_t> out;
    const uint32_t hdr[3] = {kMagic, kFormat, t ? 1u : 0u};
    out.insert(out.end(), reinterpret_cast<const uint8_t*>(hdr), reinterpret_cast<const uint8_t*>(hdr) + 12);
    if (t) {
        const uint32_t scalars[5] = {t->nStates, t->nClasses, t->nRegs, t->nSlots, t->startState};
        out.insert(out.end(), reinterpret_cast<const uint8_t*>(scalars), reinterpret_cast<const uint8_t*>(scalars) + sizeof scalars);
        putVec(out, t->classMap);
        putVec(out, t->trans);
        putVec(out, t->opsStart);
        putVec(out, t->ops);
        putVec(out, t->finalId);
        putVec(out, t->finalMap);
        putVec(out, t->startAfter);
    } else {
        putVec(out, std::vector<char>(failure.begin(), failure.end()));
    }
    (void)mkdir(dir.c_str(), 0755);
    // written under a name of its own, then renamed: a reader never sees half a file, two writers of one key write the same bytes
    char tmpName[64];
    static std::atomic<uint32_t> seq{0};
    snprintf(tmpName, sizeof tmpName, ".tmp.%d.%u", int(getpid()), seq.fetch_add(1));
    const std::string tmp = dir + "/" + key + tmpName, final = dir + "/lc_tdfa_" + key + ".bin";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    const bool wrote = fwrite(out.data(), 1, out.size(), f) == out.size();
    if (fclose(f) != 0 || !wrote || rename(tmp.c_str(), final.c_str()) != 0) {
        (void)remove(tmp.c_str());
        return;
    }
    gStored.fetch_add(1, std::memory_order_relaxed);
}

template <class Build>
TdfaTables cached(char kind, const FollowNfa& nfa, const TdfaLimits& limits, Build build) {
    const std::string dir = cacheDir();
    if (dir.empty()) return build();
    const std::string key = keyOf(kind, nfa, limits);
    {
        TdfaTables t;
        std::string failure;
        bool ok = false;
        if (load(dir + "/lc_tdfa_" + key + ".bin", t, failure, ok)) {
            if (ok) {
                gHits.fetch_add(1, std::memory_order_relaxed);
                return t;
            }
            gFailuresRecalled.fetch_add(1, std::memory_order_relaxed);
            throw RegexError(failure);
        }
    }
    gMisses.fetch_add(1, std::memory_order_relaxed);
    try {
        TdfaTables t = build();
        store(dir, key, &t, std::string());
        return t;
    } catch (const RegexError& e) {
        // only what is expensive to find out again: a construction that ran into its state or work limit
        if (std::strstr(e.what(), "limit") != nullptr) store(dir, key, nullptr, e.what());
        throw;
    }
}

}  // namespace

void lcSetTableCacheDir(const char* dir) {
    std::lock_guard<std::mutex> g(gDirMutex);
    gDirFromEnvRead = true;  // (an explicit setting wins over the environment)
    gDir = dir ? dir : "";
}
const char* lcTableCacheDir() {
    static thread_local std::string copy;
    copy = cacheDir();
    return copy.c_str();
}
TableCacheStats lcTableCacheStats() {
    TableCacheStats s;
    s.hits = gHits.load();
    s.misses = gMisses.load();
    s.stored = gStored.load();
    s.failuresRecalled = gFailuresRecalled.load();
    return s;
}

TdfaTables buildTdfa(const FollowNfa& nfa, const TdfaLimits& limits) {
    return cached('T', nfa, limits, [&] { return buildTdfaUncached(nfa, limits); });
}
TdfaTables buildScreenDfa(const FollowNfa& nfa, const TdfaLimits& limits) {
    return cached('S', nfa, limits, [&] { return buildScreenDfaUncached(nfa, limits); });
}

}  // namespace lcregex
