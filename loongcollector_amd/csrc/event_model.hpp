// event_model.hpp -- a header-compatible stand-in for the slice of LoongCollector's event model that the regex
// parse processor touches.  Same namespace, type names and member-function names as the reference, so that
// processor_parse_regex_gpu.cpp compiles unchanged against either this header or the reference's own
//   core/models/PipelineEventGroup.h:71-158   core/models/LogEvent.h:64-132   core/models/PipelineEventPtr.h:32-96
//   core/common/StringView.h:25               core/common/memory/SourceBuffer.h:29-181
// (define LC_USE_REFERENCE_HEADERS in the reference tree).  Written from the interface, not copied: only the
// observable semantics the reference unit tests pin are reproduced --
//   * contents are an ordered list of (key,value,alive); lookups scan from the back; delete is a tombstone that keeps
//     order; SetContentNoCopy overwrites in place or appends   (LogEvent.cpp:50-106)
//   * values are views into memory owned by the group's SourceBuffer (zero copy)
//   * the arena hands out NUL-terminated, 8-byte aligned copies from chunks that double from 4 KiB to 128 KiB and
//     gives requests of at least half a chunk their own block   (SourceBuffer.h:45-153)
#pragma once

#include <cstdint>
#include <cstring>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <optional>
#include <string>
#include <string_view>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace logtail {

class StringView : public std::string_view {
public:
    using std::string_view::string_view;
    StringView() = default;
    StringView(std::string_view v) : std::string_view(v) {}
    StringView(const std::string& s) : std::string_view(s) {}
    std::string to_string() const { return std::string(data(), size()); }
};

struct StringBuffer {
    char* data = nullptr;
    size_t size = 0;      // bytes in use (excluding the terminating NUL)
    size_t capacity = 0;  // bytes reserved (including the NUL)
};

// Full-size arena chunks (128 KiB) go round through a small process-wide pool instead of back to the allocator.  A group's stitch
// takes four of them (440 bytes per 10-field event), a freed 128 KiB block is unmapped or trimmed by glibc, and the next group pays
// for the same pages again: 108 first-touch faults per 1000-event group, more than the stitch itself costs (measured with getrusage,
// tests/native/host_double.cpp).  An agent keeps a bounded number of groups in flight, so a bounded pool
// (64 MiB) covers its steady state; beyond the bound chunks are freed as before.
class ArenaChunkPool {
public:
    static constexpr size_t kChunkBytes = 128 * 1024, kMaxPooled = 512;
    static ArenaChunkPool& instance() {
        static ArenaChunkPool* pool = new ArenaChunkPool;  // never destroyed: a SourceBuffer may outlive every static destructor
        return *pool;
    }
    char* take() {
#ifdef LC_REFERENCE_SHAPED_EVENT_MODEL
        return nullptr;  // (the reference's SourceBuffer has no pool: every chunk comes from the allocator, SourceBuffer.h:98-131)
#endif
        std::lock_guard<std::mutex> g(mMutex);
        if (mFree.empty()) return nullptr;
        char* p = mFree.back();
        mFree.pop_back();
        return p;
    }
    bool give(char* p) {
#ifdef LC_REFERENCE_SHAPED_EVENT_MODEL
        (void)p;
        return false;
#endif
        std::lock_guard<std::mutex> g(mMutex);
        if (mFree.size() >= kMaxPooled) return false;
        mFree.push_back(p);
        return true;
    }
    size_t pooled() {
        std::lock_guard<std::mutex> g(mMutex);
        return mFree.size();
    }

private:
    std::mutex mMutex;
    std::vector<char*> mFree;
};

class SourceBuffer {
public:
    SourceBuffer() = default;
    SourceBuffer(const SourceBuffer&) = delete;
    SourceBuffer& operator=(const SourceBuffer&) = delete;
    ~SourceBuffer() {
        for (Chunk& c : mChunks) {
            if (c.cap == kMaxChunk && ArenaChunkPool::instance().give(c.mem)) continue;
            delete[] c.mem;
        }
    }
    StringBuffer AllocateStringBuffer(size_t size) {
        StringBuffer sb;
        sb.capacity = size + 1;
        sb.data = static_cast<char*>(allocate(sb.capacity));
        sb.data[0] = '\0';
        sb.data[size] = '\0';
        sb.size = size;
        return sb;
    }
    StringBuffer CopyString(const char* s, size_t len) {
        StringBuffer sb = AllocateStringBuffer(len);
        if (len) std::memcpy(sb.data, s, len);
        sb.data[len] = '\0';
        return sb;
    }
    StringBuffer CopyString(StringView s) { return CopyString(s.data(), s.size()); }
    StringBuffer CopyString(const std::string& s) { return CopyString(s.data(), s.size()); }
    size_t chunkCount() const { return mChunks.size(); }
    // raw arena memory (8-byte aligned, lives as long as the buffer): the contents arrays of the group's events are carved
    // from here, so that stitching K fields into each of a group's events costs no malloc/free at all -- with one heap
    // allocation per event the runner threads meet in the allocator (events are created by the input thread, grown and
    // freed by the runner threads): measured 165 us -> 1650 us per 1000-event group with 8 threads
    void* AllocateRaw(size_t bytes) { return allocate(bytes); }

private:
    static constexpr size_t kFirstChunk = 4096, kMaxChunk = 128 * 1024, kAlign = 8;
    static_assert(kMaxChunk == ArenaChunkPool::kChunkBytes, "the pool holds full-size chunks");
    struct Chunk {
        char* mem = nullptr;
        size_t cap = 0, used = 0;
    };
    std::vector<Chunk> mChunks;
    size_t mNextChunk = kFirstChunk;
    size_t mCurrent = size_t(-1);

    void* allocate(size_t bytes) {
        bytes = (bytes + kAlign - 1) & ~(kAlign - 1);
        if (bytes >= mNextChunk / 2) {  // big request: its own block, the running chunk stays current
            Chunk c;
            c.mem = new char[bytes];
            c.cap = c.used = bytes;
            mChunks.push_back(c);
            if (mCurrent != size_t(-1) && mCurrent == mChunks.size() - 1) mCurrent = size_t(-1);
            return mChunks.back().mem;
        }
        if (mCurrent == size_t(-1) || mChunks[mCurrent].used + bytes > mChunks[mCurrent].cap) {
            Chunk c;
            c.mem = mNextChunk == kMaxChunk ? ArenaChunkPool::instance().take() : nullptr;
            if (!c.mem) c.mem = new char[mNextChunk];
            c.cap = mNextChunk;
            mChunks.push_back(c);
            mCurrent = mChunks.size() - 1;
            if (mNextChunk < kMaxChunk) mNextChunk *= 2;
        }
        Chunk& cur = mChunks[mCurrent];
        void* p = cur.mem + cur.used;
        cur.used += bytes;
        return p;
    }
};

enum class EventGroupMetaKey {
    UNKNOWN,
    LOG_FILE_PATH,
    LOG_FILE_PATH_RESOLVED,
    LOG_FILE_INODE,
    LOG_FILE_OFFSET_KEY,
    HAS_PART_LOG,
    SOURCE_ID
};
using GroupMetadata = std::map<EventGroupMetaKey, StringView>;
using GroupTags = std::map<StringView, StringView>;

class PipelineEventGroup;

class PipelineEvent {
public:
    enum class Type { NONE, LOG, METRIC, SPAN, RAW };
    virtual ~PipelineEvent() = default;
    Type GetType() const { return mType; }
    time_t GetTimestamp() const { return mTimestamp; }
    std::optional<uint32_t> GetTimestampNanosecond() const { return mTimestampNanosecond; }
    void SetTimestamp(time_t t) { mTimestamp = t; }
    void SetTimestamp(time_t t, uint32_t ns) {
        mTimestamp = t;
        mTimestampNanosecond = ns;
    }
    virtual size_t DataSize() const { return sizeof(mTimestamp) + sizeof(mTimestampNanosecond); }
    std::shared_ptr<SourceBuffer>& GetSourceBuffer();

protected:
    PipelineEvent(Type t, PipelineEventGroup* g) : mType(t), mGroup(g) {}
    Type mType;
    time_t mTimestamp = 0;
    std::optional<uint32_t> mTimestampNanosecond;
    PipelineEventGroup* mGroup;
};

using LogContent = std::pair<StringView, StringView>;

// The contents array of one event -- std::vector<std::pair<LogContent, bool>> in the reference (LogEvent.h:23-24) -- carved from the
// group's SourceBuffer.  Entries are two views and a flag: nothing to destroy, growth copies them into a new arena block (the old one
// stays where it is and goes with the group, like every other arena allocation), and a caller about to construct n entries gets
// their storage without having them zeroed first (std::vector::resize did, 30 stores per 10-field event before the 40 that count).
// Events made outside a group (arena == nullptr) use the heap.
template <class T>
class ArenaVector {
public:
    using value_type = T;
    using iterator = T*;
    using const_iterator = const T*;
    using const_reverse_iterator = std::reverse_iterator<const T*>;

    explicit ArenaVector(SourceBuffer* arena) : mArena(arena) {}
    ArenaVector(const ArenaVector&) = delete;
    ArenaVector& operator=(const ArenaVector&) = delete;
    ~ArenaVector() {
        if (!mArena) ::operator delete(mBegin);
    }

    size_t size() const { return mSize; }
    bool empty() const { return mSize == 0; }
    size_t capacity() const { return mCap; }
    T* data() { return mBegin; }
    const T* data() const { return mBegin; }
    T& operator[](size_t i) { return mBegin[i]; }
    const T& operator[](size_t i) const { return mBegin[i]; }
    T& back() { return mBegin[mSize - 1]; }
    iterator begin() { return mBegin; }
    iterator end() { return mBegin + mSize; }
    const_iterator begin() const { return mBegin; }
    const_iterator end() const { return mBegin + mSize; }
    const_iterator cbegin() const { return mBegin; }
    const_iterator cend() const { return mBegin + mSize; }
    const_reverse_iterator crbegin() const { return const_reverse_iterator(mBegin + mSize); }
    const_reverse_iterator crend() const { return const_reverse_iterator(mBegin); }

    void reserve(size_t n) {
        if (n > mCap) grow(n);
    }
    template <class... Args>
    T& emplace_back(Args&&... args) {
        if (mSize == mCap) grow(mCap ? size_t(mCap) * 2 : 1);  // (1, 2, 4, ...: std::vector's growth, what the reference's events do)
        return *new (mBegin + mSize++) T(std::forward<Args>(args)...);
    }
    // storage for n more entries, counted as present: the caller constructs every one of them (placement new) before anything reads
    T* appendUninitialized(size_t n) {
        if (mSize + n > mCap) grow(mSize + n);
        T* first = mBegin + mSize;
        mSize += uint32_t(n);
        return first;
    }

private:
    void grow(size_t cap) {
        T* fresh = static_cast<T*>(mArena ? mArena->AllocateRaw(cap * sizeof(T)) : ::operator new(cap * sizeof(T)));
        for (uint32_t i = 0; i < mSize; ++i) new (fresh + i) T(mBegin[i]);
        if (!mArena) ::operator delete(mBegin);
        mBegin = fresh;
        mCap = uint32_t(cap);
    }
    T* mBegin = nullptr;
    uint32_t mSize = 0, mCap = 0;
    SourceBuffer* mArena;
};
// LC_REFERENCE_SHAPED_EVENT_MODEL (build switch, bench.py end_to_end.in_agent_reference_shape_MBps): the event model shaped like the
// reference's -- contents in a heap std::vector reserved to default_log_event_capacity = 16 by the constructor (LogEvent.cpp:21-29),
// no chunk pool, and no bulk stitch (the processor then writes its K fields with K x SetContentNoCopy and drops the source with
// DelContent, exactly the calls a build against the reference headers makes: ProcessorParseRegexNative.cpp:249-251, :153-155).
// What the library's default event model gains over this shape is gained by types the agent owns; an agent build gets this shape.
#ifdef LC_REFERENCE_SHAPED_EVENT_MODEL
using ContentsContainer = std::vector<std::pair<LogContent, bool>>;
#else
using ContentsContainer = ArenaVector<std::pair<LogContent, bool>>;
#endif
static_assert(sizeof(ContentsContainer) == sizeof(std::vector<int>), "LogEvent::DataSize counts sizeof(mContents) as the reference does");
static_assert(std::is_trivially_destructible<std::pair<LogContent, bool>>::value, "ArenaVector never runs destructors");

class LogEvent : public PipelineEvent {
public:
    explicit LogEvent(PipelineEventGroup* g);

    // iterates live contents in insertion order
    class ConstContentIterator {
    public:
        ConstContentIterator(ContentsContainer::const_iterator it, const ContentsContainer& c) : mIt(it), mC(&c) {}
        const LogContent& operator*() const { return mIt->first; }
        const LogContent* operator->() const { return &mIt->first; }
        ConstContentIterator& operator++() {
            do {
                ++mIt;
            } while (mIt != mC->end() && !mIt->second);
            return *this;
        }
        bool operator==(const ConstContentIterator& o) const { return mIt == o.mIt; }
        bool operator!=(const ConstContentIterator& o) const { return mIt != o.mIt; }

    private:
        ContentsContainer::const_iterator mIt;
        const ContentsContainer* mC;
    };

    StringView GetContent(StringView key) const {
        const auto* e = findLive(key);
        return e ? e->first.second : StringView();
    }
    bool HasContent(StringView key) const { return findLive(key) != nullptr; }
    void SetContent(StringView key, StringView val);
    void SetContent(const std::string& key, const std::string& val) { SetContent(StringView(key), StringView(val)); }
    void SetContentNoCopy(StringView key, StringView val) {
        auto* e = const_cast<std::pair<LogContent, bool>*>(findLive(key));
        if (e) {
            mAllocatedContentSize += key.size() + val.size() - e->first.first.size() - e->first.second.size();
            e->first = LogContent(key, val);
        } else {
            ++mContentCnt;
            mAllocatedContentSize += key.size() + val.size();
            appendLive(key, val);
        }
    }
    // Bulk form of K x SetContentNoCopy for a caller that KNOWS none of the keys is present yet (SURVEY.md section 8(f)
    // rank 4): one reservation, no per-key reverse scan.  Not part of the reference's LogEvent; the parse processor uses it
    // only when the event holds nothing but the source content and its Keys are pairwise distinct.
    template <class KeyIt, class ValIt>
    void AppendContentsNoCopy(KeyIt keys, ValIt vals, size_t n) {
        mContents.reserve(mContents.size() + n);
        for (size_t i = 0; i < n; ++i, ++keys, ++vals) {
            mAllocatedContentSize += keys->size() + vals->size();
            appendLive(*keys, *vals);
        }
        mContentCnt += n;
    }
    // The K fields of a parsed line in one go: value k is raw[c[2k], c[2k+1]) -- or the empty view at the end of `raw` for a group
    // that took no part in the match (c[2k] < 0: boost's {last, last, matched = false}).  Same preconditions and the same result as
    // AppendContentsNoCopy; the views are built in place instead of going through a caller's array.
    // With dropKey (a key that is none of `keys`): followed by DelContent(*dropKey), whose scan from the back then starts below the new
    // entries -- and whose one-byte tombstone store comes after the array has moved, not right before the move reads it back.
#ifndef LC_REFERENCE_SHAPED_EVENT_MODEL
    void AppendCapturesNoCopy(const StringView* keys, size_t n, StringView raw, const int32_t* c, const StringView* dropKey = nullptr) {
        const size_t old = mContents.size();
        std::pair<LogContent, bool>* out = mContents.appendUninitialized(n);
        // The arena is a bump allocator and a group's events are stitched one after the other: the arrays of the next events will be
        // carved right behind this one.  Asking for those lines (for writing) four events ahead takes the stores of a 1000-event
        // group off fresh memory's latency: 57 -> 46 us per group on the build machine (tests/native/host_double.cpp hd_bench_stitch).
        {
            const size_t span = (old + n) * sizeof(*out);
            const char* ahead = reinterpret_cast<const char*>(out) + 4 * span;
            for (size_t q = 0; q < span; q += 64) __builtin_prefetch(ahead + q, 1);
        }
        size_t bytes = 0;
        for (size_t k = 0; k < n; ++k) {
            const int32_t b = c[2 * k], e = c[2 * k + 1];
            const char* at = b < 0 ? raw.data() + raw.size() : raw.data() + b;
            const size_t len = b < 0 ? 0 : size_t(e - b);
            bytes += keys[k].size() + len;
            new (out + k) std::pair<LogContent, bool>(std::piecewise_construct,
                                                      std::forward_as_tuple(std::piecewise_construct, std::forward_as_tuple(keys[k]),
                                                                            std::forward_as_tuple(at, len)),
                                                      std::forward_as_tuple(true));
        }
        size_t live = mContentCnt + n;
        if (dropKey) {
            std::pair<LogContent, bool>* const first = mContents.data();
            for (std::pair<LogContent, bool>* e = first + old; e != first;) {
                --e;
                if (e->second && e->first.first == *dropKey) {
                    e->second = false;
                    --live;
                    bytes -= e->first.first.size() + e->first.second.size();  // (modulo 2^64: the += below makes it exact)
                    break;
                }
            }
        }
        mAllocatedContentSize += bytes;
        mContentCnt = live;
    }
#endif
    // HasContent + GetContent in one scan: the live value of `key`, or nullptr (an empty value is not a missing one)
    const StringView* FindContent(StringView key) const {
        const auto* e = findLive(key);
        return e ? &e->first.second : nullptr;
    }
    void DelContent(StringView key) {
        auto* e = const_cast<std::pair<LogContent, bool>*>(findLive(key));
        if (e) {
            e->second = false;
            --mContentCnt;
            mAllocatedContentSize -= e->first.first.size() + e->first.second.size();
        }
    }
    bool Empty() const { return mContentCnt == 0; }
    size_t Size() const { return mContentCnt; }
    ConstContentIterator cbegin() const {
        auto it = mContents.cbegin();
        while (it != mContents.cend() && !it->second) ++it;
        return ConstContentIterator(it, mContents);
    }
    ConstContentIterator cend() const { return ConstContentIterator(mContents.cend(), mContents); }
    ConstContentIterator begin() const { return cbegin(); }
    ConstContentIterator end() const { return cend(); }
    void SetPosition(uint64_t offset, uint64_t size) {
        mFileOffset = offset;
        mRawSize = size;
    }
    std::pair<uint64_t, uint64_t> GetPosition() const { return {mFileOffset, mRawSize}; }
    size_t DataSize() const override { return PipelineEvent::DataSize() + sizeof(mContents) + mAllocatedContentSize; }

private:
    // (piecewise: emplace_back(LogContent(key, val), true) builds the inner pair on the stack with 8-byte stores and copies it with
    // 16-byte loads -- a store-forwarding stall per entry, three times the cost of the append itself)
    void appendLive(StringView key, StringView val) {
#ifdef LC_REFERENCE_SHAPED_EVENT_MODEL
        mContents.emplace_back(std::make_pair(key, val), true);  // (as LogEvent.cpp:93 has it -- the stall described above included)
        return;
#endif
        mContents.emplace_back(std::piecewise_construct,
                               std::forward_as_tuple(std::piecewise_construct, std::forward_as_tuple(key), std::forward_as_tuple(val)),
                               std::forward_as_tuple(true));
    }
    const std::pair<LogContent, bool>* findLive(StringView key) const {
        for (auto it = mContents.crbegin(); it != mContents.crend(); ++it)
            if (it->second && it->first.first == key) return &*it;
        return nullptr;
    }
    ContentsContainer mContents;
    size_t mAllocatedContentSize = 0;
    size_t mContentCnt = 0;
    uint64_t mFileOffset = 0, mRawSize = 0;
};

// any non-log event kind; the parse processor only needs to recognise "not a LogEvent"
class RawEvent : public PipelineEvent {
public:
    explicit RawEvent(PipelineEventGroup* g) : PipelineEvent(Type::RAW, g) {}
    StringView GetContent() const { return mContent; }
    void SetContentNoCopy(StringView c) { mContent = c; }
    size_t DataSize() const override { return PipelineEvent::DataSize() + mContent.size(); }

private:
    StringView mContent;
};

class PipelineEventPtr {
public:
    PipelineEventPtr() = default;
    explicit PipelineEventPtr(std::unique_ptr<PipelineEvent>&& p) : mData(std::move(p)) {}
    template <class T>
    bool Is() const;
    template <class T>
    T& Cast() {
        return static_cast<T&>(*mData);
    }
    template <class T>
    const T& Cast() const {
        return static_cast<const T&>(*mData);
    }
    PipelineEvent* operator->() { return mData.get(); }
    const PipelineEvent* operator->() const { return mData.get(); }
    explicit operator bool() const { return bool(mData); }

private:
    std::unique_ptr<PipelineEvent> mData;
};
template <>
inline bool PipelineEventPtr::Is<LogEvent>() const {
    return mData && mData->GetType() == PipelineEvent::Type::LOG;
}
template <>
inline bool PipelineEventPtr::Is<RawEvent>() const {
    return mData && mData->GetType() == PipelineEvent::Type::RAW;
}

using EventsContainer = std::vector<PipelineEventPtr>;

class PipelineEventGroup {
public:
    explicit PipelineEventGroup(const std::shared_ptr<SourceBuffer>& sb) : mSourceBuffer(sb) {}
    PipelineEventGroup(const PipelineEventGroup&) = delete;
    PipelineEventGroup& operator=(const PipelineEventGroup&) = delete;

    const EventsContainer& GetEvents() const { return mEvents; }
    EventsContainer& MutableEvents() { return mEvents; }
    void SwapEvents(EventsContainer& other) { mEvents.swap(other); }
    LogEvent* AddLogEvent() {
        auto e = std::make_unique<LogEvent>(this);
        LogEvent* raw = e.get();
        mEvents.emplace_back(std::move(e));
        return raw;
    }
    // (the reference draws events from the group's pool, PipelineEventGroup.h: CreateLogEvent(bool fromPool); no pool here)
    std::unique_ptr<LogEvent> CreateLogEvent(bool = false) { return std::make_unique<LogEvent>(this); }
    RawEvent* AddRawEvent() {
        auto e = std::make_unique<RawEvent>(this);
        RawEvent* raw = e.get();
        mEvents.emplace_back(std::move(e));
        return raw;
    }
    std::shared_ptr<SourceBuffer>& GetSourceBuffer() { return mSourceBuffer; }

    void SetMetadata(EventGroupMetaKey key, const std::string& val) {
        StringBuffer b = mSourceBuffer->CopyString(val);
        mMetadata[key] = StringView(b.data, b.size);
    }
    void SetMetadataNoCopy(EventGroupMetaKey key, StringView val) { mMetadata[key] = val; }
    StringView GetMetadata(EventGroupMetaKey key) const {
        auto it = mMetadata.find(key);
        return it == mMetadata.end() ? StringView() : it->second;
    }
    bool HasMetadata(EventGroupMetaKey key) const { return mMetadata.count(key) != 0; }
    void DelMetadata(EventGroupMetaKey key) { mMetadata.erase(key); }
    const GroupMetadata& GetAllMetadata() const { return mMetadata; }

    void SetTag(const std::string& key, const std::string& val) {
        StringBuffer k = mSourceBuffer->CopyString(key), v = mSourceBuffer->CopyString(val);
        mTags[StringView(k.data, k.size)] = StringView(v.data, v.size);
    }
    const GroupTags& GetTags() const { return mTags; }

    size_t DataSize() const {
        size_t n = DataSizeWithoutEvents();
        for (const auto& e : mEvents) n += e->DataSize();
        return n;
    }
    // DataSize() minus the events' own sizes (for a caller that has summed those while it had the events in hand)
    size_t DataSizeWithoutEvents() const {
        size_t n = sizeof(mEvents);
        for (const auto& kv : mTags) n += kv.first.size() + kv.second.size();
        return n;
    }

    // fixture format of the reference unit tests (PipelineEventGroup::FromJsonString / ToJsonString,
    // core/models/PipelineEventGroup.h:140-146, LogEvent.cpp:169-209)
    bool FromJsonString(const std::string& json, std::string* error = nullptr);
    std::string ToJsonString() const;

private:
    std::shared_ptr<SourceBuffer> mSourceBuffer;
    GroupMetadata mMetadata;
    GroupTags mTags;
    EventsContainer mEvents;
};

inline std::shared_ptr<SourceBuffer>& PipelineEvent::GetSourceBuffer() { return mGroup->GetSourceBuffer(); }
#ifdef LC_REFERENCE_SHAPED_EVENT_MODEL
inline LogEvent::LogEvent(PipelineEventGroup* g) : PipelineEvent(Type::LOG, g) { mContents.reserve(16); }
#else
inline LogEvent::LogEvent(PipelineEventGroup* g)
    : PipelineEvent(Type::LOG, g),
      mContents(g ? g->GetSourceBuffer().get() : nullptr) {}
#endif
inline void LogEvent::SetContent(StringView key, StringView val) {
    StringBuffer k = GetSourceBuffer()->CopyString(key), v = GetSourceBuffer()->CopyString(val);
    SetContentNoCopy(StringView(k.data, k.size), StringView(v.data, v.size));
}

}  // namespace logtail
