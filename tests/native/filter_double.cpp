// tests/native/filter_double.cpp -- TEST INFRASTRUCTURE ONLY: the HOST side of processor_filter_regex_gpu on a box without a GPU.
//
// csrc/processor_filter_gpu.cpp (Init's precedence of Include / FilterKey+FilterRegex / ConditionExp, the expression tree, which events
// survive, the non-UTF-8 blanking: ProcessorFilterNative.cpp:30-486) talks to the HIP runtime itself -- its own stream, pinned staging,
// one lc_regex_match_device_multi call per group.  This translation unit stands in for that runtime on the CPU: "device" memory is host
// memory, a stream is a token, a copy is a memcpy; the match call is answered by the CPU oracle's regex (oracle/bt_regex.h, full match =
// boost::regex_match).  tests/test_filter_host_double.py builds  processor_filter_gpu.cpp + event_model.cpp + this file  ->
// tests/_build/libfilter_double.so  and runs the PRODUCT's host code beside the reference's own ProcessorFilterNative.cpp compiled from
// /root/reference (oracle/_ref/libref_processor.so).  The device side is the -m gpu tests' business.
//
// It lives under tests/, is built only by the test that uses it and is never linked into loongcollector_amd/lib.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/lc_processor.h"
#include "../../include/lc_regex_gpu.h"
#include "../../loongcollector_amd/csrc/event_model.hpp"
#include "../../oracle/bt_regex.h"

// ---------------------------------------------------------------------------------------------- the HIP runtime, on the host
extern "C" {
hipError_t hipMalloc(void** p, size_t n) {
    *p = std::calloc(1, n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) {
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) {
    *p = std::malloc(n ? n : 1);
    if (*p) std::memset(*p, 0x01, n ? n : 1);  // pinned memory is NOT handed out zeroed: every byte reads LC_MATCH / 0x01010101
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void* p) {
    std::free(p);
    return hipSuccess;
}
static void runQueued();
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) {
    runQueued();  // (stream order: the copy sits behind the kernels queued before it)
    std::memcpy(dst, src, n);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) {
    *s = reinterpret_cast<hipStream_t>(std::malloc(8));
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    std::free(s);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) {
    runQueued();
    return hipSuccess;
}
const char* hipGetErrorString(hipError_t) { return "fake HIP runtime of the test double"; }
}  // extern "C"

// ---------------------------------------------------------------------------------------------- the library's runtime, on the host
struct lc_regex {
    orx_prog* prog = nullptr;
    int marks = 0;
};
static thread_local std::string tLastError;
static uint64_t gGaveUp = 0;

void lcNoteGaveUp(uint64_t n) { gGaveUp += n; }
bool lcRuntimeUsable() { return true; }
void lcRegisterExitHook() {}
void lcSetJobTableInPlace(bool) {}
int lcHostEntryDevice(int* dev) {
    if (dev) *dev = 0;
    return LC_OK;
}
// The device is ASYNCHRONOUS here too: a match call and the trip's signal are QUEUED, and run when the host waits for them -- unless
// the word the host waits on ALREADY reads the trip's number, in which case the wait returns at once and the host reads whatever the
// status block holds, exactly what a spinning runner thread would do on the device (the completion word used to lie
// behind the status bytes, where an earlier, longer group's bytes could spell the number).  The queued work is then dropped.
struct QueuedJob {
    lc_match_job job;
};
struct QueuedSignal {
    uint32_t* flag;
    uint32_t seq;
};
static thread_local std::vector<QueuedJob> tQueuedJobs;
static thread_local std::vector<QueuedSignal> tQueuedSignals;
static void runJob(const lc_match_job& J);
static void runQueued() {
    for (const QueuedJob& q : tQueuedJobs) runJob(q.job);
    tQueuedJobs.clear();
    for (const QueuedSignal& s : tQueuedSignals) *s.flag = s.seq;
    tQueuedSignals.clear();
}
static uint64_t gEarlyReturns = 0;
extern "C" uint64_t fd_early_returns(void) { return gEarlyReturns; }
int lcQueueTripSignal(uint32_t* hFlag, uint32_t seq, hipStream_t) {
    tQueuedSignals.push_back({hFlag, seq});
    return LC_OK;
}
int lcAwaitTripSignal(const uint32_t* hFlag, uint32_t seq, hipStream_t) {
    if (*hFlag == seq) {  // the host saw "done" before the device has run anything
        ++gEarlyReturns;
        tQueuedJobs.clear();
        tQueuedSignals.clear();
        return LC_OK;
    }
    runQueued();
    return *hFlag == seq ? LC_OK : LC_ERR_ARG;
}

extern "C" int lc_device_count(void) { return 1; }
extern "C" const char* lc_last_error(void) { return tLastError.c_str(); }
extern "C" int lc_regex_compile(const char* pattern, size_t n, uint32_t flags, int, lc_regex_t** out, char* err, size_t errcap) {
    if (!pattern || !out) return LC_ERR_ARG;
    *out = nullptr;
    unsigned oflags = 0;
    if (flags & LC_SYNTAX_ICASE) oflags |= ORX_ICASE;
    orx_prog* p = orx_compile(pattern, n, oflags, err, errcap);
    if (!p) return LC_ERR_SYNTAX;
    auto* re = new lc_regex;
    re->prog = p;
    re->marks = orx_mark_count(p);
    *out = re;
    return LC_OK;
}
extern "C" void lc_regex_free(lc_regex_t* re) {
    if (!re) return;
    orx_free(re->prog);
    delete re;
}
extern "C" int lc_upload_pinned(const void* src, void* dst, size_t nbytes, void*) {
    std::memcpy(dst, src, nbytes);
    return LC_OK;
}
extern "C" int lc_regex_match_device_multi(const lc_match_job* jobs, uint32_t njobs, void*) {
    for (uint32_t j = 0; j < njobs; ++j) tQueuedJobs.push_back({jobs[j]});
    return LC_OK;
}
static void runJob(const lc_match_job& J) {
    {
        std::vector<int32_t> what(size_t(J.re->marks + 1) * 2);
        for (uint32_t i = 0; i < J.n; ++i) {
            const uint32_t len = J.d_len ? J.d_len[i] : J.d_off[i + 1] - J.d_off[i] - J.sep_bytes;
            const int r = orx_fullmatch(J.re->prog, J.d_data + J.d_off[i], len, what.data());
            J.d_status[i] = r == 1 ? LC_MATCH : r == 0 ? LC_NOMATCH : LC_GAVE_UP;
            for (uint32_t g = 0; g < J.ngroups; ++g) {
                const bool have = r == 1 && int(g) < J.re->marks;
                J.d_caps[(size_t(i) * J.ngroups + g) * 2] = have ? what[(g + 1) * 2] : -1;
                J.d_caps[(size_t(i) * J.ngroups + g) * 2 + 1] = have ? what[(g + 1) * 2 + 1] : -1;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- the harness
extern "C" {
// fixture JSON in -> lc_filter_process -> fixture JSON out (malloc'ed; fd_free)
char* fd_process_json(lc_filter_t* f, const char* groupJson, char* err, size_t errcap) {
    logtail::PipelineEventGroup group(std::make_shared<logtail::SourceBuffer>());
    std::string error;
    if (!group.FromJsonString(groupJson, &error)) {
        std::snprintf(err, errcap, "%s", error.c_str());
        return nullptr;
    }
    const int rc = lc_filter_process(f, &group);
    if (rc != LC_OK) {
        std::snprintf(err, errcap, "lc_filter_process failed: %d (%s)", rc, lc_last_error());
        return nullptr;
    }
    return strdup(group.ToJsonString().c_str());
}
void fd_free(void* p) { std::free(p); }
}  // extern "C"
