// gpu_runtime.hip -- gfx950 kernels and the device half of the C ABI (include/lc_regex_gpu.h).
//
// Kernels (hand-written HIP for CDNA4, wave64):
//   tdfa_match_kernel : one log line per lane.  The pattern's tagged-DFA tables are staged once per workgroup
//                       into LDS; every lane walks its own line with aligned 16-byte global loads, one
//                       class lookup + one transition lookup (both LDS) per byte, and the few capture-offset
//                       register moves attached to a transition.  Registers live in LDS as [reg][lane] so that
//                       data-dependent register numbers never spill to scratch and never bank-conflict.
//   nfa_match_kernel  : one log line per wavefront, one lane per live NFA thread (priority order == lane order);
//                       follow lists in LDS, ballot/mbcnt compaction, ds_bpermute capture transfer.
// Byte-scan work: no MFMA.  The bound that matters is LDS lookup throughput / latency, then HBM.
#include <hip/hip_runtime.h>

#include <type_traits>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <thread>
#include <string>
#include <vector>

#include "device_tables.h"
#include "grok_runtime.hpp"
#include "runtime_internal.hpp"
#include "nfa_kernel.hpp"
#include "nfa_wide_kernel.hpp"
#include "nfa_decide_kernel.hpp"
#include "bt_kernel.hpp"
#include "regex_handle.hpp"
#include "sched_kernel.hpp"
#include "screen_kernel.hpp"
#include "tdfa_l2_kernel.hpp"
#include "tdfa_l2_layout.h"
#include "split_kernel.hpp"
#include "pipeline_kernel.hpp"
#include "tdfa_stream_kernel.hpp"
#include "gather_pool.hpp"
#include "device_binding.hpp"

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string tlsError;
static int hipFail(hipError_t e, const char* what) {
    tlsError = std::string(what) + ": " + hipGetErrorString(e);
    return LC_ERR_HIP;
}
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess) return hipFail(e_, #expr); \
    } while (0)

extern "C" const char* lc_last_error(void) { return tlsError.c_str(); }
// (runtime_internal.hpp: the other device translation units report through the same thread-local string)
void lcSetLastError(const std::string& msg) { tlsError = msg; }
int lcHipFail(hipError_t e, const char* what) { return hipFail(e, what); }

// Names of the match kernels the calling thread launched since it last asked (smoke() prints them, so that the GPU-box log
// shows which native code ran; not a profiler: names only, duplicates folded, capped)
static thread_local std::string tlsKernelLog;
void lcNoteKernel(const char* name);
static void noteKernel(const char* name) { lcNoteKernel(name); }
void lcNoteKernel(const char* name) {
    if (tlsKernelLog.size() > 512 || tlsKernelLog.find(name) != std::string::npos) return;
    if (!tlsKernelLog.empty()) tlsKernelLog += ", ";
    tlsKernelLog += name;
}
extern "C" size_t lc_launched_kernels(char* buf, size_t cap) {
    const size_t n = tlsKernelLog.size();
    if (buf && cap) {
        const size_t k = n < cap - 1 ? n : cap - 1;
        std::memcpy(buf, tlsKernelLog.data(), k);
        buf[k] = 0;
    }
    tlsKernelLog.clear();
    return n;
}

// Per-thread device resources (staging pipeline, Grok buffers, the decide pool) are freed by thread_local destructors.  A
// thread that ends while the process is already exiting must not call into a HIP runtime that may be gone: an atexit
// hook registered at first use (so it runs BEFORE the runtime's own teardown) turns those destructors into no-ops.
// lc_thread_release() frees the calling thread's resources explicitly; a host that recycles runner threads calls it.
static std::atomic<bool> gProcessExiting{false};
static void lcMarkExiting() { gProcessExiting.store(true); }
// Hardware queues.  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): streams beyond that share a
// queue and their kernels run one after the other.  A Grok batch queues its entries' latency-bound kernels on up to 16 worker streams
// and gains from 16 queues (profiles/round3_grok_streams.txt: 16 Ki values 6.39 ms with 8 streams on 4 queues, 3.85 ms with 16 on
// 16); the paths that are one stream per runner thread LOSE 15-20 % at 16 threads with 16 queues (same file).  The variable is
// process-wide and read once, when the runtime initialises: it is the HOST's decision.  The library never touches the environment
// on its own (round 3 did, from a constructor and from the Grok processor's Init: removed); a host that wants more queues either
// exports GPU_MAX_HW_QUEUES itself or calls lc_runtime_prefer_hw_queues(n) before the first HIP call of the process.
extern "C" int lc_runtime_prefer_hw_queues(int n) {
    if (n < 1 || n > 64) return LC_ERR_ARG;
    static std::mutex mu;  // (setenv is not safe against itself)
    std::lock_guard<std::mutex> lock(mu);
    if (getenv("GPU_MAX_HW_QUEUES")) return LC_OK;  // the host (or an earlier call) has an opinion already: kept
    char buf[16];
    snprintf(buf, sizeof buf, "%d", n);
    return setenv("GPU_MAX_HW_QUEUES", buf, 0) == 0 ? LC_OK : LC_ERR_ARG;
}

void lcRegisterExitHook() {
    static std::once_flag once;
    std::call_once(once, [] { atexit(lcMarkExiting); });
}
bool lcRuntimeUsable() { return !gProcessExiting.load(); }

extern "C" int lc_device_count(void) {
    // (a positive answer does not change during the process's life: every match call asks, from every runner thread, and a
    // runtime call per group is a shared lock per group)
    static std::atomic<int> known{0};
    int n = known.load(std::memory_order_relaxed);
    if (n > 0) return n;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    if (n > 0) known.store(n, std::memory_order_relaxed);
    return n;
}

// ------------------------------------------------------------------------------------------------ thread -> device
// The policy lives in device_binding.hpp (HIP-free: tests/native/binding_race.cpp runs it against a two-device double under
// ThreadSanitizer); here it meets the HIP runtime.  Entry points that take DEVICE pointers never switch devices: the caller owns
// the placement, and lcDeviceEntryDevice refuses a pointer of another device.
namespace {
struct HipDeviceApi {
    static int count() { return lc_device_count(); }
    static bool get(int* dev) {
        const hipError_t e = hipGetDevice(dev);
        if (e != hipSuccess) (void)hipFail(e, "hipGetDevice");
        return e == hipSuccess;
    }
    static bool set(int dev) {
        const hipError_t e = hipSetDevice(dev);
        if (e != hipSuccess) (void)hipFail(e, "hipSetDevice");
        return e == hipSuccess;
    }
};
typedef lcbind::Binder<HipDeviceApi, kLcMaxDevices> DeviceBinder;
DeviceBinder& binder() {
    static DeviceBinder* b = new DeviceBinder();  // (never destroyed: threads that end during process exit still return their ordinals)
    return *b;
}
thread_local DeviceBinder::Thread tlsBind;
static_assert(lcbind::kInherit == LC_BIND_INHERIT && lcbind::kRoundRobin == LC_BIND_ROUND_ROBIN && lcbind::kFixed == LC_BIND_FIXED, "policy values");
int bindRc(int rc, const std::string& err) {  // device_binding.hpp's codes -> the C ABI's
    if (rc >= 0) return LC_OK;
    if (!err.empty()) tlsError = err;
    return rc == lcbind::kErrArg ? LC_ERR_ARG : rc == lcbind::kErrNoDevice ? LC_ERR_NO_DEVICE : LC_ERR_HIP;
}
}  // namespace

extern "C" int lc_runtime_device_for_ordinal(uint32_t ordinal, int ndevices) { return DeviceBinder::deviceForOrdinal(ordinal, ndevices); }

extern "C" int lc_runtime_set_bind_policy(int policy, int device) { return binder().setPolicy(policy, device) == lcbind::kOk ? LC_OK : LC_ERR_ARG; }

extern "C" int lc_runtime_bind_policy(void) { return binder().policy(); }

extern "C" int lc_runtime_bind_thread(int policy) {
    std::string err;
    const int d = binder().bindThread(tlsBind, policy, &err);
    return d >= 0 ? d : -bindRc(d, err);
}

extern "C" int lc_runtime_set_thread_device(int device) {
    std::string err;
    return bindRc(binder().setThreadDevice(tlsBind, device, &err), err);
}

extern "C" int lc_runtime_thread_device(void) { return tlsBind.device; }

int lcHostEntryDevice(int* dev) {
    std::string err;
    return bindRc(binder().hostEntryDevice(tlsBind, dev, &err), err);
}

int lcDeviceEntryDevice(const void* d_ptr, int* dev) {
    HIP_TRY(hipGetDevice(dev));
    if (*dev >= kLcMaxDevices) return LC_ERR_ARG;
    // where the caller's buffer lives: asked once per (pointer, current device), a caller hands over the same buffers batch after batch
    static thread_local const void* lastPtr = nullptr;
    static thread_local int lastDev = -1;
    if (!d_ptr || (d_ptr == lastPtr && lastDev == *dev)) return LC_OK;
    if (lc_device_count() > 1) {
        hipPointerAttribute_t attr;
        const hipError_t e = hipPointerGetAttributes(&attr, d_ptr);
        if (e != hipSuccess) {
            (void)hipGetLastError();  // (not a pointer the runtime knows: e.g. fine-grained host memory mapped by someone else; not ours to judge)
        } else if (attr.type == hipMemoryTypeDevice && attr.device != *dev) {
            tlsError = "device pointer belongs to device " + std::to_string(attr.device) + ", the calling thread's current device is " +
                       std::to_string(*dev);
            return LC_ERR_ARG;
        }
    }
    lastPtr = d_ptr;
    lastDev = *dev;
    return LC_OK;
}

// ------------------------------------------------------------------------------------------------ device tables
enum { kBlobNfa = 0, kBlobTdfa = 1, kBlobTdfaWide = 2, kBlobScreen = 3, kBlobTdfaL2 = 4, kBlobBt = 5 };
static int ensureUploaded(lc_regex* re, int dev, int which, void** out) {
    std::lock_guard<std::mutex> g(re->deviceMutex);
    void** slot = which == kBlobTdfa ? &re->dTdfaBlob[dev]
                  : which == kBlobTdfaWide ? &re->dTdfaWideBlob[dev]
                  : which == kBlobScreen ? &re->dScreenBlob[dev]
                  : which == kBlobTdfaL2 ? &re->dTdfaL2Blob[dev]
                  : which == kBlobBt ? &re->dBtBlob[dev] : &re->dNfaBlob[dev];
    if (!*slot) {
        const std::vector<uint32_t>& blob = which == kBlobTdfa ? re->tdfaBlob
                                            : which == kBlobTdfaWide ? re->tdfaWideBlob
                                            : which == kBlobScreen ? re->screenBlob
                                            : which == kBlobTdfaL2 ? re->tdfaL2Blob
                                            : which == kBlobBt ? re->btBlob : re->nfaBlob;
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, blob.size() * 4 + 16));  // + one word behind the tables: the compact kernel's long-line flag
        hipError_t e = hipMemset(p, 0, blob.size() * 4 + 16);
        if (e == hipSuccess) e = hipMemcpy(p, blob.data(), blob.size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(p);
            return hipFail(e, "hipMemcpy(tables)");
        }
        *slot = p;
    }
    *out = *slot;
    return LC_OK;
}

// The lazy automaton's device copy for `dev` (regex_handle.hpp LcLazyTdfa): uploaded when the handle's version is newer than the copy;
// the older copy retires -- launches of other threads may still read it -- and is freed with the handle.  hostBlobOut: the header words.
static int ensureLazyUploaded(lc_regex* re, int dev, void** out, std::vector<uint32_t>* hostHeader) {
    LcLazyTdfa& Z = re->lazy;
    std::lock_guard<std::mutex> g(Z.m);
    if (Z.blob.empty() || Z.disabled) {
        *out = nullptr;
        return LC_OK;
    }
    if (!Z.dBlob[dev] || Z.dVersion[dev] != Z.version) {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, Z.blob.size() * 4 + 16));
        const hipError_t e = hipMemcpy(p, Z.blob.data(), Z.blob.size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(p);
            return hipFail(e, "hipMemcpy(lazy tables)");
        }
        if (Z.dBlob[dev]) Z.retired.emplace_back(dev, Z.dBlob[dev]);
        Z.dBlob[dev] = p;
        Z.dVersion[dev] = Z.version;
    }
    *out = Z.dBlob[dev];
    hostHeader->assign(Z.blob.begin(), Z.blob.begin() + TL_HEADER_WORDS);
    Z.launches.fetch_add(1, std::memory_order_relaxed);
    return LC_OK;
}

void lcReleaseDeviceTables(lc_regex* re) {
    int cur = 0;
    bool haveCur = hipGetDevice(&cur) == hipSuccess;
    {
        LcLazyTdfa& Z = re->lazy;
        for (int d = 0; d < kLcMaxDevices; ++d)
            if (Z.dBlob[d]) {
                Z.retired.emplace_back(d, Z.dBlob[d]);
                Z.dBlob[d] = nullptr;
            }
        for (const auto& r : Z.retired)
            if (hipSetDevice(r.first) == hipSuccess) (void)hipFree(r.second);
        Z.retired.clear();
    }
    for (int d = 0; d < kLcMaxDevices; ++d) {
        if (re->dTdfaBlob[d] || re->dNfaBlob[d] || re->dTdfaWideBlob[d] || re->dScreenBlob[d] || re->dTdfaL2Blob[d] || re->dBtBlob[d]) {
            if (hipSetDevice(d) == hipSuccess) {
                if (re->dBtBlob[d]) (void)hipFree(re->dBtBlob[d]);
                if (re->dTdfaL2Blob[d]) (void)hipFree(re->dTdfaL2Blob[d]);
                if (re->dScreenBlob[d]) (void)hipFree(re->dScreenBlob[d]);
                if (re->dTdfaBlob[d]) (void)hipFree(re->dTdfaBlob[d]);
                if (re->dTdfaWideBlob[d]) (void)hipFree(re->dTdfaWideBlob[d]);
                if (re->dNfaBlob[d]) (void)hipFree(re->dNfaBlob[d]);
            }
            re->dTdfaBlob[d] = re->dNfaBlob[d] = re->dTdfaWideBlob[d] = re->dScreenBlob[d] = re->dTdfaL2Blob[d] = re->dBtBlob[d] = nullptr;
        }
    }
    if (haveCur) (void)hipSetDevice(cur);
}

// A caller that wants the match's last kernel to signal completion through host memory arms this before the match call
// (runHostPipeline's zero-copy path).  The launcher that can honour it marks it consumed; otherwise the caller queues
// lc_signal_kernel behind the match.
struct DoneRequest {
    uint32_t* counter = nullptr;  // device word, zero between launches
    uint32_t* flag = nullptr;     // pinned host word
    uint32_t seq = 0;
    bool armed = false, consumed = false;
};
static thread_local DoneRequest tlsDone;
static thread_local bool tlsJobTableInPlace = false;  // lcSetJobTableInPlace (zero-copy trips of the processors, below)

// The kernel behind a (workgroup size, table format) pair: the interleaved-issue kernel (tdfa_stream_kernel.hpp) for the
// class-indexed tables with or without the byte-pair extension, the phase-separated one (tdfa_kernel.hpp) for byte-indexed
// rows -- and for everything when LC_TDFA_STREAM=0 is set (A/B measurements).
template <int BLOCK, bool PAIR, bool COMPACT = false, bool BYTEROWS = false>
static int launchTdfaBlock(const void* dBlob, uint32_t blobBytes, uint32_t regBytes, size_t lds, const uint8_t* d_data,
                           const uint32_t* d_off, const uint32_t* d_len, uint32_t sep, uint32_t minLen, uint32_t n, const uint32_t* d_n, const uint32_t* d_order, const uint32_t* d_resume, uint32_t ngroups,
                           int32_t* d_caps, uint8_t* d_status, hipStream_t stream, uint32_t* longFlag = nullptr, uint32_t seq = 0,
                           uint32_t nregsWord = 0, bool pairOne = false) {
    static const bool streamOff = [] {
        const char* e = getenv("LC_TDFA_STREAM");
        return e && e[0] == '0';
    }();
    // tables without any general register program (TD_NREGS_NO_GENERAL: the usual case once multi-stamp programs are folded,
    // regex_handle.cpp) run the instantiation that neither tracks nor replays them; LC_TDFA_NOGEN=0 keeps the checking one
    static const bool noGenOff = [] {
        const char* e = getenv("LC_TDFA_NOGEN");
        return e && e[0] == '0';
    }();
    const bool noGen = !PAIR && !BYTEROWS && !streamOff && !noGenOff && (nregsWord & TD_NREGS_NO_GENERAL) != 0;
    // COMPACT tiles (rows of exactly 64 bytes) are filled by LDS-DMA: no staging VGPRs (113 -> 89: 20 instead of 16 waves per CU),
    // no ds_write_b128 (tdfa_stream_kernel.hpp, kLabDmaStage; round 3: 0.217 -> 0.208 ms on the headline batch).  LC_TDFA_DMA=0:
    // the register-staged original (A/B measurements).
    static const bool dmaOff = [] {
        const char* e = getenv("LC_TDFA_DMA");
        return e && e[0] == '0';
    }();
    const bool dma = COMPACT && !PAIR && !BYTEROWS && !streamOff && !dmaOff;
    bool cmapA8 = false;
    auto kern = tdfa_match_kernel<BLOCK, PAIR, COMPACT, BYTEROWS>;
    if constexpr (!BYTEROWS) {
        if (!streamOff) kern = tdfa_stream_kernel<BLOCK, COMPACT, PAIR>;
        if constexpr (!PAIR) {
            if (noGen) kern = tdfa_stream_kernel<BLOCK, COMPACT, false, kTdfaNoGeneralPrograms>;
            if constexpr (COMPACT) {
                if (dma) kern = noGen ? tdfa_stream_kernel<BLOCK, true, false, kTdfaNoGeneralPrograms | kLabDmaStage>
                                      : tdfa_stream_kernel<BLOCK, true, false, kLabDmaStage>;
            }
        } else {
            // a ONE-STAMP pair table (LC_TDFA_PAIR=2, device_tables.h TP_FORMAT 1) is only understood by the instantiation made for it
            if (pairOne) {
                if (streamOff) {
                    tlsError = "one-stamp pair tables need the stream kernel (LC_TDFA_STREAM=0 is set)";
                    return LC_ERR_UNSUPPORTED;
                }
                kern = tdfa_stream_kernel<BLOCK, COMPACT, true, kTdfaNoGeneralPrograms | kLabPairOne | (COMPACT ? kLabDmaStage : 0)>;
                // (round 6) ... with the first byte's class from a u8 copy of cmapA that the workgroup builds behind its tiles (272 bytes):
                // ASCII bytes then sit on 32 different banks, where the u16 table put b and b + 64 on one.  Only where the extra LDS does
                // not cost a resident workgroup.  MEASURED AND LEFT OFF (LC_TDFA_CMAPA8=1 switches it on): 0.1753 against 0.1719 ms per 1 Mi
                // lines -- the conflicts of the class map are not what the chain link waits for (profiles/round6_tdfa_why_not.md section 8).
                const char* a8Env = getenv("LC_TDFA_CMAPA8");
                const size_t half = kLcLdsPerCu / 2;
                if (a8Env && a8Env[0] == '1' && (lds + kTdfaCmapA8Bytes <= half || lds > half) && lds + kTdfaCmapA8Bytes <= kLcLdsPerCu) {
                    kern = tdfa_stream_kernel<BLOCK, COMPACT, true, kTdfaNoGeneralPrograms | kLabPairOne | kLabCmapA8 | (COMPACT ? kLabDmaStage : 0)>;
                    lds += kTdfaCmapA8Bytes;
                    cmapA8 = true;
                }
            }
        }
    }
    static thread_local size_t ldsAttrSet[kLcMaxDevices][7] = {};  // the attribute belongs to (function, device)
    const int which = cmapA8 ? 6 : (PAIR && pairOne) ? 5 : dma ? (noGen ? 4 : 3) : noGen ? 2 : (!BYTEROWS && !streamOff) ? 1 : 0;
    int devNow = 0;
    if (lds > 64 * 1024) HIP_TRY(hipGetDevice(&devNow));
    if (lds > 64 * 1024 && devNow < kLcMaxDevices && lds > ldsAttrSet[devNow][which]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        ldsAttrSet[devNow][which] = lds;
    }
    // (measurement knob: LC_TDFA_EXTRA_LDS=<bytes> of unused LDS per workgroup lowers the number of resident workgroups -- how the
    // kernel's time scales with the lines in flight per CU says whether it waits for latency or for a pipe)
    static const size_t extraLds = [] {
        const char* e = getenv("LC_TDFA_EXTRA_LDS");
        return e ? size_t(atol(e)) : size_t(0);
    }();
    if (extraLds) {
        lds += extraLds;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    }
    uint32_t grid = (n + BLOCK - 1) / BLOCK;
    bool persistSel = false;
    // (round 6) PERSISTENT WAVEFRONTS for the one-stamp COMPACT kernel on batches of several rounds of workgroups: the launch holds as
    // many workgroups as the chip does at once, each wavefront goes on with its lines of the next block (tdfa_stream_kernel.hpp
    // kLabPersist).  MEASURED AND LEFT OFF (LC_TDFA_PERSIST=1 switches it on; the GPU parity suite runs it): 0.1825 against 0.1700 ms per
    // 1 Mi lines, 0.344 against 0.316 ms per 2 Mi -- as with round 3's persistent workgroups, the dispatcher starting a new workgroup in
    // the middle of the resident ones' loops beats 4 096 wavefronts that start together and stay in step through every block
    // (profiles/round6_tdfa_why_not.md section 9).
    if constexpr (COMPACT && PAIR && !BYTEROWS && BLOCK == 512) {
        const char* pe = getenv("LC_TDFA_PERSIST");
        const bool persistOn = pe && pe[0] == '1';
        if (persistOn && pairOne && !streamOff && !cmapA8 && !extraLds && minLen == 0) {
            static thread_local uint32_t resident[kLcMaxDevices] = {};  // workgroups the device holds at once (LDS-bound: per CU)
            int devP = 0;
            HIP_TRY(hipGetDevice(&devP));
            if (devP < kLcMaxDevices && !resident[devP]) {
                hipDeviceProp_t prop;
                HIP_TRY(hipGetDeviceProperties(&prop, devP));
                // (LDS-bound, and never more than the 32 wavefronts of a CU: four workgroups of 512)
                resident[devP] = uint32_t(prop.multiProcessorCount) * uint32_t(std::min<size_t>(4, std::max<size_t>(1, kLcLdsPerCu / lds)));
            }
            const uint32_t slots = devP < kLcMaxDevices ? resident[devP] : 0u;
            if (slots && grid > slots) {
                kern = tdfa_stream_kernel<512, true, true, kTdfaNoGeneralPrograms | kLabPairOne | kLabDmaStage | kLabPersist>;
                static thread_local size_t persistAttr[kLcMaxDevices] = {};
                if (lds > 64 * 1024 && lds > persistAttr[devP]) {
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
                    persistAttr[devP] = lds;
                }
                grid = slots;
                persistSel = true;
            }
        }
    }
    // (the mop-up launch behind a COMPACT one: an instantiation whose workgroups take the line blocks in turn, tdfa_stream_kernel.hpp
    // kLabMopUp -- for the workgroup size such a launch has in practice)
    if constexpr (BLOCK == 256 && !COMPACT && !BYTEROWS) {
        if (minLen && !streamOff && !cmapA8) {
            if constexpr (PAIR) kern = pairOne ? tdfa_stream_kernel<256, false, true, kTdfaNoGeneralPrograms | kLabPairOne | kLabMopUp>
                                               : tdfa_stream_kernel<256, false, true, kLabMopUp>;
            else kern = noGen ? tdfa_stream_kernel<256, false, false, kTdfaNoGeneralPrograms | kLabMopUp> : tdfa_stream_kernel<256, false, false, kLabMopUp>;
            grid = std::min(grid, 256u);
            static thread_local size_t mopAttrSet[kLcMaxDevices][4] = {};  // (function, device), as above
            const int mopWhich = PAIR ? (pairOne ? 0 : 1) : (noGen ? 2 : 3);
            if (lds > 64 * 1024 && devNow < kLcMaxDevices && lds > mopAttrSet[devNow][mopWhich]) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
                mopAttrSet[devNow][mopWhich] = lds;
            }
        }
    }
    noteKernel(persistSel ? "tdfa_stream_kernel<compact,nogeneral,pair1,dma,persist>" : cmapA8 ? (COMPACT ? "tdfa_stream_kernel<compact,nogeneral,pair1,dma,a8>" : "tdfa_stream_kernel<nogeneral,pair1,a8>")
               : (PAIR && pairOne) ? (COMPACT ? "tdfa_stream_kernel<compact,nogeneral,pair1,dma>" : "tdfa_stream_kernel<nogeneral,pair1>") : dma ? (noGen ? "tdfa_stream_kernel<compact,nogeneral,dma>" : "tdfa_stream_kernel<compact,dma>") : noGen ? (COMPACT ? "tdfa_stream_kernel<compact,nogeneral>" : "tdfa_stream_kernel<nogeneral>") : which ? (PAIR ? (COMPACT ? "tdfa_stream_kernel<compact,pair>" : "tdfa_stream_kernel<pair>") : (COMPACT ? "tdfa_stream_kernel<compact>" : "tdfa_stream_kernel"))
                     : (BYTEROWS ? "tdfa_match_kernel<byterows>" : "tdfa_match_kernel"));
    // (hipLaunchKernel reports the launch's own status: no second runtime call to fetch it)
    const uint32_t* blobArg = static_cast<const uint32_t*>(dBlob);
    // completion signal requested by the zero-copy host path (tlsDone, below): this launch carries it when it is the match's
    // only launch (no mop-up launch follows, nothing runs behind it)
    uint32_t* doneCounter = nullptr;
    uint32_t* doneFlag = nullptr;
    uint32_t doneSeq = 0;
    if (tlsDone.armed && minLen == 0 && longFlag == nullptr) {
        doneCounter = tlsDone.counter;
        doneFlag = tlsDone.flag;
        doneSeq = tlsDone.seq;
        tlsDone.consumed = true;
    }
    // (Round 3, tried and dropped: PERSISTENT WAVES -- the launch sized to the resident workgroups, every wave going on with the
    // next 64 lines: tables staged once, a wave's result stores overlapping its next first loads.  Measured on the headline
    // batch: 0.243 ms against 0.227 -- the dispatcher already starts a new workgroup the moment one retires, in the middle of
    // the other resident workgroups' loops, which hides the prologue better than a wave serialising its own epilogue and
    // prologue does.)
    void* args[] = {&d_data, &d_off, &d_len, &sep, &minLen, &n, &d_n, &d_order, &d_resume, &blobArg, &blobBytes, &regBytes, &ngroups,
                    &d_caps, &d_status, &longFlag, &seq, &doneCounter, &doneFlag, &doneSeq};
    HIP_TRY(hipLaunchKernel(reinterpret_cast<const void*>(kern), dim3(grid), dim3(BLOCK), args, lds, stream));
    return LC_OK;
}

// the byte-pair extension of a packed blob is a ONE-STAMP table (device_tables.h TP_FORMAT)
static bool lcPairOneFormat(const std::vector<uint32_t>& blob) {
    const uint32_t po = blob[TD_OFF_PAIR];
    return po != 0 && po / 4 + TP_FORMAT < blob.size() && blob[po / 4 + TP_FORMAT] == 1;
}

// The two walks of an automaton whose tables live in global memory (tdfa_l2_kernel.hpp), chosen per launch.  `hostBlob`: the blob's
// words on the host (header fields); `dBlob`: its device copy.  pendingFlag / seq: lazy automata only (TL_MISS) -- where a line that
// stepped on an uncomputed transition raises the launch's pending flag.  launchTdfa (the LDS kernels) serves a handle that asked for the
// wave walk (`waveByChoice`) when LC_TDFA_WAVE_MAX=0 takes the wave walk away.
static int launchTdfa(lc_regex* re, int dev, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len,
                      uint32_t sep, uint32_t n, const uint32_t* d_n, const uint32_t* d_order, const uint32_t* d_resume, uint32_t ngroups, int32_t* d_caps,
                      uint8_t* d_status, hipStream_t stream);
static int launchTdfaL2Family(lc_regex* re, const uint32_t* hostBlob, const void* dBlob, bool waveByChoice, int dev, const uint8_t* d_data,
                              const uint32_t* d_off, const uint32_t* d_len, uint32_t sep, uint32_t n, const uint32_t* d_n,
                              const uint32_t* d_order, const uint32_t* d_resume, uint32_t ngroups, int32_t* d_caps, uint8_t* d_status,
                              hipStream_t stream, uint32_t* pendingFlag, uint32_t seq) {
    const uint32_t nRegs = hostBlob[TL_NREGS];
    // Small and medium batches wait for their longest value: ONE VALUE PER WAVEFRONT (tdfa_wave_kernel: wave-uniform state, quiet
    // runs crossed 256 bytes at a time).  Large batches are about values in flight: one value per lane.  LC_TDFA_WAVE_MAX: the
    // largest batch that takes the wave kernel (0 = never; A/B measurements).
    // (read at every launch: the GPU tests run both kernels in one process)
    const char* waveEnv = getenv("LC_TDFA_WAVE_MAX");
    const uint32_t waveMax = uint32_t(waveEnv ? atol(waveEnv) : 65536);
    const bool perWave = n <= waveMax && (waveMax != 0 || !waveByChoice);
    if (waveByChoice && !perWave)  // (LC_TDFA_WAVE_MAX=0: the LDS kernels, as if the handle had not asked)
        return launchTdfa(re, dev, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, stream);
    size_t lds = perWave ? size_t(nRegs) * kTdfaWaveValues * 4 : size_t(nRegs) * kTdfaL2Block * 4;
    // the register programs (opsStart + ops, contiguous in the blob) ride in LDS when the batch is small (tdfa_l2_kernel.hpp)
    uint32_t stageBytes = 0;
    static const bool stageOffAll = getenv("LC_TDFA_L2_NO_STAGE") != nullptr;
    {
        const uint32_t progBytes = (hostBlob[TL_OFF_FINALID] - hostBlob[TL_OFF_OPSSTART] + 3u) & ~3u;
        if (!stageOffAll && n <= 32768 && progBytes <= 40 * 1024 && lds + progBytes <= 60 * 1024) stageBytes = progBytes;
    }
    if (perWave) {
        // (round 5) a small automaton rides in LDS whole: transition table + register programs (tdfa_l2_kernel.hpp LT)
        // MEASURED AND LEFT OFF (LC_TDFA_WAVE_LDS_TRANS=1 switches it on): once the walk's state lives in SGPRs the transition read is a
        // scalar load through the scalar cache, as fast as the LDS read + readfirstlane, without staging up to 48 KB per four values
        // (CISCOFW105003 on its 314 values: 0.389 ms from L2, 0.411 ms from LDS; profiles/round5_wave_step.txt)
        const bool transOff = [] {  // (read per launch: the GPU tests run both forms)
            const char* v = getenv("LC_TDFA_WAVE_LDS_TRANS");
            return !(v && v[0] == '1');
        }();
        const uint32_t allBytes = (hostBlob[TL_OFF_FINALID] - hostBlob[TL_OFF_TRANS] + 3u) & ~3u;
        const bool ldsTrans = !transOff && !stageOffAll && n <= 32768 && allBytes <= 48 * 1024 && lds + allBytes <= 60 * 1024;
        if (ldsTrans) stageBytes = allBytes;
        lds += stageBytes;
        static thread_local size_t waveLdsAttrSet[2][kLcMaxDevices] = {};
        if (lds > 48 * 1024 && dev < kLcMaxDevices && lds > waveLdsAttrSet[ldsTrans][dev]) {
            HIP_TRY(hipFuncSetAttribute(ldsTrans ? reinterpret_cast<const void*>(tdfa_wave_kernel<true>)
                                                 : reinterpret_cast<const void*>(tdfa_wave_kernel<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
            waveLdsAttrSet[ldsTrans][dev] = lds;
        }
        noteKernel(pendingFlag ? "tdfa_l2_kernel:wave:lazy" : ldsTrans ? "tdfa_l2_kernel:wave:lds" : "tdfa_l2_kernel:wave");
        if (ldsTrans)
            hipLaunchKernelGGL(tdfa_wave_kernel<true>, dim3((n + kTdfaWaveValues - 1) / kTdfaWaveValues), dim3(kTdfaWaveBlock), lds, stream,
                               d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, static_cast<const uint32_t*>(dBlob), ngroups, d_caps,
                               d_status, stageBytes, pendingFlag, seq);
        else
            hipLaunchKernelGGL(tdfa_wave_kernel<false>, dim3((n + kTdfaWaveValues - 1) / kTdfaWaveValues), dim3(kTdfaWaveBlock), lds, stream,
                               d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, static_cast<const uint32_t*>(dBlob), ngroups, d_caps,
                               d_status, stageBytes, pendingFlag, seq);
        HIP_TRY(hipGetLastError());
        return LC_OK;
    }
    lds += stageBytes;
    static thread_local size_t ldsAttrSet[kLcMaxDevices] = {};
    if (lds > 48 * 1024 && dev < kLcMaxDevices && lds > ldsAttrSet[dev]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tdfa_l2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        ldsAttrSet[dev] = lds;
    }
    noteKernel(pendingFlag ? "tdfa_l2_kernel:lazy" : "tdfa_l2_kernel");
    // (no completion signal of its own: a caller that polls queues lc_signal_kernel behind it, see tlsDone)
    hipLaunchKernelGGL(tdfa_l2_kernel, dim3((n + kTdfaL2Block - 1) / kTdfaL2Block), dim3(kTdfaL2Block), lds, stream, d_data, d_off,
                       d_len, sep, n, d_n, d_order, d_resume, static_cast<const uint32_t*>(dBlob), ngroups, d_caps, d_status, stageBytes,
                       pendingFlag, seq);
    HIP_TRY(hipGetLastError());
    return LC_OK;
}

static int launchTdfa(lc_regex* re, int dev, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len,
                      uint32_t sep, uint32_t n, const uint32_t* d_n, const uint32_t* d_order, const uint32_t* d_resume, uint32_t ngroups, int32_t* d_caps,
                      uint8_t* d_status, hipStream_t stream) {
    void* dBlob = nullptr;
    int rc = ensureUploaded(re, dev, kBlobTdfa, &dBlob);
    if (rc != LC_OK) return rc;
    // COMPACT variant (16-bit offset registers, more lines in flight per CU): it takes every line shorter than 64 KiB; the
    // 32-bit kernel below then only looks at what is left (usually nothing: its workgroups read their lines' lengths and
    // leave).  Two launches only pay on batches large enough to fill the chip several times over.
    constexpr uint32_t kCompactMinLines = 1u << 16;
    uint32_t minLen = 0;
    uint32_t* longFlag = nullptr;
    uint32_t seq = 0;
    if (!re->tdfaWideBlob.empty() && (re->tdfaWideForced || n >= kCompactMinLines)) {
        void* dWide = nullptr;
        rc = ensureUploaded(re, dev, kBlobTdfaWide, &dWide);
        if (rc != LC_OK) return rc;
        const uint32_t wideBytes = uint32_t(re->tdfaWideBlob.size() * 4);
        longFlag = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(dWide) + wideBytes);
        seq = ++re->tdfaWideSeq[dev];
        if (seq == 0) {  // the 32-bit sequence wrapped: start over below every flag value seen so far
            HIP_TRY(hipMemsetAsync(longFlag, 0, 4, stream));
            seq = ++re->tdfaWideSeq[dev];
        }
        const int wb = re->tdfaWideBlock;
        const uint32_t wRegBytes = uint32_t(size_t(re->tdfaWidePackedRegs + 1) * size_t(wb) * 2);
        const size_t wLds = lcTdfaCompactLdsBytes(wideBytes, re->tdfaWidePackedRegs, wb);
        if (wb == kLcTdfaWideBlock)
            rc = launchTdfaBlock<kLcTdfaWideBlock, false, true, true>(dWide, wideBytes, wRegBytes, wLds, d_data, d_off, d_len, sep, 0, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, stream, longFlag, seq, re->tdfaWideBlob[TD_NREGS]);
        else if (wb == 512 && re->tdfaWideBlob[TD_OFF_PAIR])
            rc = launchTdfaBlock<512, true, true, false>(dWide, wideBytes, wRegBytes, wLds, d_data, d_off, d_len, sep, 0, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, stream, longFlag, seq, re->tdfaWideBlob[TD_NREGS], lcPairOneFormat(re->tdfaWideBlob));
        else if (wb == 512)
            rc = launchTdfaBlock<512, false, true, false>(dWide, wideBytes, wRegBytes, wLds, d_data, d_off, d_len, sep, 0, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, stream, longFlag, seq, re->tdfaWideBlob[TD_NREGS]);
        else if (re->tdfaWideBlob[TD_OFF_PAIR])
            rc = launchTdfaBlock<256, true, true, false>(dWide, wideBytes, wRegBytes, wLds, d_data, d_off, d_len, sep, 0, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, stream, longFlag, seq, re->tdfaWideBlob[TD_NREGS], lcPairOneFormat(re->tdfaWideBlob));
        else
            rc = launchTdfaBlock<256, false, true, false>(dWide, wideBytes, wRegBytes, wLds, d_data, d_off, d_len, sep, 0, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, stream, longFlag, seq, re->tdfaWideBlob[TD_NREGS]);
        if (rc != LC_OK) return rc;
        minLen = kTdfaWideMaxLine + 1;
    }
    const uint32_t blobBytes = uint32_t(re->tdfaBlob.size() * 4);
    const int block = re->tdfaBlock;  // the blob's register offsets are encoded for this workgroup size
    if (block == 0) {
        tlsError = "tdfa tables + registers exceed LDS";
        return LC_ERR_UNSUPPORTED;
    }
    const size_t lds = lcTdfaLdsBytes(blobBytes, re->tdfaPackedRegs, block);
    const uint32_t regBytes = uint32_t(lcTdfaRegBytes(re->tdfaPackedRegs, block));
    // small automata carry a byte-pair transition table: half as many dependent LDS lookups per byte
    static const bool pairOff = getenv("LC_TDFA_NO_PAIR") != nullptr;
    const bool pair = re->tdfaBlob[TD_OFF_PAIR] != 0 && !pairOff;
    switch (block) {
        case 256: return pair ? launchTdfaBlock<256, true>(dBlob, blobBytes, regBytes, lds, d_data, d_off, d_len, sep, minLen, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, stream, longFlag, seq, re->tdfaBlob[TD_NREGS], lcPairOneFormat(re->tdfaBlob)) : launchTdfaBlock<256, false>(dBlob, blobBytes, regBytes, lds, d_data, d_off, d_len, sep, minLen, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, stream, longFlag, seq, re->tdfaBlob[TD_NREGS]);
        case 128: return pair ? launchTdfaBlock<128, true>(dBlob, blobBytes, regBytes, lds, d_data, d_off, d_len, sep, minLen, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, stream, longFlag, seq, re->tdfaBlob[TD_NREGS], lcPairOneFormat(re->tdfaBlob)) : launchTdfaBlock<128, false>(dBlob, blobBytes, regBytes, lds, d_data, d_off, d_len, sep, minLen, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, stream, longFlag, seq, re->tdfaBlob[TD_NREGS]);
        default: return pair ? launchTdfaBlock<64, true>(dBlob, blobBytes, regBytes, lds, d_data, d_off, d_len, sep, minLen, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, stream, longFlag, seq, re->tdfaBlob[TD_NREGS], lcPairOneFormat(re->tdfaBlob)) : launchTdfaBlock<64, false>(dBlob, blobBytes, regBytes, lds, d_data, d_off, d_len, sep, minLen, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, stream, longFlag, seq, re->tdfaBlob[TD_NREGS]);
    }
}


// ------------------------------------------------------------------------------------------------ decide pool
// Scratch of the depth-first decide kernel (nfa_decide_kernel.hpp): one pool per host thread and device, allocated the first
// time the thread launches an NFA program whose thread lists can overflow.  Launches of one thread on different streams
// share the pool, so they are chained through an event (the decide launches only: the match kernels still overlap).
namespace {
struct DecidePool {
    uint8_t* p = nullptr;
    size_t bytes = 0;
    int device = -1;
    hipEvent_t lastUse = nullptr;
    hipStream_t lastStream = nullptr;
    bool used = false;
    uint64_t settled = 0, gaveUp = 0;  // accumulated by lc_decide_stats
    ~DecidePool() { release(); }
    void release() {
        if (p && lcRuntimeUsable()) {
            (void)hipSetDevice(device);
            (void)hipDeviceSynchronize();
            (void)hipFree(p);
            if (lastUse) (void)hipEventDestroy(lastUse);
        }
        p = nullptr;
        bytes = 0;
        device = -1;
        lastUse = nullptr;
        used = false;
    }
};
// One pool per STREAM SLOT of the thread: slot 0 for ordinary calls; the Grok matcher runs its entries on a few worker streams
// (grok_device.hip) and selects slot 1.. before it queues an entry's launches, so that entries on different streams do not
// wait for each other's decide launches.  Worker pools are a quarter of the size; all are allocated on first use.
constexpr int kDecideSlots = 17;  // slot 0 + one per Grok worker stream (grok_device.hip kGrokMaxStreams)
thread_local DecidePool tlsDecidePools[kDecideSlots];
thread_local int tlsDecideSlot = 0;

std::atomic<int> gNfaDfsMode{-1};  // -1: LC_NFA_DFS decides; 0 / 1: set by lc_nfa_set_dfs

// Frame stacks of nfa_dfs_kernel (one line per lane): one pool per host thread and device, sized to the launch (about a
// frame per byte of text), grow-only, capped by LC_NFA_DFS_POOL_MB (default 4096).
struct DfsPool {
    uint8_t* p = nullptr;
    size_t bytes = 0;
    int device = -1;
    hipEvent_t lastUse = nullptr;
    hipStream_t lastStream = nullptr;
    bool used = false;
    ~DfsPool() { release(); }
    void release() {
        if (p && lcRuntimeUsable()) {
            (void)hipSetDevice(device);
            (void)hipDeviceSynchronize();
            (void)hipFree(p);
            if (lastUse) (void)hipEventDestroy(lastUse);
        }
        p = nullptr;
        bytes = 0;
        device = -1;
        lastUse = nullptr;
        used = false;
    }
};
thread_local DfsPool tlsDfsPool;

size_t decidePoolBytes() {
    static const size_t v = [] {
        const char* e = getenv("LC_DECIDE_POOL_MB");
        long mb = e ? atol(e) : 256;
        if (mb < 8) mb = 8;
        return size_t(mb) << 20;
    }();
    return tlsDecideSlot == 0 ? v : std::max<size_t>(v / 4, size_t(8) << 20);
}
}  // namespace
void lcSetDecideSlot(int slot) { tlsDecideSlot = slot >= 0 && slot < kDecideSlots ? slot : 0; }

// The thread-list kernels of this launch may have left lines LC_OVERFLOW: settle them (plan + walk, both no-ops unless the
// overflow flag carries this launch's sequence number).
static int launchDecide(lc_regex* re, int dev, const void* dBlob, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len,
                        uint32_t sep, uint32_t n, const uint32_t* d_n, const uint32_t* d_order, const uint32_t* d_resume,
                        uint32_t ngroups, int32_t* d_caps, uint8_t* d_status, hipStream_t stream, uint32_t* overflowFlag,
                        uint32_t seq, bool forced = false) {
    static const bool off = getenv("LC_NFA_NO_DECIDE") != nullptr;
    if (off && !forced) return LC_OK;
    DecidePool& pool = tlsDecidePools[tlsDecideSlot];
    lcRegisterExitHook();
    if (pool.device != dev || !pool.p) {
        pool.release();
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&pool.p), decidePoolBytes()));
        pool.bytes = decidePoolBytes();
        pool.device = dev;
        HIP_TRY(hipEventCreateWithFlags(&pool.lastUse, hipEventDisableTiming));
        HIP_TRY(hipMemsetAsync(pool.p, 0, kDecideHeaderBytes, stream));
    }
    if (pool.used && pool.lastStream != stream) HIP_TRY(hipStreamWaitEvent(stream, pool.lastUse, 0));
    const DecideShape shape{re->decideClosedCap, re->decideMaxEnter};
    const uint32_t nPos = uint32_t(re->nfa.positions.size());
    noteKernel("nfa_decide_kernel");
    hipLaunchKernelGGL(nfa_decide_plan_kernel, dim3(1), dim3(256), 0, stream, d_off, d_len, sep, n, d_n, d_order, d_resume, nPos, shape,
                       d_status, overflowFlag, seq, pool.p, uint64_t(pool.bytes));
    hipLaunchKernelGGL(nfa_decide_kernel, dim3(kDecideWorkers), dim3(64), 0, stream, d_data, d_off, d_len, sep, d_resume,
                       static_cast<const uint32_t*>(dBlob), shape, ngroups, d_caps, d_status, overflowFlag, seq, pool.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(pool.lastUse, stream));
    pool.lastStream = stream;
    pool.used = true;
    return LC_OK;
}

// which part of the NFA engine's chain a launch queues: all of it, the thread-list kernel only (lines that overflow keep LC_OVERFLOW),
// or the second chance only (nfa_wide_kernel + the depth-first decide kernels, for the lines that still say LC_OVERFLOW)
// round 5 -- kNfaWideFirst: nfa_wide_kernel over EVERY line as the first chance (lines that need more than 128 threads keep
// LC_OVERFLOW); kNfaDecideOnly: what is left behind a wide-first launch (the decide kernels alone); kNfaWideChain: both.  For
// callers that know the pattern overflows 64 threads on their data; programs the wide kernel cannot run take the usual kernels.
enum { kNfaWholeChain = 0, kNfaFirstChance = 1, kNfaSecondChance = 2, kNfaWideFirst = 3, kNfaDecideOnly = 4, kNfaWideChain = 5 };
static thread_local uint32_t* tlsWideNote = nullptr;  // lcSetWideNote: where the next wide launch reports "more than 64 threads were needed"
void lcSetWideNote(uint32_t* note) { tlsWideNote = note; }

template <int NS, bool ATOMIC, bool GLOBAL, int BLOCK = kNfaBlock>
static int launchNfaSlots(const void* dBlob, uint32_t blobBytes, uint32_t nPos, size_t lds, const uint8_t* d_data,
                          const uint32_t* d_off, const uint32_t* d_len, uint32_t sep, uint32_t n, const uint32_t* d_n, const uint32_t* d_order, const uint32_t* d_resume, uint32_t ngroups,
                          int32_t* d_caps, uint8_t* d_status, hipStream_t stream, uint32_t* overflowFlag, uint32_t seq,
                          const uint32_t* pendingFlag, int chance = 0, int wideLdsMode = 0) {
    static thread_local size_t ldsAttrSet[kLcMaxDevices] = {};  // the attribute belongs to (function, device)
    int devNow = 0;
    if (lds > 64 * 1024) HIP_TRY(hipGetDevice(&devNow));
    if (lds > 64 * 1024 && devNow < kLcMaxDevices && lds > ldsAttrSet[devNow]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&nfa_match_kernel<NS, ATOMIC, GLOBAL, BLOCK>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        ldsAttrSet[devNow] = lds;
    }
    constexpr uint32_t kWaves = BLOCK / 64;
    const uint32_t grid = (n + kWaves - 1) / kWaves;
    uint32_t* const wideNote = tlsWideNote;
    tlsWideNote = nullptr;
    if (chance == kNfaDecideOnly) return LC_OK;
    bool wideFirst = chance == kNfaWideFirst || chance == kNfaWideChain;
    if constexpr (ATOMIC || NS > 64) wideFirst = false;
    // behind a first engine that left lines pending (the lazy automaton, the depth-first walk) the narrow kernel takes exactly those
    // lines and the wide kernel is its second chance, as in the whole chain -- "wide first" walks every line whatever its status says
    const bool behindFirstEngine = pendingFlag != nullptr && wideFirst;
    if (behindFirstEngine) wideFirst = false;
    {
        static const bool wideOff = getenv("LC_NFA_NO_WIDE") != nullptr;
        const size_t wideLds0 = (size_t((nPos + 3) & ~3u) + 3 * kNfaWideThreads + 64) * 4;
        if (wideOff || !overflowFlag || wideLds0 > 64 * 1024) wideFirst = false;
    }
    if (chance != kNfaSecondChance && !wideFirst) {
        noteKernel(ATOMIC ? "nfa_match_kernel<atomic>" : "nfa_match_kernel");
        hipLaunchKernelGGL((nfa_match_kernel<NS, ATOMIC, GLOBAL, BLOCK>), dim3(grid), dim3(BLOCK), lds, stream, d_data, d_off, d_len, sep, n,
                           d_n, d_order, d_resume, static_cast<const uint32_t*>(dBlob), blobBytes, ngroups, d_caps, d_status, overflowFlag,
                           seq, pendingFlag);
        HIP_TRY(hipGetLastError());
    }
    if (chance == kNfaFirstChance || (chance == kNfaWideFirst && !wideFirst && !behindFirstEngine)) return LC_OK;
    // Second chance for the lines that needed more than 64 live threads (nfa_wide_kernel.hpp: two threads per lane), for
    // patterns without atomic groups whose capture offsets fit twice into a lane's registers.  Its workgroups return at
    // once unless the launch above raised the overflow flag.
    if constexpr (!ATOMIC && NS <= 64) {
        static const bool wideOff = getenv("LC_NFA_NO_WIDE") != nullptr;
        const size_t wideLds = (size_t((nPos + 3) & ~3u) + 3 * kNfaWideThreads + 64) * 4;
        if (!wideOff && overflowFlag && wideLds <= 64 * 1024) {
            noteKernel(wideFirst ? "nfa_wide_kernel:first" : "nfa_wide_kernel");
            // (wideLdsMode: the batch is small and program + scratch fit the CU's LDS -- the program is staged, launchNfa decides)
            if (wideLdsMode && wideLds + blobBytes <= kLcLdsPerCu) {
                static thread_local size_t wideAttrSet[kLcMaxDevices] = {};
                const size_t need = wideLds + blobBytes;
                if (need > 64 * 1024) {
                    HIP_TRY(hipGetDevice(&devNow));
                    if (devNow < kLcMaxDevices && need > wideAttrSet[devNow]) {
                        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&nfa_wide_kernel<NS, true>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, int(need)));
                        wideAttrSet[devNow] = need;
                    }
                }
                hipLaunchKernelGGL((nfa_wide_kernel<NS, true>), dim3(n), dim3(64), need, stream, d_data, d_off, d_len, sep, n, d_n, d_order,
                                   d_resume, static_cast<const uint32_t*>(dBlob), blobBytes, ngroups, d_caps, d_status, overflowFlag, seq,
                                   wideFirst ? 1u : 0u, wideNote);
            } else {
                hipLaunchKernelGGL((nfa_wide_kernel<NS, false>), dim3(n), dim3(64), wideLds, stream, d_data, d_off, d_len, sep, n, d_n, d_order,
                                   d_resume, static_cast<const uint32_t*>(dBlob), blobBytes, ngroups, d_caps, d_status, overflowFlag, seq,
                                   wideFirst ? 1u : 0u, wideNote);
            }
            HIP_TRY(hipGetLastError());
        }
    }
    return LC_OK;
}

static int launchNfa(lc_regex* re, int dev, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len,
                     uint32_t sep, uint32_t n, const uint32_t* d_n, const uint32_t* d_order, const uint32_t* d_resume, uint32_t ngroups, int32_t* d_caps,
                     uint8_t* d_status, hipStream_t stream, bool decideOnly = false, int chance = kNfaWholeChain, uint32_t* seqInOut = nullptr) {
    if (re->nfaBlob.empty()) {
        tlsError = "pattern has no NFA program";
        return LC_ERR_UNSUPPORTED;
    }
    if (chance == kNfaWholeChain) {  // LC_NFA_WIDE_FIRST=1: every whole chain starts with the wide kernel (parity tests; read per launch)
        const char* wf = getenv("LC_NFA_WIDE_FIRST");
        if (wf && wf[0] == '1') chance = kNfaWideChain;
    }
    void* dBlob = nullptr;
    int rc = ensureUploaded(re, dev, kBlobNfa, &dBlob);
    if (rc != LC_OK) return rc;
    const uint32_t blobBytes = uint32_t(re->nfaBlob.size() * 4);  // the whole upload: the overflow flag sits behind it
    // what a kernel that keeps the program in LDS stages: everything in front of the class lists (device_tables.h NF_STAGE_BYTES)
    const uint32_t stageBytes = re->nfaBlob[NF_STAGE_BYTES] ? re->nfaBlob[NF_STAGE_BYTES] : blobBytes;
    const bool atomic = re->nfa.atomicCount > 0;
    size_t lds = lcNfaLdsBytes(stageBytes, uint32_t(re->nfa.positions.size()), atomic);
    // Program too big for LDS -- or so big that fewer than three workgroups would fit per CU: the tables stay in HBM (L2) and
    // only the scratch is LDS; four or eight lines per CU with LDS-speed tables lose against a dozen with L2-speed tables
    // (configs[2], 256 Ki lines: CISCOFW106001, 89.6 KB with its tables, 32 -> 9.4 ms; CRONLOG 6.2 -> 3.2 ms; below 52 KB
    // nothing moves).  LC_NFA_GLOBAL_KB overrides the bound.
    static const size_t globalAbove = [] {
        const char* e = getenv("LC_NFA_GLOBAL_KB");
        return size_t(e ? atoi(e) : 52) * 1024;
    }();
    // (round 4) ... on LARGE batches.  A small batch -- a Grok entry's few hundred candidates, an event group -- does not fill the
    // chip either way and waits for its longest value, i.e. for the latency of a byte-step: with the program in LDS a step's
    // dependent table reads (follow list bounds, paths, masks) cost LDS latency instead of L2 latency.  LC_NFA_SMALL_BATCH: the
    // largest batch that stages whatever fits the CU's 160 KiB (0 = never; A/B measurements).
    static const uint32_t smallBatch = [] {
        const char* e = getenv("LC_NFA_SMALL_BATCH");
        return uint32_t(e ? atol(e) : 0);  // (measured on configs[2], round 4: 4.57 vs 4.84 ms at 1000 values, 8.16 vs 8.18 ms at 16 Ki -- staging
                                           // 100+ KiB per workgroup of four values eats what the faster steps save: off by default)
    }();
    // (round 5) What kept the r4 experiment from paying: the entries that ARE the long poles of a Grok step (CISCOFW313005 and its
    // kin: 3 200 positions, 127 KB of program) do not fit LDS next to FOUR waves' election marks (4 x 14 KB) and stayed in L2.  Two or
    // one value per workgroup do fit.  A batch that the chip takes in ONE round of such workgroups stages its program: block 256 if
    // that fits, else 128, else 64 lanes.  LC_NFA_STAGE_SMALL=0 switches it off (A/B measurements).
    // Measured (profiles/round5_nfa_staging.txt): alone on the chip, CISCOFW313005's launch gains 7 % from the staged program (2.75 ->
    // 2.56 ms) -- the step was not the table reads but the candidate-owner loop (nfa_kernel.hpp, fixed in the same round); inside a Grok
    // step, where sixteen such entries run side by side, a 155 KB workgroup per CU serialises them (16 Ki values: 5.3 -> 6.2 ms).
    // So: OFF by default, LC_NFA_STAGE_SMALL=1 switches it on (A/B measurements).
    static const bool stageSmall = [] {
        const char* e = getenv("LC_NFA_STAGE_SMALL");
        return e && e[0] == '1';
    }();
    int block = kNfaBlock;
    bool global = lds > kLcLdsPerCu || (lds > globalAbove && n > smallBatch);
    const uint32_t nPosAll = uint32_t(re->nfa.positions.size());
    if (global && stageSmall && !atomic && re->nfa.slotCount() <= 64) {
        for (int waves : {4, 2, 1}) {
            const size_t need = lcNfaLdsBytes(stageBytes, nPosAll, atomic, uint32_t(waves));
            if (need > kLcLdsPerCu) continue;
            const size_t perCu = kLcLdsPerCu / need;                   // workgroups a CU holds
            const size_t oneRound = size_t(256) * perCu * size_t(waves);  // values the chip walks at once
            if (n <= oneRound) {
                block = 64 * waves;
                lds = need;
                global = false;
            }
            break;  // (fewer values per workgroup only where more do not fit)
        }
    }
    const bool wideStage = stageSmall && n <= 4096;  // (the second chance walks one value per workgroup anyway)
    if (global) lds -= stageBytes;
    if (lds > 160 * 1024) {
        tlsError = "nfa tables exceed LDS";
        return LC_ERR_UNSUPPORTED;
    }
    const int slots = re->nfa.slotCount();
    const uint32_t nPos = uint32_t(re->nfa.positions.size());
    // one word behind the tables (ensureUploaded): raised by the kernel to this launch's sequence number when a line overflows
    uint32_t* overflowFlag = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(dBlob) + blobBytes);
    uint32_t seq;
    if (chance == kNfaSecondChance || chance == kNfaDecideOnly) {
        seq = seqInOut ? *seqInOut : 0;  // the first-chance launch's number: its overflow flag is what the kernels test
        if (!seq) return LC_OK;
    } else {
        seq = ++re->nfaSeq[dev];
        if (seq == 0) {  // the 32-bit sequence wrapped: start over below every flag value seen so far
            HIP_TRY(hipMemsetAsync(overflowFlag, 0, 8, stream));
            seq = ++re->nfaSeq[dev];
        }
        if (seqInOut) *seqInOut = seq;
    }
    // ---- optional first engine: the depth-first walk, one line per lane (nfa_dfs_kernel).  What it leaves pending (budget,
    // pool) is what the thread-list kernels below look at.  OFF by default: with its frames in HBM a walk step costs ~12 us
    // against ~2 us for a thread-list byte-step, and on the batch sizes measured (8-64 Ki lines, where every launch is bound by
    // its longest line) it loses 5-10x (profiles/round2_grok_dfs_vs_threadlist.txt).  LC_NFA_DFS=1 or lc_nfa_set_dfs(1).
    const uint32_t* pendingFlag = nullptr;
    // ---- the LAZY automaton first (round 6, regex_handle.hpp LcLazyTdfa): every value walks the partial tagged DFA -- a table read
    // per byte instead of a thread-list step -- and only the values that step on an uncomputed transition (LC_PENDING, the launch's
    // pending flag raised) are this launch's business for the kernels below.  Same protocol as the depth-first walk's.
    bool lazyFront = false;
    if (!decideOnly && chance != kNfaSecondChance && chance != kNfaDecideOnly && re->lazyReady.load(std::memory_order_acquire)) {
        const char* lazyEnv = getenv("LC_LAZY_TDFA");  // (read per launch: the parity tests run with and without in one process)
        if (!(lazyEnv && lazyEnv[0] == '0')) {
            void* dLazy = nullptr;
            std::vector<uint32_t> lazyHeader;
            rc = ensureLazyUploaded(re, dev, &dLazy, &lazyHeader);
            if (rc != LC_OK) return rc;
            if (dLazy) {
                uint32_t* pf = overflowFlag + 1;
                rc = launchTdfaL2Family(re, lazyHeader.data(), dLazy, false, dev, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, ngroups,
                                        d_caps, d_status, stream, pf, seq);
                if (rc != LC_OK) return rc;
                pendingFlag = pf;
                lazyFront = true;
            }
        }
    }
    static const bool dfsEnv = [] {
        const char* e = getenv("LC_NFA_DFS");
        return e && e[0] == '1';
    }();
    const int dfsMode = gNfaDfsMode.load(std::memory_order_relaxed);
    if (!decideOnly && !lazyFront && (dfsMode < 0 ? dfsEnv : dfsMode != 0)) {
        static const size_t poolCap = [] {
            const char* e = getenv("LC_NFA_DFS_POOL_MB");
            long mb = e ? atol(e) : 4096;
            if (mb < 32) mb = 32;
            return size_t(mb) << 20;
        }();
        static const uint32_t stepsPerByte = [] {
            const char* e = getenv("LC_NFA_DFS_STEPS_PER_BYTE");
            const long v = e ? atol(e) : 64;
            return uint32_t(v < 1 ? 1 : v);
        }();
        const DecideShape shape{re->decideClosedCap, re->decideMaxEnter};
        const size_t frameBytes = (size_t(kDecideFrameWords) + shape.closedCap + size_t(kDecideNodeWords) * shape.maxEnter) * 4;
        size_t want = size_t(n) * 1024 * frameBytes + (1u << 20);  // about a frame per byte, ~1 KiB lines
        if (want < (size_t(32) << 20)) want = size_t(32) << 20;
        if (want > poolCap) want = poolCap;
        DfsPool& pool = tlsDfsPool;
        lcRegisterExitHook();
        if (pool.device != dev || !pool.p || pool.bytes < want) {
            pool.release();
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&pool.p), want));
            pool.bytes = want;
            pool.device = dev;
            HIP_TRY(hipEventCreateWithFlags(&pool.lastUse, hipEventDisableTiming));
        }
        if (pool.used && pool.lastStream != stream) HIP_TRY(hipStreamWaitEvent(stream, pool.lastUse, 0));
        HIP_TRY(hipMemsetAsync(pool.p, 0, sizeof(DfsPoolHeader), stream));
        uint32_t* pf = overflowFlag + 1;
        noteKernel("nfa_dfs_kernel");
        hipLaunchKernelGGL(nfa_dfs_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume,
                           static_cast<const uint32_t*>(dBlob), shape, ngroups, d_caps, d_status, pf, seq, pool.p, uint64_t(pool.bytes),
                           stepsPerByte);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(pool.lastUse, stream));
        pool.lastStream = stream;
        pool.used = true;
        pendingFlag = pf;
    }
    if (decideOnly) {  // LC_ENGINE_DECIDE: every line goes to the depth-first walk
        hipLaunchKernelGGL(nfa_decide_mark_all_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, d_n, d_order, d_status,
                           overflowFlag, seq);
        HIP_TRY(hipGetLastError());
        return launchDecide(re, dev, dBlob, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, stream,
                            overflowFlag, seq, true);
    }
    // kernel instance by capture slots carried per thread (VGPRs), atomic groups, tables in LDS or read from HBM
    auto launch = [&](auto ns) {
        constexpr int NS = decltype(ns)::value;
        auto go = [&](auto a, auto g) {
            constexpr bool A = decltype(a)::value, G = decltype(g)::value;
            if constexpr (!G && !A && NS <= 64) {  // (staged programs without atomic groups only -- an opt-in experiment does not get 24 more instantiations)
                if (block == 128)
                    return launchNfaSlots<NS, A, G, 128>(dBlob, stageBytes, nPos, lds, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, ngroups,
                                                         d_caps, d_status, stream, overflowFlag, seq, pendingFlag, chance, wideStage);
                if (block == 64)
                    return launchNfaSlots<NS, A, G, 64>(dBlob, stageBytes, nPos, lds, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, ngroups,
                                                        d_caps, d_status, stream, overflowFlag, seq, pendingFlag, chance, wideStage);
            }
            return launchNfaSlots<NS, A, G>(dBlob, stageBytes, nPos, lds, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, ngroups, d_caps,
                                            d_status, stream, overflowFlag, seq, pendingFlag, chance, wideStage);
        };
        if (atomic && global) return go(std::true_type{}, std::true_type{});
        if (atomic) return go(std::true_type{}, std::false_type{});
        if (global) return go(std::false_type{}, std::true_type{});
        return go(std::false_type{}, std::false_type{});
    };
    if (slots <= 8) rc = launch(std::integral_constant<int, 8>{});
    else if (slots <= 16) rc = launch(std::integral_constant<int, 16>{});
    else if (slots <= 32) rc = launch(std::integral_constant<int, 32>{});
    else if (slots <= 64) rc = launch(std::integral_constant<int, 64>{});
    else if (slots <= 128) rc = launch(std::integral_constant<int, 128>{});  // 4 tag words per aux entry
    // up to 320 slots (160 groups): 10 tag words per aux entry; a thread's offsets no longer fit the 256 architected VGPRs
    // of a lane, the rest lives in accumulation registers
    else rc = launch(std::integral_constant<int, 320>{});
    if (rc != LC_OK) return rc;
    // Can a thread list overflow at all?  Without atomic groups a list holds at most one thread per position.
    const bool wideApplies = !atomic && slots <= 64 && (size_t((nPos + 3) & ~3u) + 3 * kNfaWideThreads + 64) * 4 <= 64 * 1024;
    const bool canOverflow = atomic || nPos > (wideApplies ? uint32_t(kNfaWideThreads) : 64u);
    if (!canOverflow || chance == kNfaFirstChance || chance == kNfaWideFirst) return LC_OK;
    return launchDecide(re, dev, dBlob, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, stream,
                        overflowFlag, seq);
}

// Groups written "(?=(S*))" (regex_ast.hpp Node::runCapture, Grok's "(?=%{GREEDYDATA:message})"): the automata stamp only
// where the group begins; it ends where the run of S bytes that starts there ends.  One lane per matched line walks that
// run -- a few of the line's bytes once more, through L2 right after the match kernel has read them.
struct RunSet {
    uint32_t w[8];  // 256-bit byte set
};
__global__ __launch_bounds__(256) void run_capture_kernel(const uint8_t* __restrict__ data, const uint32_t* __restrict__ off,
                                                          const uint32_t* __restrict__ len, uint32_t sepBytes, uint32_t nLines,
                                                          const uint32_t* __restrict__ nLinesPtr,
                                                          const uint32_t* __restrict__ order, uint32_t group, RunSet set,
                                                          uint32_t nGroupsOut, int32_t* __restrict__ caps,
                                                          const uint8_t* __restrict__ status) {
    if (nLinesPtr) {
        const uint32_t dyn = *nLinesPtr;
        nLines = dyn < nLines ? dyn : nLines;
    }
    const uint32_t slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= nLines) return;
    const uint32_t line = order ? order[slot] : slot;
    if (status[line] != LC_MATCH) return;
    int32_t* c = caps + size_t(line) * 2 * nGroupsOut + 2 * group;
    if (c[0] < 0) return;  // the group did not take part in the match
    const uint32_t o = off[line];
    const uint32_t L = len ? len[line] : off[line + 1] - o - sepBytes;
    uint32_t e = uint32_t(c[0]);
    // (16 bytes per load once the address is aligned: the run of a GREEDYDATA group is the rest of the value -- 4 KiB of one
    // dependent byte load per iteration was 0.8 ms of a Grok step)
    const uint8_t* base = data + size_t(o);
    bool stop = false;
    while (e < L && !stop && ((reinterpret_cast<uintptr_t>(base) + e) & 15u)) {
        const uint32_t b = base[e];
        if (!((set.w[b >> 5] >> (b & 31u)) & 1u)) stop = true;
        else ++e;
    }
    while (e + 16 <= L && !stop) {
        const uint4 v = *reinterpret_cast<const uint4*>(base + e);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t run = 16;  // bytes of this load that belong to the run
#pragma unroll
        for (int j = 15; j >= 0; --j) {
            const uint32_t b = (w[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
            if (!((set.w[b >> 5] >> (b & 31u)) & 1u)) run = uint32_t(j);
        }
        e += run;
        stop = run < 16;
    }
    while (e < L && !stop) {
        const uint32_t b = base[e];
        if (!((set.w[b >> 5] >> (b & 31u)) & 1u)) stop = true;
        else ++e;
    }
    c[1] = int32_t(e);
}

int lcEnsureScreenUploaded(lc_regex* re, int dev, const uint32_t** out) {
    void* p = nullptr;
    int rc = ensureUploaded(re, dev, kBlobScreen, &p);
    *out = static_cast<const uint32_t*>(p);
    return rc;
}

int lcScreenOnStream(lc_regex* re, int dev, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len, uint32_t n,
                     const uint32_t* d_in, uint32_t* d_out, uint32_t* d_counters, void* streamPtr) {
    if (re->screenBlob.empty()) {
        tlsError = "handle carries no screen table";
        return LC_ERR_UNSUPPORTED;
    }
    if (n == 0) return LC_OK;
    void* dBlob = nullptr;
    int rc = ensureUploaded(re, dev, kBlobScreen, &dBlob);
    if (rc != LC_OK) return rc;
    noteKernel("dfa_screen_kernel");
    hipLaunchKernelGGL(dfa_screen_kernel, dim3((n + kScreenBlock - 1) / kScreenBlock), dim3(kScreenBlock), 0,
                       static_cast<hipStream_t>(streamPtr), d_data, d_off, d_len, d_in, n, static_cast<const uint32_t*>(dBlob), d_out,
                       d_counters);
    HIP_TRY(hipGetLastError());
    return LC_OK;
}

extern "C" int lc_regex_screen_device(lc_regex_t* re, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len,
                                      uint32_t n, const uint32_t* d_lines, uint32_t* d_out, uint32_t* d_count, void* stream) {
    if (!re || !d_data || !d_off || !d_len || !d_out || !d_count) return LC_ERR_ARG;
    if (lc_device_count() <= 0) {
        tlsError = "no HIP device";
        return LC_ERR_NO_DEVICE;
    }
    int dev = 0;
    {
        const int rcDev = lcDeviceEntryDevice(d_data, &dev);
        if (rcDev != LC_OK) return rcDev;
    }
    return lcScreenOnStream(re, dev, d_data, d_off, d_len, n, d_lines, d_out, d_count, stream);
}

static int lcMatchChainOnStream(lc_regex* re, int engine, int dev, const uint8_t* d_data, const uint32_t* d_off,
                         const uint32_t* d_len, uint32_t sep, uint32_t n, const uint32_t* d_n, const uint32_t* d_order, const uint32_t* d_resume, uint32_t ngroups,
                         int32_t* d_caps, uint8_t* d_status, void* streamPtr, int chance, uint32_t* seqInOut);

// ---- several automata over their own values in ONE launch (runtime_internal.hpp; the Grok plan's round 0)
bool lcNfaWideApplies(const lc_regex* re);
bool lcWaveJobPrepare(lc_regex* re, int dev, uint32_t n, bool stagePrograms, TdfaWaveJob* job, uint32_t* ldsBytes, uint32_t* seqOut, int* rc) {
    *rc = LC_OK;
    *seqOut = 0;
    if (!re || n == 0 || !re->nfa.runGroups.empty()) return false;  // (run captures: a kernel of the handle's own behind the match)
    {
        const char* waveEnv = getenv("LC_TDFA_WAVE_MAX");
        if (n > uint32_t(waveEnv ? atol(waveEnv) : 65536)) return false;
        const char* fusedEnv = getenv("LC_GROK_FUSED_ROUND0");  // (read per batch: the parity tests run both forms in one process)
        if (fusedEnv && fusedEnv[0] == '0') return false;
    }
    const uint32_t* hostHdr = nullptr;
    std::vector<uint32_t> lazyHeader;
    void* dBlob = nullptr;
    job->missFlag = nullptr;
    job->seq = 0;
    job->missStatus = 4;
    if (re->engine == LC_ENGINE_TDFA) {
        if (re->tdfaL2Blob.empty() || !(re->preferWave || !re->hasTdfa)) return false;
        *rc = ensureUploaded(re, dev, kBlobTdfaL2, &dBlob);
        if (*rc != LC_OK) return false;
        hostHdr = re->tdfaL2Blob.data();
    } else if (re->engine == LC_ENGINE_NFA) {
        const char* lazyEnv = getenv("LC_LAZY_TDFA");
        if ((lazyEnv && lazyEnv[0] == '0') || !re->lazyReady.load(std::memory_order_acquire) || !lcNfaWideApplies(re)) return false;
        // the thread-list program too: its overflow word is what a miss raises, and the second chance reads its tables
        void* dNfa = nullptr;
        *rc = ensureUploaded(re, dev, kBlobNfa, &dNfa);
        if (*rc != LC_OK) return false;
        *rc = ensureLazyUploaded(re, dev, &dBlob, &lazyHeader);
        if (*rc != LC_OK || !dBlob) return false;
        hostHdr = lazyHeader.data();
        uint32_t* overflowFlag = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(dNfa) + re->nfaBlob.size() * 4);
        uint32_t seq = ++re->nfaSeq[dev];
        if (seq == 0) return false;  // (the sequence wrapped: this batch goes the usual way, which resets the words)
        job->missFlag = overflowFlag;
        job->seq = seq;
        job->missStatus = LC_OVERFLOW;
        *seqOut = seq;
    } else {
        return false;
    }
    const uint32_t nRegs = hostHdr[TL_NREGS];
    uint32_t lds = nRegs * kTdfaWaveValues * 4, stage = 0;
    static const bool stageOffAll = getenv("LC_TDFA_L2_NO_STAGE") != nullptr;
    const uint32_t progBytes = (hostHdr[TL_OFF_FINALID] - hostHdr[TL_OFF_OPSSTART] + 3u) & ~3u;
    if (!stageOffAll && stagePrograms && n <= 32768 && progBytes <= 40 * 1024 && lds + progBytes <= 60 * 1024) stage = progBytes;
    if (lds + stage > 60 * 1024) return false;
    job->blob = static_cast<const uint32_t*>(dBlob);
    job->stageBytes = stage;
    *ldsBytes = lds + stage;
    return true;
}

size_t lcWaveJobTableBytes() { return kTdfaWaveMaxJobs * 4 + kTdfaWaveMaxJobs * sizeof(TdfaWaveJob); }

int lcLaunchWaveJobs(const uint8_t* d_data, const TdfaWaveJob* jobs, uint32_t nJobs, uint32_t totalBlocks, uint32_t ldsBytes, void* hTable,
                     void* dTable, int dev, hipStream_t st) {
    if (!nJobs || !totalBlocks) return LC_OK;
    if (nJobs > kTdfaWaveMaxJobs) {
        tlsError = "wave jobs: more than 64 in one launch";
        return LC_ERR_ARG;
    }
    uint32_t* first = static_cast<uint32_t*>(hTable);
    for (uint32_t j = 0; j < kTdfaWaveMaxJobs; ++j) first[j] = j < nJobs ? jobs[j].firstBlock : 0xFFFFFFFFu;
    std::memcpy(first + kTdfaWaveMaxJobs, jobs, size_t(nJobs) * sizeof(TdfaWaveJob));
    HIP_TRY(hipMemcpyAsync(dTable, hTable, kTdfaWaveMaxJobs * 4 + size_t(nJobs) * sizeof(TdfaWaveJob), hipMemcpyHostToDevice, st));
    // (LC_GROK_FUSED_LDS=<bytes>: at least that much LDS per workgroup -- fewer workgroups, so fewer wavefronts, per CU: the walk is scalar
    // code, and a SIMD issues one scalar instruction every four cycles for ALL its wavefronts; A/B measurements)
    if (const char* padEnv = getenv("LC_GROK_FUSED_LDS")) ldsBytes = std::max<uint32_t>(ldsBytes, uint32_t(atol(padEnv)));
    static thread_local size_t ldsAttrSet[kLcMaxDevices] = {};
    if (ldsBytes > 48 * 1024 && dev < kLcMaxDevices && ldsBytes > ldsAttrSet[dev]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tdfa_wave_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes)));
        ldsAttrSet[dev] = ldsBytes;
    }
    noteKernel("tdfa_wave_multi_kernel");
    hipLaunchKernelGGL(tdfa_wave_multi_kernel, dim3(totalBlocks), dim3(kTdfaWaveBlock), ldsBytes, st, d_data, static_cast<const uint32_t*>(dTable), nJobs);
    HIP_TRY(hipGetLastError());
    return LC_OK;
}

int lcMatchOnStream(lc_regex* re, int engine, int dev, const uint8_t* d_data, const uint32_t* d_off,
                         const uint32_t* d_len, uint32_t sep, uint32_t n, const uint32_t* d_n, const uint32_t* d_order, const uint32_t* d_resume, uint32_t ngroups,
                         int32_t* d_caps, uint8_t* d_status, void* streamPtr) {
    return lcMatchChainOnStream(re, engine, dev, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, streamPtr,
                                kNfaWholeChain, nullptr);
}
// The engine's main kernel only; *seq = what lcMatchSecondChanceOnStream needs to finish the lines that came back LC_OVERFLOW
// (0: this engine has no second chance -- a DFA decides every line).  The Grok matcher queues the second chance only for the
// entries whose first chance reported overflows (grok_device.hip).
int lcMatchFirstOnStream(lc_regex* re, int engine, int dev, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len,
                         uint32_t sep, uint32_t n, const uint32_t* d_n, const uint32_t* d_order, const uint32_t* d_resume, uint32_t ngroups,
                         int32_t* d_caps, uint8_t* d_status, uint32_t* seq, void* streamPtr) {
    *seq = 0;
    return lcMatchChainOnStream(re, engine, dev, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, streamPtr,
                                kNfaFirstChance, seq);
}
int lcMatchSecondChanceOnStream(lc_regex* re, int engine, int dev, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len,
                                uint32_t sep, uint32_t n, const uint32_t* d_n, const uint32_t* d_order, const uint32_t* d_resume,
                                uint32_t ngroups, int32_t* d_caps, uint8_t* d_status, uint32_t seq, void* streamPtr) {
    if (!seq || engine != LC_ENGINE_NFA) return LC_OK;
    return lcMatchChainOnStream(re, engine, dev, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, streamPtr,
                                kNfaSecondChance, &seq);
}

int lcMatchWideFirstOnStream(int part, lc_regex* re, int engine, int dev, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len,
                             uint32_t sep, uint32_t n, const uint32_t* d_n, const uint32_t* d_order, const uint32_t* d_resume,
                             uint32_t ngroups, int32_t* d_caps, uint8_t* d_status, uint32_t* seq, uint32_t* wideNote, void* streamPtr) {
    if (part == 1) {
        if (!seq || !*seq || engine != LC_ENGINE_NFA) return LC_OK;
        return lcMatchChainOnStream(re, engine, dev, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, streamPtr,
                                    kNfaDecideOnly, seq);
    }
    if (seq) *seq = 0;
    if (engine == LC_ENGINE_NFA) lcSetWideNote(wideNote);
    const int rc = lcMatchChainOnStream(re, engine, dev, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, ngroups, d_caps, d_status,
                                        streamPtr, part == 0 ? kNfaWideFirst : kNfaWideChain, seq);
    lcSetWideNote(nullptr);
    return rc;
}
bool lcNfaWideApplies(const lc_regex* re) {
    static const bool wideOff = getenv("LC_NFA_NO_WIDE") != nullptr;
    if (!re || wideOff || re->nfaBlob.empty() || re->nfa.atomicCount > 0 || re->nfa.slotCount() > 64) return false;
    const uint32_t nPos = uint32_t(re->nfa.positions.size());
    return (size_t((nPos + 3) & ~3u) + 3 * kNfaWideThreads + 64) * 4 <= 64 * 1024;
}

// The backtracking engine (bt_kernel.hpp): patterns with back-references.  The scratch pool of the launch -- 64 KB per lane in flight
// -- is allocated and freed in stream order, so concurrent callers of one handle never share a stack.
// Scratch of the SMALL launches (an event group's thousand lines: a few MB), kept per calling thread and stream: launches on one
// stream are ordered, so the block is never shared by two kernels in flight.  (Stream-ordered allocation from a pool -- the large
// launches' way, below -- hands a block just freed on another stream to the next caller WITH a dependency on that stream: sixteen
// runner threads with a group each ran one after the other, 2.3 ms per group where one thread took 0.14.)
namespace {
struct BtScratchCache {
    struct Entry {
        int dev;
        hipStream_t stream;
        uint32_t* p;
        size_t words;
    };
    std::vector<Entry> entries;
    ~BtScratchCache() {
        if (!lcRuntimeUsable()) return;
        for (const Entry& e : entries)
            if (hipSetDevice(e.dev) == hipSuccess) (void)hipFree(e.p);
    }
    uint32_t* get(int dev, hipStream_t stream, size_t words) {
        for (Entry& e : entries)
            if (e.dev == dev && e.stream == stream) {
                if (e.words >= words) return e.p;
                (void)hipFree(e.p);  // (waits for what is in flight)
                e.p = nullptr;
                e.words = 0;
                if (hipMalloc(reinterpret_cast<void**>(&e.p), words * 4) != hipSuccess) return nullptr;
                e.words = words;
                return e.p;
            }
        if (entries.size() >= 8) {  // (a thread that walks through many streams: the oldest block goes)
            (void)hipFree(entries.front().p);
            entries.erase(entries.begin());
        }
        uint32_t* p = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&p), words * 4) != hipSuccess) return nullptr;
        entries.push_back({dev, stream, p, words});
        return p;
    }
};
thread_local BtScratchCache tlsBtScratch;
constexpr size_t kBtCachedWordsMax = size_t(64) << 18;  // 64 MB: above it the pool
}  // namespace

static int launchBt(lc_regex* re, int dev, const uint8_t* d_data, const uint32_t* d_off, const uint32_t* d_len, uint32_t sep, uint32_t n,
                    const uint32_t* d_n, const uint32_t* d_order, const uint32_t* d_resume, uint32_t ngroups, int32_t* d_caps, uint8_t* d_status,
                    hipStream_t stream) {
    if (re->btBlob.empty()) {
        tlsError = "pattern has no backtracking program";
        return LC_ERR_UNSUPPORTED;
    }
    void* dBlob = nullptr;
    const int rc = ensureUploaded(re, dev, kBlobBt, &dBlob);
    if (rc != LC_OK) return rc;
    const uint32_t blobWords = uint32_t(re->btBlob.size());
    const uint32_t stageWords = blobWords * 4 <= kBtStageMaxBytes ? blobWords : 0u;
    uint32_t maxLanes = kBtMaxLanes, firstSlice = kBtSliceWords;
    if (const char* e = getenv("LC_BT_LANES")) maxLanes = std::max<uint32_t>(kBtBlock, uint32_t(strtoul(e, nullptr, 10)) / kBtBlock * kBtBlock);  // (A/B measurements)
    if (const char* e = getenv("LC_BT_SLICE_WORDS")) firstSlice = std::max<uint32_t>(256u, uint32_t(strtoul(e, nullptr, 10)));
    const uint32_t blocks = std::min<uint32_t>((n + kBtBlock - 1) / kBtBlock, maxLanes / kBtBlock);
    // (pass 2 takes the few values that filled their slice: an eighth of pass 1's lanes with eight times the slice -- the same pool)
    const uint32_t retryBlocks = std::max<uint32_t>(1u, std::min<uint32_t>(blocks / 8u, kBtRetryLanes / kBtBlock));
    uint32_t budget = kBtDefaultBudget;
    if (const char* e = getenv("LC_BT_BUDGET")) budget = uint32_t(strtoul(e, nullptr, 10));  // (read per launch: tests)
    const uint32_t need = re->btBlob[BT_NCAPS] + re->btBlob[BT_NLOOP] + 64u;
    uint32_t sliceWords = firstSlice;
    if (need + 128u > sliceWords) sliceWords = kBtRetrySliceWords;  // (hundreds of groups: every lane gets the large slice; fewer lanes)
    const uint32_t firstBlocks = sliceWords != kBtRetrySliceWords ? blocks : retryBlocks;
    if (need > kBtRetrySliceWords) {
        tlsError = "backtracking program: captures and loop registers exceed a lane's scratch";
        return LC_ERR_UNSUPPORTED;
    }
    // The scratch comes from a memory pool of the library's own, which KEEPS what launches give back (release threshold: none): with
    // the runtime's default pool every synchronisation trimmed the pool -- a device-wide free -- and sixteen runner threads with a
    // 1000-line group each took 2.3 ms per group where one took 0.14 (profiles/round6_bt_engine.txt).
    static hipMemPool_t btPool[kLcMaxDevices] = {};
    static std::once_flag btPoolOnce[kLcMaxDevices];
    if (dev >= 0 && dev < kLcMaxDevices)
        std::call_once(btPoolOnce[dev], [&] {
            hipMemPoolProps props = {};
            props.allocType = hipMemAllocationTypePinned;
            props.handleTypes = hipMemHandleTypeNone;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = dev;
            hipMemPool_t pool = nullptr;
            if (hipMemPoolCreate(&pool, &props) == hipSuccess) {
                uint64_t keep = ~uint64_t(0);
                (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
                btPool[dev] = pool;
            } else {
                (void)hipGetLastError();  // (no pool of our own: the runtime's default one)
            }
        });
    hipMemPool_t pool = dev >= 0 && dev < kLcMaxDevices ? btPool[dev] : nullptr;
    uint32_t* scratch = nullptr;
    const size_t poolWords = kBtPoolHeaderWords + std::max(size_t(firstBlocks) * kBtBlock * sliceWords, size_t(retryBlocks) * kBtBlock * kBtRetrySliceWords);
    const bool cached = poolWords <= kBtCachedWordsMax;
    if (cached) {
        scratch = tlsBtScratch.get(dev, stream, poolWords);
        if (!scratch) return hipFail(hipErrorOutOfMemory, "hipMalloc(backtracking scratch)");
    } else if (pool) {
        HIP_TRY(hipMallocFromPoolAsync(reinterpret_cast<void**>(&scratch), poolWords * 4, pool, stream));
    } else {
        HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&scratch), poolWords * 4, stream));
    }
    hipError_t launched = hipMemsetAsync(scratch, 0, kBtPoolHeaderWords * 4, stream);
    lcNoteKernel("bt_match_kernel");
    if (launched == hipSuccess) {
        hipLaunchKernelGGL(bt_match_kernel, dim3(firstBlocks), dim3(kBtBlock), stageWords * 4, stream, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume,
                           static_cast<const uint32_t*>(dBlob), blobWords, stageWords, ngroups, d_caps, d_status, scratch, sliceWords, budget, 0u);
        launched = hipGetLastError();
    }
    if (launched == hipSuccess && sliceWords != kBtRetrySliceWords) {  // pass 2: the values that filled their slice (none: the kernel returns at once)
        hipLaunchKernelGGL(bt_match_kernel, dim3(retryBlocks), dim3(kBtBlock), stageWords * 4, stream, d_data, d_off, d_len, sep, n, d_n, d_order,
                           d_resume, static_cast<const uint32_t*>(dBlob), blobWords, stageWords, ngroups, d_caps, d_status, scratch,
                           kBtRetrySliceWords, budget, 1u);
        launched = hipGetLastError();
    }
    if (!cached) (void)hipFreeAsync(scratch, stream);
    HIP_TRY(launched);
    return LC_OK;
}

static int lcMatchChainOnStream(lc_regex* re, int engine, int dev, const uint8_t* d_data, const uint32_t* d_off,
                         const uint32_t* d_len, uint32_t sep, uint32_t n, const uint32_t* d_n, const uint32_t* d_order, const uint32_t* d_resume, uint32_t ngroups,
                         int32_t* d_caps, uint8_t* d_status, void* streamPtr, int chance, uint32_t* seqInOut) {
    hipStream_t stream = static_cast<hipStream_t>(streamPtr);
    int rc;
    if (engine == LC_ENGINE_BT || re->engine == LC_ENGINE_BT) {
        if (engine != LC_ENGINE_BT) {
            tlsError = "pattern runs on the backtracking engine only (back-references)";
            return LC_ERR_UNSUPPORTED;
        }
        return launchBt(re, dev, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, stream);
    }
    if (!re->nfa.runGroups.empty()) tlsDone.armed = false;  // run_capture_kernel runs behind the match: it cannot signal
    // (a handle that asked for it -- lcPreferWaveTdfa: the Grok matcher's entries -- takes the wave-per-value kernel on small batches
    // even though its automaton fits the LDS kernels: those walk one value per lane, and a batch of a few hundred 4 KiB values waits
    // 0.3-1.4 ms for the longest of them)
    const bool waveByChoice = engine == LC_ENGINE_TDFA && re->hasTdfa && re->preferWave && !re->tdfaL2Blob.empty() && n <= 16384;
    if (engine == LC_ENGINE_TDFA && (waveByChoice || !re->hasTdfa) && !re->tdfaL2Blob.empty()) {
        // the automaton is too large for the LDS kernels: tables in global memory, one line per lane (tdfa_l2_kernel.hpp)
        void* dBlob = nullptr;
        rc = ensureUploaded(re, dev, kBlobTdfaL2, &dBlob);
        if (rc != LC_OK) return rc;
        rc = launchTdfaL2Family(re, re->tdfaL2Blob.data(), dBlob, waveByChoice, dev, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, ngroups,
                                d_caps, d_status, stream, nullptr, 0u);
    } else if (engine == LC_ENGINE_TDFA) {
        if (!re->hasTdfa) {
            tlsError = "pattern has no TDFA: " + re->tdfaError;
            return LC_ERR_UNSUPPORTED;
        }
        rc = launchTdfa(re, dev, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, stream);
    } else {
        rc = launchNfa(re, dev, d_data, d_off, d_len, sep, n, d_n, d_order, d_resume, ngroups, d_caps, d_status, stream,
                       engine == LC_ENGINE_DECIDE, engine == LC_ENGINE_DECIDE ? kNfaWholeChain : chance, seqInOut);
    }
    if (rc != LC_OK) return rc;
    for (const auto& rg : re->nfa.runGroups) {
        if (uint32_t(rg.first) >= ngroups) continue;  // the caller did not ask for this group
        RunSet set;
        for (int k = 0; k < 8; ++k) set.w[k] = uint32_t(rg.second.w[size_t(k) / 2] >> (32 * (k & 1)));
        hipLaunchKernelGGL(run_capture_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_data, d_off, d_len, sep, n, d_n,
                           d_order, uint32_t(rg.first), set, ngroups, d_caps, d_status);
        HIP_TRY(hipGetLastError());
    }
    return LC_OK;
}

extern "C" int lc_regex_match_device_engine(lc_regex_t* re, int engine, const uint8_t* d_data, const uint32_t* d_off,
                                            const uint32_t* d_len, uint32_t sep_bytes, uint32_t n, uint32_t ngroups,
                                            int32_t* d_caps, uint8_t* d_status, void* stream) {
    if (!re) return LC_ERR_ARG;
    if (n == 0) return LC_OK;
    if (!d_data || !d_off || !d_caps || !d_status) return LC_ERR_ARG;
    if (lc_device_count() <= 0) {
        tlsError = "no HIP device";
        return LC_ERR_NO_DEVICE;
    }
    int dev = 0;
    {
        const int rcDev = lcDeviceEntryDevice(d_data, &dev);
        if (rcDev != LC_OK) return rcDev;
    }
    if (engine == LC_ENGINE_AUTO) engine = re->engine;
    return lcMatchOnStream(re, engine, dev, d_data, d_off, d_len, sep_bytes, n, nullptr, nullptr, nullptr, ngroups, d_caps, d_status,
                         static_cast<hipStream_t>(stream));
}

// ---- lc_regex_match_device_multi: a job table (TdfaJob[] + u16 blockToJob[]) is packed in pinned host memory and copied to its
// device mirror on the launch stream; the kernel reads the device copy only.  A small ring of them per thread, each guarded by an
// event, so that a caller may queue several multi-launches before it synchronises
namespace {
struct JobTableRing {
    static constexpr int kTables = 8;
    struct Table {
        uint8_t* host = nullptr;  // pinned
        uint8_t* dev = nullptr;   // device mirror
        size_t cap = 0;           // bytes
        size_t bytes = 0;         // bytes of the table the device mirror holds (0: none)
        hipStream_t stream = nullptr;  // the stream of its last upload / launch
        hipEvent_t done = nullptr;
        bool inFlight = false;
    } t[kTables];
    int next = 0, device = -1;
    ~JobTableRing() { release(); }
    void release() {
        if (device < 0 || !lcRuntimeUsable()) {
            device = -1;
            return;
        }
        (void)hipSetDevice(device);
        for (auto& x : t) {
            if (x.inFlight) (void)hipEventSynchronize(x.done);
            if (x.done) (void)hipEventDestroy(x.done);
            (void)hipHostFree(x.host);
            (void)hipFree(x.dev);
            x = Table();
        }
        device = -1;
    }
};
thread_local JobTableRing tlsJobTables;

template <int BLOCK, bool PAIR1>
int launchTdfaMulti(const TdfaJob* table, const uint16_t* blockToJob, uint32_t totalBlocks, size_t lds, hipStream_t stream) {
    auto kern = tdfa_stream_multi_kernel<BLOCK, false, PAIR1>;
    static thread_local size_t ldsAttrSet[kLcMaxDevices] = {};
    int devNow = 0;
    if (lds > 64 * 1024) HIP_TRY(hipGetDevice(&devNow));
    if (lds > 64 * 1024 && devNow < kLcMaxDevices && lds > ldsAttrSet[devNow]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        ldsAttrSet[devNow] = lds;
    }
    noteKernel(PAIR1 ? "tdfa_stream_multi_kernel<pair1>" : "tdfa_stream_multi_kernel");
    uint32_t* nullCounter = nullptr;
    uint32_t* nullFlag = nullptr;
    uint32_t zero = 0;
    void* args[] = {&table, &blockToJob, &nullCounter, &nullFlag, &zero};
    HIP_TRY(hipLaunchKernel(reinterpret_cast<const void*>(kern), dim3(totalBlocks), dim3(BLOCK), args, lds, stream));
    return LC_OK;
}
}  // namespace

extern "C" int lc_regex_match_device_multi(const lc_match_job* jobs, uint32_t njobs, void* streamPtr) {
    if (!jobs && njobs) return LC_ERR_ARG;
    if (njobs == 0) return LC_OK;
    if (lc_device_count() <= 0) {
        tlsError = "no HIP device";
        return LC_ERR_NO_DEVICE;
    }
    int dev = 0;
    {
        const int rcDev = lcDeviceEntryDevice(jobs[0].d_data, &dev);
        if (rcDev != LC_OK) return rcDev;
    }
    hipStream_t stream = static_cast<hipStream_t>(streamPtr);
    for (uint32_t i = 0; i < njobs; ++i) {
        const lc_match_job& j = jobs[i];
        if (!j.re) return LC_ERR_ARG;
        if (j.n && (!j.d_data || !j.d_off || !j.d_caps || !j.d_status)) return LC_ERR_ARG;
    }
    lcRegisterExitHook();
    JobTableRing& ring = tlsJobTables;
    if (ring.device != dev) {
        ring.release();
        ring.device = dev;
    }
    // one packed launch per workgroup size the tables were packed for (almost always one: 256)
    for (int variant = 0; variant < 6; ++variant) {
        const int block = variant < 2 ? 256 : variant < 4 ? 128 : 64;
        const bool pairOne = (variant & 1) != 0;  // jobs whose standard tables carry a one-stamp pair table go together
        std::vector<TdfaJob> packed;
        uint32_t blocks = 0;
        size_t lds = 0;
        for (uint32_t i = 0; i < njobs; ++i) {
            const lc_match_job& j = jobs[i];
            lc_regex* re = j.re;
            if (j.n == 0 || re->engine != LC_ENGINE_TDFA || !re->hasTdfa || re->tdfaBlock != block || !re->nfa.runGroups.empty()) continue;
            static const bool pairOff = getenv("LC_TDFA_NO_PAIR") != nullptr;
            if ((lcPairOneFormat(re->tdfaBlob) && !pairOff) != pairOne) continue;
            void* dBlob = nullptr;
            const int rc = ensureUploaded(re, dev, kBlobTdfa, &dBlob);
            if (rc != LC_OK) return rc;
            const uint32_t blobBytes = uint32_t(re->tdfaBlob.size() * 4);
            TdfaJob t;
            t.data = j.d_data;
            t.off = j.d_off;
            t.len = j.d_len;
            t.blob = static_cast<const uint32_t*>(dBlob);
            t.caps = j.d_caps;
            t.status = j.d_status;
            t.sepBytes = j.sep_bytes;
            t.nLines = j.n;
            t.blobBytes = blobBytes;
            t.regBytes = uint32_t(lcTdfaRegBytes(re->tdfaPackedRegs, block));
            t.nGroupsOut = j.ngroups;
            t.firstBlock = blocks;
            blocks += (j.n + uint32_t(block) - 1) / uint32_t(block);
            const size_t need = lcTdfaLdsBytes(blobBytes, re->tdfaPackedRegs, block);
            lds = need > lds ? need : lds;
            packed.push_back(t);
        }
        if (packed.empty()) continue;
        if (packed.size() > 0xFFFFu) {
            tlsError = "lc_regex_match_device_multi: more than 65535 jobs in one call";
            return LC_ERR_ARG;
        }
        // the table as the kernel reads it: TdfaJob[] | u16 blockToJob[]
        const size_t mapAt = (packed.size() * sizeof(TdfaJob) + 15) & ~size_t(15), tableBytes = mapAt + size_t(blocks) * 2;
        static thread_local std::vector<uint8_t> image;
        image.assign(tableBytes, 0);
        std::memcpy(image.data(), packed.data(), packed.size() * sizeof(TdfaJob));
        {
            uint16_t* map = reinterpret_cast<uint16_t*>(image.data() + mapAt);
            for (size_t k = 0; k < packed.size(); ++k) {
                const uint32_t end = k + 1 < packed.size() ? packed[k + 1].firstBlock : blocks;
                for (uint32_t b = packed[k].firstBlock; b < end; ++b) map[b] = uint16_t(k);
            }
        }
        // A caller hands over the same buffers turn after turn (staging is reused), so the table of this turn is usually one the device
        // already holds: a slot whose image is byte-identical is used again without a copy -- if its upload is ordered before this
        // launch (same stream) or known to be complete.  The copy was a command of its own in front of every packed launch (~5 us
        // of a 48 us turn).
        JobTableRing::Table* hit = nullptr;
        for (auto& x : ring.t) {
            if (!x.dev || x.bytes != tableBytes || std::memcmp(x.host, image.data(), tableBytes) != 0) continue;
            if (x.inFlight && x.stream != stream) {
                if (hipEventQuery(x.done) != hipSuccess) {
                    (void)hipGetLastError();
                    continue;
                }
                x.inFlight = false;
            }
            hit = &x;
            break;
        }
        if (!hit) {
            hit = &ring.t[ring.next];
            ring.next = (ring.next + 1) % JobTableRing::kTables;
        }
        JobTableRing::Table& tab = *hit;
        if (!tab.done) HIP_TRY(hipEventCreateWithFlags(&tab.done, hipEventDisableTiming));
        const bool reuse = tab.dev && tab.bytes == tableBytes && std::memcmp(tab.host, image.data(), tableBytes) == 0;
        const bool inPlace = tlsJobTableInPlace && blocks <= 64;
        if (!reuse) {
            if (tab.inFlight) {
                HIP_TRY(hipEventSynchronize(tab.done));
                tab.inFlight = false;
            }
            tab.bytes = 0;
            if (tableBytes > tab.cap) {
                (void)hipHostFree(tab.host);
                (void)hipFree(tab.dev);
                tab.host = tab.dev = nullptr;
                tab.cap = 0;
                const size_t cap = tableBytes * 2 + 4096;
                HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&tab.host), cap, hipHostMallocDefault));
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&tab.dev), cap));
                tab.cap = cap;
            }
            std::memcpy(tab.host, image.data(), tableBytes);
            if (!inPlace) {
                HIP_TRY(hipMemcpyAsync(tab.dev, tab.host, tableBytes, hipMemcpyHostToDevice, stream));
                tab.bytes = tableBytes;
            }
        }
        tab.stream = stream;
        // (in place: the pinned image itself -- valid for this launch only, so it is never taken for a device copy later: bytes = 0)
        const uint8_t* tableAt = (inPlace && !reuse) ? tab.host : tab.dev;
        const TdfaJob* dJobs = reinterpret_cast<const TdfaJob*>(tableAt);
        const uint16_t* dMap = reinterpret_cast<const uint16_t*>(tableAt + mapAt);
        int rc = LC_OK;
        switch (variant) {
            case 0: rc = launchTdfaMulti<256, false>(dJobs, dMap, blocks, lds, stream); break;
            case 1: rc = launchTdfaMulti<256, true>(dJobs, dMap, blocks, lds, stream); break;
            case 2: rc = launchTdfaMulti<128, false>(dJobs, dMap, blocks, lds, stream); break;
            case 3: rc = launchTdfaMulti<128, true>(dJobs, dMap, blocks, lds, stream); break;
            case 4: rc = launchTdfaMulti<64, false>(dJobs, dMap, blocks, lds, stream); break;
            default: rc = launchTdfaMulti<64, true>(dJobs, dMap, blocks, lds, stream); break;
        }
        if (rc != LC_OK) return rc;
        HIP_TRY(hipEventRecord(tab.done, stream));
        tab.inFlight = true;
    }
    // everything else: one launch sequence per job, as lc_regex_match_device would do
    for (uint32_t i = 0; i < njobs; ++i) {
        const lc_match_job& j = jobs[i];
        lc_regex* re = j.re;
        if (j.n == 0) continue;
        if (re->engine == LC_ENGINE_TDFA && re->hasTdfa && re->nfa.runGroups.empty() &&
            (re->tdfaBlock == 256 || re->tdfaBlock == 128 || re->tdfaBlock == 64))
            continue;  // went with a packed launch
        const int rc = lcMatchOnStream(re, re->engine, dev, j.d_data, j.d_off, j.d_len, j.sep_bytes, j.n, nullptr, nullptr, nullptr,
                                       j.ngroups, j.d_caps, j.d_status, stream);
        if (rc != LC_OK) return rc;
    }
    return LC_OK;
}

extern "C" int lc_regex_match_device_from(lc_regex_t* re, int engine, const uint8_t* d_data, const uint32_t* d_off,
                                          const uint32_t* d_len, uint32_t sep_bytes, uint32_t n, const uint32_t* d_lines,
                                          const uint32_t* d_nlines, const uint32_t* d_from, uint32_t ngroups,
                                          int32_t* d_caps, uint8_t* d_status, void* stream) {
    if (!re) return LC_ERR_ARG;
    if (n == 0) return LC_OK;
    if (!d_data || !d_off || !d_caps || !d_status) return LC_ERR_ARG;
    if (d_from && re->nfa.searchPrefix < 0) {
        tlsError = "resume offsets need a pattern compiled with LC_SYNTAX_SEARCH";
        return LC_ERR_ARG;
    }
    if (lc_device_count() <= 0) {
        tlsError = "no HIP device";
        return LC_ERR_NO_DEVICE;
    }
    int dev = 0;
    {
        const int rcDev = lcDeviceEntryDevice(d_data, &dev);
        if (rcDev != LC_OK) return rcDev;
    }
    if (engine == LC_ENGINE_AUTO) engine = re->engine;
    return lcMatchOnStream(re, engine, dev, d_data, d_off, d_len, sep_bytes, n, d_nlines, d_lines, d_from, ngroups, d_caps,
                           d_status, static_cast<hipStream_t>(stream));
}

extern "C" int lc_regex_match_device_dyn(lc_regex_t* re, int engine, const uint8_t* d_data, const uint32_t* d_off,
                                         uint32_t sep_bytes, const uint32_t* d_nlines, uint32_t max_lines,
                                         uint32_t ngroups, int32_t* d_caps, uint8_t* d_status, void* stream) {
    if (!re || !d_nlines) return LC_ERR_ARG;
    if (max_lines == 0) return LC_OK;
    if (!d_data || !d_off || !d_caps || !d_status) return LC_ERR_ARG;
    if (lc_device_count() <= 0) {
        tlsError = "no HIP device";
        return LC_ERR_NO_DEVICE;
    }
    int dev = 0;
    {
        const int rcDev = lcDeviceEntryDevice(d_data, &dev);
        if (rcDev != LC_OK) return rcDev;
    }
    if (engine == LC_ENGINE_AUTO) engine = re->engine;
    return lcMatchOnStream(re, engine, dev, d_data, d_off, nullptr, sep_bytes, max_lines, d_nlines, nullptr, nullptr, ngroups, d_caps,
                         d_status, static_cast<hipStream_t>(stream));
}

extern "C" size_t lc_sched_scratch_bytes(uint32_t max_lines) { return (size_t(max_lines) + 2 * kSchedBuckets) * 4; }

// runtime_internal.hpp: order[] = the lines sorted by length bucket, longest first; work = 2 * kSchedBuckets words
int lcLengthOrderOnStream(const uint32_t* d_off, const uint32_t* d_len, uint32_t sep_bytes, uint32_t n, uint32_t* work, uint32_t* order,
                          hipStream_t st) {
    uint32_t* hist = work;
    uint32_t* cursor = hist + kSchedBuckets;
    HIP_TRY(hipMemsetAsync(hist, 0, kSchedBuckets * 4, st));
    const uint32_t grid = std::min<uint32_t>((n + kSchedBlock - 1) / kSchedBlock, 2048u);
    hipLaunchKernelGGL(sched_hist_kernel, dim3(grid), dim3(kSchedBlock), 0, st, d_off, d_len, sep_bytes, n, nullptr, hist);
    hipLaunchKernelGGL(sched_scan_kernel, dim3(1), dim3(kSchedBuckets), 0, st, hist, cursor);
    hipLaunchKernelGGL(sched_scatter_kernel, dim3(grid), dim3(kSchedBlock), 0, st, d_off, d_len, sep_bytes, n, nullptr, cursor, order);
    HIP_TRY(hipGetLastError());
    return LC_OK;
}

extern "C" int lc_regex_match_device_ragged(lc_regex_t* re, int engine, const uint8_t* d_data, const uint32_t* d_off,
                                            const uint32_t* d_len, uint32_t sep_bytes, uint32_t n,
                                            const uint32_t* d_nlines, uint32_t ngroups, int32_t* d_caps,
                                            uint8_t* d_status, void* d_scratch, size_t scratch_bytes, void* stream) {
    if (!re) return LC_ERR_ARG;
    if (n == 0) return LC_OK;
    if (!d_data || !d_off || !d_caps || !d_status || !d_scratch || scratch_bytes < lc_sched_scratch_bytes(n))
        return LC_ERR_ARG;
    if (lc_device_count() <= 0) {
        tlsError = "no HIP device";
        return LC_ERR_NO_DEVICE;
    }
    int dev = 0;
    {
        const int rcDev = lcDeviceEntryDevice(d_data, &dev);
        if (rcDev != LC_OK) return rcDev;
    }
    if (engine == LC_ENGINE_AUTO) engine = re->engine;
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint32_t* hist = static_cast<uint32_t*>(d_scratch);
    uint32_t* cursor = hist + kSchedBuckets;
    uint32_t* order = cursor + kSchedBuckets;
    HIP_TRY(hipMemsetAsync(hist, 0, kSchedBuckets * 4, st));
    const uint32_t grid = std::min<uint32_t>((n + kSchedBlock - 1) / kSchedBlock, 2048u);
    hipLaunchKernelGGL(sched_hist_kernel, dim3(grid), dim3(kSchedBlock), 0, st, d_off, d_len, sep_bytes, n, d_nlines, hist);
    hipLaunchKernelGGL(sched_scan_kernel, dim3(1), dim3(kSchedBuckets), 0, st, hist, cursor);
    hipLaunchKernelGGL(sched_scatter_kernel, dim3(grid), dim3(kSchedBlock), 0, st, d_off, d_len, sep_bytes, n, d_nlines,
                       cursor, order);
    HIP_TRY(hipGetLastError());
    return lcMatchOnStream(re, engine, dev, d_data, d_off, d_len, sep_bytes, n, d_nlines, order, nullptr, ngroups, d_caps, d_status,
                         st);
}

// ------------------------------------------------------------------------------------------------ line split
extern "C" size_t lc_split_scratch_bytes(uint64_t nbytes) {
    const uint64_t nBlocks = (nbytes + kSplitBytesPerBlock - 1) / kSplitBytesPerBlock;
    return size_t(nBlocks + 4) * 4;
}

extern "C" int lc_split_lines_device(const uint8_t* d_data, uint64_t nbytes, uint8_t split_char, uint32_t* d_off,
                                     uint32_t off_capacity, uint32_t* d_nlines, void* d_scratch, size_t scratch_bytes,
                                     void* stream) {
    if (!d_off || !d_nlines || off_capacity < 2) return LC_ERR_ARG;
    if (nbytes >= (uint64_t(1) << 32) - 1) return LC_ERR_ARG;  // offsets are 32-bit, like the match kernels'
    if (lc_device_count() <= 0) {
        tlsError = "no HIP device";
        return LC_ERR_NO_DEVICE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (nbytes == 0) {
        HIP_TRY(hipMemsetAsync(d_nlines, 0, 4, st));
        return LC_OK;
    }
    if (!d_data || !d_scratch || scratch_bytes < lc_split_scratch_bytes(nbytes)) return LC_ERR_ARG;
    const uint32_t nBlocks = uint32_t((nbytes + kSplitBytesPerBlock - 1) / kSplitBytesPerBlock);
    uint32_t* blockHits = static_cast<uint32_t*>(d_scratch);
    uint32_t* nHits = blockHits + nBlocks;
    hipLaunchKernelGGL(split_count_kernel, dim3(nBlocks), dim3(kSplitBlock), 0, st, d_data, nbytes, uint32_t(split_char),
                       blockHits);
    hipLaunchKernelGGL(split_scan_kernel, dim3(1), dim3(1024), 0, st, blockHits, nBlocks, nHits);
    hipLaunchKernelGGL(split_scatter_kernel, dim3(nBlocks), dim3(kSplitBlock), 0, st, d_data, nbytes,
                       uint32_t(split_char), blockHits, nHits, d_off, off_capacity, d_nlines);
    HIP_TRY(hipGetLastError());
    return LC_OK;
}

extern "C" int lc_upload_pinned(const void* pinned_src, void* d_dst, size_t nbytes, void* stream) {
    if (nbytes == 0) return LC_OK;
    if (!pinned_src || !d_dst || (reinterpret_cast<uintptr_t>(pinned_src) & 15) || (reinterpret_cast<uintptr_t>(d_dst) & 15)) return LC_ERR_ARG;
    const uint64_t n16 = (uint64_t(nbytes) + 15) / 16;  // (the block is read in whole 16-byte pieces: pad the source)
    const uint32_t grid = uint32_t(std::min<uint64_t>((n16 + 255) / 256, 1024));
    hipLaunchKernelGGL(pinned_upload_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const uint4*>(pinned_src),
                       static_cast<uint4*>(d_dst), n16);
    HIP_TRY(hipGetLastError());
    return LC_OK;
}

// ---- the filter step on the parser's capture spans (pipeline_kernel.hpp)
extern "C" int lc_span_filter_device(const lc_span_filter_t* filters, uint32_t nfilters, const uint8_t* d_data, const uint32_t* d_off,
                                     uint32_t sep_bytes, const uint32_t* d_nlines, uint32_t max_lines, uint32_t ngroups,
                                     const int32_t* d_caps, const uint8_t* d_status, int32_t* d_packed, uint32_t packed_cap_rows,
                                     uint32_t* d_counts, void* stream) {
    if (nfilters > kSpanFilterMax || (nfilters && !filters) || !d_nlines || !d_counts || ngroups == 0) return LC_ERR_ARG;
    if (max_lines == 0) return LC_OK;
    if (!d_data || !d_off || !d_caps || !d_status || (packed_cap_rows && !d_packed)) return LC_ERR_ARG;
    if (lc_device_count() <= 0) {
        tlsError = "no HIP device";
        return LC_ERR_NO_DEVICE;
    }
    int dev = 0;
    {
        const int rcDev = lcDeviceEntryDevice(d_data, &dev);
        if (rcDev != LC_OK) return rcDev;
    }
    SpanFilterArgs args{};
    args.n = nfilters;
    for (uint32_t f = 0; f < nfilters; ++f) {
        lc_regex* re = filters[f].re;
        if (!re || filters[f].group == 0 || filters[f].group > ngroups) return LC_ERR_ARG;
        if (re->screenBlob.empty()) {
            tlsError = "span filter: the rule's regex carries no yes/no DFA (lc_regex_prepare_span_filter)";
            return LC_ERR_UNSUPPORTED;
        }
        void* blob = nullptr;
        int rc = ensureUploaded(re, dev, kBlobScreen, &blob);
        if (rc != LC_OK) return rc;
        args.f[f].blob = static_cast<const uint32_t*>(blob);
        args.f[f].group = filters[f].group;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(d_counts, 0, 16, st));
    noteKernel("span_filter_pack_kernel");
    hipLaunchKernelGGL(span_filter_pack_kernel, dim3((max_lines + kSpanFilterBlock - 1) / kSpanFilterBlock), dim3(kSpanFilterBlock), 0, st,
                       d_data, d_off, sep_bytes, d_nlines, max_lines, ngroups, d_caps, d_status, args, d_packed, packed_cap_rows, d_counts);
    HIP_TRY(hipGetLastError());
    return LC_OK;
}

extern "C" int lc_regex_match_device(lc_regex_t* re, const uint8_t* d_data, const uint32_t* d_off,
                                     const uint32_t* d_len, uint32_t sep_bytes, uint32_t n, uint32_t ngroups,
                                     int32_t* d_caps, uint8_t* d_status, void* stream) {
    return lc_regex_match_device_engine(re, LC_ENGINE_AUTO, d_data, d_off, d_len, sep_bytes, n, ngroups, d_caps,
                                        d_status, stream);
}

// ------------------------------------------------------------------------------------------------ host batches
namespace {

// Completion of a zero-copy batch is signalled by the GPU itself: a one-lane kernel queued behind the match kernels stores the
// batch's sequence number into a pinned word the host thread polls.  (hipStreamSynchronize from many runner threads at once
// convoys inside the runtime: measured 7 GB/s with 2 threads, 1.2 GB/s with 32.)
__global__ void lc_signal_kernel(uint32_t* flag, uint32_t seq) {
    __atomic_store_n(flag, seq, __ATOMIC_RELEASE);
}
std::atomic<int> gZeroCopyWaiters{0};
}  // namespace

// The two halves of a ZERO-COPY device trip's ending, for the processors that run one outside runHostPipeline (filter, columnar,
// multiline: round 5).  Their kernels read the thread's pinned staging and write its pinned result block through the mapping; what is
// left of a trip is to learn that the last kernel is done.  lcQueueTripSignal puts a one-lane kernel behind everything queued on the
// stream that stores `seq` into the pinned word; lcAwaitTripSignal spins on the word while few threads wait and blocks in the runtime
// when many do (a spinning thread burns a core the others stitch on) -- the policy runHostPipeline measured in round 2.
int lcQueueTripSignal(uint32_t* hFlag, uint32_t seq, hipStream_t stream) {
    void* fargs[] = {&hFlag, &seq};
    HIP_TRY(hipLaunchKernel(reinterpret_cast<const void*>(lc_signal_kernel), dim3(1), dim3(1), fargs, 0, stream));
    return LC_OK;
}
int lcAwaitTripSignal(const uint32_t* hFlag, uint32_t seq, hipStream_t stream) {
    static const bool pollOff = getenv("LC_HOST_NO_POLL") != nullptr;
    const int ahead = gZeroCopyWaiters.fetch_add(1, std::memory_order_relaxed);
    hipError_t waitErr = hipSuccess;
    if (pollOff || ahead >= 4) {
        waitErr = hipStreamSynchronize(stream);
    } else {
        const volatile uint32_t* flag = hFlag;
        unsigned spins = 0;
        while (*flag != seq) {
            __builtin_ia32_pause();
            if (++spins > 40000u) {  // ~1 ms: a long kernel, or something is wrong -- the runtime's wait reports errors
                waitErr = hipStreamSynchronize(stream);
                break;
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    gZeroCopyWaiters.fetch_sub(1, std::memory_order_relaxed);
    HIP_TRY(waitErr);
    return LC_OK;
}
// lc_regex_match_device_multi on behalf of a zero-copy trip: the (small) job table is read by the kernel where the host wrote it, in
// pinned memory -- no copy command per group on the SDMA queue, where the groups of all runner threads would meet.  A dozen workgroups
// fetching 80 bytes over PCIe is nothing like the 250 that searched the table there in round 4.
void lcSetJobTableInPlace(bool on) { tlsJobTableInPlace = on; }

namespace {

struct Slot {
    uint32_t* hFlag = nullptr;  // pinned: sequence number of the last finished zero-copy batch
    uint32_t* dDone = nullptr;  // device: workgroups of the signalling launch that have finished (zero between launches)
    uint32_t flagSeq = 0;
    uint8_t* hData = nullptr;  // pinned
    uint32_t* hOff = nullptr;
    uint32_t* hLen = nullptr;
    int32_t* hCaps = nullptr;
    uint8_t* hStatus = nullptr;
    uint8_t* dData = nullptr;
    uint32_t* dOff = nullptr;
    uint32_t* dLen = nullptr;
    int32_t* dCaps = nullptr;
    uint8_t* dStatus = nullptr;
    size_t dataCap = 0, lineCap = 0, capsCap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    // what is in flight
    uint32_t first = 0, count = 0;
    bool busy = false;
    unsigned gatherWays = 1;  // gather_pool.hpp: how many ways the chunk's host copies are split (1: the calling thread alone)
};

struct HostPipeline {
    int device = -1;
    Slot slots[2];
    ~HostPipeline() { release(); }
    void release() {
        if (device < 0) return;
        if (!lcRuntimeUsable()) {  // process exit: the runtime may be gone, the OS reclaims everything
            device = -1;
            return;
        }
        (void)hipSetDevice(device);
        for (auto& s : slots) {
            if (s.stream) (void)hipStreamSynchronize(s.stream);
            (void)hipHostFree(s.hData); (void)hipHostFree(s.hOff); (void)hipHostFree(s.hLen);
            (void)hipHostFree(s.hCaps); (void)hipHostFree(s.hStatus); (void)hipHostFree(s.hFlag); (void)hipFree(s.dDone);
            (void)hipFree(s.dData); (void)hipFree(s.dOff); (void)hipFree(s.dLen);
            (void)hipFree(s.dCaps); (void)hipFree(s.dStatus);
            if (s.done) (void)hipEventDestroy(s.done);
            if (s.stream) (void)hipStreamDestroy(s.stream);
            s = Slot();
        }
        device = -1;
    }
};

thread_local HostPipeline* tlsPipe = nullptr;  // for lc_thread_release

constexpr size_t kChunkBytes = 32u << 20;   // payload bytes per pipelined chunk
constexpr uint32_t kChunkLines = 1u << 18;  // and at most this many lines

int growSlot(Slot& s, size_t dataBytes, size_t lines, size_t capsInts) {
    if (!s.stream) {
        HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.hFlag), 64, hipHostMallocDefault));
        *s.hFlag = 0;
        s.flagSeq = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.dDone), 64));
        HIP_TRY(hipMemset(s.dDone, 0, 64));
    }
    if (dataBytes > s.dataCap) {
        (void)hipHostFree(s.hData); (void)hipFree(s.dData);
        s.hData = nullptr; s.dData = nullptr; s.dataCap = 0;
        size_t cap = dataBytes + (dataBytes >> 2) + 4096;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.hData), cap, hipHostMallocDefault));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.dData), cap));
        s.dataCap = cap;
    }
    if (lines > s.lineCap) {
        (void)hipHostFree(s.hOff); (void)hipHostFree(s.hLen); (void)hipHostFree(s.hStatus);
        (void)hipFree(s.dOff); (void)hipFree(s.dLen); (void)hipFree(s.dStatus);
        s.hOff = s.hLen = nullptr; s.hStatus = nullptr; s.dOff = s.dLen = nullptr; s.dStatus = nullptr; s.lineCap = 0;
        size_t cap = lines + (lines >> 2) + 64;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.hOff), cap * 4, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.hLen), cap * 4, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.hStatus), cap, hipHostMallocDefault));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.dOff), cap * 4));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.dLen), cap * 4));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.dStatus), cap));
        s.lineCap = cap;
    }
    if (capsInts >= s.capsCap) {  // (>=: a status-only caller has capsInts == 0 and still gets a valid pointer)
        (void)hipHostFree(s.hCaps); (void)hipFree(s.dCaps);
        s.hCaps = nullptr; s.dCaps = nullptr; s.capsCap = 0;
        size_t cap = capsInts + (capsInts >> 2) + 64;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.hCaps), cap * 4, hipHostMallocDefault));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.dCaps), cap * 4));
        s.capsCap = cap;
    }
    return LC_OK;
}

int drainSlot(Slot& s, uint32_t ngroups, int32_t* caps, uint8_t* status) {
    if (!s.busy) return LC_OK;
    HIP_TRY(hipEventSynchronize(s.done));
    if (ngroups) lcgather::parallelCopy(caps + size_t(s.first) * 2 * ngroups, s.hCaps, size_t(s.count) * 2 * ngroups * 4, s.gatherWays);
    std::memcpy(status + s.first, s.hStatus, s.count);
    s.busy = false;
    return LC_OK;
}

}  // namespace

namespace {

// describes where the lines of a host batch live; both public entry points funnel into runHostPipeline
struct LineSource {
    const uint8_t* base = nullptr;         // (off,len) form: line i = base + off[i]
    const uint32_t* off = nullptr;
    const uint8_t* const* ptrs = nullptr;  // views form: line i = ptrs[i]
    const uint32_t* len = nullptr;
    const uint8_t* at(uint32_t i) const { return ptrs ? ptrs[i] : base + off[i]; }
};

int runHostPipeline(lc_regex_t* re, const LineSource& src, uint32_t n, uint32_t ngroups, int32_t* caps,
                    uint8_t* status) {
    if (n == 0) return LC_OK;
    if (lc_device_count() <= 0) {
        tlsError = "no HIP device";
        return LC_ERR_NO_DEVICE;
    }
    // the thread's device: its binding (lc_runtime_bind_thread; by default the thread's ordinal modulo the visible devices)
    int dev = 0;
    {
        const int rcDev = lcHostEntryDevice(&dev);
        if (rcDev != LC_OK) return rcDev;
    }
    static thread_local HostPipeline pipe;
    lcRegisterExitHook();
    tlsPipe = &pipe;
    if (pipe.device != dev) {
        pipe.release();
        pipe.device = dev;
    }
    // Whatever way this call ends, no slot may stay `busy`: the next call on this thread would otherwise drain a stale
    // chunk (its first/count) into the new caller's buffers.  On an error path the guard waits for both streams and
    // drops what was in flight without copying anything out.
    struct DrainGuard {
        HostPipeline& p;
        ~DrainGuard() {
            bool dropped = false;
            for (auto& s : p.slots) {
                if (!s.busy) continue;
                if (s.stream) (void)hipStreamSynchronize(s.stream);
                s.busy = false;
                s.first = s.count = 0;
                dropped = true;
            }
            if (dropped) (void)hipGetLastError();  // (error paths only: this is a runtime call, and every group passes here)
        }
    } guard{pipe};
    uint32_t next = 0;
    int which = 0;
    int rc = LC_OK;
    // ---- small batches (one chunk: what ProcessorRunner hands over, ~1000 lines): ZERO-COPY.  The lines are gathered into
    // the slot's pinned staging and the kernels read them -- and write the capture table -- straight through the pinned
    // mapping: no hipMemcpyAsync at all, one kernel launch per group.  Measured with tools/inagent_bench.cpp (1000-line groups,
    // lc_processor_process, profiles/round2_inagent.txt): the copy pipeline's five small copies per group all pass through the
    // device's SDMA queue, where the groups of ALL runner threads serialise (1.9 GB/s with 1 thread, 7.8 GB/s with 16); a
    // kernel that streams its 64-byte stages over PCIe overlaps transfer and compute by construction and leaves the threads
    // independent (2.5 GB/s with 1 thread, 16.6-20 GB/s with 16-32).  Batches of several chunks keep the copy pipeline below
    // (its copies overlap with the kernels and use the bus better: 21 GB/s on 1 Mi lines).
    // LC_HOST_ZEROCOPY=0: never; =2: the block goes up with one copy, only the results are zero-copy (measured slower).
    static const int zeroCopyEnv = [] {
        const char* e = getenv("LC_HOST_ZEROCOPY");
        return e ? atoi(e) : -1;
    }();
    {
        size_t bytes = 0;
        for (uint32_t i = 0; i < n && bytes <= kChunkBytes; ++i) bytes += src.len[i];
        const bool oneChunk = n <= kChunkLines && bytes <= kChunkBytes;
        if (zeroCopyEnv != 0 && oneChunk) {
            Slot& s = pipe.slots[0];
            const uint8_t* lo = src.at(0);
            const uint8_t* hi = src.at(n - 1) + src.len[n - 1];
            bool contiguous = hi >= lo && size_t(hi - lo) <= bytes + 2ull * n;
            for (uint32_t i = 1; i < n && contiguous; ++i) {
                const uint8_t* prevEnd = src.at(i - 1) + src.len[i - 1];
                const uint8_t* cur = src.at(i);
                contiguous = cur >= prevEnd && size_t(cur - prevEnd) <= 2;
            }
            const size_t stageBytes = contiguous ? size_t(hi - lo) : bytes;
            // one staging block: the lines, then the (offset, length) tables -- so that the copy-up variant needs ONE copy
            const size_t tableAt = (stageBytes + 16 + 63) & ~size_t(63);
            const size_t blockBytes = tableAt + 8 * size_t(n);
            if ((rc = growSlot(s, blockBytes, n, size_t(n) * 2 * ngroups)) != LC_OK) return rc;
            uint32_t* hOff = reinterpret_cast<uint32_t*>(s.hData + tableAt);
            uint32_t* hLen = hOff + n;
            if (contiguous) {
                std::memcpy(s.hData, lo, stageBytes);
                for (uint32_t i = 0; i < n; ++i) {
                    hOff[i] = uint32_t(src.at(i) - lo);
                    hLen[i] = src.len[i];
                }
            } else {
                size_t at = 0;
                for (uint32_t i = 0; i < n; ++i) {
                    std::memcpy(s.hData + at, src.at(i), src.len[i]);
                    hOff[i] = uint32_t(at);
                    hLen[i] = src.len[i];
                    at += src.len[i];
                }
            }
            std::memset(s.hData + stageBytes, 0, 16);
            static const bool pollOff = getenv("LC_HOST_NO_POLL") != nullptr;
            const uint32_t doneSeqNo = ++s.flagSeq;
            // Whatever way this block is left: the completion request is disarmed (a later, unrelated launch of this thread must
            // not signal a stale flag), and after an error nothing queued here may still be reading the staging block or writing
            // the pinned capture table when the next call reuses them.
            struct ZeroCopyGuard {
                hipStream_t stream;
                bool queued = false, ok = false;
                ~ZeroCopyGuard() {
                    tlsDone = DoneRequest{};
                    if (queued && !ok) {
                        (void)hipStreamSynchronize(stream);
                        (void)hipGetLastError();
                    }
                }
            } zc{s.stream};
            tlsDone = DoneRequest{s.dDone, s.hFlag, doneSeqNo, !pollOff, false};
            zc.queued = true;
            // (the backtracking engine reads a value in small steps, some of them more than once: through the pinned mapping every
            // step would be a PCIe read of its own -- its group is copied up first, in one piece)
            if (zeroCopyEnv == 2 || re->engine == LC_ENGINE_BT) {
                HIP_TRY(hipMemcpyAsync(s.dData, s.hData, blockBytes, hipMemcpyHostToDevice, s.stream));
                rc = lcMatchOnStream(re, re->engine, dev, s.dData, reinterpret_cast<uint32_t*>(s.dData + tableAt),
                                     reinterpret_cast<uint32_t*>(s.dData + tableAt) + n, 0, n, nullptr, nullptr, nullptr, ngroups,
                                     s.hCaps, s.hStatus, s.stream);
            } else {
                rc = lcMatchOnStream(re, re->engine, dev, s.hData, hOff, hLen, 0, n, nullptr, nullptr, nullptr, ngroups, s.hCaps,
                                     s.hStatus, s.stream);
            }
            const bool signalled = tlsDone.consumed;
            tlsDone = DoneRequest{};
            if (rc != LC_OK) return rc;
            // Waiting.  Few runner threads: spin on the pinned flag word the match kernel's last workgroup stores (the kernels of
            // a 1000-line group last ~60 us; the runtime's wait costs ~40 us more per group than a spin: 2.5 vs 2.1 GB/s with one
            // thread).  Many: spinning threads burn the cores the others stitch on, and the runtime's blocking wait scales better
            // (20.0 vs 16.6 GB/s with 16 threads on a 16-core quota) -- tools/inagent_bench.cpp, profiles/round2_inagent.txt.
            std::atomic<int>& waiters = gZeroCopyWaiters;
            const int ahead = waiters.fetch_add(1, std::memory_order_relaxed);
            hipError_t waitErr = hipSuccess;
            if (pollOff || ahead >= 4) {
                waitErr = hipStreamSynchronize(s.stream);
            } else {
                uint32_t seq = doneSeqNo;
                if (!signalled) {  // the match was more than one launch (or not a TDFA launch): a one-lane kernel behind it signals
                    void* fargs[] = {&s.hFlag, &seq};
                    waitErr = hipLaunchKernel(reinterpret_cast<const void*>(lc_signal_kernel), dim3(1), dim3(1), fargs, 0, s.stream);
                }
                volatile uint32_t* flag = s.hFlag;
                unsigned spins = 0;
                while (waitErr == hipSuccess && *flag != seq) {
                    __builtin_ia32_pause();
                    if (++spins > 40000u) {  // ~1 ms: long kernel, or something is wrong -- the runtime's wait reports errors
                        waitErr = hipStreamSynchronize(s.stream);
                        break;
                    }
                }
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
            }
            waiters.fetch_sub(1, std::memory_order_relaxed);
            HIP_TRY(waitErr);
            if (ngroups) std::memcpy(caps, s.hCaps, size_t(n) * 2 * ngroups * 4);
            std::memcpy(status, s.hStatus, n);
            zc.ok = true;
            return LC_OK;
        }
    }
    while (next < n) {
        // carve a chunk: up to kChunkLines lines / kChunkBytes payload bytes
        uint32_t cnt = 0;
        size_t bytes = 0;
        while (next + cnt < n && cnt < kChunkLines && (cnt == 0 || bytes + src.len[next + cnt] <= kChunkBytes)) {
            bytes += src.len[next + cnt];
            ++cnt;
        }
        Slot& s = pipe.slots[which];
        if ((rc = drainSlot(s, ngroups, caps, status)) != LC_OK) return rc;
        // contiguous fast path: the chunk's lines sit back to back in memory with at most a separator byte or two
        // between them (what one LogFileReader buffer looks like after ProcessorSplitLogStringNative): one memcpy
        const uint8_t* lo = src.at(next);
        const uint8_t* hi = src.at(next + cnt - 1) + src.len[next + cnt - 1];
        bool contiguous = hi >= lo && size_t(hi - lo) <= bytes + 2ull * cnt;
        for (uint32_t i = 1; i < cnt && contiguous; ++i) {
            const uint8_t* prevEnd = src.at(next + i - 1) + src.len[next + i - 1];
            const uint8_t* cur = src.at(next + i);
            contiguous = cur >= prevEnd && size_t(cur - prevEnd) <= 2;
        }
        const size_t stageBytes = contiguous ? size_t(hi - lo) : bytes;
        if ((rc = growSlot(s, stageBytes + 16, cnt, size_t(cnt) * 2 * ngroups)) != LC_OK) return rc;
        // Round 6: the chunk's host copies split over a few helper threads (gather_pool.hpp): one thread gathers ~10 GB/s, the bus
        // takes 55 -- the path was gather-bound at 16-20 GB/s for three rounds
        const unsigned ways = (n > cnt || next > 0) ? lcgather::GatherPool::instance().width() : 1u;  // (only batches of several chunks)
        s.gatherWays = ways;
        if (contiguous) {
            lcgather::parallelCopy(s.hData, lo, stageBytes, ways);
            for (uint32_t i = 0; i < cnt; ++i) {
                s.hOff[i] = uint32_t(src.at(next + i) - lo);
                s.hLen[i] = src.len[next + i];
            }
        } else if (ways > 1 && bytes >= (size_t(1) << 20)) {
            // per-line gather: the lines dealt in runs of equal byte count, each run to its place
            std::vector<uint32_t> cut(ways + 1, cnt);
            std::vector<size_t> cutAt(ways + 1, bytes);
            {
                size_t at = 0;
                unsigned w = 0;
                for (uint32_t i = 0; i < cnt; ++i) {
                    while (w < ways && at >= bytes / ways * w) {
                        cut[w] = i;
                        cutAt[w] = at;
                        ++w;
                    }
                    s.hOff[i] = uint32_t(at);
                    s.hLen[i] = src.len[next + i];
                    at += src.len[next + i];
                }
            }
            lcgather::GatherPool::instance().run(ways, [&](unsigned w) {
                size_t at = cutAt[w];
                for (uint32_t i = cut[w]; i < cut[w + 1]; ++i) {
                    std::memcpy(s.hData + at, src.at(next + i), src.len[next + i]);
                    at += src.len[next + i];
                }
            });
        } else {
            size_t at = 0;
            for (uint32_t i = 0; i < cnt; ++i) {
                std::memcpy(s.hData + at, src.at(next + i), src.len[next + i]);
                s.hOff[i] = uint32_t(at);
                s.hLen[i] = src.len[next + i];
                at += src.len[next + i];
            }
        }
        HIP_TRY(hipMemcpyAsync(s.dData, s.hData, stageBytes, hipMemcpyHostToDevice, s.stream));
        HIP_TRY(hipMemcpyAsync(s.dOff, s.hOff, size_t(cnt) * 4, hipMemcpyHostToDevice, s.stream));
        HIP_TRY(hipMemcpyAsync(s.dLen, s.hLen, size_t(cnt) * 4, hipMemcpyHostToDevice, s.stream));
        rc = lc_regex_match_device(re, s.dData, s.dOff, s.dLen, 0, cnt, ngroups, s.dCaps, s.dStatus, s.stream);
        if (rc != LC_OK) return rc;
        if (ngroups)
            HIP_TRY(hipMemcpyAsync(s.hCaps, s.dCaps, size_t(cnt) * 2 * ngroups * 4, hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(hipMemcpyAsync(s.hStatus, s.dStatus, cnt, hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(hipEventRecord(s.done, s.stream));
        s.first = next;
        s.count = cnt;
        s.busy = true;
        next += cnt;
        which ^= 1;
    }
    for (auto& s : pipe.slots)
        if ((rc = drainSlot(s, ngroups, caps, status)) != LC_OK) return rc;
    return LC_OK;
}

}  // namespace

extern "C" void lc_thread_release(void) {
    tlsBind.unbind();  // the thread's ordinal goes back to the deal; its next host entry binds it afresh
    tlsDfsPool.release();
    if (tlsPipe) tlsPipe->release();
    tlsJobTables.release();
    for (auto& pool : tlsDecidePools) pool.release();
    lcGrokThreadRelease();
    lcPipelineThreadRelease();
    lcMultilineThreadRelease();
    lcFilterThreadRelease();
}

extern "C" void lc_nfa_set_dfs(int on) { gNfaDfsMode.store(on < 0 ? -1 : (on ? 1 : 0), std::memory_order_relaxed); }

extern "C" int lc_dfs_stats(uint64_t lines[2]) {
    if (!lines) return LC_ERR_ARG;
    lines[0] = lines[1] = 0;
    DfsPool& pool = tlsDfsPool;
    if (!pool.p || !pool.used) return LC_OK;
    HIP_TRY(hipSetDevice(pool.device));
    HIP_TRY(hipEventSynchronize(pool.lastUse));
    DfsPoolHeader h;
    HIP_TRY(hipMemcpy(&h, pool.p, sizeof h, hipMemcpyDeviceToHost));
    lines[0] = h.cursor;
    lines[1] = h.pending;
    return LC_OK;
}

extern "C" int lc_decide_stats(uint64_t lines[2]) {
    if (!lines) return LC_ERR_ARG;
    lines[0] = lines[1] = 0;
    // (per-launch numbers live in the pool headers; a caller that wants them per batch calls this after each batch.  Summed over
    // the thread's stream slots: the Grok matcher spreads its entries over several)
    for (DecidePool& pool : tlsDecidePools) {
        if (!pool.p || !pool.used) continue;
        HIP_TRY(hipSetDevice(pool.device));
        HIP_TRY(hipEventSynchronize(pool.lastUse));
        DecidePlan plan;
        HIP_TRY(hipMemcpy(&plan, pool.p, sizeof plan, hipMemcpyDeviceToHost));
        lines[0] += plan.count;
        lines[1] += plan.gaveUp;
    }
    return LC_OK;
}

extern "C" int lc_regex_match_host(lc_regex_t* re, const uint8_t* data, const uint32_t* off, const uint32_t* len,
                                   uint32_t n, uint32_t ngroups, int32_t* caps, uint8_t* status) {
    if (!re) return LC_ERR_ARG;
    if (n == 0) return LC_OK;
    if (!data || !off || !len || (ngroups && !caps) || !status) return LC_ERR_ARG;  // ngroups == 0: status only
    LineSource src;
    src.base = data;
    src.off = off;
    src.len = len;
    return runHostPipeline(re, src, n, ngroups, caps, status);
}

extern "C" int lc_regex_match_host_views(lc_regex_t* re, const uint8_t* const* lines, const uint32_t* len, uint32_t n,
                                         uint32_t ngroups, int32_t* caps, uint8_t* status) {
    if (!re) return LC_ERR_ARG;
    if (n == 0) return LC_OK;
    if (!lines || !len || (ngroups && !caps) || !status) return LC_ERR_ARG;  // ngroups == 0: status only
    LineSource src;
    src.ptrs = lines;
    src.len = len;
    return runHostPipeline(re, src, n, ngroups, caps, status);
}

