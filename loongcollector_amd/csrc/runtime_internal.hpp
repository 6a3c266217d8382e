// runtime_internal.hpp -- what the device translation units share (implemented in gpu_runtime.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

struct lc_regex;

void lcSetLastError(const std::string& msg);           // the thread's lc_last_error() text
int lcHipFail(hipError_t e, const char* what);         // sets the error text, returns LC_ERR_HIP
void lcNoteKernel(const char* name);                   // lc_launched_kernels() log
void lcRegisterExitHook();                             // thread_local device resources: see gpu_runtime.hip
bool lcRuntimeUsable();
void lcGrokThreadRelease();                            // grok_device.hip: the calling thread's Grok buffers
void lcPipelineThreadRelease();                        // processor_pipeline_gpu.cpp: the calling thread's staging and stream
void lcMultilineThreadRelease();                       // multiline_device.hip: the same for the multiline processors
void lcFilterThreadRelease();                          // processor_filter_gpu.cpp: the same for the filter
// The device a HOST entry point (processors, lc_*_match_host, multiline, filter, pipeline) runs on for the calling thread: the thread's
// binding (lc_runtime_bind_thread; first call binds by the process-wide policy), made current for the thread.  LC_OK or an error code.
int lcHostEntryDevice(int* dev);
// The device a DEVICE-pointer entry point runs on: the caller's current HIP device; LC_ERR_ARG when d_ptr lives on another device.
int lcDeviceEntryDevice(const void* d_ptr, int* dev);
// the ending of a zero-copy device trip (gpu_runtime.hip): a one-lane kernel behind everything on `stream` stores seq into the pinned word;
// the host spins on it (few waiters) or blocks in the runtime (many)
int lcQueueTripSignal(uint32_t* hFlag, uint32_t seq, hipStream_t stream);
int lcAwaitTripSignal(const uint32_t* hFlag, uint32_t seq, hipStream_t stream);
// the calling thread's next lc_regex_match_device_multi calls let the kernel read their (small) job tables from pinned memory
void lcSetJobTableInPlace(bool on);
// the decide pool the calling thread's next NFA launches use (0 = default; 1.. = worker streams of the Grok matcher)
void lcSetDecideSlot(int slot);
// where the calling thread's NEXT wide-kernel launch reports that it had work (a device word set to 1; nullptr = nowhere)
void lcSetWideNote(uint32_t* note);
// device copy of a screen handle's yes/no DFA (screen_kernel_layout.h)
int lcEnsureScreenUploaded(lc_regex* re, int dev, const uint32_t** out);

// ---- several automata over their own values in ONE launch (tdfa_l2_kernel.hpp tdfa_wave_multi_kernel; the Grok plan's round 0)
// lcWaveJobPrepare: can `re` walk `n` values one per wavefront from tables in global memory -- a complete automaton that lives there
// (or asked for the wave walk), or a thread-list handle with a lazy automaton whose misses the WIDE kernel can take as a second chance?
// If so fills the job's engine side (tables on `dev`, staging, and for lazy automata the flag word / sequence number / LC_OVERFLOW its
// misses raise: what lcMatchSecondChanceOnStream takes as `seq`) and returns true; the caller fills off / len / resume / caps / status /
// n / nGroupsOut / firstBlock.  false: the handle goes its usual way (lcMatchFirstOnStream).  rc != LC_OK: a device error.
struct TdfaWaveJob;
// stagePrograms: the register programs ride in LDS (a launch of a few thousand values waits for its longest value; a launch that fills the
// chip several times over is better off with the waves the LDS would cost)
bool lcWaveJobPrepare(lc_regex* re, int dev, uint32_t n, bool stagePrograms, TdfaWaveJob* job, uint32_t* ldsBytes, uint32_t* seqOut, int* rc);
// the jobs (firstBlock ascending, at most 64) as one launch on `st`; dTable / hTable: device and pinned buffers of lcWaveJobTableBytes()
size_t lcWaveJobTableBytes();
int lcLaunchWaveJobs(const uint8_t* d_data, const TdfaWaveJob* jobs, uint32_t nJobs, uint32_t totalBlocks, uint32_t ldsBytes, void* hTable,
                     void* dTable, int dev, hipStream_t st);

// order[] = the lines 0..n-1 sorted by length bucket (32 bytes), longest first (sched_kernel.hpp); work: 512 words
int lcLengthOrderOnStream(const uint32_t* d_off, const uint32_t* d_len, uint32_t sep_bytes, uint32_t n, uint32_t* work, uint32_t* order,
                          hipStream_t st);

#define LC_HIP_TRY(expr)                                    \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) return lcHipFail(e_, #expr);  \
    } while (0)
