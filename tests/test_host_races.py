"""The product's HOST code under runner threads, under ThreadSanitizer.

Runner threads share a processor instance (ProcessQueueManager.cpp:167-205: queues are not pinned to threads; the reference keeps one
boost::regex per thread for that reason, ProcessorParseRegexNative.cpp:255-257).  tests/native/race_driver.cpp runs 8 threads on ONE instance
of every processor -- parse (stitched and columnar), filter, the fused pipeline, the multiline splitter, the merge processor -- with the
device calls answered by the CPU doubles, compares every group with the answer the same code gave single-threaded and the counters with
their sums; built with -fsanitize=thread, any report fails the test.  (What the device side does under threads -- streams, staging,
tables shared between them -- is the -m gpu tests' business: test_concurrent_process_calls_on_one_instance and its neighbours.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tsan_usable():
    if shutil.which("g++") is None:
        return False
    r = subprocess.run(["gcc", "-print-file-name=libtsan.so"], stdout=subprocess.PIPE, text=True)
    return os.path.isabs(r.stdout.strip()) and os.path.exists(r.stdout.strip())


@pytest.mark.skipif(not _tsan_usable(), reason="needs g++ with libtsan")
def test_runner_threads_on_shared_instances_under_thread_sanitizer():
    from loongcollector_amd import build as native_build
    objdir = os.path.join(ROOT, "loongcollector_amd", "lib", "obj")
    if not os.path.exists(os.path.join(objdir, "grok_defaults.inc")):
        native_build.build_native()
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "race_driver")
    csrc = os.path.join(ROOT, "loongcollector_amd", "csrc")
    native = os.path.join(ROOT, "tests", "native")
    srcs = [os.path.join(native, "race_driver.cpp")] + [os.path.join(csrc, f) for f in (
        "c_processor_slot.cpp", "processor_pipeline_gpu.cpp", "processor_parse_regex_gpu.cpp", "processor_filter_gpu.cpp", "multiline_events.cpp",
        "multiline_gpu.cpp", "event_model.cpp")] + [os.path.join(ROOT, "oracle", "bt_regex.c")]
    deps = srcs + [os.path.join(native, f) for f in ("pipeline_double.cpp", "filter_double.cpp", "multiline_double.cpp")] + [
        os.path.join(csrc, h) for h in ("event_model.hpp", "processor_parse_regex_gpu.hpp", "processor_filter_gpu.hpp", "processor_pipeline_gpu.hpp",
                                        "multiline_gpu.hpp", "multiline_scan.hpp", "trip_buffers.hpp")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-w", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include",
                               "-I", os.path.join(ROOT, "include"), "-I", csrc, "-I", objdir, "-o", exe] + srcs + ["-lpthread"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    env.pop("LD_PRELOAD", None)   # (tools/asan_check.sh runs the suite with ASan's runtime preloaded: two sanitizer runtimes do not share a process)
    r = subprocess.run([exe, "8", "40"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    assert "0 mismatching groups, 0 counters off" in r.stdout, r.stdout


@pytest.mark.skipif(not _tsan_usable(), reason="needs g++ with libtsan")
def test_the_group_commit_of_concurrent_runner_threads_under_thread_sanitizer():
    """csrc/group_combiner.hpp (round 6): the groups of the runner threads that call lc_grok_match_host together travel as ONE device
    batch (core/runner/ProcessorRunner.cpp:138-142: synchronous, one group per call, concurrent callers on one instance).  The combiner
    holds no HIP: tests/native/combiner_race.cpp gives it a device that is a function and runs 16 threads x 200 groups through it under
    ThreadSanitizer -- every job gets the rows of ITS values whatever batch it travelled in, batches never overlap, one thread alone
    never lingers, sixteen find each other (> 0.6 x 16 jobs per batch), a failing batch fails exactly its jobs, stop() with callers in
    flight answers or refuses every one of them."""
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "combiner_race")
    src = os.path.join(ROOT, "tests", "native", "combiner_race.cpp")
    hdr = os.path.join(ROOT, "loongcollector_amd", "csrc", "group_combiner.hpp")
    hdr2 = os.path.join(ROOT, "loongcollector_amd", "csrc", "gather_pool.hpp")   # (part (d) of the driver: the host-fed path's gather helpers)
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in (src, hdr, hdr2)):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-I", os.path.dirname(hdr), "-o", exe, src, "-lpthread"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe, "16", "200"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, (r.returncode, r.stdout[-800:], r.stderr[-2000:])
    assert "0 checks failed" in r.stdout, r.stdout


@pytest.mark.skipif(not _tsan_usable(), reason="needs g++ with libtsan")
def test_thread_to_gpu_placement_on_a_two_device_double_under_thread_sanitizer():
    """csrc/device_binding.hpp (ProcessorRunner.h:40: runner thread k -> GPU k mod n): no box of this project has
    ever shown more than one GPU, so the multi-device branches of tests/test_gpu_binding.py have never run.  Their CPU twin:
    tests/native/binding_race.cpp gives the policy a HIP double with TWO devices -- the deal in order of first entry, ordinals that come
    back with lc_thread_release and at thread exit (200 helper threads do not skew the deal), a host library that moves the thread's
    current device between two calls, the fixed / inherit policies, a thread its host has placed, devices out of range, no device --
    and sixteen threads that enter, release and re-enter at once, under ThreadSanitizer."""
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "binding_race")
    src = os.path.join(ROOT, "tests", "native", "binding_race.cpp")
    hdr = os.path.join(ROOT, "loongcollector_amd", "csrc", "device_binding.hpp")
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in (src, hdr)):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-I", os.path.dirname(hdr), "-o", exe, src, "-lpthread"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "0 checks failed" in r.stdout, (r.returncode, r.stdout[-800:], r.stderr[-1500:])
