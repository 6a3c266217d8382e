"""The shim the agent would load, type-checked against the reference's OWN headers.

`-DLC_USE_REFERENCE_HEADERS` swaps the stand-in event model (csrc/event_model.hpp) for core/models/LogEvent.h /
PipelineEventGroup.h and makes the dlsym slot's init() take the `const Json::Value*` the agent passes
(core/plugin/processor/DynamicCProcessorProxy.cpp:25-40).  Until round 2 that variant had never been through a compiler.
Boost, JsonCpp, spdlog and the protoc-generated checkpoint.pb.h / sls_logs.pb.h are not in this image: tests/refhdr/ holds
declarations-only stand-ins for exactly those headers (string_view, the few boost headers core/common/Lock.h names, json.h,
spdlog.h, the two .pb.h), everything else on the include path is the reference tree itself.  Since round 3 the parse processor
also includes the reference's CollectionPipelineContext / AlarmManager / AppConfig: the slot keeps the context the agent hands
to init() and raises the REGEX_MATCH_ALARM paths of RegexLogLineParser there (ProcessorParseRegexNative.cpp:196-244).
`g++ -fsyntax-only` parses, resolves overloads and instantiates every template the sources use -- i.e. each call the
shim makes on LogEvent / PipelineEventGroup / SourceBuffer / Json::Value exists with those argument types.

/root/reference does not exist on the GPU box: the test is skipped there.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/core"
CSRC = os.path.join(ROOT, "loongcollector_amd", "csrc")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("g++") is None,
                                reason="needs the reference tree (/root/reference) and g++")

SOURCES = ["processor_parse_regex_gpu.cpp", "c_processor_slot.cpp", "processor_filter_gpu.cpp", "processor_grok_gpu.cpp", "processor_pipeline_gpu.cpp",
           "processor_go_regex_gpu.cpp", "multiline_gpu.cpp", "multiline_events.cpp"]


def _syntax_only(src, extra=()):
    from loongcollector_amd import build as native_build
    objdir = os.path.join(ROOT, "loongcollector_amd", "lib", "obj")
    if not os.path.exists(os.path.join(objdir, "grok_defaults.inc")):
        native_build.build_native()  # (writes the generated include processor_grok_gpu.cpp needs)
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-DLC_USE_REFERENCE_HEADERS", "-I", os.path.join(ROOT, "tests", "refhdr"),
           "-I", REF, "-I", os.path.join(REF, "config"), "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-I", objdir, *extra, os.path.join(CSRC, src)]
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


@pytest.mark.parametrize("src", SOURCES)
def test_shim_type_checks_against_the_reference_headers(src):
    # (the fused pipeline talks to the HIP runtime itself: its own stream and pinned staging)
    extra = ("-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include") if src in ("processor_pipeline_gpu.cpp", "processor_filter_gpu.cpp") else ()
    r = _syntax_only(src, extra)
    assert r.returncode == 0, r.stdout[-3000:]


def test_the_reference_headers_are_really_the_ones_parsed():
    """Guard against a silently ignored switch: with the reference headers, LogEvent has no AppendContentsNoCopy (the stand-in's
    bulk stitch entry) -- a translation unit that calls it must FAIL to compile, and the compiler's suggestion must be a member
    only the reference's LogEvent has (AppendContentNoCopy, core/models/LogEvent.h)."""
    probe = os.path.join(ROOT, "scratch", "refhdr_probe.cpp")
    os.makedirs(os.path.dirname(probe), exist_ok=True)
    with open(probe, "w") as f:
        f.write('#include "processor_parse_regex_gpu.hpp"\n'
                "void probe(logtail::LogEvent& e, logtail::StringView* k) { e.AppendContentsNoCopy(k, k, 1); }\n")
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-DLC_USE_REFERENCE_HEADERS", "-I", os.path.join(ROOT, "tests", "refhdr"), "-I", REF,
           "-I", os.path.join(REF, "config"), "-I", os.path.join(ROOT, "include"), "-I", CSRC, probe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0 and "no member named" in r.stdout and "AppendContentNoCopy" in r.stdout, r.stdout[-2000:]
    with open(os.path.join(CSRC, "event_model.hpp")) as f:
        assert "AppendContentNoCopy(" not in f.read()  # (the suggestion cannot have come from the stand-in)
