"""Build liblc_regex_gpu.so (host compilers + gfx950 kernels + C ABI) in-tree with hipcc.

    python -m loongcollector_amd.build          # or __graft_entry__.build()

The .so lands in loongcollector_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "liblc_regex_gpu.so")
# the same library with the stand-in event model built in the REFERENCE's shape (csrc/event_model.hpp LC_REFERENCE_SHAPED_EVENT_MODEL:
# heap std::vector contents, no chunk pool, K x SetContentNoCopy + DelContent).  bench.py loads it in a process of its own for
# end_to_end.in_agent_reference_shape_MBps -- what an agent build's event model lets one runner thread do; never the default.
LIB_REFSHAPE = os.path.join(LIBDIR, "liblc_regex_gpu_refshape.so")

SOURCES = ["regex_parse.cpp", "atomic_elide.cpp", "follow_nfa.cpp", "tdfa.cpp", "table_cache.cpp", "screen_dfa.cpp", "bt_program.cpp", "regex_handle.cpp", "gpu_runtime.hip", "grok_device.hip", "multiline_device.hip"]
OPTIONAL_SOURCES = ["event_model.cpp", "processor_parse_regex_gpu.cpp", "grok.cpp", "grok_literal_index.cpp", "processor_grok_gpu.cpp", "processor_filter_gpu.cpp", "processor_go_regex_gpu.cpp", "multiline_gpu.cpp", "multiline_events.cpp", "processor_pipeline_gpu.cpp", "c_processor_slot.cpp"]


# what the on-disk table cache's stamp is made of: the constructions, the structures they read and write, the cache's own format
TABLE_SHAPING_SOURCES = ["tdfa.cpp", "screen_dfa.cpp", "tdfa.hpp", "follow_nfa.hpp", "regex_ast.hpp", "table_cache.cpp"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    out = [os.path.join(CSRC, s) for s in SOURCES]
    out += [os.path.join(CSRC, s) for s in OPTIONAL_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(HERE, "..", "include", "lc_regex_gpu.h"))
    deps.append(os.path.join(HERE, "..", "include", "lc_processor.h"))
    deps.append(os.path.join(HERE, "..", "include", "lc_grok.h"))
    deps.append(os.path.join(HERE, "..", "include", "lc_go_regex.h"))
    deps.append(os.path.join(HERE, "..", "include", "lc_multiline.h"))
    deps.append(os.path.join(HERE, "data", "grok_default_patterns.txt"))
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


CORPUS_SRC = os.path.join(HERE, "..", "tools", "corpus_gen.cpp")
CORPUS_LIB = os.path.join(LIBDIR, "libcorpus_gen.so")


def build_corpus_gen():
    """tools/corpus_gen.cpp (bench / test tooling: the headline corpus generated line by line from std::mt19937_64) -> lib/libcorpus_gen.so"""
    os.makedirs(LIBDIR, exist_ok=True)
    if os.path.exists(CORPUS_LIB) and os.path.getmtime(CORPUS_LIB) > os.path.getmtime(CORPUS_SRC):
        return CORPUS_LIB
    cxx = shutil.which("g++") or _hipcc()
    subprocess.check_call([cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-o", CORPUS_LIB, CORPUS_SRC])
    return CORPUS_LIB


def build_native(force=False, verbose=False, force_sources=()):
    """force_sources: basenames of sources to recompile even if their objects look current (smoke() uses it to show a real
    gfx950 compile on the GPU box)."""
    if not force and not force_sources and not needs_build() and os.path.exists(LIB_REFSHAPE) and \
            os.path.getmtime(LIB_REFSHAPE) >= os.path.getmtime(LIB):
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    # the Grok default library (data/grok_default_patterns.txt) is compiled in as one raw string literal
    inc = os.path.join(objdir, "grok_defaults.inc")
    with open(os.path.join(HERE, "data", "grok_default_patterns.txt"), encoding="utf-8") as f:
        blob = 'R"GROKDATA(' + f.read() + ')GROKDATA"\n'
    if not os.path.exists(inc) or open(inc, encoding="utf-8").read() != blob:
        with open(inc, "w", encoding="utf-8") as f:
            f.write(blob)
    # the stamp of the on-disk table cache (csrc/table_cache.cpp): a hash of every source that shapes the compiled tables, so that an
    # edit of tdfa.cpp alone also invalidates what an older build of the library has written
    import hashlib
    hh = hashlib.sha256()
    for name in TABLE_SHAPING_SOURCES:
        with open(os.path.join(CSRC, name), "rb") as f:
            hh.update(name.encode() + b"\0" + f.read() + b"\0")
    stamp = '"lc-tables-%s"\n' % hh.hexdigest()[:32]
    stamp_inc = os.path.join(objdir, "table_sources_stamp.inc")
    if not os.path.exists(stamp_inc) or open(stamp_inc).read() != stamp:
        with open(stamp_inc, "w") as f:
            f.write(stamp)
    common = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-I", os.path.join(HERE, "..", "include"),
              "-I", CSRC, "-I", objdir] + os.environ.get("LC_EXTRA_CXXFLAGS", "").split()
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if (not force and os.path.basename(src) not in force_sources and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and (os.path.basename(src) != "table_cache.cpp" or os.path.getmtime(obj) > os.path.getmtime(stamp_inc))
                and all(os.path.getmtime(obj) > os.path.getmtime(os.path.join(CSRC, h))
                        for h in os.listdir(CSRC) if h.endswith((".h", ".hpp")))):
            continue
        cmd = [_hipcc(), "--offload-arch=gfx950"] + common
        if src.endswith(".hip"):
            cmd += ["-x", "hip"]
        cmd += ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, out.decode()))
        if verbose and out:
            print(out.decode(), file=sys.stderr)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    # the reference-shaped variant: only the host translation units that see the event model are compiled again (with the switch);
    # the regex compilers and every gfx950 object are the ones just linked above
    shape_objs, procs = [], []
    for src, obj in zip(sources(), objs):
        base = os.path.basename(src)
        if base not in OPTIONAL_SOURCES:
            shape_objs.append(obj)
            continue
        sobj = os.path.join(objdir, base + ".refshape.o")
        shape_objs.append(sobj)
        if (not force and os.path.exists(sobj) and os.path.getmtime(sobj) > os.path.getmtime(src)
                and all(os.path.getmtime(sobj) > os.path.getmtime(os.path.join(CSRC, h))
                        for h in os.listdir(CSRC) if h.endswith((".h", ".hpp")))):
            continue
        cmd = [_hipcc(), "--offload-arch=gfx950"] + common + ["-DLC_REFERENCE_SHAPED_EVENT_MODEL", "-c", src, "-o", sobj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed for %s (reference-shaped variant):\n%s" % (src, out.decode()))
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_REFSHAPE] + shape_objs)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
