"""The drop-in boundary (SURVEY.md section 8 b) driven by the REFERENCE's own code.

oracle/_ref/libref_processor.so holds, compiled from /root/reference: common/DynamicLibHelper.cpp (DynamicLibLoader), collection_pipeline/
plugin/creator/DynamicCProcessorCreator.cpp, plugin/processor/DynamicCProcessorProxy.cpp and collection_pipeline/plugin/instance/
ProcessorInstance.cpp -- everything between the plugin registry and a dynamic plugin's three function pointers (the registry itself
registers every plugin of the agent and is not compiled; its directory rule and its symbol / version check, PluginRegistry.cpp:239-243,
270-290, are restated in the harness).  The PLUGIN is the product's dlsym slot in the form an agent build takes: csrc/c_processor_slot.cpp +
csrc/processor_parse_regex_gpu.cpp compiled with LC_USE_REFERENCE_HEADERS against the reference's headers -- init() reads the Json::Value the
proxy hands over, keeps the CollectionPipelineContext, process() works on the reference's own PipelineEventGroup -- with the five device
calls answered by the CPU oracle (tests/native/host_double.cpp, HD_DOUBLES_ONLY): there is no GPU here, and the device side is the
-m gpu tests' business.

Checked: where the loader looks for the file; the version check; name(); init with a good and with a refused config (and the teardown
behind a refused one); the same event groups through the reference's own processor_parse_regex_native and through the plugin, both wrapped
in the reference's ProcessorInstance: the same events and the same in / out event and byte counters; alarms arriving at the agent's
AlarmManager.  CPU only; skipped where the reference tree is not present (the GPU box)."""
import ctypes
import json
import os
import random
import shutil
import subprocess

import pytest

from test_reference_neighbours import RefPlugin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/core"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (/root/reference): the slot's code is compiled from there")

PLUGIN_TYPE = "processor_parse_regex_gpu"


@pytest.fixture(scope="module")
def agent():
    """-> (the reference library with the slot harness bound, the "process execution dir" the plugin was installed under)"""
    L = RefPlugin.lib()
    vp, cp, sz = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t
    L.refp_dyn_load.restype = vp
    L.refp_dyn_load.argtypes = [cp, cp, cp, cp, sz]
    L.refp_static_instance.restype = vp
    L.refp_static_instance.argtypes = [cp, cp, sz]
    L.refp_dyn_process_json.restype = vp
    L.refp_dyn_process_json.argtypes = [vp, cp, cp, sz]
    L.refp_dyn_counters_json.restype = vp
    L.refp_dyn_counters_json.argtypes = [vp]
    L.refp_dyn_name.restype = cp
    L.refp_dyn_name.argtypes = [vp]
    L.refp_dyn_unload.argtypes = [vp]
    exec_dir = os.path.join(ROOT, "tests", "_build", "agent")
    os.makedirs(exec_dir, exist_ok=True)
    # PluginRegistry.cpp:239-243 + DynamicLibHelper.cpp:74: dlopen(GetProcessExecutionDir() + "/plugins" + "lib" + type + ".so") -- the
    # directory is used as a PREFIX, there is no '/' behind it: the file the agent opens is "<exec dir>/pluginslib<type>.so"
    so = os.path.join(exec_dir, "pluginslib%s.so" % PLUGIN_TYPE)
    csrc = os.path.join(ROOT, "loongcollector_amd", "csrc")
    srcs = [os.path.join(csrc, "c_processor_slot.cpp"), os.path.join(csrc, "processor_parse_regex_gpu.cpp"), os.path.join(ROOT, "tests", "native", "host_double.cpp")]
    ref_lib = os.path.join(ROOT, "oracle", "_ref")
    deps = srcs + [os.path.join(csrc, "processor_parse_regex_gpu.hpp"), os.path.join(ref_lib, "libref_processor.so"), os.path.join(ref_lib, "libref_models.so")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        from loongcollector_amd import build as native_build
        objdir = os.path.join(ROOT, "loongcollector_amd", "lib", "obj")
        if not os.path.exists(os.path.join(objdir, "grok_defaults.inc")):
            native_build.build_native()
        # (undefined symbols are the agent's: AlarmManager, AppConfig, the context, the event model -- here libref_processor / libref_models)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-g", "-fPIC", "-shared", "-w", "-DLC_USE_REFERENCE_HEADERS", "-DHD_DOUBLES_ONLY",
                               "-include", "set", "-include", "memory", "-I", os.path.join(ROOT, "oracle", "ref_processor", "stubs"), "-I", os.path.join(ROOT, "oracle"),
                               "-I", os.path.join(ROOT, "tests", "refhdr"), "-I", REF, "-I", os.path.join(REF, "config"), "-I", os.path.join(ROOT, "include"),
                               "-I", csrc, "-I", objdir, "-o", so] + srcs +
                              ["-L" + os.path.join(ROOT, "oracle"), "-loracle", "-L" + ref_lib, "-lref_processor", "-lref_models",
                               "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath," + ref_lib, "-Wl,-z,defs"])
    return L, exec_dir


class Slot:
    def __init__(self, L, h):
        self.L, self.h = L, h

    @classmethod
    def dynamic(cls, agent, config, exec_dir=None, plugin_type=PLUGIN_TYPE):
        L, default_dir = agent
        err = ctypes.create_string_buffer(1024)
        h = L.refp_dyn_load((exec_dir or default_dir).encode(), plugin_type.encode(), json.dumps(config).encode(), err, 1024)
        if not h:
            raise ValueError(err.value.decode("utf-8", "replace"))
        return cls(L, h)

    @classmethod
    def static(cls, agent, config):
        L, _ = agent
        err = ctypes.create_string_buffer(1024)
        h = L.refp_static_instance(json.dumps(config).encode(), err, 1024)
        if not h:
            raise ValueError(err.value.decode("utf-8", "replace"))
        return cls(L, h)

    def name(self):
        return self.L.refp_dyn_name(self.h).decode()

    def process(self, fixture):
        err = ctypes.create_string_buffer(512)
        p = self.L.refp_dyn_process_json(self.h, json.dumps(fixture).encode(), err, 512)
        assert p, err.value
        try:
            d = json.loads(ctypes.string_at(p).decode("utf-8"), object_pairs_hook=list)
        finally:
            self.L.refp_free(p)
        out = []
        for ev in dict(d or []).get("events", []):
            ev = dict(ev)
            ev["contents"] = [tuple(kv) for kv in ev.get("contents", [])]
            out.append(sorted(ev.items(), key=lambda kv: kv[0]))
        return out

    def counters(self):
        p = self.L.refp_dyn_counters_json(self.h)
        try:
            c = json.loads(ctypes.string_at(p).decode())
        finally:
            self.L.refp_free(p)
        c.pop("total_process_time_ms", None)
        return c

    def unload(self):
        if self.h:
            self.L.refp_dyn_unload(self.h)   # ~DynamicCProcessorProxy: finalize(plugin_state); ~DynamicCProcessorCreator: CloseLib
            self.h = None

    def alarms(self):
        self.L.refp_take_alarms.restype = ctypes.c_void_p
        p = self.L.refp_take_alarms()
        try:
            return ctypes.string_at(p).decode("utf-8", "replace")
        finally:
            self.L.refp_free(p)


CONFIG = {"SourceKey": "content", "Regex": r"(\w+)\t(\w+).*", "Keys": ["key1", "key2"]}


def test_the_reference_s_loader_finds_checks_and_names_the_plugin(agent, tmp_path):
    L, exec_dir = agent
    for d in (exec_dir, exec_dir + "/"):            # (the agent's own value ends in '/': GetProcessExecutionDir() + "crash_dump.dmp" elsewhere)
        s = Slot.dynamic(agent, CONFIG, exec_dir=d)
        assert s.name() == PLUGIN_TYPE              # processor_interface.name: the Type a pipeline config names
        got = s.process({"events": [{"contents": {"content": "value1\tvalue2"}, "timestamp": 1, "type": 1}]})
        assert dict(got[0])["contents"] == [("key1", "value1"), ("key2", "value2")]
        s.unload()
    # a plugins/ DIRECTORY is not where the loader looks (no '/' between the directory and "lib", PluginRegistry.cpp:239-243)
    wrong = tmp_path / "exec"
    (wrong / "plugins").mkdir(parents=True)
    shutil.copy(os.path.join(exec_dir, "pluginslib%s.so" % PLUGIN_TYPE), str(wrong / "plugins" / ("lib%s.so" % PLUGIN_TYPE)))
    with pytest.raises(ValueError, match="open plugin"):
        Slot.dynamic(agent, CONFIG, exec_dir=str(wrong))
    # ... and a library without the data symbol, or of another interface version, is turned away by the checks of LoadProcessorPlugin
    nosym = tmp_path / "nosym"
    nosym.mkdir()
    src = nosym / "x.c"
    src.write_text("int nothing_here = 1;\n")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(nosym / "pluginslibempty.so"), str(src)])
    with pytest.raises(ValueError, match="load method"):
        Slot.dynamic(agent, CONFIG, exec_dir=str(nosym), plugin_type="empty")
    src.write_text("struct { int version; const char* name; const char* language; void* a; void* b; void* c; } processor_interface = {99, \"old\", \"c\", 0, 0, 0};\n")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(nosym / "pluginslibold.so"), str(src)])
    with pytest.raises(ValueError, match="version mismatch: expected 100, actual 99"):
        Slot.dynamic(agent, CONFIG, exec_dir=str(nosym), plugin_type="old")


def test_a_refused_config_fails_init_and_survives_the_proxy_s_teardown(agent):
    """init() != 0 -> ProcessorInstance::Init false -> the pipeline is not built (CollectionPipeline.cpp:142-144).  The proxy's destructor
    calls finalize(plugin_state) all the same (DynamicCProcessorProxy.cpp:25-28) on an instance it never initialised (:21-23): the slot sets
    plugin_state to null before anything can fail"""
    for bad in ({"SourceKey": "content"}, {"SourceKey": "content", "Regex": "(", "Keys": ["a"]}, {"Regex": "(a)", "Keys": []}):
        with pytest.raises(ValueError, match="Init returned false"):
            Slot.dynamic(agent, bad)
    Slot.dynamic(agent, CONFIG).unload()            # (the library is still good for a new instance)


def _groups(rng, n_groups):
    lines = ["GET\t200 rest", "POST\t404", "nomatch", "", "a\tb", "x\ty z\tw", "tab\t", "\tlead", "ünï\tcode é"]
    for _ in range(n_groups):
        events = []
        for k in range(rng.randint(0, 12)):
            kind = rng.random()
            contents = [["content", rng.choice(lines)]] if kind < 0.8 else [["other", "v"]] if kind < 0.9 else [["content", rng.choice(lines)], ["extra", "1"]]
            ev = {"contents": contents, "timestamp": 1700000000 + k, "type": 1}
            if rng.random() < 0.5:
                ev["timestampNanosecond"] = k
            if rng.random() < 0.3:
                ev["fileOffset"], ev["rawSize"] = 100 * k, 17
            events.append(ev)
        g = {"events": events}
        if rng.random() < 0.5:
            g["tags"] = {"host": "h1"}
        if rng.random() < 0.3:
            g["metadata"] = {"log.file.offset": "__file_offset__"}
        yield g


@pytest.mark.parametrize("options", [
    {},
    {"KeepingSourceWhenParseFail": True, "KeepingSourceWhenParseSucceed": True, "RenamedSourceKey": "raw"},
    {"KeepingSourceWhenParseFail": True, "CopingRawLog": True},
    {"KeepingSourceWhenParseSucceed": True},
    {"KeepingSourceWhenParseFail": True, "Keys": ["key1", "key2", "key3"]},        # more keys than groups
])
def test_the_plugin_beside_the_static_processor_in_the_reference_s_processor_instance(agent, options):
    """ProcessorInstance::Process (ProcessorInstance.cpp:46-63) around the reference's own processor_parse_regex_native and around the
    dynamic plugin: the same events (every field the fixture writer prints, contents in order) and the same in / out event and byte
    counters -- what the pipeline and the self-monitor see does not change when only `Type` changes"""
    config = dict(CONFIG, **options)
    dyn, sta = Slot.dynamic(agent, config), Slot.static(agent, config)
    dyn.alarms()
    rng = random.Random(11)
    n = 0
    for g in _groups(rng, 150):
        a = dyn.process(g)
        alarms_dyn = dyn.alarms()
        b = sta.process(g)
        alarms_sta = sta.alarms()
        assert a == b, g
        assert alarms_dyn == alarms_sta, g          # (REGEX_MATCH_ALARM through the context the proxy handed to init)
        n += len(a)
    assert n > 300
    cd, cs = dyn.counters(), sta.counters()
    # (the static processor's own four counters hang on the Plugin base's MetricsRecordRef, which the C ABI does not reach -- SURVEY 8b;
    # the plugin keeps its equivalents in plugin_state: lc_processor_counters)
    instance = ("in_events_total", "out_events_total", "in_size_bytes", "out_size_bytes")
    assert sorted(cd) == sorted(instance)
    assert cd == {k: cs[k] for k in instance} and cd["in_size_bytes"] > 0
    assert cd["in_events_total"] == cd["out_events_total"] if options.get("KeepingSourceWhenParseFail") else cd["in_events_total"] > cd["out_events_total"]
    dyn.unload()
    sta.unload()


def test_generated_parser_configs_through_the_slot_beside_the_static_processor(agent):
    """Generated configs -- regexes with groups that may not take part, none, or whole-line mode; fewer / more Keys than groups; Keys that
    collide with the source key, the renamed source key and the legacy raw-log keys; every option on and off; other source keys -- accepted
    or refused alike by the plugin's init (through the reference's proxy) and the reference's own Init, and on accepted ones: the same events,
    the same alarms, the same instance counters.  (A 10 000-group run of the same generator: no difference.)"""
    rng = random.Random(1)
    regexes = [(r"(\w+)\t(\w+).*", 2), (r"(\w+) (\d{3}) (.*)", 3), (r"(\S+)", 1), ("(.*)", 1), (r"(a)|(b)", 2), (r"(\w+)\t?(\w*)", 2), (r"no groups", 0),
               (r"(?:x)(\d+)y", 1)]
    key_pool = ["k1", "k2", "k3", "content", "raw", "__raw_log__", "__raw__", "other"]
    lines = ["GET\t200 rest", "POST 404 ua", "nomatch", "", "a\tb", "x\ty z", "a", "b", "x12y", "no groups", "GET 200 curl", "ünï\tcödé"]
    groups = refused = 0
    for trial in range(300):
        rx, ngroups = rng.choice(regexes)
        nkeys = rng.choice([ngroups, ngroups, ngroups, max(0, ngroups - 1), ngroups + 1, 0])
        config = {"SourceKey": rng.choice(["content", "content", "other"]), "Regex": rx, "Keys": [rng.choice(key_pool) for _ in range(nkeys)]}
        for opt in ("KeepingSourceWhenParseFail", "KeepingSourceWhenParseSucceed", "CopingRawLog"):
            if rng.random() < 0.5:
                config[opt] = rng.random() < 0.7
        if rng.random() < 0.4:
            config["RenamedSourceKey"] = rng.choice(key_pool + [""])
        try:
            sta = Slot.static(agent, config)
        except ValueError:
            sta = None
        try:
            dyn = Slot.dynamic(agent, config)
        except ValueError:
            dyn = None
        assert (sta is None) == (dyn is None), (config, "the reference refuses" if sta is None else "the reference accepts")
        if sta is None:
            refused += 1
            continue
        dyn.alarms()
        for _ in range(4):
            events = []
            for k in range(rng.randint(0, 10)):
                kind = rng.random()
                contents = [[config["SourceKey"], rng.choice(lines)]]
                if kind > 0.8:
                    contents.append([rng.choice(key_pool), "pre-existing"])
                if kind > 0.95:
                    contents = [["elsewhere", "v"]]
                events.append({"contents": contents, "timestamp": 1 + k, "type": 1})
            g = {"events": events}
            if rng.random() < 0.3:
                g["metadata"] = {"log.file.offset": "__file_offset__"}
            a = dyn.process(g)
            alarms_dyn = dyn.alarms()
            b = sta.process(g)
            assert a == b, (config, g)
            assert alarms_dyn == sta.alarms(), (config, g)
            groups += 1
        cd, cs = dyn.counters(), sta.counters()
        assert cd == {k: cs[k] for k in cd}, config
        dyn.unload()
        sta.unload()
    assert groups > 800 and refused > 50
