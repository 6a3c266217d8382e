"""The GPU-only scripts cannot run in the CPU suite; they can at least be parsed: a syntax error in bench.py or in a tool it spawns
(tools/grok_config2.py) would cost the driver's one bench line."""
import glob
import os
import py_compile
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_script_compiles():
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] + sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")))
    assert len(files) > 10
    for f in files:
        py_compile.compile(f, doraise=True)


def test_bench_and_its_subprocess_parse_their_arguments():
    for script in ("bench.py", os.path.join("tools", "grok_config2.py"), os.path.join("tools", "grok_bench.py"),
                   os.path.join("tools", "grok_inagent_bench.py")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, script), "--help"], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and "usage" in out.stdout.lower(), (script, out.stderr[-400:])
