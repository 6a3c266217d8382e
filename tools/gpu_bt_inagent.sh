cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6
g++ -O2 -std=c++17 -I include tools/inagent_bench.cpp -o scratch/inagent_bench -L loongcollector_amd/lib -llc_regex_gpu -lpthread
export LD_LIBRARY_PATH=loongcollector_amd/lib:/opt/rocm/lib
{
echo "# in-agent parse (lc_processor_process, 1000-line groups of 512 B lines, N runner threads sharing one instance): regex A on its tagged DFA,"
echo "# and regex A with an optional back-reference behind it -- (?:\\1)? -- which puts the whole pattern on the device backtracking engine"
echo "## regex A (tagged DFA)"
timeout 120 scratch/inagent_bench 128000 1000 1 16 2>&1 | grep -v amdgpu.ids
echo "## regex A(?:\\1)? (LC_ENGINE_BT)"
INAGENT_REGEX='([\d\.]+) \S+ \S+ \[(\S+) \S+\] \"(\w+) ([^\\"]*)\" ([\d\.]+) (\d+) (\d+) (\d+|-) \"([^\\"]*)\" \"([^\\"]*)\"(?:\1)?' timeout 120 scratch/inagent_bench 128000 1000 1 16 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/r6/bt_inagent.txt | cut -c1-200
