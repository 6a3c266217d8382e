#!/bin/bash
# One-off probe of the GPU box: is there a boost / Go / PCRE the oracle could be pinned to?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/probe.txt
mkdir -p $R/gpurun_out
{
echo "== date"; date -u
echo "== nproc / cpu"; nproc; lscpu | grep -E "Model name|Socket|NUMA node\(s\)|Thread|Core" 
echo "== mem"; free -g | head -2
echo "== gpu"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -12
rocm-smi --showmeminfo vram 2>/dev/null | head -8
echo "== boost headers"; ls /usr/include/boost/regex.hpp /usr/local/include/boost/regex.hpp /opt/conda/include/boost/regex.hpp 2>&1
find / -xdev \( -name "regex.hpp" -path "*boost*" -o -name "libboost_regex*" -o -name "boost-cpp*" \) 2>/dev/null | head -20
echo "== go / java / rust"; which go gofmt javac rustc cargo 2>&1; ls /usr/local/go /usr/lib/go* 2>&1 | head
find / -xdev -type d -name "regexp2*" 2>/dev/null | head
echo "== pcre"; ls -l /opt/conda/include/pcre.h /opt/conda/lib/libpcre.so* /usr/lib/x86_64-linux-gnu/libpcre* 2>&1
echo "== re2 / hyperscan / jsoncpp"; find / -xdev \( -name "re2.h" -o -name "libre2*" -o -name "hs.h" -o -name "json.h" -path "*json/*" -o -name "libjsoncpp*" \) 2>/dev/null | head
echo "== reference dir"; ls /root/reference 2>&1 | head -3
echo "== hipcc"; which hipcc; hipcc --version | head -3
echo "== pcie"; lspci 2>/dev/null | grep -i -E "amd.*(instinct|display|processing)" | head -3
rocm-smi --showbus 2>/dev/null | head -8
} > $O 2>&1
cat $O
