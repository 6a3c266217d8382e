#!/bin/bash
# tools/asan_check.sh: the host half of the library (regex parser, follow-NFA / tagged-DFA builders, table packers, Grok
# library, processors) rebuilt with AddressSanitizer + UBSan and the CPU test-suite run against it.  No GPU needed.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
CL=/opt/rocm/lib/llvm/bin/clang++
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
OUT=${TMPDIR:-/tmp}/lc_asan; mkdir -p $OUT
cd $R/loongcollector_amd
python -m loongcollector_amd.build > /dev/null 2>&1 || (cd $R && python -m loongcollector_amd.build)
rm -f $OUT/*.o
for s in csrc/*.cpp; do
  $CL -std=c++17 -O1 -g -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -I ../include -I csrc -I lib/obj -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -c $s -o $OUT/$(basename $s).o &
done; wait
$CL -shared -fPIC -fsanitize=address,undefined -shared-libsan -o $OUT/liblc_asan.so $OUT/*.o lib/obj/*.hip.o -L/opt/rocm/lib -lamdhip64
cd $R
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 LC_REGEX_GPU_LIB=$OUT/liblc_asan.so python -m pytest tests -q -m "not gpu" \
  --deselect tests/test_shard_gloo.py "$@"   # (spawned gloo ranks would need the preload too)
