// micro-benchmark: dependent LDS read chains (t = lds[t + col]) -- how many wave-steps per cycle per CU, as a
// function of waves per CU and of the address pattern.   hipcc --offload-arch=gfx950 -O3 scratch/lds_chain_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t __attribute__((address_space(3))) * LdsWordPtr;
// modes 3..5 = mode 2 with the write under a lane mask: 3: one lane in 16 (what a capture stamp that only fires on tagged
// transitions looks like), 4: no lane at all (EXEC = 0, instruction still issued), 5: a 16-bit write from every lane
template <int MODE>  // 0: all lanes same address, 1: per-lane pseudo-random rows (36 rows x 12 cols), 2: +1 independent extra read and write per step
__global__ void chain(uint32_t* out, int steps, int tableWords) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < tableWords; i += blockDim.x) {
        // entry = byte address of a pseudo-random next row start (row = 12 words)
        uint32_t r = (i * 2654435761u >> 7) % 36u;
        lds[i] = r * 48u;
    }
    __syncthreads();
    uint32_t t = (MODE == 0) ? 0u : ((threadIdx.x * 7u) % 36u) * 48u;
    uint32_t col = (MODE == 0) ? 0u : ((threadIdx.x * 5u) % 12u) * 4u;
    uint32_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        t = *reinterpret_cast<LdsWordPtr>(t + col);
        if (MODE == 2) {
            acc += *reinterpret_cast<LdsWordPtr>(((threadIdx.x + s) & 63u) * 4u);
            *reinterpret_cast<LdsWordPtr>(2048u + threadIdx.x * 4u) = acc;
        }
        if (MODE >= 3) {
            acc += *reinterpret_cast<LdsWordPtr>(((threadIdx.x + s) & 63u) * 4u);
            if (MODE == 5) {
                *reinterpret_cast<uint16_t __attribute__((address_space(3)))*>(2048u + threadIdx.x * 2u) = uint16_t(acc);
            } else {
                const bool on = MODE == 3 ? ((threadIdx.x + s) & 15u) == 0u : (acc == 0x7fffffffu && s < 0);
                // v_cmpx-style predication: the store is issued every step, under a (mostly empty) lane mask
                const unsigned long long mask = __ballot(on);
                asm volatile("s_mov_b64 exec, %0\n\tds_write_b32 %1, %2\n\ts_mov_b64 exec, -1" ::"s"(mask), "v"(2048u + threadIdx.x * 4u), "v"(acc) : "memory");
            }
        }
        if (MODE != 0) col = (col + 4u) % 48u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = t + acc;
}
template <int MODE>
void run(int blocksPerCu, int threads, int ldsBytes) {
    uint32_t* d;
    int grid = 256 * blocksPerCu;
    hipMalloc(&d, size_t(grid) * threads * 4);
    int steps = 4096;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(chain<MODE>, dim3(grid), dim3(threads), ldsBytes, 0, d, steps, 36 * 12);
    hipEventRecord(a);
    hipLaunchKernelGGL(chain<MODE>, dim3(grid), dim3(threads), ldsBytes, 0, d, steps, 36 * 12);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double waveSteps = double(grid) * (threads / 64) * steps / 256.0;  // per CU
    double cycles = ms * 1e-3 * 2.4e9;
    printf("mode %d  waves/CU %3d  ms %.3f  cycles per wave-step per CU %.2f  (per-wave step latency ~%.0f cyc)\n", MODE,
           blocksPerCu * threads / 64, ms, cycles / waveSteps, cycles / steps);
    hipFree(d);
}
int main() {
    for (int bpc : {1, 2, 4, 8}) run<0>(bpc, 256, 20000);
    for (int bpc : {1, 2, 4, 8}) run<1>(bpc, 256, 20000);
    for (int bpc : {1, 2, 4, 8}) run<2>(bpc, 256, 20000);
    for (int bpc : {2, 4, 8}) run<3>(bpc, 256, 20000);
    for (int bpc : {2, 4, 8}) run<4>(bpc, 256, 20000);
    for (int bpc : {2, 4, 8}) run<5>(bpc, 256, 20000);
    return 0;
}
