// LDS read-modify-write microbenchmark (gfx950): what one CU sustains for the access pattern of the sparse scorer.
//
// sparse_score_kernel applies a posting as  sc[doc - doc0] += w * v  on a 32 KB LDS score tile: a ds_read_b32 and a
// ds_write_b32 at an address that is random with respect to the banks.  This program measures the chip-wide rate of
// exactly that pattern with NO global-memory traffic (indices come from a per-thread LCG), four independent
// read-modify-writes in flight per thread like the kernel's groups, for 1..4 workgroups of 256 threads per CU, plus:
//   seq   the same loop on consecutive addresses (conflict-free ceiling of the instruction pair)
//   b128  random 16-byte slots of a [2048][4] tile with ds_read_b128 / ds_write_b128 (four queries per posting)
// It prints one JSON line; bench.py reads the committed copy (profiles/r02_lds_rmw.json) as the `peak` of the
// sparse scorer's LDS roofline.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_rmw.hip -o /tmp/lds_rmw && /tmp/lds_rmw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define TILE 8192
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: random b32, 1: sequential b32, 2: random b128
template <int MODE>
__global__ __launch_bounds__(256) void rmw_kernel(int iters, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) float sc[TILE];
    const int tid = threadIdx.x;
    for (int d = tid; d < TILE; d += 256) sc[d] = 0.f;
    __syncthreads();
    unsigned s = (blockIdx.x * 256u + tid) * 2654435761u + 12345u;
    const float w = 1.0001f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 2) {
            int d[4];
            f32x4 cur[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s = s * 1664525u + 1013904223u;
                d[u] = (s >> 10) & (TILE / 4 - 1);
                cur[u] = *reinterpret_cast<const f32x4*>(&sc[d[u] * 4]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                cur[u] += w;
                *reinterpret_cast<f32x4*>(&sc[d[u] * 4]) = cur[u];
            }
        } else {
            int d[4];
            float cur[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (MODE == 0) {
                    s = s * 1664525u + 1013904223u;
                    d[u] = (s >> 10) & (TILE - 1);
                } else {
                    d[u] = (it * 1024 + u * 256 + tid) & (TILE - 1);
                }
                cur[u] = sc[d[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) sc[d[u]] = cur[u] + w;
        }
    }
    __syncthreads();
    float acc = 0.f;
    for (int d = tid; d < TILE; d += 256) acc += sc[d];
    if (acc == -1.f) sink[blockIdx.x] = acc;   // keeps the loop alive
}

template <int MODE>
static double run(int wg_per_cu, int n_cu, int iters, float* sink) {
    const int grid = wg_per_cu * n_cu;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(rmw_kernel<MODE>, dim3(grid), dim3(256), 0, 0, iters / 8, sink);   // warm-up
    CHECK(hipDeviceSynchronize());
    double best = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(rmw_kernel<MODE>, dim3(grid), dim3(256), 0, 0, iters, sink);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        const double rate = (double)grid * 256.0 * 4.0 * iters / (ms * 1e-3);
        best = rate > best ? rate : best;
    }
    return best;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    float* sink;
    CHECK(hipMalloc(&sink, 4 * 4096));
    const int iters = 200000;
    printf("{\"device\": \"%s\", \"cus\": %d, \"tile_bytes\": %d, \"unit\": \"LDS read-modify-writes per second, chip-wide\"", p.gcnArchName, n_cu, TILE * 4);
    for (int wg = 1; wg <= 4; ++wg) {
        printf(", \"random_b32_wg%d\": %.4e", wg, run<0>(wg, n_cu, iters, sink));
        fflush(stdout);
    }
    printf(", \"sequential_b32_wg4\": %.4e", run<1>(4, n_cu, iters, sink));
    for (int wg = 1; wg <= 4; wg *= 2) printf(", \"random_b128_slots_wg%d\": %.4e", wg, run<2>(wg, n_cu, iters, sink));
    printf("}\n");
    return 0;
}
