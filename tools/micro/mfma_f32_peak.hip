// Microbenchmark: sustained rate of the fp32 MFMAs from registers (no memory traffic), random data:
// v_mfma_f32_32x32x2_f32 (64 cycles, 4096 FLOP) and v_mfma_f32_16x16x4_f32 (32 cycles, 2048 FLOP) - the same nominal
// FLOP/cycle; what differs is the clock the chip holds under each.
// hipcc --offload-arch=gfx950 -O3 -o mfma_f32_peak mfma_f32_peak.hip && ./mfma_f32_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512, 2) void k32(float* out, const float* in, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = in[threadIdx.x], b = in[threadIdx.x + 512];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-9f;
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(512, 2) void k16(float* out, const float* in, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = in[threadIdx.x], b = in[threadIdx.x + 512];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        a += 1e-9f;
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)
template <typename F>
static int sustained(const char* name, F launch, double flop_per_launch) {
    // the chip lowers its clock under a long MFMA load: report the first launches and the rate after ~2 s of them
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int phase = 0; phase < 2; ++phase) {
        const int n = phase == 0 ? 2 : 150;
        CK(hipEventRecord(e0));
        for (int i = 0; i < n; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        if (phase == 0) { printf("%s first 2 launches: %.1f TFLOP/s\n", name, 2 * flop_per_launch / ms / 1e9); continue; }
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s after %d launches back to back: %.1f TFLOP/s (%.2f ms each)\n", name, n, 20 * flop_per_launch / ms / 1e9, ms / 20);
    }
    return 0;
}
int main() {
    float *out, *in; CK(hipMalloc(&out, 4 * 512 * 2048)); CK(hipMalloc(&in, 4 * 1024));
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (rand() / (float)RAND_MAX) - 0.5f;
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    const int iters = 20000, blocks = 512;
    for (int rep = 0; rep < 2; ++rep) {
        if (sustained("32x32x2 (8 acc) ", [&] { hipLaunchKernelGGL(k32<8>, dim3(blocks), dim3(512), 0, 0, out, in, iters); },
                      (double)blocks * 8 * iters * 8 * 4096.0)) return 1;
        if (sustained("16x16x4 (32 acc)", [&] { hipLaunchKernelGGL(k16<32>, dim3(blocks), dim3(512), 0, 0, out, in, iters / 2); },
                      (double)blocks * 8 * (iters / 2) * 32 * 2048.0)) return 1;
    }
    return 0;
}
