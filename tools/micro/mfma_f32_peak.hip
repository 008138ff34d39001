// Microbenchmark: sustained rate of v_mfma_f32_32x32x2_f32 from registers (no memory traffic), random data.
// hipcc --offload-arch=gfx950 -O3 -o mfma_f32_peak mfma_f32_peak.hip && ./mfma_f32_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512, 2) void k(float* out, const float* in, int iters) {
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
int main() {
    float *out, *in; hipMalloc(&out, 4 * 512 * 2048); hipMalloc(&in, 4 * 1024);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (rand() / (float)RAND_MAX) - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512}) {
        const int iters = 20000;
        hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(512), 0, 0, out, in, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(512), 0, 0, out, in, iters); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)blocks * 8 /*waves*/ * iters * 8 /*acc*/ * 4096.0;
        printf("blocks=%d (8 waves each, 8 accumulators): %.1f TFLOP/s  (%.2f ms)\n", blocks, flop / ms / 1e9, ms);
    }
    return 0;
}
