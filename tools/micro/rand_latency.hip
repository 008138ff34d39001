// Dependent random-read latency over footprints of different sizes (gfx950): separates plain HBM latency from address-
// translation misses.  One wave per workgroup, `waves` workgroups; every lane chases its own chain
//   i = (buf[i] ^ salt) % n   for `iters` steps, buf[i] = hash(i),
// so each step is one dependent 4-byte load at an unpredictable address.  Prints ns per step per footprint.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/rand_latency.hip -o /tmp/rand_latency && /tmp/rand_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill(unsigned* buf, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        buf[i] = (unsigned)(i * 2654435761ull >> 7) ^ (unsigned)(i >> 3);
}
// lanes_active: 1 = one lane per wave chases (one line per step), 64 = every lane its own line
__global__ __launch_bounds__(64) void chase(const unsigned* __restrict__ buf, size_t n, int iters, int lanes_active, unsigned* sink) {
    const int lane = threadIdx.x;
    if (lane >= lanes_active) return;
    size_t i = ((size_t)blockIdx.x * 64 + lane) * 7919u % n;
    unsigned acc = 0;
    for (int k = 0; k < iters; ++k) {
        const unsigned v = buf[i];
        acc ^= v;
        i = ((size_t)v * 40503u + (size_t)k * 977u + blockIdx.x) % n;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const size_t sizes_mb[] = {16, 256, 2048, 4608, 9216, 32768};
    unsigned* sink;
    CHECK(hipMalloc(&sink, 4));
    printf("{");
    for (size_t smb : sizes_mb) {
        const size_t n = smb * 1024 * 1024 / 4;
        unsigned* buf;
        CHECK(hipMalloc(&buf, n * 4));
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, buf, n);
        CHECK(hipDeviceSynchronize());
        for (int lanes : {1, 64}) {
            for (int waves : {256, 2048}) {
                const int iters = 2000;
                hipEvent_t e0, e1;
                CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                hipLaunchKernelGGL(chase, dim3(waves), dim3(64), 0, 0, buf, n, 200, lanes, sink);
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(chase, dim3(waves), dim3(64), 0, 0, buf, n, iters, lanes, sink);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                printf("\"%zuMB_lanes%d_waves%d_ns_per_step\": %.1f, ", smb, lanes, waves, ms * 1e6 / iters);
            }
        }
        CHECK(hipFree(buf));
    }
    printf("\"note\": \"dependent random 4-byte loads, one chain per active lane\"}\n");
    return 0;
}
