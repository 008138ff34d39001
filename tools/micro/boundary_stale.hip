// Does a word written by a plain store at the end of launch L read back fresh at the start of launch L+1, through each load path?
// One workgroup per counter (6980 counters, 32 to a 128-byte line, the workgroups of a line spread over the 8 XCDs), a streaming
// kernel between the launches.  build: hipcc --offload-arch=gfx950 -O3 -o boundary_stale boundary_stale.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void rmw_kernel(int* cnt, int* bad, int L, int spin, unsigned rot) {
    const int q = (int)((blockIdx.x + (unsigned)L * rot) % gridDim.x);   // rot != 0: the workgroup (and XCD) of a counter changes every launch
    int v;
    const int* p = cnt + q;
    int zero = 0;
    if (MODE == 0) {
        v = *p;                                   // uniform address, no stores before it: a scalar load
    } else if (MODE == 1) {
        asm volatile("global_load_dword %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(zero), "s"(p) : "memory");
    } else if (MODE == 2) {
        asm volatile("global_load_dword %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(zero), "s"(p) : "memory");
    } else if (MODE == 3) {
        asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)\n\tglobal_load_dword %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(zero), "s"(p) : "memory");
    } else {
        asm volatile("global_load_dword %0, %1, %2 sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(zero), "s"(p) : "memory");
    }
    v = __builtin_amdgcn_readfirstlane(v);
    if (v != L && threadIdx.x == 0) {
        const int n = atomicAdd(bad, 1);
        if (n < 8) { bad[1 + 3 * n] = q; bad[2 + 3 * n] = v; bad[3 + 3 * n] = L; }
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) cnt[q] = L + 1;
}

__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ src, float* __restrict__ dst, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 x = src[i];
        acc += x.x + x.y + x.z + x.w;
    }
    if (acc == 12345.678f) dst[blockIdx.x] = acc;
}

template <int MODE>
static void run(const char* name, int nq, int launches, size_t stream_bytes, float4* buf, float* dst, unsigned rot) {
    int *cnt, *bad;
    CK(hipMalloc(&cnt, nq * sizeof(int)));
    CK(hipMalloc(&bad, 64 * sizeof(int)));
    CK(hipMemset(cnt, 0, nq * sizeof(int)));
    CK(hipMemset(bad, 0, 64 * sizeof(int)));
    for (int L = 0; L < launches; ++L) {
        hipLaunchKernelGGL(rmw_kernel<MODE>, dim3(nq), dim3(256), 0, 0, cnt, bad, L, L % 3, rot);
        if (stream_bytes) hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, 0, buf, dst, stream_bytes / 16);
    }
    CK(hipDeviceSynchronize());
    int h[64];
    CK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-28s rot %u stream %6zu MB: %d stale reads in %d launches x %d workgroups", name, rot, stream_bytes >> 20, h[0], launches, nq);
    for (int i = 0; i < (h[0] < 4 ? h[0] : 4); ++i) printf("  [q %d read %d at launch %d]", h[1 + 3 * i], h[2 + 3 * i], h[3 + 3 * i]);
    printf("\n");
    fflush(stdout);
    CK(hipFree(cnt));
    CK(hipFree(bad));
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 4000;
    const int nq = 6980;
    float4* buf;
    float* dst;
    const size_t cap = (size_t)1 << 30;
    CK(hipMalloc(&buf, cap));
    CK(hipMalloc(&dst, 4096 * sizeof(float)));
    CK(hipMemset(buf, 0, cap));
    const size_t sizes[3] = {0, (size_t)64 << 20, cap};
    for (unsigned rot = 0; rot < 4; rot += (rot == 0 ? 1 : 2))
        for (int s = 0; s < 3; ++s) {
            run<0>("scalar load", nq, launches, sizes[s], buf, dst, rot);
            run<1>("vector load", nq, launches, sizes[s], buf, dst, rot);
            run<2>("vector load sc1", nq, launches, sizes[s], buf, dst, rot);
            run<4>("vector load sc0 sc1", nq, launches, sizes[s], buf, dst, rot);
            run<3>("buffer_inv sc1 + vector load", nq, launches, sizes[s], buf, dst, rot);
        }
    return 0;
}
