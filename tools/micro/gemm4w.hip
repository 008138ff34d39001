// Prototype k-loop (round 3 -> round 4 direction): C[M, N] = A[M, K] . B[N, K]^T in bf16 on 256 x 256 workgroup tiles with FOUR waves,
// each owning 128 x 128 (v_mfma_f32_32x32x16_bf16, 256 accumulator registers per lane), against the shipped kernels' EIGHT waves of
// 128 x 64.  Per k-step of 64 a CU then reads 128 KB of fragments out of LDS instead of 192 KB (+ 64 KB of LDS-DMA writes either
// way): at the MFMA peak the 8-wave layout needs 2048 LDS cycles per 2062 MFMA cycles - both pipes saturate together, which is the
// ~0.5 ceiling every LDS-staged loop of this repository sits at - the 4-wave layout 1536.
// Measured (MI355X, random operands, bf16 output; variant 0 = minimal two-phase loop, 1 = fragments of the next sub-step read under
// the MFMAs, one barrier per k-step, issue order pinned):
//   16384 x 8192 x 2048   0.322 / 0.344 of 2.5 PFLOP/s      61440 x 6144 x 2048   0.370 / 0.396
//    8192 x 8192 x 16384  0.388 / 0.407   (32 k-steps per tile -> 256: prologue and epilogue amortised, this is the k-loop)
// i.e. as written the 4-wave loop reaches 0.41 where the shipped 8-wave loops reach 0.51: with ONE wave per SIMD every wait for an
// LDS-DMA piece or at the k-step barrier idles that SIMD's MFMA pipe, which the second resident wave of the 8-wave layout covers.
// Where the time goes (8192 x 8192 x 16384; variants 2-5 are timing diagnostics with wrong results):
//   4: the MFMAs alone                                   0.758   <- what this chip sustains on the 32x32x16 pipe at all
//   5: + the fragment reads out of LDS                   0.622
//   3: + the LDS-DMA of the next k-step (no waits)       0.47
//   2: + the k-step barrier                              0.41    (1: + the wait for the LDS-DMA pieces: no further change)
// so the barrier costs 0.06, the LDS-DMA stream 0.15 and the fragment reads 0.14 - none of it waiting for memory.  The conflict-free
// swizzle for 128-byte rows ((row >> 1) & 7 instead of row & 7) changes nothing: bank conflicts are not what the reads cost.
// Lower LDS traffic alone does not pay; with one wave per SIMD the loop needs one operand to bypass LDS (direct global -> VGPR
// fragments) or a second resident wave - round 4 material.
// build: hipcc --offload-arch=gfx950 -O3 -o gemm4w gemm4w.hip ; run: ./gemm4w [M N K [variant]]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;      // 64 KB
#ifndef SWZ
#define SWZ(row) (((row) >> 1) & 7)      // 128-byte rows: a 16-lane group of a ds_read_b128 (16 consecutive rows, one k-chunk) then covers all 16 bank groups;
#endif                                   // (row & 7) leaves rows r and r + 8 on the same one (2-way conflict)

template <int VARIANT>
__global__ __launch_bounds__(256, 1) void gemm4w_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B,
                                                        unsigned short* __restrict__ C, int M, int N, int K, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                       // 2 x 2 waves of 128 x 128
    // XCD-aware tile order: consecutive workgroups of one XCD share a row panel of A
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
    const int swz = (nwg % 8 == 0) ? (bid % 8) * (nwg / 8) + bid / 8 : bid;
    const int tm = swz / tiles_n, tn = swz % tiles_n;
    const int64_t a_base = (int64_t)tm * BM * K, b_base = (int64_t)tn * BN * K;

    const int srow = lane >> 3;
    int aoff[8], boff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = wave * 64 + i * 8 + srow;
        const int schunk = (lane & 7) ^ SWZ(r);
        aoff[i] = r * K + schunk * 8;
        boff[i] = r * K + schunk * 8;
    }
    auto stage = [&](int st, int k0) {
        unsigned char* ab = smem + st * STAGE_BYTES + (wave * 64) * 128;
        unsigned char* bb = smem + st * STAGE_BYTES + BM * 128 + (wave * 64) * 128;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(A + a_base + k0 + aoff[i]), (lds_void_ptr)(ab + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(B + b_base + k0 + boff[i]), (lds_void_ptr)(bb + i * 1024), 16, 0, 0);
    };
    const int frow = lane & 31, fk = lane >> 5;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / BK;
    if constexpr (VARIANT == 0) {
        // the minimal two-phase form: wait, barrier, issue the next stage, then reads and MFMAs as the compiler orders them
        stage(0, 0);
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nk) stage(buf ^ 1, (kt + 1) * BK);
            const unsigned char* at = smem + buf * STAGE_BYTES;
            const unsigned char* bt = at + BM * 128;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {                           // four k16 sub-steps
                bf16x8 af[4], bfr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ra = wm * 128 + i * 32 + frow, rb = wn * 128 + i * 32 + frow;
                    af[i] = *reinterpret_cast<const bf16x8*>(at + ra * 128 + (((kk * 2 + fk) ^ SWZ(ra)) * 16));
                    bfr[i] = *reinterpret_cast<const bf16x8*>(bt + rb * 128 + (((kk * 2 + fk) ^ SWZ(rb)) * 16));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
            buf ^= 1;
        }
    } else {
        // software-pipelined: the fragments of sub-step s + 1 are read under the MFMAs of sub-step s (two register sets), one
        // barrier per k-step in front of its last sub-step, the LDS-DMA of k-step kt + 2 issued right behind it; issue order pinned
        bf16x8 fa[2][4], fb[2][4];
        auto load_frags = [&](int st, int kk, bf16x8 (&xa)[4], bf16x8 (&xb)[4]) {
            const unsigned char* at = smem + st * STAGE_BYTES;
            const unsigned char* bt = at + BM * 128;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wm * 128 + i * 32 + frow, rb = wn * 128 + i * 32 + frow;
                xa[i] = *reinterpret_cast<const bf16x8*>(at + ra * 128 + (((kk * 2 + fk) ^ SWZ(ra)) * 16));
                xb[i] = *reinterpret_cast<const bf16x8*>(bt + rb * 128 + (((kk * 2 + fk) ^ SWZ(rb)) * 16));
            }
        };
#define MMA_ALL(SET)                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                     \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SET][i], fb[SET][j], acc[i][j], 0, 0, 0);
#define SGB(MASK, CNT, ID) __builtin_amdgcn_sched_group_barrier(MASK, CNT, ID)
        stage(0, 0);
        if (nk > 1) stage(1, BK);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");           // stage 0 landed (this wave's pieces), stage 1 in flight
        __syncthreads();
        load_frags(0, 0, fa[0], fb[0]);
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if constexpr (VARIANT >= 4) {          // timing diagnostics (wrong results): 4 = MFMAs only, 5 = MFMAs + fragment reads, no LDS-DMA
                if constexpr (VARIANT == 5) load_frags(buf, 1, fa[1], fb[1]);
                MMA_ALL(0)
                if constexpr (VARIANT == 5) load_frags(buf, 2, fa[0], fb[0]);
                MMA_ALL(1)
                if constexpr (VARIANT == 5) load_frags(buf, 3, fa[1], fb[1]);
                MMA_ALL(0)
                if constexpr (VARIANT == 5) load_frags(buf ^ 1, 0, fa[0], fb[0]);
                MMA_ALL(1)
                buf ^= 1;
                continue;
            }
            load_frags(buf, 1, fa[1], fb[1]);
            MMA_ALL(0)
#pragma unroll
            for (int i = 0; i < 8; ++i) { SGB(0x008, 1, 0); SGB(0x100, 1, 0); }
            SGB(0x008, 8, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_frags(buf, 2, fa[0], fb[0]);
            MMA_ALL(1)
#pragma unroll
            for (int i = 0; i < 8; ++i) { SGB(0x008, 1, 1); SGB(0x100, 1, 1); }
            SGB(0x008, 8, 1);
            __builtin_amdgcn_sched_barrier(0);
            load_frags(buf, 3, fa[1], fb[1]);
            MMA_ALL(0)
#pragma unroll
            for (int i = 0; i < 8; ++i) { SGB(0x008, 1, 2); SGB(0x100, 1, 2); }
            SGB(0x008, 8, 2);
            __builtin_amdgcn_sched_barrier(0);
            // k-step kt + 1 has landed (every wave waits for its own pieces, then the barrier); all reads of `buf` are done
            if constexpr (VARIANT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // 2, 3: timing diagnostics, wrong results
            if constexpr (VARIANT != 3) __syncthreads();
            if (kt + 1 < nk) load_frags(buf ^ 1, 0, fa[0], fb[0]);
            if (kt + 2 < nk) stage(buf, (kt + 2) * BK);
            MMA_ALL(1)
#pragma unroll
            for (int i = 0; i < 8; ++i) { SGB(0x008, 1, 3); SGB(0x100, 1, 3); }
#pragma unroll
            for (int i = 0; i < 8; ++i) { SGB(0x008, 1, 3); SGB(0x010, 2, 3); }
            __builtin_amdgcn_sched_barrier(0);
            buf ^= 1;
        }
#undef MMA_ALL
#undef SGB
    }
    // C as bf16 (round to nearest even): element (row = 8 blk + 4 (lane / 32) + j, col = lane % 32) of each 32 x 32 block
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tm * BM + wm * 128 + i * 32 + (r >> 2) * 8 + fk * 4 + (r & 3);
                const int col = tn * BN + wn * 128 + j * 32 + frow;
                const uint32_t u = __float_as_uint(acc[i][j][r]);
                C[(int64_t)row * N + col] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
            }
}

static unsigned short f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static float bf2f(unsigned short h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    const int M = argc > 3 ? atoi(argv[1]) : 16384, N = argc > 3 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 2048;
    if (M % BM || N % BN || K % BK) { printf("M, N multiples of 256 and K of 64\n"); return 1; }
    std::vector<unsigned short> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hA) v = f2bf(rnd());
    for (auto& v : hB) v = f2bf(rnd());
    unsigned short *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, hC.size() * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    const int lds = 2 * STAGE_BYTES;
    const int tiles_n = N / BN, grid = (M / BM) * tiles_n;
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int variant = argc > 4 ? atoi(argv[4]) : 1;
    auto launch = [&]() {
        if (variant == 0) hipLaunchKernelGGL(gemm4w_kernel<0>, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, tiles_n);
        else if (variant == 1) hipLaunchKernelGGL(gemm4w_kernel<1>, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, tiles_n);
        else if (variant == 2) hipLaunchKernelGGL(gemm4w_kernel<2>, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, tiles_n);
        else if (variant == 3) hipLaunchKernelGGL(gemm4w_kernel<3>, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, tiles_n);
        else if (variant == 4) hipLaunchKernelGGL(gemm4w_kernel<4>, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, tiles_n);
        else hipLaunchKernelGGL(gemm4w_kernel<5>, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, tiles_n);
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) launch();
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int it = 0; it < reps; ++it) launch();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 2000; ++t) {
        s = s * 1664525u + 1013904223u; const int m = (s >> 4) % M;
        s = s * 1664525u + 1013904223u; const int n = (s >> 4) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hB[(size_t)n * K + k]);
        const double got = bf2f(hC[(size_t)m * N + n]);
        const double err = fabs(got - ref) / (fabs(ref) + 1.0);
        if (err > worst) worst = err;
    }
    const double tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12;
    printf("gemm4w variant %d, %d x %d x %d: %.3f ms, %.1f TFLOP/s = %.3f of 2500; worst sampled relative error %.2e (bf16 output)\n", variant, M, N, K, ms, tf, tf / 2500.0, worst);
    return 0;
}
