// Prototype (round 3 -> round 4 direction), 8 waves of 128 x 64 on a 256 x 256 tile like the shipped loops (v_mfma_f32_16x16x32_bf16):
//   variant 0: both operands staged through LDS by LDS-DMA (the shipped data path, minimal two-phase schedule)
//   variant 1: the N-side operand (4 fragments per k32 and wave) loaded straight from global memory into registers, one k-step
//              ahead; only the M side goes through LDS: 160 KB of LDS traffic per k-step instead of 256 KB
//   variant 2: as 1 with the N-side operand packed in fragment order (a wave's load is one contiguous KB)
//   variant 3: packed, one register set per k32 half (see the measurements below)
// Same schedule in all, so the difference is the data path.
// Measured (8192 x 8192 x 16384): variant 0 0.436 of 2.5 PFLOP/s (the shipped loops' tuned schedule reaches 0.51 on the same data path);
// variants 1 / 2 0.141 / 0.101 - NOT a verdict on the data path: 128 accumulators + two prefetch sets of the direct operand (64) +
// fragments + addresses exceed the 256 registers a wave has at two waves per SIMD, hipcc spills 190-250 bytes per lane inside the
// k-loop (scratch_load / scratch_store between the MFMAs) and everything waits on that.
//   variant 3: ONE register set per k32 half of the (packed) direct operand, refilled right behind the MFMAs that consumed it,
//              counted vmcnt waits: 196 registers, no spills, correct: 0.433 against variant 0's 0.430 in the same run - the direct
//              operand costs nothing even half a k-step ahead, and with this minimal schedule it gains nothing either: 37 % less LDS
//              traffic is not what this schedule waits for.  Whether it pays under the shipped loops' pinned schedule (0.51) is
//              the open question for round 4.  build: hipcc --offload-arch=gfx950 -O3 -o gemm8w gemm8w.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

constexpr int BM = 256, BN = 256, BK = 64;
#define SWZ(row) ((row) & 7)

template <int VARIANT>
__global__ __launch_bounds__(512, 2) void gemm8w_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B,
                                                        unsigned short* __restrict__ C, int M, int N, int K, int tiles_n, const unsigned short* __restrict__ Bp) {
    constexpr int STAGE_BYTES = (VARIANT == 0 ? BM + BN : BM) * BK * 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 2, wm = wave & 3;                       // M halves of 128 rows x N quarters of 64 columns
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
    const int swz = (nwg % 8 == 0) ? (bid % 8) * (nwg / 8) + bid / 8 : bid;
    const int tm = swz / tiles_n, tn = swz % tiles_n;
    const int64_t a_base = (int64_t)tm * BM * K, b_base = (int64_t)tn * BN * K;
    const int srow = lane >> 3;
    int aoff[4], boff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wave * 32 + i * 8 + srow;
        const int schunk = (lane & 7) ^ SWZ(r);
        aoff[i] = r * K + schunk * 8;
        boff[i] = r * K + schunk * 8;
    }
    auto stage = [&](int st, int k0) {
        unsigned char* ab = smem + st * STAGE_BYTES + (wave * 32) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(A + a_base + k0 + aoff[i]), (lds_void_ptr)(ab + i * 1024), 16, 0, 0);
        if constexpr (VARIANT == 0) {
            unsigned char* bb = smem + st * STAGE_BYTES + BM * 128 + (wave * 32) * 128;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)(B + b_base + k0 + boff[i]), (lds_void_ptr)(bb + i * 1024), 16, 0, 0);
        }
    };
    const int frow = lane & 15, fg = lane >> 4;
    // variant 1: fragment (kk, j) of the N side for this lane: row wm * 64 + j * 16 + frow, k-chunk 4 kk + fg
    bf16x8 nreg[2][2][4];
    auto gload = [&](int set, int k0) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if constexpr (VARIANT == 2)      // B packed in fragment order: [N / 16][K / 32][64 lanes][8]: one contiguous KB per wave load
                    nreg[set][kk][j] = *reinterpret_cast<const bf16x8*>(Bp + ((((int64_t)(tn * 16 + wm * 4 + j)) * (K / 32) + (k0 / 32 + kk)) * 64 + lane) * 8);
                else
                    nreg[set][kk][j] = *reinterpret_cast<const bf16x8*>(B + b_base + (int64_t)(wm * 64 + j * 16 + frow) * K + k0 + (4 * kk + fg) * 8);
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = K / BK;
    stage(0, 0);
    if constexpr (VARIANT == 1 || VARIANT == 2) gload(0, 0);
    if constexpr (VARIANT == 3) {
        // one register set per k32 half of the direct operand (packed layout), each refilled right after the MFMAs that consumed it
        // were issued: 32 registers instead of 64; counted vmcnt waits (issue order per k-step: nh[1] refill, LDS-DMA, nh[0] refill)
        bf16x8 nh[2][4];
        auto gl = [&](int kk, int k0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                nh[kk][j] = *reinterpret_cast<const bf16x8*>(Bp + ((((int64_t)(tn * 16 + wm * 4 + j)) * (K / 32) + (k0 / 32 + kk)) * 64 + lane) * 8);
        };
        auto mma_half = [&](const unsigned char* at, int kk) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bf16x8 mf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = wn * 128 + (h * 4 + i) * 16 + frow;
                    mf[i] = *reinterpret_cast<const bf16x8*>(at + r * 128 + (((4 * kk + fg) ^ SWZ(r)) * 16));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[h * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf[i], nh[kk][j], acc[h * 4 + i][j], 0, 0, 0);
            }
        };
        // prologue (stage(0) is already issued above): nh[0], nh[1] of k-step 0
        gl(0, 0);
        gl(1, 0);
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            // outstanding, oldest first: LDS-DMA(kt) x4, nh[0](kt) x4, nh[1](kt) x4 -> the first two groups must have landed
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nk) stage(buf ^ 1, (kt + 1) * BK);
            const unsigned char* at = smem + buf * STAGE_BYTES;
            mma_half(at, 0);
            if (kt + 1 < nk) gl(0, (kt + 1) * BK);
            // outstanding: nh[1](kt) x4, LDS-DMA(kt + 1) x4, nh[0](kt + 1) x4 -> nh[1](kt) must have landed
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            mma_half(at, 1);
            if (kt + 1 < nk) gl(1, (kt + 1) * BK);
            buf ^= 1;
        }
    } else {
    // two k-steps per trip so that the register sets of the direct operand are indexed statically (a run-time index would put
    // them in scratch memory); nk is even
#define KSTEP(KT, CUR, NXT)                                                                                               \
    {                                                                                                                     \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                  \
        __syncthreads();                                                                                                  \
        if ((KT) + 1 < nk) {                                                                                              \
            stage(NXT, ((KT) + 1) * BK);                                                                                  \
            if constexpr (VARIANT >= 1) gload(NXT, ((KT) + 1) * BK);                                                      \
        }                                                                                                                 \
        const unsigned char* at = smem + (CUR) * STAGE_BYTES;                                                             \
        const unsigned char* bt = at + BM * 128;                                                                          \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                                \
            bf16x8 nf[4];                                                                                                 \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                               \
                if constexpr (VARIANT == 0) {                                                                             \
                    const int r = wm * 64 + j * 16 + frow;                                                                \
                    nf[j] = *reinterpret_cast<const bf16x8*>(bt + r * 128 + (((4 * kk + fg) ^ SWZ(r)) * 16));             \
                } else {                                                                                                  \
                    nf[j] = nreg[CUR][kk][j];                                                                             \
                }                                                                                                         \
            }                                                                                                             \
            _Pragma("unroll") for (int h = 0; h < 2; ++h) {       /* M fragments four at a time: 16 registers, not 32 */   \
                bf16x8 mf[4];                                                                                             \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                           \
                    const int r = wn * 128 + (h * 4 + i) * 16 + frow;                                                     \
                    mf[i] = *reinterpret_cast<const bf16x8*>(at + r * 128 + (((4 * kk + fg) ^ SWZ(r)) * 16));             \
                }                                                                                                         \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                         \
                        acc[h * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf[i], nf[j], acc[h * 4 + i][j], 0, 0, 0); \
            }                                                                                                             \
        }                                                                                                                 \
    }
    for (int kt = 0; kt < nk; kt += 2) {
        KSTEP(kt, 0, 1)
        KSTEP(kt + 1, 1, 0)
    }
#undef KSTEP
    }
    // element (row = 4 (lane / 16) + r, col = lane % 16) of each 16 x 16 block
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = tm * BM + wn * 128 + i * 16 + fg * 4 + r;
                const int col = tn * BN + wm * 64 + j * 16 + frow;
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
    std::vector<unsigned short> hBp((size_t)N * K);
    for (int nb = 0; nb < N / 16; ++nb)
        for (int kb = 0; kb < K / 32; ++kb)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e)
                    hBp[((((size_t)nb * (K / 32) + kb) * 64 + l) * 8) + e] = hB[(size_t)(nb * 16 + (l & 15)) * K + kb * 32 + (l >> 4) * 8 + e];
    unsigned short* dBp;
    CK(hipMalloc(&dBp, hBp.size() * 2));
    CK(hipMemcpy(dBp, hBp.data(), hBp.size() * 2, hipMemcpyHostToDevice));
    const int variant = argc > 4 ? atoi(argv[4]) : 1;
    const int lds = 2 * (variant == 0 ? BM + BN : BM) * BK * 2;
    const int tiles_n = N / BN, grid = (M / BM) * tiles_n;
    CK(hipFuncSetAttribute((const void*)gemm8w_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM + BN) * BK * 2));
    CK(hipFuncSetAttribute((const void*)gemm8w_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BM * BK * 2));
    CK(hipFuncSetAttribute((const void*)gemm8w_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BM * BK * 2));
    CK(hipFuncSetAttribute((const void*)gemm8w_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BM * BK * 2));
    auto launch = [&]() {
        if (variant == 0) hipLaunchKernelGGL(gemm8w_kernel<0>, dim3(grid), dim3(512), lds, 0, dA, dB, dC, M, N, K, tiles_n, dBp);
        else if (variant == 1) hipLaunchKernelGGL(gemm8w_kernel<1>, dim3(grid), dim3(512), lds, 0, dA, dB, dC, M, N, K, tiles_n, dBp);
        else if (variant == 2) hipLaunchKernelGGL(gemm8w_kernel<2>, dim3(grid), dim3(512), lds, 0, dA, dB, dC, M, N, K, tiles_n, dBp);
        else hipLaunchKernelGGL(gemm8w_kernel<3>, dim3(grid), dim3(512), lds, 0, dA, dB, dC, M, N, K, tiles_n, dBp);
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
    printf("gemm8w variant %d, %d x %d x %d: %.3f ms, %.1f TFLOP/s = %.3f of 2500; worst sampled relative error %.2e (bf16 output)\n", variant, M, N, K, ms, tf, tf / 2500.0, worst);
    return 0;
}
