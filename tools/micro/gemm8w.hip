// Prototype (round 3 -> round 4 direction), 8 waves of 128 x 64 on a 256 x 256 tile like the shipped loops (v_mfma_f32_16x16x32_bf16):
//   variant 0: both operands staged through LDS by LDS-DMA (the shipped data path, minimal two-phase schedule)
//   variant 1: the N-side operand (4 fragments per k32 and wave) loaded straight from global memory into registers, one k-step
//              ahead; only the M side goes through LDS: 160 KB of LDS traffic per k-step instead of 256 KB
//   variant 2: as 1 with the N-side operand packed in fragment order (a wave's load is one contiguous KB)
//   variant 3: packed, one register set per k32 half (see the measurements below)
// Same schedule in all, so the difference is the data path.
//   variant 4 / 5: the shipped loops' pinned four-phase schedule, both operands through LDS / the N side direct (as variant 3)
// run: ./gemm8w M N K variant [krep]   (krep > 1: the timing pass walks the k range krep times without the epilogue stores, so that
// the operands stay L2-resident and the k-loop is compute-bound: 8192 x 8192 x 16384 streams 8.9 GB per GEMM and is HBM-limited at ~0.43)
// Measured, 8192 x 8192 x 2048 with krep 8 (fraction of 2.5 PFLOP/s; all variants validated against the host on 2000 samples):
//   0: minimal schedule, LDS both     0.516        4: pinned schedule, LDS both     0.520   <- the shipped loops' 0.51 reproduced
//   3: minimal schedule, direct N     0.493        5: pinned schedule, direct N     0.517
//   1 / 2 (two prefetch sets): 190-250 bytes of spills per lane inside the k-loop, 0.10-0.14
//   6: pinned schedule, global_load_dwordx4 -> registers -> ds_write_b128 instead of LDS-DMA     0.484
// Timing diagnostics of variant 4 (wrong results; variant code 40 + DIAG bits, see the kernel):
//   no k-step barrier / wait 0.560    no fragment reads 0.566    NO LDS-DMA 0.693    no LDS-DMA + no barrier 0.703
//   no LDS-DMA + no fragment reads 0.772    MFMAs only 0.778    half the LDS-DMA pieces 0.580    LDS-DMA with the nt policy 0.418
//   LDS-DMA of a cache-hot source (k = 0 every step) 0.563
// Reading: the MFMA pipe sustains 0.78 on this chip; fragment reads and the barrier cost 0.04 each; the operand DELIVERY costs 0.17
// (proportional to the pieces issued, a third of it the L2 / fabric fetch), whichever way it is done - LDS-DMA (best), registers +
// ds_write (0.484), or half of it as direct fragment loads (0.517: as many vector-memory instructions as before).  Delivered bytes per
// flop are fixed by the 256 x 256 tile, and a larger tile does not fit the register file: ~0.52 is this chip's ceiling for a bf16
// GEMM loop fed from memory, not a tuning gap of these kernels.
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

// DIAG (variant 4 only, timing diagnostics with wrong results): bit 0 = no k-step barrier and no LDS-DMA wait, bit 1 = no LDS-DMA,
// bit 2 = no fragment reads (the MFMAs run on whatever the registers hold)
template <int VARIANT, int DIAG = 0>
__global__ __launch_bounds__(512, 2) void gemm8w_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B,
                                                        unsigned short* __restrict__ C, int M, int N, int K, int tiles_n, const unsigned short* __restrict__ Bp, int do_store, int krep) {
    constexpr int STAGE_BYTES = ((VARIANT == 0 || VARIANT == 4 || VARIANT == 6) ? BM + BN : BM) * BK * 2;
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
    // DIAG bit 3: half of the LDS-DMA pieces, bit 4: nt policy on them, bit 5: every k-step stages k = 0 (the source stays cache-hot)
    auto stage = [&](int st, int k0) {
        constexpr int NP = (DIAG & 8) ? 2 : 4;
        constexpr int AUX = (DIAG & 16) ? 2 : 0;
        if constexpr ((DIAG & 32) != 0) k0 = 0;
        unsigned char* ab = smem + st * STAGE_BYTES + (wave * 32) * 128;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(A + a_base + k0 + aoff[i]), (lds_void_ptr)(ab + i * 1024), 16, 0, AUX);
        if constexpr (VARIANT == 0 || VARIANT == 4 || VARIANT == 6) {
            unsigned char* bb = smem + st * STAGE_BYTES + BM * 128 + (wave * 32) * 128;
#pragma unroll
            for (int i = 0; i < NP; ++i)
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)(B + b_base + k0 + boff[i]), (lds_void_ptr)(bb + i * 1024), 16, 0, AUX);
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
    const int nk1 = K / BK, nk = nk1 * krep;      // krep > 1 (timing only): the k range walked krep times, operands L2-resident
#define KOF(KT) (((KT) % nk1) * BK)
    stage(0, 0);
    if constexpr (VARIANT == 1 || VARIANT == 2) gload(0, 0);
    if constexpr (VARIANT == 6) {
        // the pinned schedule with the classic staging: global_load_dwordx4 into registers (issued behind the k-step barrier), written
        // to LDS by ds_write_b128 one k-step later (in front of the next barrier) - no LDS-DMA
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        bf16x8 wx[4], wy[4], n0[4], n1[4];
        u32x4 ga[4], gb[4];
        const int c8 = lane & 7;
        auto gfetch = [&](int k0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wave * 32 + i * 8 + srow;
                ga[i] = *reinterpret_cast<const u32x4*>(A + a_base + (int64_t)r * K + k0 + c8 * 8);
                gb[i] = *reinterpret_cast<const u32x4*>(B + b_base + (int64_t)r * K + k0 + c8 * 8);
            }
        };
        auto lstore = [&](int st) {
            unsigned char* at = smem + st * STAGE_BYTES;
            unsigned char* bt = at + BM * 128;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wave * 32 + i * 8 + srow;
                *reinterpret_cast<u32x4*>(at + r * 128 + ((c8 ^ SWZ(r)) * 16)) = ga[i];
                *reinterpret_cast<u32x4*>(bt + r * 128 + ((c8 ^ SWZ(r)) * 16)) = gb[i];
            }
        };
        auto load_w = [&](int st, int kk, int h, bf16x8 (&wf)[4]) {
            const unsigned char* at = smem + st * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wn * 128 + (h * 4 + i) * 16 + frow;
                wf[i] = *reinterpret_cast<const bf16x8*>(at + r * 128 + (((4 * kk + fg) ^ SWZ(r)) * 16));
            }
        };
        auto load_n = [&](int st, int kk, bf16x8 (&nf)[4]) {
            const unsigned char* bt = smem + st * STAGE_BYTES + BM * 128;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wm * 64 + j * 16 + frow;
                nf[j] = *reinterpret_cast<const bf16x8*>(bt + r * 128 + (((4 * kk + fg) ^ SWZ(r)) * 16));
            }
        };
#define MMA_HALF(HH, WF, NF)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                             \
            acc[(HH) * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[i], NF[j], acc[(HH) * 4 + i][j], 0, 0, 0);
#define SGB(MASK, CNT, ID) __builtin_amdgcn_sched_group_barrier(MASK, CNT, ID)
        // prologue (stage(0) by LDS-DMA was issued above and is simply waited for): k-step 1 goes through the registers
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (nk > 1) gfetch(KOF(1));
        load_w(0, 0, 0, wx);
        load_n(0, 0, n0);
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            load_w(buf, 0, 1, wy);
            load_n(buf, 1, n1);
            MMA_HALF(0, wx, n0)
#pragma unroll
            for (int i = 0; i < 8; ++i) { SGB(0x008, 1, 0); SGB(0x100, 1, 0); }
            SGB(0x008, 8, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_w(buf, 1, 0, wx);
            MMA_HALF(1, wy, n0)
#pragma unroll
            for (int i = 0; i < 4; ++i) { SGB(0x008, 1, 1); SGB(0x100, 1, 1); }
            SGB(0x008, 12, 1);
            __builtin_amdgcn_sched_barrier(0);
            load_w(buf, 1, 1, wy);
            if (kt + 1 < nk) lstore(buf ^ 1);           // k-step kt + 1: fetched one k-step ago, nobody reads that buffer now
            MMA_HALF(0, wx, n1)
#pragma unroll
            for (int i = 0; i < 4; ++i) { SGB(0x008, 1, 2); SGB(0x100, 1, 2); }
#pragma unroll
            for (int i = 0; i < 8; ++i) { SGB(0x008, 1, 2); SGB(0x200, 1, 2); }
            SGB(0x008, 4, 2);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            if (kt + 1 < nk) {
                load_w(buf ^ 1, 0, 0, wx);
                load_n(buf ^ 1, 0, n0);
            }
            if (kt + 2 < nk) gfetch(KOF(kt + 2));
            MMA_HALF(1, wy, n1)
#pragma unroll
            for (int i = 0; i < 8; ++i) { SGB(0x008, 1, 3); SGB(0x100, 1, 3); }
#pragma unroll
            for (int i = 0; i < 8; ++i) { SGB(0x008, 1, 3); SGB(0x020, 1, 3); }
            __builtin_amdgcn_sched_barrier(0);
            buf ^= 1;
        }
#undef MMA_HALF
#undef SGB
    } else
    if constexpr (VARIANT == 4 || VARIANT == 5) {
        // the shipped loops' schedule (dense_split.hip / gemm_bf16.hip): four phases of 16 MFMAs per k-step, fragments of the next
        // phase read under the current one, one barrier per k-step, LDS-DMA of k-step kt + 2 behind it, issue order pinned.
        // 4: both operands through LDS; 5: the N side (4 fragments per k32) straight from the packed copy, one register set per half
        constexpr bool DIRECT = VARIANT == 5;
        bf16x8 wx[4], wy[4], n0[4], n1[4];
        if constexpr ((DIAG & 4) != 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { wx[i] = wy[i] = n0[i] = n1[i] = *reinterpret_cast<const bf16x8*>(A + (lane + 64 * i) * 8); }
        }
        auto load_w = [&](int st, int kk, int h, bf16x8 (&wf)[4]) {
            if constexpr ((DIAG & 4) != 0) return;
            const unsigned char* at = smem + st * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wn * 128 + (h * 4 + i) * 16 + frow;
                wf[i] = *reinterpret_cast<const bf16x8*>(at + r * 128 + (((4 * kk + fg) ^ SWZ(r)) * 16));
            }
        };
        auto load_n = [&](int st, int kk, int k0, bf16x8 (&nf)[4]) {
            if constexpr ((DIAG & 4) != 0) return;
            if constexpr (DIRECT) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    nf[j] = *reinterpret_cast<const bf16x8*>(Bp + ((((int64_t)(tn * 16 + wm * 4 + j)) * (K / 32) + (k0 / 32 + kk)) * 64 + lane) * 8);
            } else {
                const unsigned char* bt = smem + st * STAGE_BYTES + BM * 128;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = wm * 64 + j * 16 + frow;
                    nf[j] = *reinterpret_cast<const bf16x8*>(bt + r * 128 + (((4 * kk + fg) ^ SWZ(r)) * 16));
                }
            }
        };
#define MMA_HALF(HH, WF, NF)                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                             \
            acc[(HH) * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[i], NF[j], acc[(HH) * 4 + i][j], 0, 0, 0);
#define SGB(MASK, CNT, ID) __builtin_amdgcn_sched_group_barrier(MASK, CNT, ID)
        // prologue: stage(0) was issued above; stage 1, then the first fragments
        // issue order as in the steady state: n0, LDS-DMA, n1 (the counted waits below rely on it)
        if constexpr (DIRECT) load_n(0, 0, 0, n0);
        if (nk > 1) stage(1, KOF(1));
        if constexpr (DIRECT) load_n(0, 1, 0, n1);
        if constexpr (DIRECT) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // stage 0 landed
        __syncthreads();
        load_w(0, 0, 0, wx);
        if constexpr (!DIRECT) load_n(0, 0, 0, n0);
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const int k0 = KOF(kt), k1 = KOF(kt + 1), k2 = KOF(kt + 2);
            // P0
            load_w(buf, 0, 1, wy);
            if constexpr (!DIRECT) load_n(buf, 1, k0, n1);
            if constexpr (DIRECT) {      // n0 of this k-step (older than the last LDS-DMA and n1; in the last k-step only n1 is younger)
                if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
            MMA_HALF(0, wx, n0)
#pragma unroll
            for (int i = 0; i < (DIRECT ? 4 : 8); ++i) { SGB(0x008, 1, 0); SGB(0x100, 1, 0); }
            SGB(0x008, DIRECT ? 12 : 8, 0);
            __builtin_amdgcn_sched_barrier(0);
            // P1
            load_w(buf, 1, 0, wx);
            MMA_HALF(1, wy, n0)
#pragma unroll
            for (int i = 0; i < 4; ++i) { SGB(0x008, 1, 1); SGB(0x100, 1, 1); }
            SGB(0x008, 12, 1);
            __builtin_amdgcn_sched_barrier(0);
            // P2 (n0 is free: its refill for the next k-step goes out here)
            load_w(buf, 1, 1, wy);
            if constexpr (DIRECT) { if (kt + 1 < nk) load_n(0, 0, k1, n0); }
            if constexpr (DIRECT) { if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }   // n1 of this k-step and the LDS-DMA in front of it
            MMA_HALF(0, wx, n1)
#pragma unroll
            for (int i = 0; i < 4; ++i) { SGB(0x008, 1, 2); SGB(0x100, 1, 2); }
            SGB(0x008, 12, 2);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!DIRECT && (DIAG & 1) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr ((DIAG & 1) == 0) __syncthreads();
            // P3
            if (kt + 1 < nk) {
                load_w(buf ^ 1, 0, 0, wx);
                if constexpr (!DIRECT) load_n(buf ^ 1, 0, k1, n0);
            }
            if constexpr ((DIAG & 2) == 0) { if (kt + 2 < nk) stage(buf, k2); }
            MMA_HALF(1, wy, n1)
            if constexpr (DIRECT) { if (kt + 1 < nk) load_n(0, 1, k1, n1); }
            __builtin_amdgcn_sched_barrier(0);
            buf ^= 1;
        }
#undef MMA_HALF
#undef SGB
    } else
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
            if (kt + 1 < nk) stage(buf ^ 1, KOF(kt + 1));
            const unsigned char* at = smem + buf * STAGE_BYTES;
            mma_half(at, 0);
            if (kt + 1 < nk) gl(0, KOF(kt + 1));
            // outstanding: nh[1](kt) x4, LDS-DMA(kt + 1) x4, nh[0](kt + 1) x4 -> nh[1](kt) must have landed
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            mma_half(at, 1);
            if (kt + 1 < nk) gl(1, KOF(kt + 1));
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
            stage(NXT, KOF((KT) + 1));                                                                                    \
            if constexpr (VARIANT >= 1) gload(NXT, KOF((KT) + 1));                                                        \
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
    // element (row = 4 (lane / 16) + r, col = lane % 16) of each 16 x 16 block; do_store = 0 times the k-loop alone (the 2-byte
    // scattered stores of this prototype's epilogue are slow), with one store that keeps the accumulators alive
    if (!do_store) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (sum == 1234.5678f) C[0] = 1;
        return;
    }
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
    const int lds = 2 * ((variant == 0 || variant == 4 || variant == 6 || variant >= 40) ? BM + BN : BM) * BK * 2;
    const int tiles_n = N / BN, grid = (M / BM) * tiles_n;
    CK(hipFuncSetAttribute((const void*)gemm8w_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM + BN) * BK * 2));
    CK(hipFuncSetAttribute((const void*)gemm8w_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BM * BK * 2));
    CK(hipFuncSetAttribute((const void*)gemm8w_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BM * BK * 2));
    CK(hipFuncSetAttribute((const void*)gemm8w_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BM * BK * 2));
    CK(hipFuncSetAttribute((const void*)gemm8w_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM + BN) * BK * 2));
    CK(hipFuncSetAttribute((const void*)gemm8w_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BM * BK * 2));
    CK(hipFuncSetAttribute((const void*)gemm8w_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM + BN) * BK * 2));
#define DIAGATTR(D) CK(hipFuncSetAttribute((const void*)gemm8w_kernel<4, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM + BN) * BK * 2));
    DIAGATTR(1) DIAGATTR(2) DIAGATTR(3) DIAGATTR(4) DIAGATTR(6) DIAGATTR(7) DIAGATTR(8) DIAGATTR(16) DIAGATTR(32)
#undef DIAGATTR
    int do_store = 1, krep = 1;
    const int krep_timing = argc > 5 ? atoi(argv[5]) : 1;
    auto launch = [&]() {
        if (variant == 0) hipLaunchKernelGGL(gemm8w_kernel<0>, dim3(grid), dim3(512), lds, 0, dA, dB, dC, M, N, K, tiles_n, dBp, do_store, krep);
        else if (variant == 1) hipLaunchKernelGGL(gemm8w_kernel<1>, dim3(grid), dim3(512), lds, 0, dA, dB, dC, M, N, K, tiles_n, dBp, do_store, krep);
        else if (variant == 2) hipLaunchKernelGGL(gemm8w_kernel<2>, dim3(grid), dim3(512), lds, 0, dA, dB, dC, M, N, K, tiles_n, dBp, do_store, krep);
        else if (variant == 3) hipLaunchKernelGGL(gemm8w_kernel<3>, dim3(grid), dim3(512), lds, 0, dA, dB, dC, M, N, K, tiles_n, dBp, do_store, krep);
        else if (variant == 4) hipLaunchKernelGGL(gemm8w_kernel<4>, dim3(grid), dim3(512), lds, 0, dA, dB, dC, M, N, K, tiles_n, dBp, do_store, krep);
        else if (variant == 5) hipLaunchKernelGGL(gemm8w_kernel<5>, dim3(grid), dim3(512), lds, 0, dA, dB, dC, M, N, K, tiles_n, dBp, do_store, krep);
        else if (variant == 6) hipLaunchKernelGGL(gemm8w_kernel<6>, dim3(grid), dim3(512), lds, 0, dA, dB, dC, M, N, K, tiles_n, dBp, do_store, krep);
#define DIAGCASE(D) else if (variant == 40 + D) hipLaunchKernelGGL((gemm8w_kernel<4, D>), dim3(grid), dim3(512), lds, 0, dA, dB, dC, M, N, K, tiles_n, dBp, do_store, krep);
        DIAGCASE(1) DIAGCASE(2) DIAGCASE(3) DIAGCASE(4) DIAGCASE(6) DIAGCASE(7) DIAGCASE(8) DIAGCASE(16) DIAGCASE(32)
#undef DIAGCASE
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) launch();
    CK(hipDeviceSynchronize());
    const int reps = 20;
    float ms, ms_loop;
    CK(hipEventRecord(e0));
    for (int it = 0; it < reps; ++it) launch();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
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
    do_store = 0;
    krep = krep_timing;
    for (int it = 0; it < 3; ++it) launch();
    CK(hipEventRecord(e0));
    for (int it = 0; it < reps; ++it) launch();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms_loop, e0, e1));
    ms_loop /= reps;
    const double tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12, tfl = 2.0 * M * N * K * krep / (ms_loop * 1e-3) / 1e12;
    printf("gemm8w variant %d, %d x %d x %d: %.3f ms, %.1f TFLOP/s = %.3f of 2500 (k-loop alone, no epilogue stores, k range walked %d times: %.3f); worst sampled relative error %.2e (bf16 output)\n",
           variant, M, N, K, ms, tf, tf / 2500.0, krep, tfl / 2500.0, worst);
    return 0;
}
