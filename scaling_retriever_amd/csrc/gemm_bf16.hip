// bf16 x bf16 -> fp32 MFMA GEMM with fused epilogues for the Llama encoder (gfx950).
//
// y[M,N] = A[M,K] @ W[N,K]^T  (nn.Linear).  Replaces the cuBLAS GEMMs HF's
// LlamaModel issues under torch.autocast(bf16) (scaling_retriever/indexer.py:46-52).
//
// Tile 128(n) x 128(m) x 64(k), 4 waves (2 x 2), each wave 64 x 64 as 4 x 4 blocks of
// v_mfma_f32_16x16x32_bf16.  The WEIGHT tile is the MFMA A operand and the activation
// tile the B operand, so an accumulator lane owns one token row m (lane & 15) and 4
// consecutive output features n in its 4 registers: epilogue stores are 8 B (bf16x4) or
// 16 B (fp32x4) per lane, gate/up and RoPE partners sit in the same lane.
// Staging: global_load_lds 16 B (LDS-DMA), two LDS stages, one barrier per k-step; the
// XOR swizzle (16-B chunk ^ (row & 7)) is applied on the global SOURCE address and on the
// ds_read_b128 address (LDS image is lane-linear).  Grid is n-tile-fastest so that, with
// round-robin XCD placement, an XCD keeps touching the same 1/8 of W (L2-resident).
#include "kernels.h"

typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef __attribute__((address_space(1))) const void* gbl_void_ptr;

#define G_BM 128
#define G_BN 128
#define G_BK 64
#define G_STAGE_BYTES (2 * 128 * 64 * 2)  // W tile + A tile, bf16

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    const int tiles_n = (g.N + G_BN - 1) / G_BN;
    const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
    const int n0 = tile_n * G_BN, m0 = tile_m * G_BM;
    const int K = g.K;

    // ---- staging addresses: wave w moves rows [32w, 32w+32) of each tile, 8 rows per instruction
    const int srow = lane >> 3;                       // row inside an 8-row piece
    const int schunk = (lane & 7) ^ (srow & 7);       // source 16-B chunk (swizzle on the source)
    const bf16_t* wsrc[4];
    const bf16_t* asrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int rn = n0 + wave * 32 + i * 8 + srow;
        rn = rn < g.N ? rn : g.N - 1;
        int rm = m0 + wave * 32 + i * 8 + srow;
        rm = rm < g.M ? rm : g.M - 1;
        wsrc[i] = g.W + (int64_t)rn * K + schunk * 8;
        asrc[i] = g.A + (int64_t)rm * K + schunk * 8;
    }
    auto stage = [&](int st, int k0) {
        unsigned char* wbase = smem + st * G_STAGE_BYTES + (wave * 32) * 128;
        unsigned char* abase = wbase + 128 * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(wsrc[i] + k0), (lds_void_ptr)(wbase + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(asrc[i] + k0), (lds_void_ptr)(abase + i * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fg = lane >> 4;
    const int nk = K / G_BK;
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * G_BK);
        const unsigned char* wt = smem + cur * G_STAGE_BYTES;
        const unsigned char* at = wt + 128 * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int pos = ((4 * kk + fg) ^ (frow & 7)) * 16;
            mfma_bf16x8 wf[4], af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                wf[i] = *reinterpret_cast<const mfma_bf16x8*>(wt + (wn * 64 + i * 16 + frow) * 128 + pos);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                af[j] = *reinterpret_cast<const mfma_bf16x8*>(at + (wm * 64 + j * 16 + frow) * 128 + pos);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogues: lane owns token m = .. + (lane & 15), features n = .. + 4 * (lane >> 4) + r
    if constexpr (EPI == EPI_STORE_BF16 || EPI == EPI_STORE_F32 || EPI == EPI_RESID_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + j * 16 + frow;
            if (m >= g.M) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + wn * 64 + i * 16 + fg * 4;
                if (n >= g.N) continue;
                const f32x4 v = acc[i][j];
                if constexpr (EPI == EPI_STORE_BF16) {
                    bf16x4 o;
                    o[0] = (short)f32_to_bf16(v[0]);
                    o[1] = (short)f32_to_bf16(v[1]);
                    o[2] = (short)f32_to_bf16(v[2]);
                    o[3] = (short)f32_to_bf16(v[3]);
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(g.C) + (int64_t)m * g.N + n) = o;
                } else if constexpr (EPI == EPI_STORE_F32) {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + (int64_t)m * g.N + n) = v;
                } else {
                    f32x4* p = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + (int64_t)m * g.N + n);
                    f32x4 x = *p;
                    x += v;
                    *p = x;
                }
            }
        }
    } else if constexpr (EPI == EPI_SWIGLU) {
        const int half_n = g.N >> 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + j * 16 + frow;
            if (m >= g.M) continue;
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                const int n = (n0 >> 1) + wn * 32 + (i >> 1) * 16 + fg * 4;
                if (n >= half_n) continue;
                const f32x4 gt = acc[i][j], up = acc[i + 1][j];
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sg = gt[r] / (1.f + __expf(-gt[r]));
                    o[r] = (short)f32_to_bf16(sg * up[r]);
                }
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(g.C) + (int64_t)m * half_n + n) = o;
            }
        }
    } else {  // EPI_SEGMAX: tile of logits -> LDS, per-column segmented max over token rows, atomicMax
        float* L = reinterpret_cast<float*>(smem);            // [128 m][128 n], column index XOR-swizzled by row
        int* seq_s = reinterpret_cast<int*>(smem + 2 * G_STAGE_BYTES);  // [128]
        if (tid < 128) seq_s[tid] = (m0 + tid < g.M) ? g.seq_of[m0 + tid] : -1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ml = wm * 64 + j * 16 + frow;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int nl = wn * 64 + i * 16 + fg * 4;
                *reinterpret_cast<f32x4*>(L + ml * 128 + (nl ^ ((ml & 15) << 2))) = acc[i][j];
            }
        }
        __syncthreads();
        const int col = tid & 127, rbeg = (tid >> 7) * 64;
        const int n = n0 + col;
        if (n < g.N) {
            float* out = reinterpret_cast<float*>(g.C);
            int cur_seq = -1;
            float cur_max = 0.f;
            for (int r = rbeg; r < rbeg + 64; ++r) {
                const int sq = seq_s[r];
                if (sq == -1) break;      // past the last token
                if (sq < 0) continue;     // masked token: not part of any max
                const float v = L[r * 128 + (col ^ ((r & 15) << 2))];
                if (sq != cur_seq) {
                    if (cur_seq >= 0 && cur_max > 0.f)
                        atomicMax(reinterpret_cast<int*>(out + (int64_t)cur_seq * g.out_ld + n), __float_as_int(cur_max));
                    cur_seq = sq;
                    cur_max = 0.f;
                }
                cur_max = v > cur_max ? v : cur_max;
            }
            if (cur_seq >= 0 && cur_max > 0.f)
                atomicMax(reinterpret_cast<int*>(out + (int64_t)cur_seq * g.out_ld + n), __float_as_int(cur_max));
        }
    }
}

template <int EPI>
static int launch_one(const GemmArgs& g, hipStream_t s) {
    constexpr size_t lds = 2 * G_STAGE_BYTES + (EPI == EPI_SEGMAX ? 512 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        SR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<EPI>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int64_t tiles = ceil_div64(g.N, G_BN) * ceil_div64(g.M, G_BM);
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI>), dim3((unsigned)tiles), dim3(256), lds, s, g);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

int launch_gemm_bf16(GemmEpilogue epi, const GemmArgs& g, hipStream_t s) {
    SR_REQUIRE(g.M >= 0 && g.N > 0 && g.K > 0, "gemm: bad shape M=%d N=%d K=%d", g.M, g.N, g.K);
    if (g.M == 0) return SR_OK;
    SR_REQUIRE(g.K % G_BK == 0, "gemm: K=%d must be a multiple of %d", g.K, G_BK);
    SR_REQUIRE(g.N % 16 == 0 && (epi != EPI_SWIGLU || g.N % 32 == 0), "gemm: N=%d must be a multiple of 16 (32 for SwiGLU)", g.N);
    switch (epi) {
        case EPI_STORE_BF16: return launch_one<EPI_STORE_BF16>(g, s);
        case EPI_RESID_F32: return launch_one<EPI_RESID_F32>(g, s);
        case EPI_SWIGLU: return launch_one<EPI_SWIGLU>(g, s);
        case EPI_SEGMAX: return launch_one<EPI_SEGMAX>(g, s);
        case EPI_STORE_F32: return launch_one<EPI_STORE_F32>(g, s);
    }
    sr_set_error("gemm: unknown epilogue %d", (int)epi);
    return SR_ERR_INVALID;
}
