// bf16 x bf16 -> fp32 MFMA GEMM with fused epilogues for the Llama encoder (gfx950).
//
// y[M,N] = A[M,K] @ W[N,K]^T  (nn.Linear).  Replaces the cuBLAS GEMMs HF's
// LlamaModel issues under torch.autocast(bf16) (scaling_retriever/indexer.py:46-52).
//
// Tile configurations of one kernel template (BK = 64, v_mfma_f32_16x16x32_bf16):
//   256(n) x 256(m), 8 waves (2 x 4), wave tile 128 x 64, 128 KB LDS, 1 workgroup per CU, software-pipelined k-loop:
//       half the L2->LDS bytes per FLOP of the small tile; runs whole rounds of 256 tiles (see the tile plan below)
//   128(n) x 128(m), 4 waves (2 x 2), wave tile 64 x 64, 64 KB LDS, 2 workgroups per CU: small problems and ragged tails
//    64(n) x 16/32/64(m), ONE wave, four LDS stages: up to 64 token rows (online queries) - the work is streaming W once
//   128(n) x  64(m), 4 waves, three 24 KB LDS stages (LDS-DMA issued three k-steps ahead, counted vmcnt before a bare
//       s_barrier), 2 workgroups per CU: tails of fewer than 384 tiles
// The WEIGHT tile is the MFMA A operand and the activation tile the B operand, so an
// accumulator lane owns one token row m (lane & 15) and 4 consecutive output features n in
// its 4 registers: epilogue stores are 8 B (bf16x4) or 16 B (fp32x4) per lane, gate/up
// partners sit in the same lane.
// Staging: global_load_lds 16 B (LDS-DMA), two LDS stages, one barrier per k-step; the XOR
// swizzle (16-B chunk ^ (row & 7)) is applied on the global SOURCE address and on the
// ds_read_b128 address (the LDS image is lane-linear).  Workgroups are persistent over tiles; the tile order is
// feature-tile-fastest (with round-robin XCD placement an XCD keeps touching the same 1/8 of W) unless W is far
// larger than the caches, then token-tile-fastest (launch_cfg).
#include "kernels.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 mfma_f16x8 __attribute__((ext_vector_type(8)));
// one MFMA step of the k-loop: the fragments are 16 bytes of either bf16 or fp16
template <bool F16>
__device__ __forceinline__ f32x4 sr_mma(const mfma_bf16x8& w, const mfma_bf16x8& a, const f32x4& c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mfma_f16x8, w), __builtin_bit_cast(mfma_f16x8, a), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, c, 0, 0, 0);
}
// RoPE on the accumulators, HF rotate_half layout: first half of a head  x1 c - x2 s,  second half  x2 c + x1 s.  Products and sum
// rounded separately, as PyTorch's eager `q * cos + rotate_half(q) * sin` does - and so that every epilogue variant (direct or staged
// stores, any tile configuration) produces the same bits whatever the compiler would otherwise fuse.
__device__ __forceinline__ f32x4 sr_rope_lo(const f32x4& x1, const f32x4& x2, const f32x4& c, const f32x4& s) {
#pragma clang fp contract(off)
    f32x4 y;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float a = x1[r] * c[r], b = x2[r] * s[r]; y[r] = a - b; }
    return y;
}
__device__ __forceinline__ f32x4 sr_rope_hi(const f32x4& x1, const f32x4& x2, const f32x4& c, const f32x4& s) {
#pragma clang fp contract(off)
    f32x4 y;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float a = x2[r] * c[r], b = x1[r] * s[r]; y[r] = a + b; }
    return y;
}
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef __attribute__((address_space(1))) const void* gbl_void_ptr;

#define G_BK 64

// WAVES_N x WAVES_M waves; each wave owns (16 * NB) features x (16 * MB) tokens.
// PIPE: fragment double-buffering - the ds_reads of the next half k-step are in flight while the MFMAs of the current
// half run, stages are issued two k-steps ahead, still one barrier per k-step (placed between the two halves).
// KL = 1: the FOUR-wave k-loop of the 256 x 256 tile (2 x 2 waves of 128 x 128, one wave per SIMD, the 256 accumulator registers of a
// lane in AGPRs), written instruction by instruction: see the KL == 1 branch below.
typedef int sr_i32x4 __attribute__((ext_vector_type(4)));
template <int EPI, int WAVES_N, int WAVES_M, int NB, int MB, bool PIPE = false, int NS = 2, int KL = 0>
__global__ __launch_bounds__(64 * WAVES_N * WAVES_M, KL == 1 ? 1 : (WAVES_N * WAVES_M) / 4 * (WAVES_N * WAVES_M == 4 ? 2 : 1))
void gemm_bf16_kernel(GemmArgs g) {
    constexpr bool F16 = EPI >= EPI_H_FIRST;          // fp16 plane operands + row scales (fp32 regime), see kernels.h
    constexpr int BEPI = !F16 ? EPI : (EPI == EPI_QKV_ROPE_F32_H ? EPI_QKV_ROPE_F32 : (EPI == EPI_RESID_F32_H ? EPI_RESID_F32
                                      : (EPI == EPI_SWIGLU_F32_H ? EPI_SWIGLU_F32 : (EPI == EPI_SWIGLU_SPLIT_H ? EPI_SWIGLU_SPLITH_BASE : EPI_SEGMAX))));
    constexpr int NW = WAVES_N * WAVES_M;
    constexpr int BN = 16 * NB * WAVES_N, BM = 16 * MB * WAVES_M;
    constexpr int W_BYTES = BN * 128, A_BYTES = BM * 128, STAGE_BYTES = W_BYTES + A_BYTES;
    constexpr int W_INSTR = BN / 8 / NW, A_INSTR = BM / 8 / NW;   // 8-row (1 KB) LDS-DMA pieces per wave
    static_assert(BN % (8 * NW) == 0 && BM % (8 * NW) == 0, "tile rows must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WAVES_M, wm = wave % WAVES_M;
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    const int n_tiles = tiles_n * tiles_m;
    const int K = g.K;
    constexpr bool CROSS_PREFETCH = true;

    // ---- staging addresses: wave w moves rows [w*R, (w+1)*R) of each tile, 8 rows per instruction
    const int srow = lane >> 3;                       // row inside an 8-row piece
    const int schunk = (lane & 7) ^ (srow & 7);       // source 16-B chunk (swizzle on the source)
    const bf16_t* wsrc[W_INSTR];
    const bf16_t* asrc[A_INSTR];
    // ---- slot -> tile.  A workgroup takes slots blockIdx.x, blockIdx.x + G, ...  Workgroups are dealt to the 8 XCDs
    // round-robin (XCD = blockIdx.x % 8), and every XCD has its own 4 MB L2.  In the plain feature-tile-fastest order the G / 8
    // tiles an XCD works on at a time are G / 8 DIFFERENT token tiles of one feature tile (N = 2048: 8 feature tiles, slot
    // % 8 = XCD), so only the W half of the staged bytes can hit in its L2 and every token tile is fetched into all 8 L2s.
    // xcd_order deals each XCD a compact block of bmt x bnt tiles (bmt * bnt = G / 8; 4 x 8 at G = 256): 12 operand tiles
    // behind 32 output tiles instead of 33.  Only which workgroup computes which tile changes - not a bit of the result.
    const int G = (int)gridDim.x;
    const bool swz = g.xcd_order && !g.m_fastest && (G & 7) == 0;
    auto fits = [&](int b) { return tiles_n % b == 0 && (G / 8) % b == 0; };
    const int bnt = !swz ? 1 : (fits(8) ? 8 : fits(4) ? 4 : fits(2) ? 2 : 1);
    const int bmt = !swz ? 1 : (G / 8) / bnt;
    const int slot_limit = !swz ? n_tiles : ((((tiles_m + bmt - 1) / bmt) * bmt * tiles_n + G - 1) / G) * G;
    auto slot_tile = [&](int slot, int& tn, int& tm) -> bool {
        if (!swz) {
            tn = g.m_fastest ? slot / tiles_m : slot % tiles_n;
            tm = g.m_fastest ? slot % tiles_m : slot / tiles_n;
            return true;
        }
        const int r = slot / G, w = slot - r * G;
        const int c = r * G + (w & 7) * (G >> 3) + (w >> 3);      // XCD x of round r: compact indices [r G + x G/8, + G/8)
        const int per = bmt * bnt, blk = c / per, w2 = c - blk * per;
        const int nbn = tiles_n / bnt, bm = blk / nbn, bn = blk - bm * nbn;
        tm = bm * bmt + w2 / bnt;
        tn = bn * bnt + w2 % bnt;
        return tm < tiles_m;
    };
    auto next_slot = [&](int slot) {      // first slot >= `slot` of this workgroup that holds a tile; slot_limit when none
        for (; slot < slot_limit; slot += G) {
            int tn, tm;
            if (slot_tile(slot, tn, tm)) break;
        }
        return slot < slot_limit ? slot : slot_limit;
    };
    auto set_tile = [&](int tile) {
        int tn_i, tm_i;
        (void)slot_tile(tile, tn_i, tm_i);
        const int tn0 = tn_i * BN, tm0 = tm_i * BM;
#pragma unroll
        for (int i = 0; i < W_INSTR; ++i) {
            int rn = tn0 + (wave * W_INSTR + i) * 8 + srow;
            rn = rn < g.N ? rn : g.N - 1;
            wsrc[i] = g.W + (int64_t)rn * K + schunk * 8;
        }
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            int rm = tm0 + (wave * A_INSTR + i) * 8 + srow;
            rm = rm < g.M ? rm : g.M - 1;
            asrc[i] = g.A + (int64_t)rm * K + schunk * 8;
        }
    };
    auto stage = [&](int st, int k0) {
        unsigned char* wbase = smem + st * STAGE_BYTES + (wave * W_INSTR) * 1024;
        unsigned char* abase = smem + st * STAGE_BYTES + W_BYTES + (wave * A_INSTR) * 1024;
#pragma unroll
        for (int i = 0; i < W_INSTR; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(wsrc[i] + k0), (lds_void_ptr)(wbase + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(asrc[i] + k0), (lds_void_ptr)(abase + i * 1024), 16, 0, 0);
    };

    f32x4 acc[NB][MB];
    const int frow = lane & 15, fg = lane >> 4;
    const int nk = K / G_BK;
    int tile = next_slot((int)blockIdx.x);
    if (tile >= slot_limit) return;
    const bool stamp = g.stamps != nullptr && tid == 0;
    unsigned long long* stp = g.stamps ? g.stamps + (size_t)blockIdx.x * 64 : nullptr;   // up to 16 tiles x 4 stamps
    int titer = 0;
    if (stamp) stp[0] = __builtin_amdgcn_s_memrealtime();
    if constexpr (KL == 0) {
        set_tile(tile);
        stage(0, 0);
        if constexpr (PIPE) stage(1, G_BK);     // PIPE needs nk >= 2 (checked at launch)
        if constexpr (PIPE && NS > 2) {
#pragma unroll
            for (int st = 2; st < NS; ++st) stage(st, st * G_BK);     // NS stages need nk >= NS (checked at launch)
        }
        __syncthreads();
    }
    int buf = 0;
    // ---- KL == 1 state (see the branch in the tile loop) ----
    // LDS-DMA through buffer loads: a buffer descriptor per operand (base = the tile's first row, bounds = its valid rows: rows past
    // the end of the matrix read as zeros - their outputs are never stored), a per-lane byte offset (piece, row inside the 8-row
    // piece, swizzled 16-byte chunk: the same 8 registers for every tile and both operands; it is what gfx9 range-checks) and ONE
    // scalar offset, the k position inside the rows, which advances by 128 bytes per k-step: no vector instruction computes an
    // address inside the k-loop
    [[maybe_unused]] sr_i32x4 srd_w, srd_a;
    [[maybe_unused]] uint32_t kl_koff = 0, kl_m0 = 0, kl_bw0 = 0, kl_bw1 = 0, kl_ba0 = 0, kl_ba1 = 0;
    [[maybe_unused]] uint32_t kl_voff[8];
    [[maybe_unused]] auto kl_rebase = [&](int t) {        // descriptors of tile slot t, k = 0
        int tn_i, tm_i;
        (void)slot_tile(t, tn_i, tm_i);
        const int tn0 = tn_i * BN, tm0 = tm_i * BM;
        const uint64_t pw = reinterpret_cast<uint64_t>(g.W + (int64_t)tn0 * K), pa = reinterpret_cast<uint64_t>(g.A + (int64_t)tm0 * K);
        const int rows_w = (g.N - tn0) < BN ? (g.N - tn0) : BN, rows_a = (g.M - tm0) < BM ? (g.M - tm0) : BM;
        srd_w[0] = (int)(uint32_t)pw; srd_w[1] = (int)(uint32_t)(pw >> 32); srd_w[2] = rows_w * K * 2; srd_w[3] = 0x00020000;
        srd_a[0] = (int)(uint32_t)pa; srd_a[1] = (int)(uint32_t)(pa >> 32); srd_a[2] = rows_a * K * 2; srd_a[3] = 0x00020000;
        kl_koff = 0;
    };
    auto load_frags = [&](int st, int kk, mfma_bf16x8 (&wf)[NB], mfma_bf16x8 (&af)[MB]) {
        const unsigned char* wt = smem + st * STAGE_BYTES;
        const unsigned char* at = wt + W_BYTES;
        const int pos = ((4 * kk + fg) ^ (frow & 7)) * 16;
#pragma unroll
        for (int i = 0; i < NB; ++i)
            wf[i] = *reinterpret_cast<const mfma_bf16x8*>(wt + (wn * NB * 16 + i * 16 + frow) * 128 + pos);
#pragma unroll
        for (int j = 0; j < MB; ++j)
            af[j] = *reinterpret_cast<const mfma_bf16x8*>(at + (wm * MB * 16 + j * 16 + frow) * 128 + pos);
    };
    auto mfma_all = [&](const mfma_bf16x8 (&wf)[NB], const mfma_bf16x8 (&af)[MB]) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < MB; ++j)
                acc[i][j] = sr_mma<F16>(wf[i], af[j], acc[i][j]);
    };
    if constexpr (KL == 1) {
        const uint32_t lds0 = (uint32_t)(size_t)(lds_void_ptr)smem;
        const int wave_s = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int i = 0; i < 8; ++i) kl_voff[i] = (uint32_t)(((wave_s * 8 + i) * 8 + srow) * K * 2 + schunk * 16);
        kl_m0 = lds0 + (uint32_t)wave_s * 8192u;                          // this wave's 8 pieces of stage 0's weight rows
        // fragment addresses: row (wave's first + frow) of block 0, k-half 0 / 1; blocks are 2048 bytes apart (immediates)
        const uint32_t c0 = (uint32_t)((fg ^ (frow & 7)) * 16), c1 = (uint32_t)(((4 + fg) ^ (frow & 7)) * 16);
        kl_bw0 = lds0 + (uint32_t)((wn * 128 + frow) * 128) + c0;
        kl_bw1 = lds0 + (uint32_t)((wn * 128 + frow) * 128) + c1;
        kl_ba0 = lds0 + (uint32_t)W_BYTES + (uint32_t)((wm * 128 + frow) * 128) + c0;
        kl_ba1 = lds0 + (uint32_t)W_BYTES + (uint32_t)((wm * 128 + frow) * 128) + c1;
    }
    // Persistent over tiles (grid <= one round of resident workgroups): the first k-tile(s) of the NEXT output tile are
    // prefetched during the last k-step(s) of the current one, so only the epilogue's stores stay exposed between tiles.
    [[maybe_unused]] bool kl_first = true;
    for (; tile < slot_limit;) {
    int tn_i, tm_i;
    (void)slot_tile(tile, tn_i, tm_i);
    const int n0 = tn_i * BN, m0 = tm_i * BM;
    const int tile_next = next_slot(tile + G);
    const bool has_next = tile_next < slot_limit;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int sj[MB];      // EPI_SEGMAX: sequence ids of this lane's token rows, fetched here so that the k-loop covers the latency
    if constexpr (BEPI == EPI_SEGMAX) {
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int m = m0 + wm * MB * 16 + j * 16 + frow;
            sj[j] = m < g.M ? g.seq_of[m] : -1;
        }
    }
    if (stamp && titer < 16) stp[titer * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    if (stamp && titer == 1) stp[62] = __builtin_amdgcn_s_memtime();        // shader-clock ticks at the start of tile 1's k-loop
    if constexpr (KL == 1) {
        // 256 x 256 tile on FOUR waves (2 x 2, wave tile 128 features x 128 tokens, v_mfma_f32_16x16x32: 8 x 8 blocks = 256
        // accumulator registers per lane, one wave per SIMD).  Per k-step of 64 a CU then reads 128 KB of fragments out of LDS
        // instead of the 8-wave layout's 192 KB (+ 64 KB of LDS-DMA writes either way): 1 536 LDS cycles against 2 048 MFMA cycles
        // instead of 2 048 against 2 048 - the 8-wave loops sit where both pipes saturate together (0.48-0.52 of the MFMA peak).
        // It is the configuration the vendor library runs on these shapes (hipBLASLt's MT256x256x64 MI16x16x1 kernel: 256
        // threads, 130 KB of LDS, accumulators in AGPRs; 1.1-1.28x the 8-wave loop, profiles/r04_gemm_vs_hipblaslt.json).  With ONE
        // wave per SIMD nothing hides a stall and hipcc's register allocator shuffles the 256 accumulators between VGPRs and
        // AGPRs when it is left to place them, so the k-step is written instruction by instruction (volatile asm statements are
        // emitted in program order; "+a" pins an accumulator to its AGPR quad for the whole loop):
        //   * two register sets of fragments (k-half 0 / 1: 8 + 8 ds_read_b128 each), every read issued under the MFMAs of the
        //     k-half before its first use, one read per 2 MFMAs;
        //   * the LDS-DMA pieces of k-step kt + 2 go into the stage k-step kt is read from as soon as a barrier says every wave
        //     has read it (40 MFMAs into the k-step) - a piece has ~1.6 k-steps to land, not < 1 - one piece per 3 MFMAs;
        //   * a second barrier behind a COUNTED vmcnt (the 16 pieces just issued stay in flight) publishes k-step kt + 1 in the
        //     other stage 24 MFMAs into k-half 1, and the first fragments of k-step kt + 1 are read under the remaining 40;
        //   * the last two k-steps of a tile fetch the NEXT tile's first two, and the very last one already reads its first
        //     fragments: the loop runs across tile boundaries, only the epilogue sits in between.
        // Every output element is the same k-ordered MFMA chain as in the other tile configurations: bit-identical results.
        static_assert(PIPE && NS == 2 && NW == 4 && NB == 8 && MB == 8 && WAVES_N == 2, "KL = 1 is the 2 x 2 x (128 x 128) configuration");
        sr_i32x4 w0[8], a0[8], w1[8], a1[8];      // fragments (16 bytes = 8 bf16 / fp16 per lane), k-half 0 and 1
        // accumulators of weight row blocks 0-6 in AGPRs (224 of the 256), those of row block 7 in VGPRs: with every AGPR pinned
        // the register allocator has no room for a single copy and spills accumulators to scratch inside the loop
#define KL_MMA(C, WF, AF, INV)                                                                                      \
        do {                                                                                                        \
            if constexpr (F16) {                                                                                    \
                if (INV) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(C) : "v"(WF), "v"(AF)); \
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(C) : "v"(WF), "v"(AF));              \
            } else {                                                                                                \
                if (INV) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(C) : "v"(WF), "v"(AF)); \
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(C) : "v"(WF), "v"(AF));             \
            }                                                                                                       \
        } while (0)
#define KL_DSR(DST, BASE, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(BASE), "n"(OFF))
// KL_DMA writes m0 (the LDS destination of the DMA) and says so in its clobber list: the compiler may use m0 itself anywhere around the asm
// (an LDS-DMA builtin, s_movrel, sendmsg) and will not carry a value in it across this statement.  The loads it issues are NOT tracked by the
// compiler's vmcnt / lgkmcnt bookkeeping: every wait for them in the four-wave loop is written by hand (the `s_waitcnt vmcnt(16)` / `lgkmcnt(0)` + `s_barrier` asm at the KL_B1_AT / KL_B2_AT slots), and nothing outside the asm
// blocks reads the LDS bytes they fill before such a wait + barrier.
#define KL_DMA(SRD, VOFF, KOFF, M0B, M0OFF)                                                                         \
        asm volatile("s_add_u32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(VOFF), "s"(SRD), "s"(KOFF), "s"(M0B), "n"(M0OFF) : "memory", "scc", "m0")
        // MFMA number n (0..63) of a k-half: row block n / 8 of the weights x token block n % 8
#define KL_MMA_N(n, WS, AS) KL_MMA(acc[(n) >> 3][(n) & 7], WS[(n) >> 3], AS[(n) & 7], (((n) >> 3) == 7))
        // fragment read number r (0..15): 0-7 weight blocks, 8-15 token blocks, of k-half KH
#define KL_DSR_N(r, WS, AS, BW, BA)                                                 \
        do {                                                                         \
            if ((r) < 8) KL_DSR(WS[(r) & 7], BW, ((r) & 7) * 2048);                  \
            else KL_DSR(AS[(r) & 7], BA, ((r) & 7) * 2048);                          \
        } while (0)
        // The k-step as a list of events behind its 128 MFMAs (0-63: k-half 0 on (w0, a0), 64-127: k-half 1 on (w1, a1)):
        //   reads of k-half 1 (this stage): one per KL_RD1_EVERY MFMAs from the start;
        //   barrier 1 after MFMA KL_B1_AT: every wave has read all of this stage;
        //   the 16 LDS-DMA pieces of the k-step two ahead (8 weight, 8 activation) into this stage: one per KL_DMA_EVERY MFMAs;
        //   barrier 2 after MFMA KL_B2_AT behind vmcnt(16): the NEXT k-step's pieces (issued one k-step ago) have landed;
        //   reads of the next k-step's k-half 0 (other stage): one per KL_RD2_EVERY MFMAs; lgkmcnt(0) at the end.
        // The LDS pipe is 75 % busy over a k-step (128 KB of fragment reads + 64 KB of landing pieces against 2 048 MFMA cycles),
        // so where the reads sit matters: clustered reads make the wait in front of a barrier stall the MFMA stream.
#ifndef KL_RD1_EVERY
#define KL_RD1_EVERY 2
#define KL_B1_AT 39
#define KL_DMA_EVERY 3
#define KL_B2_AT 87
#define KL_RD2_EVERY 2
#endif
        static_assert(16 * KL_RD1_EVERY - 1 <= KL_B1_AT && KL_B1_AT + 16 * KL_DMA_EVERY <= KL_B2_AT && KL_B2_AT + 16 * KL_RD2_EVERY <= 127,
                      "k-step schedule");
        auto kstep = [&]() {
            // the descriptors and offsets are wave-uniform, but they are loop-carried through the tile switch and the register
            // allocator is free to keep them in VGPRs: readfirstlane pins what the buffer instructions need in SGPRs
            sr_i32x4 sw, sa;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sw[c] = __builtin_amdgcn_readfirstlane(srd_w[c]);
                sa[c] = __builtin_amdgcn_readfirstlane(srd_a[c]);
            }
            const uint32_t koff = (uint32_t)__builtin_amdgcn_readfirstlane((int)kl_koff);
            const uint32_t m0b = (uint32_t)__builtin_amdgcn_readfirstlane((int)kl_m0);
            // (short unrolled loops: one 128-iteration loop is unrolled too late for the accumulator array to leave memory)
#define KL_MMA_G(n)                                        \
            do {                                           \
                if ((n) < 64) KL_MMA_N((n) & 63, w0, a0);  \
                else KL_MMA_N((n) & 63, w1, a1);           \
            } while (0)
#pragma unroll
            for (int r = 0; r < 16; ++r) {                 // MFMAs 0 .. 16 RD1 - 1, a read of k-half 1 behind every RD1-th
#pragma unroll
                for (int t = 0; t < KL_RD1_EVERY; ++t) KL_MMA_G(r * KL_RD1_EVERY + t);
                KL_DSR_N(r, w1, a1, kl_bw1, kl_ba1);
            }
#pragma unroll
            for (int n = 16 * KL_RD1_EVERY; n <= KL_B1_AT; ++n) KL_MMA_G(n);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave has read all of this stage
#pragma unroll
            for (int r = 0; r < 16; ++r) {                 // a piece of the k-step two ahead behind every DMA-th MFMA
#pragma unroll
                for (int t = 0; t < KL_DMA_EVERY; ++t) KL_MMA_G(KL_B1_AT + 1 + r * KL_DMA_EVERY + t);
                if (r < 8) KL_DMA(sw, kl_voff[r & 7], koff, m0b, (r & 7) * 1024);
                else KL_DMA(sa, kl_voff[r & 7], koff, m0b, W_BYTES + (r & 7) * 1024);
            }
#pragma unroll
            for (int n = KL_B1_AT + 1 + 16 * KL_DMA_EVERY; n <= KL_B2_AT; ++n) KL_MMA_G(n);
            // the 16 pieces of the NEXT k-step were issued one k-step ago: they have landed once at most the 16 just issued are
            // outstanding; the barrier publishes every wave's share
            asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
            kl_bw0 ^= 0x10000u; kl_bw1 ^= 0x10000u; kl_ba0 ^= 0x10000u; kl_ba1 ^= 0x10000u;      // the other stage from here on
            kl_m0 ^= 0x10000u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {                 // a read of the next k-step's k-half 0 behind every RD2-th MFMA
#pragma unroll
                for (int t = 0; t < KL_RD2_EVERY; ++t) KL_MMA_G(KL_B2_AT + 1 + r * KL_RD2_EVERY + t);
                KL_DSR_N(r, w0, a0, kl_bw0, kl_ba0);
            }
#pragma unroll
            for (int n = KL_B2_AT + 1 + 16 * KL_RD2_EVERY; n < 128; ++n) KL_MMA_G(n);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        if (kl_first) {       // the workgroup's first tile: k-steps 0 and 1 into the two stages, first fragments
            kl_first = false;
            kl_rebase(tile);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                sr_i32x4 sw, sa;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    sw[c] = __builtin_amdgcn_readfirstlane(srd_w[c]);
                    sa[c] = __builtin_amdgcn_readfirstlane(srd_a[c]);
                }
                const uint32_t koff = (uint32_t)__builtin_amdgcn_readfirstlane((int)kl_koff);
                const uint32_t m0b = (uint32_t)__builtin_amdgcn_readfirstlane((int)kl_m0);
#pragma unroll
                for (int r = 0; r < 8; ++r) KL_DMA(sw, kl_voff[r], koff, m0b, r * 1024);
#pragma unroll
                for (int r = 0; r < 8; ++r) KL_DMA(sa, kl_voff[r], koff, m0b, W_BYTES + r * 1024);
                kl_koff += 128;
                kl_m0 ^= 0x10000u;
            }
            asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r) KL_DSR_N(r, w0, a0, kl_bw0, kl_ba0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // invariant at the top of k-step kt: (w0, a0) hold its k-half 0; stage kt % 2 (kl_b*, kl_m0) holds it, the other stage
        // k-step kt + 1 (landed or landing); the descriptors + kl_koff point at k-step kt + 2
        // ONE copy of the k-step in the code (the accumulators keep their registers): the last two k-steps of the workgroup's last
        // tile issue their pieces against empty descriptors - every lane is out of range, nothing is fetched, zeros land in a
        // stage nobody reads again
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 2 == nk) {
                if (has_next) kl_rebase(tile_next);
                else { srd_w[2] = 0; srd_a[2] = 0; }
            }
            kstep();
            kl_koff += 128;
        }
#undef KL_MMA
#undef KL_DSR
#undef KL_DMA
#undef KL_MMA_N
#undef KL_MMA_G
#undef KL_DSR_N
    } else if constexpr (PIPE && NS > 2) {
        // NS LDS stages (small tiles, where a workgroup has a CU's MFMA pipe to itself or shares it with one other):
        // the LDS-DMA of k-step kt + NS is issued when k-step kt's stage is released, so a piece has NS - 1 k-steps to
        // land and the wait before the barrier only covers pieces issued NS - 1 k-steps ago (counted vmcnt; the barrier is
        // a bare s_barrier - __syncthreads() would drain every outstanding piece).  Same four phases as below.
        constexpr int HB = NB / 2, PIECES = W_INSTR + A_INSTR;
        mfma_bf16x8 wx[HB], wy[HB], a0[MB], a1[MB];
        auto load_w = [&](int st, int kk, int h, mfma_bf16x8 (&wf)[HB]) {
            const unsigned char* wt = smem + st * STAGE_BYTES;
            const int pos = ((4 * kk + fg) ^ (frow & 7)) * 16;
#pragma unroll
            for (int i = 0; i < HB; ++i)
                wf[i] = *reinterpret_cast<const mfma_bf16x8*>(wt + (wn * NB * 16 + (h * HB + i) * 16 + frow) * 128 + pos);
        };
        auto load_a = [&](int st, int kk, mfma_bf16x8 (&af)[MB]) {
            const unsigned char* at = smem + st * STAGE_BYTES + W_BYTES;
            const int pos = ((4 * kk + fg) ^ (frow & 7)) * 16;
#pragma unroll
            for (int j = 0; j < MB; ++j)
                af[j] = *reinterpret_cast<const mfma_bf16x8*>(at + (wm * MB * 16 + j * 16 + frow) * 128 + pos);
        };
#define SR_MFMA_HALF(H, WF, AF)                                                                              \
        _Pragma("unroll") for (int i = 0; i < HB; ++i)                                                       \
            _Pragma("unroll") for (int j = 0; j < MB; ++j)                                                   \
                acc[(H) * HB + i][j] = sr_mma<F16>(WF[i], AF[j], acc[(H) * HB + i][j]);
        load_w(buf, 0, 0, wx);
        load_a(buf, 0, a0);
        int kt = 0;
#define SR_SGB2(MASK, N, ID) __builtin_amdgcn_sched_group_barrier(MASK, N, ID)
        constexpr int PH = HB * MB;                                   // MFMAs per phase
        constexpr int R0 = HB + MB < PH ? HB + MB : PH, R1 = HB < PH ? HB : PH;
        for (; kt + NS < nk; ++kt) {
            const int nxt = buf + 1 == NS ? 0 : buf + 1;
            load_w(buf, 0, 1, wy);
            load_a(buf, 1, a1);
            SR_MFMA_HALF(0, wx, a0)
#pragma unroll
            for (int i = 0; i < R0; ++i) { SR_SGB2(0x008, 1, 0); SR_SGB2(0x100, 1, 0); }
            if constexpr (PH > R0) SR_SGB2(0x008, PH - R0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_w(buf, 1, 0, wx);
            SR_MFMA_HALF(1, wy, a0)
#pragma unroll
            for (int i = 0; i < R1; ++i) { SR_SGB2(0x008, 1, 1); SR_SGB2(0x100, 1, 1); }
            if constexpr (PH > R1) SR_SGB2(0x008, PH - R1, 1);
            __builtin_amdgcn_sched_barrier(0);
            load_w(buf, 1, 1, wy);
            SR_MFMA_HALF(0, wx, a1)
#pragma unroll
            for (int i = 0; i < R1; ++i) { SR_SGB2(0x008, 1, 2); SR_SGB2(0x100, 1, 2); }
            if constexpr (PH > R1) SR_SGB2(0x008, PH - R1, 2);
            __builtin_amdgcn_sched_barrier(0);
            // pieces of k-steps kt + 2 .. kt + NS - 1 may still be in flight; k-step kt + 1 (read next) must have landed
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NS - 2) * PIECES) : "memory");
            load_w(nxt, 0, 0, wx);
            load_a(nxt, 0, a0);
            stage(buf, (kt + NS) * G_BK);
            SR_MFMA_HALF(1, wy, a1)
#pragma unroll
            for (int i = 0; i < R0; ++i) { SR_SGB2(0x008, 1, 3); SR_SGB2(0x100, 1, 3); }
            __builtin_amdgcn_sched_barrier(0);
            buf = nxt;
        }
#undef SR_SGB2
        for (; kt < nk; ++kt) {        // drain: every remaining k-step is already on its way
            const int nxt = buf + 1 == NS ? 0 : buf + 1;
            load_w(buf, 0, 1, wy);
            load_a(buf, 1, a1);
            SR_MFMA_HALF(0, wx, a0)
            load_w(buf, 1, 0, wx);
            SR_MFMA_HALF(1, wy, a0)
            load_w(buf, 1, 1, wy);
            SR_MFMA_HALF(0, wx, a1)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + 1 < nk) {
                load_w(nxt, 0, 0, wx);
                load_a(nxt, 0, a0);
            }
            SR_MFMA_HALF(1, wy, a1)
            buf = nxt;
        }
#undef SR_MFMA_HALF
    } else if constexpr (PIPE) {
        // 4 phases per k-step: (kk, half of the wave's feature blocks).  While a phase's 16 MFMAs run, the fragments of the
        // next phase are being read from LDS (rolling wx / wy, a0 / a1).  The k-step barrier sits before the LAST phase's
        // MFMAs: by then every read of stage `buf` has been issued and waited for, so the stage can be refilled (k-step
        // kt + 2) and the next k-step's first fragments can be requested from stage buf ^ 1 under those MFMAs.
        constexpr int HB = NB / 2;
        mfma_bf16x8 wx[HB], wy[HB], a0[MB], a1[MB];
        auto load_w = [&](int st, int kk, int h, mfma_bf16x8 (&wf)[HB]) {
            const unsigned char* wt = smem + st * STAGE_BYTES;
            const int pos = ((4 * kk + fg) ^ (frow & 7)) * 16;
#pragma unroll
            for (int i = 0; i < HB; ++i)
                wf[i] = *reinterpret_cast<const mfma_bf16x8*>(wt + (wn * NB * 16 + (h * HB + i) * 16 + frow) * 128 + pos);
        };
        auto load_a = [&](int st, int kk, mfma_bf16x8 (&af)[MB]) {
            const unsigned char* at = smem + st * STAGE_BYTES + W_BYTES;
            const int pos = ((4 * kk + fg) ^ (frow & 7)) * 16;
#pragma unroll
            for (int j = 0; j < MB; ++j)
                af[j] = *reinterpret_cast<const mfma_bf16x8*>(at + (wm * MB * 16 + j * 16 + frow) * 128 + pos);
        };
#define SR_MFMA_HALF(H, WF, AF)                                                                              \
        _Pragma("unroll") for (int i = 0; i < HB; ++i)                                                       \
            _Pragma("unroll") for (int j = 0; j < MB; ++j)                                                   \
                acc[(H) * HB + i][j] = sr_mma<F16>(WF[i], AF[j], acc[(H) * HB + i][j]);
        load_w(buf, 0, 0, wx);
        load_a(buf, 0, a0);
        int kt = 0;
        // Steady state (all but the last two k-steps): branch-free body with the issue order pinned - each phase's MFMAs
        // start at once and its LDS reads / LDS-DMA pieces are dealt out one per MFMA, so a read has the rest of its phase
        // (and the partner wave's MFMAs) to land before the next phase consumes it.  Left to itself hipcc sinks the reads
        // to just before their first use and waits for them there.
#define SR_SGB(MASK, N, ID) __builtin_amdgcn_sched_group_barrier(MASK, N, ID)
        for (; kt + 2 < nk; ++kt) {
            // reads per phase: 8 (wy, a1) / 4 (wx) / 4 (wy) / 8 (wx, a0 of the next k-step) + the 8 LDS-DMA pieces;
            // a1 is dead from the previous k-step's last phase on, so it is fetched two phases before its first use
            load_w(buf, 0, 1, wy);
            load_a(buf, 1, a1);
            SR_MFMA_HALF(0, wx, a0)
#pragma unroll
            for (int i = 0; i < HB + MB; ++i) { SR_SGB(0x008, 1, 0); SR_SGB(0x100, 1, 0); }
            SR_SGB(0x008, HB * MB - HB - MB, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_w(buf, 1, 0, wx);
            SR_MFMA_HALF(1, wy, a0)
#pragma unroll
            for (int i = 0; i < HB; ++i) { SR_SGB(0x008, 1, 1); SR_SGB(0x100, 1, 1); }
            SR_SGB(0x008, HB * MB - HB, 1);
            __builtin_amdgcn_sched_barrier(0);
            load_w(buf, 1, 1, wy);
            SR_MFMA_HALF(0, wx, a1)
#pragma unroll
            for (int i = 0; i < HB; ++i) { SR_SGB(0x008, 1, 2); SR_SGB(0x100, 1, 2); }
            SR_SGB(0x008, HB * MB - HB, 2);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // LDS-DMA of k-step kt + 1 (see below)
            __syncthreads();
            load_w(buf ^ 1, 0, 0, wx);
            load_a(buf ^ 1, 0, a0);
            stage(buf, (kt + 2) * G_BK);
            SR_MFMA_HALF(1, wy, a1)
#pragma unroll
            for (int i = 0; i < HB + MB; ++i) { SR_SGB(0x008, 1, 3); SR_SGB(0x100, 1, 3); }
#pragma unroll
            for (int i = 0; i < W_INSTR + A_INSTR; ++i) { SR_SGB(0x008, 1, 3); SR_SGB(0x010, 1, 3); }
            __builtin_amdgcn_sched_barrier(0);
            buf ^= 1;
        }
#undef SR_SGB
        for (; kt < nk; ++kt) {
            load_w(buf, 0, 1, wy);
            SR_MFMA_HALF(0, wx, a0)
            load_w(buf, 1, 0, wx);
            load_a(buf, 1, a1);
            SR_MFMA_HALF(1, wy, a0)
            load_w(buf, 1, 1, wy);
            SR_MFMA_HALF(0, wx, a1)
            // The LDS-DMA of k-step kt + 1 was issued in the PREVIOUS iteration: hipcc does not see it as pending here and
            // emits no vmcnt wait for this barrier, so drain it by hand (every wave its own pieces, then the barrier).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 2 < nk) {
                stage(buf, (kt + 2) * G_BK);
            } else if (has_next) {
                if (kt + 2 == nk) set_tile(tile_next);
                stage(buf, (kt + 2 - nk) * G_BK);
            }
            if (kt + 1 < nk) {
                load_w(buf ^ 1, 0, 0, wx);
                load_a(buf ^ 1, 0, a0);
            }
            SR_MFMA_HALF(1, wy, a1)
            buf ^= 1;
        }
#undef SR_MFMA_HALF
    } else {
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {
            stage(buf ^ 1, (kt + 1) * G_BK);
        } else if (CROSS_PREFETCH && has_next) {
            set_tile(tile_next);
            stage(buf ^ 1, 0);
        }
        mfma_bf16x8 wf[NB], af[MB];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            load_frags(buf, kk, wf, af);
            mfma_all(wf, af);
        }
        __syncthreads();
        buf ^= 1;
    }
    }

    if (stamp && titer < 16) stp[titer * 4 + 2] = __builtin_amdgcn_s_memrealtime();
    if (stamp && titer == 1) stp[63] = __builtin_amdgcn_s_memtime();        // ... and at its end: clock = d(memtime) / d(realtime) * 100 MHz
    if constexpr (F16) {       // undo the power-of-two row scales of both operands (exact)
        float sa[MB];
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int m = m0 + wm * MB * 16 + j * 16 + frow;
            sa[j] = m < g.M ? g.a_scale[m] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int n = n0 + wn * NB * 16 + i * 16 + fg * 4;
            f32x4 sw = {0.f, 0.f, 0.f, 0.f};
            if (n < g.N) sw = *reinterpret_cast<const f32x4*>(g.w_scale + n);
#pragma unroll
            for (int j = 0; j < MB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] *= sa[j] * sw[r];
        }
    }
    // ---- epilogues: lane owns token m = .. + (lane & 15), features n = .. + 4 * (lane >> 4) + r
#ifdef KL_NO_STAGING
    constexpr bool KL_STAGED = false;
#else
    constexpr bool KL_STAGED = KL == 1 && (BEPI == EPI_STORE_BF16 || BEPI == EPI_SWIGLU || BEPI == EPI_QKV_ROPE);
#endif
    // The workgroup is persistent: whatever an epilogue computes from the lane index alone (feature offsets, n % head_dim, swizzled
    // LDS offsets ...) is invariant in the TILE loop, and hipcc hoists it above the k-loop and carries it through - behind the
    // asm-pinned four-wave loop that meant accumulators spilled to scratch INSIDE the k-loop (QKV + RoPE epilogues: query encode 300
    // -> 670 ms).  The staged epilogues therefore see the lane coordinates through an opaque asm: nothing of them can be hoisted.
    [[maybe_unused]] int frow_e = frow, fg_e = fg, lane_e = lane;
    if constexpr (KL == 1) asm volatile("" : "+v"(frow_e), "+v"(fg_e), "+v"(lane_e));
    if constexpr (KL_STAGED) {
        // The bf16 outputs of the four-wave tile leave through LDS.  Stored straight from the accumulator layout a lane_e writes
        // 8 bytes and an instruction 16 rows x 32 bytes: 4 096 32-byte fragments per tile and CU, 256 CUs finishing their tiles
        // together - in-kernel stamps put that epilogue at 8 us of a 57 us tile (K = 2048), 0.2 us without the stores.  Here a
        // wave lays the 16 token rows of one block column into LDS behind the two stages and reads them back as whole rows - 16
        // bytes per lane_e, 4 rows x 256 bytes (SwiGLU: 8 rows x 128 bytes) per store instruction: full cache lines, a quarter of
        // the store instructions (8 -> 4.2 us, +6 % on the layer's GEMMs).  LDS operations of one wave execute in order: no barrier.
        constexpr int OUT_F = BEPI == EPI_SWIGLU ? 64 : 128;           // output features per token row of this wave
        constexpr int ROWB = OUT_F * 2, NP = ROWB / 16;                // bytes and 16-byte pieces per staged row
        // two 4 KB buffers per wave (block columns alternate: the conversion and writes of column j + 1 do not wait for the
        // reads of column j), rows unpadded with the 16-byte pieces XOR-swizzled by the row: 2 x 4 x 4 KB = the 32 KB the two
        // stages leave of the CU's 160 KB
        unsigned char* const stg0 = smem + 2 * STAGE_BYTES + wave * 8192;
        const int ldc = BEPI == EPI_SWIGLU ? (g.N >> 1) : g.N;
        const int nb = (BEPI == EPI_SWIGLU ? (n0 >> 1) : n0) + wn * OUT_F;      // first output feature of this wave
        unsigned char* stg = stg0;
        auto put = [&](int f_local, const f32x4& v) {                          // 4 consecutive output features of token row frow_e
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (short)f32_to_bf16(v[r]);
            *reinterpret_cast<bf16x4*>(stg + frow_e * ROWB + (((f_local >> 3) ^ (frow_e & (NP - 1))) << 4) + ((f_local & 4) << 1)) = o;
        };
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int mrow = m0 + wm * MB * 16 + j * 16;
            stg = stg0 + (j & 1) * 4096;
            if constexpr (BEPI == EPI_STORE_BF16) {
#pragma unroll
                for (int i = 0; i < NB; ++i) put(i * 16 + fg_e * 4, acc[i][j]);
            } else if constexpr (BEPI == EPI_SWIGLU) {
#pragma unroll
                for (int i = 0; i < NB; i += 2) {
                    const f32x4 gt = acc[i][j], up = acc[i + 1][j];
                    f32x4 y;
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[r] = (gt[r] * __builtin_amdgcn_rcpf(1.f + __expf(-gt[r]))) * up[r];   // as EPI_SWIGLU below
                    put((i >> 1) * 16 + fg_e * 4, y);
                }
            } else if constexpr (BEPI == EPI_QKV_ROPE) {
                // q / k heads rotated in fp32 on the accumulators (the same sr_rope_lo / sr_rope_hi on the same MFMA chains as the direct-store
                // epilogue below: the same bits), v heads stored as they are.  A wave's 128 features are whole heads (64 or 128 wide) that start
                // at a multiple of 128, so d = n % head_dim needs no division and the rotation partner of block i is block i + head_dim / 32 of
                // the same lane.  Two table loads (L2-resident, 16 bytes each) per pair of blocks; nothing here is invariant in the tile loop
                // except through frow_e / fg_e, which are opaque.
                const int mtok = mrow + frow_e;
                const int pos_t = mtok < g.M ? g.pos[mtok] : 0;
                const int hd2 = g.head_dim >> 1;
                const float* ct = g.rope_cos + (int64_t)pos_t * hd2 + fg_e * 4;
                const float* st = g.rope_sin + (int64_t)pos_t * hd2 + fg_e * 4;
                if (n0 + wn * NB * 16 < g.n_rope) {
                    if (g.head_dim == 64) {
#pragma unroll
                        for (int i = 0; i < NB; ++i) {
                            if ((i & 2) != 0) continue;                  // blocks 2, 3, 6, 7: written with their partners 0, 1, 4, 5
                            const f32x4 c = *reinterpret_cast<const f32x4*>(ct + (i & 1) * 16), sn = *reinterpret_cast<const f32x4*>(st + (i & 1) * 16);
                            const f32x4 x1 = acc[i][j], x2 = acc[i + 2][j];
                            put(i * 16 + fg_e * 4, sr_rope_lo(x1, x2, c, sn));
                            put((i + 2) * 16 + fg_e * 4, sr_rope_hi(x1, x2, c, sn));
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < NB / 2; ++i) {
                            const f32x4 c = *reinterpret_cast<const f32x4*>(ct + i * 16), sn = *reinterpret_cast<const f32x4*>(st + i * 16);
                            const f32x4 x1 = acc[i][j], x2 = acc[i + 4][j];
                            put(i * 16 + fg_e * 4, sr_rope_lo(x1, x2, c, sn));
                            put((i + 4) * 16 + fg_e * 4, sr_rope_hi(x1, x2, c, sn));
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < NB; ++i) put(i * 16 + fg_e * 4, acc[i][j]);
                }
            }
            // rows back out: lane_e -> (token row, 16-byte piece)
            constexpr int LPR = OUT_F * 2 / 16;          // lanes per row (16 or 8)
            constexpr int RPI = 64 / LPR;                // rows per instruction (4 or 8)
#pragma unroll
            for (int it = 0; it < 16 / RPI; ++it) {
                const int tok = it * RPI + lane_e / LPR, pc = lane_e % LPR;
                const sr_i32x4 v = *reinterpret_cast<const sr_i32x4*>(stg + tok * ROWB + ((pc ^ (tok & (NP - 1))) << 4));
                const int m = mrow + tok, n = nb + pc * 8;
                if (m < g.M && n < ldc) *reinterpret_cast<sr_i32x4*>(reinterpret_cast<bf16_t*>(g.C) + (int64_t)m * ldc + n) = v;
            }
            __builtin_amdgcn_sched_barrier(0);       // one block column at a time (bounds what the scheduler keeps in flight)
        }
    } else if constexpr (KL == 1 && (BEPI == EPI_STORE_F32 || BEPI == EPI_RESID_F32)) {
        // fp32 outputs of the four-wave tile (the fp32 regime's residual adds), staged through LDS like the
        // bf16 ones: a token row of this wave is 128 features x 4 B = 512 B, handled as two halves of 64 features (16 rows x 256 B
        // = one 4 KB buffer, the two buffers alternate); read back 16 bytes per lane_e, 4 rows x 256 bytes per instruction.
        // EPI_RESID_F32 adds into x with 16-byte loads and stores of whole lines instead of 16-row x 64-byte fragments.
        unsigned char* const stg0 = smem + 2 * STAGE_BYTES + wave * 8192;
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int mrow = m0 + wm * MB * 16 + j * 16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                unsigned char* stg = stg0 + h * 4096;
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int i = 4 * h + ii;
                    f32x4 y = acc[i][j];
                    *reinterpret_cast<f32x4*>(stg + frow_e * 256 + (((ii * 4 + fg_e) ^ frow_e) << 4)) = y;
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int tok = it * 4 + (lane_e >> 4), pc = lane_e & 15;
                    f32x4 v = *reinterpret_cast<const f32x4*>(stg + tok * 256 + ((pc ^ tok) << 4));
                    const int m = mrow + tok, n = n0 + wn * NB * 16 + h * 64 + pc * 4;
                    if (m < g.M && n < g.N) {
                        f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + (int64_t)m * g.N + n);
                        if constexpr (BEPI == EPI_RESID_F32) {
                            f32x4 x = *dst;
                            x += v;
                            v = x;
                        }
                        *dst = v;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);       // see the bf16 version above
            }
        }
    } else if constexpr (KL == 1 && BEPI == EPI_SWIGLU_SPLITH_BASE) {
        // fp32 regime, SwiGLU output as fp16 plane segments [f1 | f0 | f0] (see the direct-store version below), staged: a token
        // row of this wave is 64 output features = 128 B per plane; a buffer holds 16 rows x (f1 | f0) = 4 KB
        unsigned char* const stg0 = smem + 2 * STAGE_BYTES + wave * 8192;
        const int half_n = g.N >> 1;
        const int64_t ldc = 3 * (int64_t)half_n;
        const int nb = (n0 >> 1) + wn * 64;
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int mrow = m0 + wm * MB * 16 + j * 16;
            unsigned char* stg = stg0 + (j & 1) * 4096;
            const float osc = (mrow + frow_e) < g.M ? g.out_scale[mrow + frow_e] : 0.f;
#pragma unroll
            for (int i = 0; i < NB; i += 2) {
                const f32x4 gt = acc[i][j], up = acc[i + 1][j];
                bf16x4 p0, p1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y = (gt[r] / (1.f + expf(-gt[r]))) * up[r];
                    unsigned short f0, f1;
                    split_f16x2(y * osc, f0, f1);
                    p0[r] = (short)f0; p1[r] = (short)f1;
                }
                // 8-byte slot s = (i / 2) * 4 + fg_e of the row's 16 (f1) + 16 (f0): 16-byte piece s / 2, XOR-swizzled by the row
                const int sl = (i >> 1) * 4 + fg_e;
                *reinterpret_cast<bf16x4*>(stg + frow_e * 256 + ((((sl >> 1)) ^ (frow_e & 7)) << 4) + ((sl & 1) << 3)) = p1;
                *reinterpret_cast<bf16x4*>(stg + frow_e * 256 + 128 + ((((sl >> 1)) ^ (frow_e & 7)) << 4) + ((sl & 1) << 3)) = p0;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                // lane_e -> (row, plane, 16-byte piece): 4 rows x (8 pieces of f1 + 8 of f0) per instruction
                const int tok = it * 4 + (lane_e >> 4), pl = (lane_e >> 3) & 1, pc = lane_e & 7;
                const sr_i32x4 v = *reinterpret_cast<const sr_i32x4*>(stg + tok * 256 + pl * 128 + ((pc ^ (tok & 7)) << 4));
                const int m = mrow + tok, n = nb + pc * 8;
                if (m < g.M && n < half_n) {
                    bf16_t* crow = reinterpret_cast<bf16_t*>(g.C) + (int64_t)m * ldc + n;
                    if (pl == 0) {
                        *reinterpret_cast<sr_i32x4*>(crow) = v;                    // f1
                    } else {
                        *reinterpret_cast<sr_i32x4*>(crow + half_n) = v;           // f0, twice
                        *reinterpret_cast<sr_i32x4*>(crow + 2 * half_n) = v;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (BEPI == EPI_STORE_BF16 || BEPI == EPI_STORE_F32 || BEPI == EPI_RESID_F32) {
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int m = m0 + wm * MB * 16 + j * 16 + frow;
            if (m >= g.M) continue;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int n = n0 + wn * NB * 16 + i * 16 + fg * 4;
                if (n >= g.N) continue;
                const f32x4 v = acc[i][j];
                if constexpr (BEPI == EPI_STORE_BF16) {
                    bf16x4 o;
                    o[0] = (short)f32_to_bf16(v[0]);
                    o[1] = (short)f32_to_bf16(v[1]);
                    o[2] = (short)f32_to_bf16(v[2]);
                    o[3] = (short)f32_to_bf16(v[3]);
#ifdef KL_DIAG_NOSTORE      // timing diagnostic (variant builds only): the epilogue without its stores (one lane keeps the values alive)
                    if (o[0] == 12345 && o[1] == 23456 && lane == 77)
#endif
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(g.C) + (int64_t)m * g.N + n) = o;
                } else if constexpr (BEPI == EPI_STORE_F32) {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + (int64_t)m * g.N + n) = v;
                } else {
                    f32x4* p = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + (int64_t)m * g.N + n);
                    f32x4 x = *p;
                    x += v;
                    *p = x;
                }
            }
        }
    } else if constexpr (BEPI == EPI_QKV_ROPE || BEPI == EPI_QKV_ROPE_F32) {
        // q/k heads: out[d] = x[d] cos - x[d + hd/2] sin, out[d + hd/2] = x[d + hd/2] cos + x[d] sin (HF rotate_half),
        // in fp32 on the accumulators; both halves of a head live in this lane (blocks i and i + hd/32).
        // A wave's feature range (16 * NB) covers whole heads: NB * 16 % head_dim == 0 is checked at launch.
        // EPI_QKV_ROPE_F32 (fp32 regime) stores the rotated fp32 values as they are.
        constexpr bool F32OUT = BEPI == EPI_QKV_ROPE_F32;
        const int hd = g.head_dim, hb = hd / 32;   // hb = block distance between rotation partners
        auto put = [&](int64_t off, const f32x4& v) {
            if constexpr (F32OUT) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + off) = v;
            } else {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (short)f32_to_bf16(v[r]);
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(g.C) + off) = o;
            }
        };
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int m = m0 + wm * MB * 16 + j * 16 + frow;
            if (m >= g.M) continue;
            const int p = g.pos[m];
            const int64_t crow = (int64_t)m * g.N;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int n = n0 + wn * NB * 16 + i * 16 + fg * 4;
                if (n >= g.N) continue;
                if (n < g.n_rope) {
                    const int d = n % hd;
                    if (d >= hd / 2) continue;                 // written together with its first-half partner
                    // partner block index is a compile-time offset only when unrolled over i: handle both head sizes
                    f32x4 x1 = acc[i][j], x2;
                    if (hb == 2) x2 = acc[(i + 2) % NB][j]; else x2 = acc[(i + 4) % NB][j];
                    const f32x4 c = *reinterpret_cast<const f32x4*>(g.rope_cos + (int64_t)p * (hd / 2) + d);
                    const f32x4 sn = *reinterpret_cast<const f32x4*>(g.rope_sin + (int64_t)p * (hd / 2) + d);
                    put(crow + n, sr_rope_lo(x1, x2, c, sn));
                    put(crow + n + hd / 2, sr_rope_hi(x1, x2, c, sn));
                } else {
                    put(crow + n, acc[i][j]);
                }
            }
        }
    } else if constexpr (BEPI == EPI_SWIGLU) {
        const int half_n = g.N >> 1;
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int m = m0 + wm * MB * 16 + j * 16 + frow;
            if (m >= g.M) continue;
#pragma unroll
            for (int i = 0; i < NB; i += 2) {
                const int n = (n0 >> 1) + wn * NB * 8 + (i >> 1) * 16 + fg * 4;
                if (n >= half_n) continue;
                const f32x4 gt = acc[i][j], up = acc[i + 1][j];
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sg = gt[r] * __builtin_amdgcn_rcpf(1.f + __expf(-gt[r]));   // silu; rcp is ~1 ulp, output is bf16
                    o[r] = (short)f32_to_bf16(sg * up[r]);
                }
                *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(g.C) + (int64_t)m * half_n + n) = o;
            }
        }
    } else if constexpr (BEPI == EPI_SWIGLU_F32) {
        const int half_n = g.N >> 1;
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int m = m0 + wm * MB * 16 + j * 16 + frow;
            if (m >= g.M) continue;
#pragma unroll
            for (int i = 0; i < NB; i += 2) {
                const int n = (n0 >> 1) + wn * NB * 8 + (i >> 1) * 16 + fg * 4;
                if (n >= half_n) continue;
                const f32x4 gt = acc[i][j], up = acc[i + 1][j];
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = (gt[r] / (1.f + expf(-gt[r]))) * up[r];
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + (int64_t)m * half_n + n) = y;
            }
        }
    } else if constexpr (BEPI == EPI_SWIGLU_SPLITH_BASE) {
        // fp32 regime on fp16 planes: silu(gate) * up (accurate exp, true division), scaled by the row's power of two and
        // stored as [f1 | f0 | f0]: the fp32 intermediate and its row-split pass (112 -> 48 bytes of traffic per token and
        // feature) are gone.  out_scale[m] comes from the bound |silu(g) u| <= |xn|^2 max_j |wg_j||wu_j|: no overflow.
        const int half_n = g.N >> 1;
        const int64_t ldc = 3 * (int64_t)half_n;
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int m = m0 + wm * MB * 16 + j * 16 + frow;
            if (m >= g.M) continue;
            const float osc = g.out_scale[m];
#pragma unroll
            for (int i = 0; i < NB; i += 2) {
                const int n = (n0 >> 1) + wn * NB * 8 + (i >> 1) * 16 + fg * 4;
                if (n >= half_n) continue;
                const f32x4 gt = acc[i][j], up = acc[i + 1][j];
                bf16x4 p0, p1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y = (gt[r] / (1.f + expf(-gt[r]))) * up[r];
                    unsigned short f0, f1;
                    split_f16x2(y * osc, f0, f1);
                    p0[r] = (short)f0; p1[r] = (short)f1;
                }
                bf16_t* crow = reinterpret_cast<bf16_t*>(g.C) + (int64_t)m * ldc + n;
                *reinterpret_cast<bf16x4*>(crow) = p1;
                *reinterpret_cast<bf16x4*>(crow + half_n) = p0;
                *reinterpret_cast<bf16x4*>(crow + 2 * half_n) = p0;
            }
        }
    } else if constexpr (BEPI == EPI_SWIGLU_SPLIT) {
        // fp32 regime: silu(gate) * up with an accurate exp and a true division, then stored as the split-bf16 plane
        // segments the down_proj GEMM consumes: C [M, n_seg * N/2], segment sg holds plane out_map.plane[sg]
        const int half_n = g.N >> 1;
        const int nsg = g.out_map.n_seg;
        const int64_t ldc = (int64_t)nsg * half_n;
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int m = m0 + wm * MB * 16 + j * 16 + frow;
            if (m >= g.M) continue;
#pragma unroll
            for (int i = 0; i < NB; i += 2) {
                const int n = (n0 >> 1) + wn * NB * 8 + (i >> 1) * 16 + fg * 4;
                if (n >= half_n) continue;
                const f32x4 gt = acc[i][j], up = acc[i + 1][j];
                bf16x4 pl0, pl1, pl2;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y = (gt[r] / (1.f + expf(-gt[r]))) * up[r];
                    unsigned short a0, a1, a2;
                    split_bf16x3(y, a0, a1, a2);
                    pl0[r] = (short)a0; pl1[r] = (short)a1; pl2[r] = (short)a2;
                }
                bf16_t* crow = reinterpret_cast<bf16_t*>(g.C) + (int64_t)m * ldc + n;
                for (int sg = 0; sg < nsg; ++sg) {
                    const int pi = g.out_map.plane[sg];
                    *reinterpret_cast<bf16x4*>(crow + (int64_t)sg * half_n) = pi == 0 ? pl0 : (pi == 1 ? pl1 : pl2);
                }
            }
        }
    } else {  // EPI_SEGMAX: per-sequence max over this tile's token rows, straight from the accumulators
        // A lane owns token rows (lane & 15) + 16 j of its wave's slab and, per row, features 4 (lane >> 4) + r of each
        // block i: the rows of one feature sit in the 16 lanes of a DPP row (and in j).  Sequence ids ascend along the
        // rows (-2 = masked token, -1 = past the last token), so for every sequence q present in the slab - usually one
        // or two - each lane maxes its own rows of q, four DPP steps finish the max over the 16 lanes, and the lanes of
        // the row share out the (positive) results and fold them into out[q][n] with integer atomicMax - all 64 lanes
        // active per atomic instruction (a wave stalls once ~16 atomics are outstanding).  No LDS, no barrier: the
        // staging buffers stay free for the cross-tile prefetch.
        int lo = 0x7fffffff, hi = -1;
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            if (sj[j] >= 0) {
                lo = sj[j] < lo ? sj[j] : lo;
                hi = sj[j] > hi ? sj[j] : hi;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int l2 = __shfl_xor(lo, off), h2 = __shfl_xor(hi, off);
            lo = l2 < lo ? l2 : lo;
            hi = h2 > hi ? h2 : hi;
        }
        float* out = reinterpret_cast<float*>(g.C);
        auto row_max = [](float v) {
            v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));    // quad_perm [1,0,3,2]
            v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));    // quad_perm [2,3,0,1]
            v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));   // row_half_mirror
            v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));   // row_mirror
            return v;
        };
        constexpr int PER_LANE = NB * 4 / 16;       // after the row reduction all 16 lanes hold every maximum: lane f keeps
        static_assert(NB * 4 % 16 == 0, "");        // those of the (block, register) pairs f, f + 16, ... and issues their atomics
        for (int q = lo; q <= hi; ++q) {       // wave-uniform
            float mine[PER_LANE];
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = 0.f;             // log(1 + relu(x)): only positive maxima matter
#pragma unroll
                    for (int j = 0; j < MB; ++j) v = (sj[j] == q) ? fmaxf(v, acc[i][j][r]) : v;
                    v = row_max(v);
                    const int idx = i * 4 + r;
                    if ((idx & 15) == frow) mine[idx >> 4] = v;
                }
#pragma unroll
            for (int k = 0; k < PER_LANE; ++k) {
                const int idx = frow + 16 * k;
                const int n = n0 + wn * NB * 16 + (idx >> 2) * 16 + fg * 4 + (idx & 3);
                if (mine[k] > 0.f && n < g.N)
                    atomicMax(reinterpret_cast<int*>(out + (int64_t)q * g.out_ld + n), __float_as_int(mine[k]));
            }
        }
    }
    if (stamp && titer < 16) {
        __builtin_amdgcn_s_waitcnt(0);   // diagnostic build path only: let this wave's stores retire before the stamp
        stp[titer * 4 + 3] = __builtin_amdgcn_s_memrealtime();
        if (titer + 1 < 16) stp[(titer + 1) * 4] = stp[titer * 4 + 3];
    }
    ++titer;
    if constexpr (PIPE && NS > 2) {
        if (has_next) {                 // no cross-tile prefetch with NS stages: stage the next tile's first k-steps now
            __syncthreads();
            set_tile(tile_next);
#pragma unroll
            for (int st = 0; st < NS; ++st) stage(st, st * G_BK);
            __syncthreads();
            buf = 0;
        }
    }
    tile = tile_next;
    }  // tile loop
}

template <int EPI, int WAVES_N, int WAVES_M, int NB, int MB, bool PIPE = false, int NS = 2, int KL = 0>
static int launch_cfg(const GemmArgs& g_in, hipStream_t s) {
    constexpr int BN = 16 * NB * WAVES_N, BM = 16 * MB * WAVES_M;
    constexpr size_t lds = NS * (size_t)(BN + BM) * 128 + (KL == 1 ? 4 * 8192 : 0);     // KL = 1: + the waves' output staging (160 KB in all)
    static DeviceOnce attr_once;
    bool* attr_slot = attr_once.pending();
    if (attr_slot) {
        SR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<EPI, WAVES_N, WAVES_M, NB, MB, PIPE, NS, KL>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        *attr_slot = true;
    }
    GemmArgs g = g_in;
    // Tile order.  Feature-tile-fastest keeps an XCD on the same 1/8 of W (good while W fits the Infinity Cache).  A weight
    // matrix far larger than that (the 525 MB vocabulary head) would be streamed from HBM once per token tile: walk the
    // token tiles fastest instead, so that a W tile is used by all of them while it is hot and W is read once.
    g.m_fastest = ((int64_t)g.N * g.K * 2 > (128ll << 20) && (int64_t)g.N > 4 * (int64_t)g.M) ? 1 : 0;
    if (const char* e = sr_dev_getenv("SR_GEMM_MFAST")) g.m_fastest = atoi(e);     // A/B switch
    g.xcd_order = 1;
    if (const char* e = sr_dev_getenv("SR_GEMM_XCD")) g.xcd_order = atoi(e);       // A/B switch
    int64_t tiles = ceil_div64(g.N, BN) * ceil_div64(g.M, BM);
    const char* env = sr_dev_getenv("SR_GEMM_PERSIST");            // A/B switch: 0 = one workgroup per tile
    // resident workgroups on 256 CUs: one 8-wave workgroup, up to 3 of 4 waves, up to 8 single-wave ones (LDS permitting)
    constexpr int64_t by_lds = (160 * 1024) / lds;
    const int64_t slots = 256 * (WAVES_N * WAVES_M == 4 ? (by_lds > 3 ? 3 : by_lds) : (WAVES_N * WAVES_M == 1 ? (by_lds > 8 ? 8 : by_lds) : 1));
    if (!(env && *env == '0') && tiles > slots) tiles = slots;
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI, WAVES_N, WAVES_M, NB, MB, PIPE, NS, KL>), dim3((unsigned)tiles), dim3(64 * WAVES_N * WAVES_M), lds,
                       s, g);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

// ---- tile plan --------------------------------------------------------------------------------------
// Workgroups are persistent and take tiles round-robin, so a launch costs ceil(tiles / resident workgroups) rounds
// and a last round that is mostly empty is paid in full (M = 9600 tokens x N = 2048: 304 tiles of 256^2 on 256 CUs = 2
// rounds for 1.19 rounds of work).  The plan cuts the token rows once: rows [0, m_big) run as whole rounds of 256^2
// tiles, the rest as 128^2 tiles (two resident per CU, a quarter of the work each).  Every output element is still
// one k-ordered MFMA chain, so the result does not depend on the cut.
// Costs in units of one 256^2 round, from tools/quick_gemm_bench.py (1120 vs 920 TFLOP/s at M = 38400): a full round
// of 512 128^2 tiles is half the FLOPs of a 256^2 round at 1.22 x the time per FLOP; with at most one 128^2
// workgroup per CU a round finishes sooner.
static double plan_cost(int64_t big_tiles, int64_t small_tiles) {
    double c = (double)ceil_div64(big_tiles, 256);
    if (small_tiles > 0) {
        const int64_t full = small_tiles / 512, rest = small_tiles % 512;
        c += 0.61 * (double)full + (rest == 0 ? 0.0 : (rest <= 256 ? 0.45 : 0.61));
        c += 0.03;                       // second launch
    }
    return c;
}

// rows of g.M handled with 256^2 tiles (0 = none, g.M = all)
static int plan_big_rows(const GemmArgs& g, bool small_allowed) {
    const char* env = sr_dev_getenv("SR_GEMM_TILE");   // test / A-B switch: 128 | 256 | split (read per call)
    if (env && *env) {
        if (atoi(env) == 256) return g.M;
        if (atoi(env) == 128) return 0;
        if (!strncmp(env, "split:", 6)) {          // split:<rows> forces the cut (tests)
            const int rows = atoi(env + 6) / 256 * 256;
            return rows < g.M ? rows : g.M;
        }
    }
    if (!small_allowed) return g.M;
    const int64_t tn256 = ceil_div64(g.N, 256), tn128 = ceil_div64(g.N, 128);
    int best_rows = 0;
    double best = plan_cost(0, tn128 * ceil_div64(g.M, 128));
    if (tn256 * ceil_div64(g.M, 256) >= 128) {              // a handful of 256^2 tiles cannot fill the chip
        const double all_big = plan_cost(tn256 * ceil_div64(g.M, 256), 0);
        if (all_big <= best) { best = all_big; best_rows = g.M; }
        for (int rows = 256; rows < g.M; rows += 256) {
            const double c = plan_cost(tn256 * (rows / 256), tn128 * ceil_div64(g.M - rows, 128));
            if (c < best - 1e-9) { best = c; best_rows = rows; }
        }
    }
    return best_rows;
}

template <int EPI>
static int launch_big(const GemmArgs& g, hipStream_t s) {
    if constexpr (EPI == EPI_STORE_BF16 || EPI == EPI_SWIGLU || EPI == EPI_QKV_ROPE) {
        // A/B switch: SR_GEMM_BIG=s3 -> 256 x 128 tile, 8 waves, THREE 48 KB LDS stages (LDS-DMA issued three k-steps ahead)
        const char* big = sr_dev_getenv("SR_GEMM_BIG");
        if (big && big[0] == 's' && big[1] == '3' && g.K / G_BK >= 4) {
            if constexpr (EPI == EPI_QKV_ROPE) {
                if (g.head_dim == 64) return launch_cfg<EPI, 2, 4, 8, 2, true, 3>(g, s);
            } else {
                return launch_cfg<EPI, 2, 4, 8, 2, true, 3>(g, s);
            }
        }
    }
    const char* env = sr_dev_getenv("SR_GEMM_PIPE");   // A/B switch: 0 = plain double-buffered loop
    {
        // The four-wave loop is the default for the bf16 regime's plain-store / residual / SwiGLU GEMMs (o_proj, down_proj, gate-up:
        // 88 % of a layer's GEMM work; +7-10 % on those, corpus encode 7 590 -> 8 390 passages/s).  SR_GEMM_BIG=8w: the 8-wave loop
        // everywhere (A/B); SR_GEMM_BIG=4w: the four-wave loop also for the fp32 regime's fp16-plane GEMMs, which have staged
        // epilogues too but measure 2 % SLOWER on it (query encode 300 -> 307 ms: K' = 3 K k-steps per tile, the epilogue does
        // not matter there and the 8-wave loop's second wave per SIMD does) - tests force it to keep that path covered.
        // The bf16 regime's QKV + RoPE epilogue (12 % of a layer's GEMM work) joined in round 6: rotation on the accumulators, staged
        // through LDS like the plain store.  The fp32 regime's QKV epilogues (fp32 output, fp16-plane operands) stay on the 8-wave loop.
        // tests/test_abi.py holds every four-wave instantiation to zero scratch.  The loop's buffer descriptors address a tile's
        // rows with 32-bit byte offsets: 256 rows x 2 K bytes must stay below 2^31.
        const char* big = sr_dev_getenv("SR_GEMM_BIG");
        const bool want8 = big && big[0] == '8', want4 = big && big[0] == '4';
        constexpr bool STAGED_BF16 = EPI == EPI_STORE_BF16 || EPI == EPI_SWIGLU || EPI == EPI_STORE_F32 || EPI == EPI_RESID_F32 || EPI == EPI_QKV_ROPE;
        constexpr bool STAGED_F16 = EPI == EPI_RESID_F32_H || EPI == EPI_SWIGLU_SPLIT_H;
        if constexpr (STAGED_BF16 || STAGED_F16) {
            // QKV + RoPE: a wave's 128 features must lie on one side of n_rope (two heads of 64 per wave)
            const bool rope_ok = EPI != EPI_QKV_ROPE || g.n_rope % 128 == 0;
            if (g.K / G_BK >= 4 && (int64_t)g.K * 512 < (1ll << 31) && !want8 && (STAGED_BF16 || want4) && rope_ok && !(env && *env == '0'))
                return launch_cfg<EPI, 2, 2, 8, 8, true, 2, 1>(g, s);
        }
    }
    if (g.K / G_BK >= 4 && !(env && *env == '0')) return launch_cfg<EPI, 2, 4, 8, 4, true>(g, s);
    return launch_cfg<EPI, 2, 4, 8, 4, false>(g, s);
}

static bool env_off(const char* name) {
    const char* e = sr_dev_getenv(name);
    return e && *e == '0';
}

template <int EPI>
static int launch_small(const GemmArgs& g, hipStream_t s) {
    const bool pipe = g.K / G_BK >= 4 && !env_off("SR_GEMM_PIPE");      // the pipelined k-loop needs >= 4 k-steps
    if constexpr (EPI == EPI_QKV_ROPE || EPI == EPI_QKV_ROPE_F32 || EPI == EPI_QKV_ROPE_F32_H) {
        if (g.head_dim == 128) return launch_cfg<EPI, 1, 4, 8, 2>(g, s);        // 128 x 128 tile, wave = 128 features x 32 tokens
    }
    if constexpr (EPI != EPI_QKV_ROPE && EPI != EPI_QKV_ROPE_F32 && EPI != EPI_QKV_ROPE_F32_H) {
        // few 128^2 tiles (a short tail behind the 256^2 rounds, or a small problem): halve the token tile so that two or
        // three workgroups share every CU instead of one 4-wave workgroup idling half its MFMA pipe
        const int64_t t128 = ceil_div64(g.N, 128) * ceil_div64(g.M, 128);
        if (t128 < 384 && g.M > 64 && !env_off("SR_GEMM_TAIL64"))
            return pipe ? (env_off("SR_GEMM_NS3") ? launch_cfg<EPI, 2, 2, 4, 2, true>(g, s) : launch_cfg<EPI, 2, 2, 4, 2, true, 3>(g, s))
                        : launch_cfg<EPI, 2, 2, 4, 2>(g, s);
    }
    return pipe ? launch_cfg<EPI, 2, 2, 4, 4, true>(g, s) : launch_cfg<EPI, 2, 2, 4, 4>(g, s);
}

// the token rows [row0, M) of the same problem
template <int EPI>
static GemmArgs rows_from(const GemmArgs& g, int row0) {
    GemmArgs t = g;
    t.A = g.A + (int64_t)row0 * g.K;
    t.M = g.M - row0;
    const int64_t ldc = (EPI == EPI_SWIGLU || EPI == EPI_SWIGLU_F32 || EPI == EPI_SWIGLU_F32_H)
                            ? g.N / 2 : (EPI == EPI_SWIGLU_SPLIT ? (int64_t)g.out_map.n_seg * (g.N / 2) : (EPI == EPI_SWIGLU_SPLIT_H ? 3 * (int64_t)(g.N / 2) : g.N));
    const int64_t esz = (EPI == EPI_STORE_F32 || EPI == EPI_RESID_F32 || EPI == EPI_QKV_ROPE_F32 || EPI == EPI_SWIGLU_F32 ||
                         (EPI >= EPI_H_FIRST && EPI != EPI_SWIGLU_SPLIT_H)) ? 4 : 2;
    if constexpr (EPI != EPI_SEGMAX && EPI != EPI_SEGMAX_H) t.C = reinterpret_cast<unsigned char*>(g.C) + (int64_t)row0 * ldc * esz;
    if (g.a_scale) t.a_scale = g.a_scale + row0;
    if (g.out_scale) t.out_scale = g.out_scale + row0;
    if (g.seq_of) t.seq_of = g.seq_of + row0;
    if (g.pos) t.pos = g.pos + row0;
    t.stamps = nullptr;
    return t;
}

// token rows up to which the single-wave streaming configurations are used (SR_GEMM_SKINNY = 0 disables, = n overrides)
static int skinny_max_rows() {
    const char* e = sr_dev_getenv("SR_GEMM_SKINNY");
    if (!e || !*e) return 64;
    return atoi(e);
}

template <int EPI>
static int launch_one(const GemmArgs& g, hipStream_t s) {
    if constexpr (EPI == EPI_QKV_ROPE || EPI == EPI_QKV_ROPE_F32 || EPI == EPI_QKV_ROPE_F32_H) {
        SR_REQUIRE(g.head_dim == 64 || g.head_dim == 128, "gemm(qkv+rope): head_dim %d not supported", g.head_dim);
        SR_REQUIRE(g.pos && g.rope_cos && g.rope_sin && g.n_rope % g.head_dim == 0 && g.n_rope <= g.N && g.N % g.head_dim == 0,
                   "gemm(qkv+rope): bad rope arguments");
    }
    if (g.M <= skinny_max_rows() && g.K / G_BK >= 4) {
        if (g.M > 32) {
            if constexpr (EPI == EPI_QKV_ROPE || EPI == EPI_QKV_ROPE_F32 || EPI == EPI_QKV_ROPE_F32_H) {
                if (g.head_dim == 128) return launch_cfg<EPI, 1, 1, 8, 4, true, 3>(g, s);
            }
            return launch_cfg<EPI, 1, 1, 4, 4, true, 4>(g, s);
        }
        if (g.M > 16) {
            if constexpr (EPI == EPI_QKV_ROPE || EPI == EPI_QKV_ROPE_F32 || EPI == EPI_QKV_ROPE_F32_H) {
                if (g.head_dim == 128) return launch_cfg<EPI, 1, 1, 8, 2, true, 4>(g, s);
            }
            return launch_cfg<EPI, 1, 1, 4, 2, true, 4>(g, s);
        }
        // A handful of tokens (online queries): the work is streaming W once.  One WAVE per workgroup owns 64 features x 16
        // tokens (N / 64 independent workgroups instead of N / 128 four-wave ones idling on a 16-token tile), four 10 KB LDS
        // stages keep three k-steps of weights in flight per wave.  Same MFMA chain per output element as every other tile.
        if constexpr (EPI == EPI_QKV_ROPE || EPI == EPI_QKV_ROPE_F32 || EPI == EPI_QKV_ROPE_F32_H) {
            if (g.head_dim == 128) return launch_cfg<EPI, 1, 1, 8, 1, true, 4>(g, s);
        }
        return launch_cfg<EPI, 1, 1, 4, 1, true, 4>(g, s);
    }
    {
        const int big_rows = plan_big_rows(g, true);
        if (big_rows >= g.M) return launch_big<EPI>(g, s);
        if (big_rows <= 0) return launch_small<EPI>(g, s);
        GemmArgs head = g;
        head.M = big_rows;
        SR_TRY(launch_big<EPI>(head, s));
        return launch_small<EPI>(rows_from<EPI>(g, big_rows), s);
    }
}

int launch_gemm_bf16(GemmEpilogue epi, const GemmArgs& g, hipStream_t s) {
    SR_REQUIRE(g.M >= 0 && g.N > 0 && g.K > 0, "gemm: bad shape M=%d N=%d K=%d", g.M, g.N, g.K);
    if (g.M == 0) return SR_OK;
    SR_REQUIRE(g.K % G_BK == 0, "gemm: K=%d must be a multiple of %d", g.K, G_BK);
    const bool swiglu = epi == EPI_SWIGLU || epi == EPI_SWIGLU_SPLIT || epi == EPI_SWIGLU_F32 || epi == EPI_SWIGLU_F32_H || epi == EPI_SWIGLU_SPLIT_H;
    SR_REQUIRE(epi != EPI_SWIGLU_SPLIT_H || g.out_scale, "gemm(swiglu split, fp16 planes): missing output row scales");
    SR_REQUIRE(g.N % 16 == 0 && (!swiglu || g.N % 32 == 0), "gemm: N=%d must be a multiple of 16 (32 for SwiGLU)", g.N);
    SR_REQUIRE(epi < EPI_H_FIRST || (g.a_scale && g.w_scale), "gemm(fp16 planes): missing row scales");
    SR_REQUIRE(epi != EPI_SWIGLU_SPLIT || (g.out_map.n_seg >= 1 && g.out_map.n_seg <= SR_MAX_SEG), "gemm(swiglu split): bad segment map");
    switch (epi) {
        case EPI_STORE_BF16: return launch_one<EPI_STORE_BF16>(g, s);
#ifndef SR_GEMM_VARIANT_BUILD      // tools/micro/gemm_variants.sh: k-step schedule variants of the plain-store GEMM only (seconds to build)
        case EPI_RESID_F32: return launch_one<EPI_RESID_F32>(g, s);
        case EPI_SWIGLU: return launch_one<EPI_SWIGLU>(g, s);
        case EPI_SEGMAX: return launch_one<EPI_SEGMAX>(g, s);
        case EPI_STORE_F32: return launch_one<EPI_STORE_F32>(g, s);
        case EPI_QKV_ROPE: return launch_one<EPI_QKV_ROPE>(g, s);
        case EPI_QKV_ROPE_F32: return launch_one<EPI_QKV_ROPE_F32>(g, s);
        case EPI_SWIGLU_SPLIT: return launch_one<EPI_SWIGLU_SPLIT>(g, s);
        case EPI_SWIGLU_F32: return launch_one<EPI_SWIGLU_F32>(g, s);
        case EPI_QKV_ROPE_F32_H: return launch_one<EPI_QKV_ROPE_F32_H>(g, s);
        case EPI_RESID_F32_H: return launch_one<EPI_RESID_F32_H>(g, s);
        case EPI_SWIGLU_F32_H: return launch_one<EPI_SWIGLU_F32_H>(g, s);
        case EPI_SEGMAX_H: return launch_one<EPI_SEGMAX_H>(g, s);
        case EPI_SWIGLU_SPLIT_H: return launch_one<EPI_SWIGLU_SPLIT_H>(g, s);
#endif
        default: break;
    }
    sr_set_error("gemm: unknown epilogue %d", (int)epi);
    return SR_ERR_INVALID;
}
