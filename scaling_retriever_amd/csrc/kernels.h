// Internal launch interface of the encoder kernels (gfx950).
#pragma once
#include "common.h"

typedef unsigned short bf16_t;  // raw bf16 bits

enum GemmEpilogue {
    EPI_STORE_BF16 = 0,  // C[M,N] bf16 = acc
    EPI_RESID_F32 = 1,   // X[M,N] fp32 += acc            (o_proj / down_proj + residual)
    EPI_SWIGLU = 2,      // C[M,N/2] bf16 = silu(gate) * up, W rows interleaved gate/up in 16-row blocks
    EPI_SEGMAX = 3,      // out[seq_of[m], n] = max(out, acc) over the tokens of each sequence (sparse head)
    EPI_STORE_F32 = 4,   // C[M,N] fp32 = acc (tests)
    EPI_QKV_ROPE = 5,    // C[M,N] bf16 = acc with RoPE applied (fp32) to features n < n_rope (q and k heads)
    EPI_QKV_ROPE_F32 = 6,   // same, C[M,N] fp32 (fp32 regime: the attention reads fp32 operands)
    EPI_SWIGLU_SPLIT = 7,   // fp32 regime: silu(gate) * up in fp32, stored as split-bf16 plane segments (out_map)
    EPI_SWIGLU_F32 = 8,     // C[M,N/2] fp32 = silu(gate) * up (accurate exp, true division)
    // fp32 regime on fp16 planes: operands are [rows, 3K] fp16 plane segments of rows scaled by a power of two, the MFMA is
    // v_mfma_f32_16x16x32_f16 and the accumulators are multiplied by a_scale[m] * w_scale[n] (the inverse scales) before the
    // epilogue of the base behaviour runs
    EPI_H_FIRST = 9,
    EPI_QKV_ROPE_F32_H = 9, EPI_RESID_F32_H = 10, EPI_SWIGLU_F32_H = 11, EPI_SEGMAX_H = 12,
    // silu(gate) * up in fp32, multiplied by out_scale[m] (a power of two from a rigorous row bound, so that no value can
    // overflow fp16) and stored directly as the [f1 | f0 | f0] fp16 plane segments the down_proj GEMM consumes: C [M, 3 N/2]
    EPI_SWIGLU_SPLIT_H = 13,
    EPI_SWIGLU_SPLITH_BASE = 14,    // internal: the base behaviour of EPI_SWIGLU_SPLIT_H
};

// ---- fp32 regime: fp32 operands as sums of bf16 planes ---------------------------------------------------
// x = p0 + p1 + p2, each plane the bf16 rounding of what the previous ones left (the subtractions are exact in fp32),
// so three planes carry the whole 24-bit significand.  A fp32-class product a.w is the sum of plane products
// a_i . w_j; laid out along K as segments it is ONE bf16 GEMM with K' = n_seg * K (fp32 accumulate in the MFMA):
//   A' = [a_{pa[0]} | a_{pa[1]} | ...]   W' = [w_{pw[0]} | w_{pw[1]} | ...]      (smallest terms first)
//   3 planes, 6 products (error below fp32 epsilon):  pa = 2 0 1 1 0 0   pw = 0 2 1 0 1 0
//   2 planes, 3 products (~2^-17 relative):           pa = 1 0 0         pw = 0 1 0
#define SR_MAX_SEG 6
struct SplitMap {
    int n_seg;
    int plane[SR_MAX_SEG];
};
//   fp16 planes (SR_FP32_PLANES_F16 = 16): two fp16 planes of power-of-two scaled rows, the same 3-product maps; 11 + 11
//   significand bits: truncation 3 * 2^-22, below the fp32 accumulation's own rounding - an fp32 GEMM's error at half the work
#define SR_FP32_PLANES_F16 16
inline SplitMap split_map_a(int planes) {
    if (planes == 3) return SplitMap{6, {2, 0, 1, 1, 0, 0}};
    return SplitMap{3, {1, 0, 0, 0, 0, 0}};
}
inline SplitMap split_map_w(int planes) {
    if (planes == 3) return SplitMap{6, {0, 2, 1, 0, 1, 0}};
    return SplitMap{3, {0, 1, 0, 0, 0, 0}};
}

struct GemmArgs {
    const bf16_t* A;   // activations [M, K] row-major
    const bf16_t* W;   // weights [N, K] row-major (nn.Linear layout)
    int M, N, K;
    void* C;           // output (bf16 or fp32 by epilogue); ldc = N (or N/2 for SWIGLU)
    const int* seq_of; // EPI_SEGMAX: sequence id of each token row
    int64_t out_ld;    // EPI_SEGMAX: leading dimension of out (= N)
    // EPI_QKV_ROPE
    const int* pos;         // [M] rope position of each token
    const float* rope_cos;  // [max_pos, head_dim / 2]
    const float* rope_sin;
    int n_rope;             // features [0, n_rope) are rotated (q heads then k heads), the rest (v) stored as is
    int head_dim;           // 64 or 128
    unsigned long long* stamps;  // diagnostics only (tools/micro): 4 s_memrealtime stamps (100 MHz) per workgroup-tile, else null
    const float* a_scale;   // _H epilogues: [M] inverse scale of each activation row (a power of two)
    const float* w_scale;   // _H epilogues: [N] inverse scale of each weight row
    const float* out_scale; // EPI_SWIGLU_SPLIT_H: [M] forward scale of each output row
    SplitMap out_map;  // EPI_SWIGLU_SPLIT: plane of each output segment; C is [M, n_seg * N/2] bf16
    int m_fastest;     // tile order, chosen by launch_gemm_bf16: 1 = token tiles fastest (W far larger than the caches)
    int xcd_order;     // 1: the workgroups of one XCD (blockIdx % 8) work on a compact block of tiles, so that its L2 serves
                       // most operand rows (see gemm_bf16.hip); ignored with m_fastest
};

// y = A @ W^T with fused epilogue. Requirements: K % 64 == 0, N % 16 == 0 (N % 32 for SWIGLU),
// A/W/C 16-byte aligned.
int launch_gemm_bf16(GemmEpilogue epi, const GemmArgs& g, hipStream_t s);

struct AttnArgs {
    const bf16_t* qkv;      // [T, (nh + 2*nkv) * hd] packed tokens, q heads then k heads then v heads
    bf16_t* out;            // [T, nh * hd]
    const int* cu_seqlens;  // [B + 1]
    const int* pos;         // [T] rope position of each token (apply_rope only)
    const unsigned char* key_valid;  // [T] 1 = attend to this token as a key
    const float* rope_cos;  // [max_pos, hd/2]
    const float* rope_sin;
    int B, nh, nkv, hd;
    float scale;            // 1/sqrt(hd)
    int apply_rope;         // 1: rotate q/k while loading (qkv holds raw projections); 0: qkv is already rotated
    int max_seqlen;         // longest sequence of the batch (host side), 0 = unknown; <= 256 enables the fast path
};
// Bidirectional (non-causal) GQA attention over packed var-len sequences.
int launch_attention(const AttnArgs& a, hipStream_t s);

// fp32 regime (the reference encodes dense queries without autocast, eval_dense.py:94-106): fp32 q/k/v (already
// rotated), fp32 scores, softmax and P.V; the output is written as split-bf16 plane segments for the o_proj GEMM.
struct AttnF32Args {
    const float* qkv;       // [T, (nh + 2*nkv) * hd]
    float* out_f32;         // if set: plain fp32 output [T, nh * hd] (the fp16-plane regime splits whole rows afterwards)
    bf16_t* out;            // else [T, n_seg * nh * hd] bf16 plane segments
    const int* cu_seqlens;  // [B + 1]
    const unsigned char* key_valid;  // [T]
    int B, nh, nkv, hd;
    float scale;
    int max_seqlen;         // longest sequence of the batch (grid sizing)
    SplitMap out_map;
    // set by launch_attention_f32: a kernel serves the sequences with only_gt < S and (only_le == 0 or S <= only_le), so the kernel
    // - and with it the bits - of a sequence depends on ITS length, never on what else shares the batch
    int only_le = 0, only_gt = 0;
};
int launch_attention_f32(const AttnF32Args& a, hipStream_t s);
