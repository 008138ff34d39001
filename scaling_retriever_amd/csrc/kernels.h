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
};

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
    int m_fastest;     // tile order, chosen by launch_gemm_bf16: 1 = token tiles fastest (W far larger than the caches)
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
