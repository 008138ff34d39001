// Error plumbing + misc entry points of the C ABI (include/sr_hip.h).
#include "kernels.h"
#include <stdlib.h>
#include <string>
#include <vector>

static thread_local std::string g_last_error;

void sr_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int sr_cu_count() {
    static int per_dev[SR_MAX_DEVICES] = {};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= SR_MAX_DEVICES) d = 0;
    if (per_dev[d] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
        per_dev[d] = n;
    }
    return per_dev[d];
}

extern "C" const char* sr_last_error(void) { return g_last_error.c_str(); }
extern "C" int sr_version(void) { return 1; }
extern "C" int sr_max_topk(void) { return SR_MAX_TOPK; }

extern "C" int sr_gemm_bf16(const void* d_A, const void* d_W, int32_t M, int32_t N, int32_t K, int32_t epilogue, void* d_C,
                            const int32_t* d_seq_of, sr_stream stream) {
    SR_REQUIRE(d_A && d_W && d_C, "sr_gemm_bf16: null pointer");
    SR_REQUIRE(epilogue >= 0 && epilogue <= 4, "sr_gemm_bf16: unknown epilogue %d", epilogue);
    SR_REQUIRE(epilogue != EPI_SEGMAX || d_seq_of, "sr_gemm_bf16: epilogue 3 needs d_seq_of");
    GemmArgs g{};
    g.A = (const bf16_t*)d_A; g.W = (const bf16_t*)d_W; g.M = M; g.N = N; g.K = K; g.C = d_C; g.seq_of = d_seq_of; g.out_ld = N;
    if (epilogue != EPI_SEGMAX && d_seq_of && sr_dev_getenv("SR_GEMM_STAMPS")) {   // tools/micro diagnostics: d_seq_of carries the stamp buffer
        g.stamps = (unsigned long long*)d_seq_of;
        g.seq_of = nullptr;
    }
    return launch_gemm_bf16((GemmEpilogue)epilogue, g, (hipStream_t)stream);
}

extern "C" int sr_gemm_qkv_rope(const void* d_A, const void* d_W, int32_t M, int32_t N, int32_t K, void* d_C, const int32_t* d_pos,
                                const float* d_rope_cos, const float* d_rope_sin, int32_t n_rope, int32_t head_dim,
                                sr_stream stream) {
    SR_REQUIRE(d_A && d_W && d_C && d_pos && d_rope_cos && d_rope_sin, "sr_gemm_qkv_rope: null pointer");
    GemmArgs g{};
    g.A = (const bf16_t*)d_A; g.W = (const bf16_t*)d_W; g.M = M; g.N = N; g.K = K; g.C = d_C;
    g.pos = d_pos; g.rope_cos = d_rope_cos; g.rope_sin = d_rope_sin; g.n_rope = n_rope; g.head_dim = head_dim;
    return launch_gemm_bf16(EPI_QKV_ROPE, g, (hipStream_t)stream);
}

extern "C" int sr_attention_varlen(const void* d_qkv, void* d_out, const int32_t* d_cu_seqlens, const int32_t* d_pos,
                                   const uint8_t* d_key_valid, const float* d_rope_cos, const float* d_rope_sin, int32_t B,
                                   int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, sr_stream stream) {
    SR_REQUIRE(d_qkv && d_out && d_cu_seqlens && d_key_valid, "sr_attention_varlen: null pointer");
    SR_REQUIRE((d_rope_cos == nullptr) == (d_rope_sin == nullptr), "sr_attention_varlen: pass both rope tables or neither");
    SR_REQUIRE(!d_rope_cos || d_pos, "sr_attention_varlen: rope tables need d_pos");
    SR_REQUIRE(B >= 0 && num_heads > 0 && num_kv_heads > 0, "sr_attention_varlen: bad sizes");
    AttnArgs a{};
    a.qkv = (const bf16_t*)d_qkv; a.out = (bf16_t*)d_out; a.cu_seqlens = d_cu_seqlens; a.pos = d_pos; a.key_valid = d_key_valid;
    a.rope_cos = d_rope_cos; a.rope_sin = d_rope_sin; a.B = B; a.nh = num_heads; a.nkv = num_kv_heads; a.hd = head_dim;
    a.scale = 1.0f / sqrtf((float)head_dim);
    a.apply_rope = d_rope_cos ? 1 : 0;
    a.max_seqlen = 0;
    if (!a.apply_rope && B > 0) {   // test hook: fetch the lengths to pick the kernel the encoder would pick
        std::vector<int> cu((size_t)B + 1);
        SR_CHECK_HIP(hipMemcpyAsync(cu.data(), d_cu_seqlens, sizeof(int) * ((size_t)B + 1), hipMemcpyDeviceToHost, (hipStream_t)stream));
        SR_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
        for (int b = 0; b < B; ++b) a.max_seqlen = cu[b + 1] - cu[b] > a.max_seqlen ? cu[b + 1] - cu[b] : a.max_seqlen;
    }
    return launch_attention(a, (hipStream_t)stream);
}

// fp16-plane GEMM of the encoder's fp32 regime, exported for the per-kernel parity test: C fp32 [M, N] += (A' @ W'^T) *
// a_scale[m] * w_scale[n], A' / W' = [rows, K] fp16 plane segments.
extern "C" int sr_gemm_f16_scaled(const void* d_A, const void* d_W, int32_t M, int32_t N, int32_t K, const float* d_a_scale,
                                  const float* d_w_scale, float* d_C, sr_stream stream) {
    SR_REQUIRE(d_A && d_W && d_C && d_a_scale && d_w_scale, "sr_gemm_f16_scaled: null pointer");
    GemmArgs g{};
    g.A = (const bf16_t*)d_A; g.W = (const bf16_t*)d_W; g.M = M; g.N = N; g.K = K; g.C = d_C; g.a_scale = d_a_scale; g.w_scale = d_w_scale;
    return launch_gemm_bf16(EPI_RESID_F32_H, g, (hipStream_t)stream);
}
