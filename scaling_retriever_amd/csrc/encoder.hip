// LlamaBiDense / LlamaBiSparse encode on gfx950: model handle, packing plan, norms,
// pooling heads and the layer loop.  GEMMs: gemm_bf16.hip; attention: attention.hip.
//
// Reference semantics reproduced (file:line into HansiZeng/scaling-retriever):
//  * backbone   LlamaBiModel over HF LlamaModel.forward, bidirectional mask on padded
//               KEYS only, position_ids = arange(L)     modeling/bidirectional_llama.py:67-188
//  * dense head per-token L2 normalise, mean of the LAST `len` positions (literal
//               `[-length:]` slice, so left padding is assumed)  modeling/llm_encoder.py:424-443
//  * sparse head log(1 + relu(max_L(logits * H^-0.25 + (1-mask) * -1e6)))
//                                                           modeling/llm_encoder.py:186-196
//  * precision  bf16 GEMM inputs / fp32 accumulate, fp32 residual stream, norms, rope
//               tables, softmax and heads - the torch.autocast(bf16) regime of
//               indexer.py:46-52,255-256 (fp32 master weights are rounded to bf16 once).
//
// Packing: a row keeps the contiguous span of positions it needs - the keys (mask == 1)
// and, for the dense head, the last `len` positions - so a left-padded batch (the
// reference's eval setting) is computed with ZERO pad tokens, while right-padded or
// holed masks stay literal (pad query rows are computed, masked as keys).
#include "kernels.h"
#include <math.h>
#include <mutex>
#include <string>
#include <vector>

// ============================================================== small kernels ===
// ---- plan: per-row span ------------------------------------------------------
// mode 0 (dense) and 2 (both heads): span = [min(first1, L - len), L); pool_start = L - len (0 if len == 0)
// mode 1 (sparse): span = [first1, last1 + 1)
// row_shift (nullable, sr_encode_rows): row b sits row_shift[b] columns further right than in the batch it came from; positions
// (RoPE, pooling) are counted from there, so the row gets the position_ids it had in its own batch.
__global__ void plan_rows_kernel(const int64_t* __restrict__ mask, int B, int L, int mode, int* __restrict__ span_start,
                                 int* __restrict__ span_len, int* __restrict__ pool_start, int* __restrict__ row_len,
                                 const int* __restrict__ row_shift) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;  // 64 threads
    int len = 0, first = L, last = -1;
    for (int p = lane; p < L; p += 64) {
        if (mask[(int64_t)b * L + p] != 0) {
            ++len;
            first = p < first ? p : first;
            last = p > last ? p : last;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        len += __shfl_xor(len, off);
        const int f = __shfl_xor(first, off), l = __shfl_xor(last, off);
        first = f < first ? f : first;
        last = l > last ? l : last;
    }
    if (lane == 0) {
        int st, ln, ps;
        if (len == 0) {
            st = 0; ln = 0; ps = 0;
        } else if (mode != 1) {
            ps = L - len;
            st = first < ps ? first : ps;
            ln = L - st;
        } else {
            ps = first;
            st = first;
            ln = last + 1 - first;
        }
        span_start[b] = st;
        span_len[b] = ln;
        pool_start[b] = ps - (row_shift ? row_shift[b] : 0);     // compared with pos[], which carries the same shift
        row_len[b] = len;
    }
}

// exclusive scan of span_len -> cu_seqlens[B+1] (single workgroup, B <= 65536)
__global__ void plan_scan_kernel(const int* __restrict__ span_len, int B, int* __restrict__ cu) {
    __shared__ int part[256];
    const int tid = threadIdx.x;
    const int per = (B + 255) / 256;
    const int b0 = tid * per, b1 = (b0 + per) < B ? (b0 + per) : B;
    int s = 0;
    for (int b = b0; b < b1; ++b) s += span_len[b];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) { const int v = part[i]; part[i] = run; run += v; }
        cu[B] = run;
    }
    __syncthreads();
    int run = part[tid];
    for (int b = b0; b < b1; ++b) { cu[b] = run; run += span_len[b]; }
}

// per packed token: source position, token id, key flag, sequence id
__global__ void plan_tokens_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ mask, int L,
                                   const int* __restrict__ span_start, const int* __restrict__ cu, int* __restrict__ tok_id,
                                   int* __restrict__ pos, unsigned char* __restrict__ key_valid, int* __restrict__ seq_of,
                                   int vocab, int mode, const int* __restrict__ row_shift) {
    const int b = blockIdx.x;
    const int t0 = cu[b], n = cu[b + 1] - t0, st = span_start[b];
    const int shift = row_shift ? row_shift[b] : 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const int p = st + j;
        int64_t id = ids[(int64_t)b * L + p];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // clamp (HF would raise an index error)
        tok_id[t0 + j] = (int)id;
        pos[t0 + j] = p - shift < 0 ? 0 : p - shift;     // a shift beyond the row's span start is a caller error: stay in the tables
        const bool valid = mask[(int64_t)b * L + p] != 0;
        key_valid[t0 + j] = valid ? 1 : 0;
        // sparse head: masked positions inside the span take no part in the max (-2 = skip row)
        seq_of[t0 + j] = (mode != 0 && !valid) ? -2 : b;
    }
}

// ---- RMSNorm (fp32 variance) : one wave per token, optional embedding gather / pending residual -----
// x[t] = embed[tok_id[t]]            (if embed)
// x[t] += delta[t]                   (if delta: the bf16 output of the previous o_proj / down_proj GEMM - under the
//                                     reference's autocast nn.Linear returns bf16, which is then added to the fp32
//                                     residual stream; doing the add here keeps the GEMM epilogue at 2 B per element)
// xn[t] = bf16( x * rsqrt(mean(x^2) + eps) * w )
// NCH > 0: H == 256 * NCH and the row stays in registers between the two passes (all its loads in flight at once, no
// re-read of the freshly written row); NCH == 0: any H % 4 == 0, second pass re-reads the row.  Same arithmetic order.
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_kernel(float* __restrict__ x, const float* __restrict__ embed,
                                                      const int* __restrict__ tok_id, const bf16_t* __restrict__ delta,
                                                      const float* __restrict__ w, bf16_t* __restrict__ xn,
                                                      float* __restrict__ xn_f32, int T, int H, float eps) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= T) return;
    float* xr = x + (int64_t)t * H;
    const float* src = embed ? embed + (int64_t)tok_id[t] * H : xr;
    const bf16_t* dr = delta ? delta + (int64_t)t * H : nullptr;
    float ss = 0.f;
    if constexpr (NCH > 0) {
        f32x4 v[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) v[c] = *reinterpret_cast<const f32x4*>(src + lane * 4 + c * 256);
        if (dr) {
            bf16x4 d[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) d[c] = *reinterpret_cast<const bf16x4*>(dr + lane * 4 + c * 256);
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[c][e] += bf16_to_f32((unsigned short)d[c][e]);
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            ss += v[c][0] * v[c][0] + v[c][1] * v[c][1] + v[c][2] * v[c][2] + v[c][3] * v[c][3];
            if (embed || dr) *reinterpret_cast<f32x4*>(xr + lane * 4 + c * 256) = v[c];
        }
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        const float rs = 1.0f / sqrtf(ss / (float)H + eps);
        if (!w) return;   // residual add only
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = lane * 4 + c * 256;
            const f32x4 g = *reinterpret_cast<const f32x4*>(w + i);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[c][e] * rs) * g[e];
            if (xn) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (short)f32_to_bf16(y[e]);
                *reinterpret_cast<bf16x4*>(xn + (int64_t)t * H + i) = o;
            }
            if (xn_f32) *reinterpret_cast<f32x4*>(xn_f32 + (int64_t)t * H + i) = y;
        }
    } else {
        for (int i = lane * 4; i < H; i += 256) {
            f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
            if (dr) {
                const bf16x4 d = *reinterpret_cast<const bf16x4*>(dr + i);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] += bf16_to_f32((unsigned short)d[c]);
            }
            ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            if (embed || dr) *reinterpret_cast<f32x4*>(xr + i) = v;
        }
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        const float rs = 1.0f / sqrtf(ss / (float)H + eps);
        if (!w) return;   // residual add only
        for (int i = lane * 4; i < H; i += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);   // this lane's own writes (or the unchanged row)
            const f32x4 g = *reinterpret_cast<const f32x4*>(w + i);
            f32x4 y;
#pragma unroll
            for (int c = 0; c < 4; ++c) y[c] = (v[c] * rs) * g[c];
            if (xn) {
                bf16x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = (short)f32_to_bf16(y[c]);
                *reinterpret_cast<bf16x4*>(xn + (int64_t)t * H + i) = o;
            }
            if (xn_f32) *reinterpret_cast<f32x4*>(xn_f32 + (int64_t)t * H + i) = y;
        }
    }
}

static int launch_rmsnorm(float* x, const float* embed, const int* tok_id, const bf16_t* delta, const float* w, bf16_t* xn,
                          float* xn_f32, int T, int H, float eps, hipStream_t s) {
    if (T == 0) return SR_OK;
    const dim3 grid((unsigned)ceil_div64(T, 4)), block(256);
#define SR_RMS(NCH) hipLaunchKernelGGL(rmsnorm_kernel<NCH>, grid, block, 0, s, x, embed, tok_id, delta, w, xn, xn_f32, T, H, eps)
    switch (H) {
        case 256: SR_RMS(1); break;
        case 512: SR_RMS(2); break;
        case 1024: SR_RMS(4); break;
        case 2048: SR_RMS(8); break;
        case 3072: SR_RMS(12); break;
        case 4096: SR_RMS(16); break;
        default: SR_RMS(0); break;
    }
#undef SR_RMS
    SR_CHECK_LAUNCH();
    return SR_OK;
}

// ---- dense head: final RMSNorm -> per-token L2 normalise -> mean over pooled tokens -----
// (1) one wave per token: the two row statistics  rs = rsqrt(mean(x^2) + eps),  inv = 1 / max(||x rs w||, 1e-12)
__global__ __launch_bounds__(256) void dense_head_stats_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               float2* __restrict__ stats, int T, int H, float eps) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= T) return;
    const float* xr = x + (int64_t)t * H;
    float ss = 0.f;
    for (int i = lane * 4; i < H; i += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float rs = 1.0f / sqrtf(ss / (float)H + eps);
    float s2 = 0.f;
    for (int i = lane * 4; i < H; i += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
        const f32x4 g = *reinterpret_cast<const f32x4*>(w + i);
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float y = (v[c] * rs) * g[c]; s2 += y * y; }
    }
    for (int off = 32; off > 0; off >>= 1) s2 += __shfl_xor(s2, off);
    if (lane == 0) stats[t] = make_float2(rs, 1.0f / fmaxf(sqrtf(s2), 1e-12f));
}
// (2) workgroup = (sequence, 256-column slab): out[b][c] = mean over pooled tokens of x[t][c] rs_t w[c] inv_t,
//     accumulated in token order (fp32), coalesced row reads.
__global__ __launch_bounds__(256) void dense_head_pool_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float2* __restrict__ stats, const int* __restrict__ cu,
                                                              const int* __restrict__ pos, const int* __restrict__ pool_start,
                                                              float* __restrict__ out, int H) {
    const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= H) return;
    const int t0 = cu[b], n = cu[b + 1] - t0, ps = pool_start[b];
    const float wc = w[c];
    float acc = 0.f;
    int cnt = 0;
    for (int j = 0; j < n; ++j) {
        const int t = t0 + j;
        if (pos[t] < ps) continue;
        const float2 st = stats[t];
        acc += ((x[(int64_t)t * H + c] * st.x) * wc) * st.y;
        ++cnt;
    }
    out[(int64_t)b * H + c] = cnt > 0 ? acc / (float)cnt : 0.f;
}

// ---- sparse head finish: reps = log(1 + relu(bf16_round(max_logit) * H^-0.25)) ----
__global__ void sparse_finish_kernel(float* __restrict__ out, int64_t n, float scale, int round_bf16) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = out[i];  // max over tokens of positive logits, 0 if none
    if (round_bf16) v = bf16_to_f32(f32_to_bf16(v));
    v *= scale;
    out[i] = logf(fmaxf(v, 0.f) + 1.0f);
}

// ---- weights: convert / permute rows into the internal bf16 layout --------------------
// dst[row_map(r)][c] = bf16(src[r][c]);  row_map: dst_row = base + (interleave ? blk16(r) : r)
__global__ void convert_rows_kernel(const void* __restrict__ src, int src_dtype, int64_t rows, int64_t cols,
                                    bf16_t* __restrict__ dst_bf16, float* __restrict__ dst_f32, int64_t dst_row_base,
                                    int interleave /*0 none, 1 gate, 2 up*/) {
    const int64_t r = blockIdx.x;
    int64_t dr = r;
    if (interleave) dr = (r / 16) * 32 + (interleave == 2 ? 16 : 0) + (r % 16);
    dr += dst_row_base;
    for (int64_t c = threadIdx.x; c < cols; c += blockDim.x) {
        float v;
        if (src_dtype == SR_DTYPE_F32) v = reinterpret_cast<const float*>(src)[r * cols + c];
        else v = bf16_to_f32(reinterpret_cast<const bf16_t*>(src)[r * cols + c]);
        if (dst_bf16) dst_bf16[dr * cols + c] = f32_to_bf16(v);
        if (dst_f32) dst_f32[dr * cols + c] = v;
    }
}

// fp32 regime: the same rows as split-bf16 plane segments, dst[row_map(r)][sg * cols + c] = plane_{map.plane[sg]}(src[r][c])
__global__ void convert_rows_split_kernel(const void* __restrict__ src, int src_dtype, int64_t rows, int64_t cols,
                                          bf16_t* __restrict__ dst, int64_t dst_row_base, int interleave, SplitMap map) {
    const int64_t r = blockIdx.x;
    int64_t dr = r;
    if (interleave) dr = (r / 16) * 32 + (interleave == 2 ? 16 : 0) + (r % 16);
    dr += dst_row_base;
    bf16_t* drow = dst + dr * cols * map.n_seg;
    for (int64_t c = threadIdx.x; c < cols; c += blockDim.x) {
        float v;
        if (src_dtype == SR_DTYPE_F32) v = reinterpret_cast<const float*>(src)[r * cols + c];
        else v = bf16_to_f32(reinterpret_cast<const bf16_t*>(src)[r * cols + c]);
        unsigned short p[3];
        split_bf16x3(v, p[0], p[1], p[2]);
        for (int sg = 0; sg < map.n_seg; ++sg) drow[(int64_t)sg * cols + c] = p[map.plane[sg]];
    }
}

// ---- RMSNorm of the fp32 regime: one wave per token, output as split-bf16 plane segments [T, n_seg * H] -------
// x[t] = embed[tok_id[t]] (if embed);  y = (x * rsqrt(mean(x^2) + eps)) * w in fp32 (HF LlamaRMSNorm on fp32 input)
__global__ __launch_bounds__(256) void rmsnorm_split_kernel(float* __restrict__ x, const float* __restrict__ embed,
                                                            const int* __restrict__ tok_id, const float* __restrict__ w,
                                                            bf16_t* __restrict__ xs, int T, int H, float eps, SplitMap map) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= T) return;
    float* xr = x + (int64_t)t * H;
    const float* src = embed ? embed + (int64_t)tok_id[t] * H : xr;
    float ss = 0.f;
    for (int i = lane * 4; i < H; i += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        if (embed) *reinterpret_cast<f32x4*>(xr + i) = v;
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float rs = 1.0f / sqrtf(ss / (float)H + eps);
    bf16_t* orow = xs + (int64_t)t * H * map.n_seg;
    for (int i = lane * 4; i < H; i += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
        const f32x4 g = *reinterpret_cast<const f32x4*>(w + i);
        bf16x4 pl[3];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned short a0, a1, a2;
            split_bf16x3((v[c] * rs) * g[c], a0, a1, a2);
            pl[0][c] = (short)a0; pl[1][c] = (short)a1; pl[2][c] = (short)a2;
        }
        for (int sg = 0; sg < map.n_seg; ++sg) {
            const int pi = map.plane[sg];
            *reinterpret_cast<bf16x4*>(orow + (int64_t)sg * H + i) = pi == 0 ? pl[0] : (pi == 1 ? pl[1] : pl[2]);
        }
    }
}

// ---- fp16-plane regime (cfg.fp32_planes == SR_FP32_PLANES_F16): rows scaled by a power of two, two fp16 planes ----
// weights: dst[row_map(r)] = [g0 | g1 | g0] (kernels.h split_map_w), w_inv[row_map(r)] = 1 / scale of that row
__global__ __launch_bounds__(256) void convert_rows_split_h_kernel(const void* __restrict__ src, int src_dtype, int64_t rows, int64_t cols,
                                                                   bf16_t* __restrict__ dst, float* __restrict__ w_inv,
                                                                   int64_t dst_row_base, int interleave) {
    __shared__ float red[4];
    const int64_t r = blockIdx.x;
    int64_t dr = r;
    if (interleave) dr = (r / 16) * 32 + (interleave == 2 ? 16 : 0) + (r % 16);
    dr += dst_row_base;
    auto at = [&](int64_t c) {
        return src_dtype == SR_DTYPE_F32 ? reinterpret_cast<const float*>(src)[r * cols + c]
                                         : bf16_to_f32(reinterpret_cast<const bf16_t*>(src)[r * cols + c]);
    };
    float mx = 0.f;
    for (int64_t c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, fabsf(at(c)));
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = row_scale_pow2(mx);
    if (threadIdx.x == 0) w_inv[dr] = 1.0f / sc;
    bf16_t* drow = dst + dr * cols * 3;
    for (int64_t c = threadIdx.x; c < cols; c += 256) {
        unsigned short f0, f1;
        split_f16x2(at(c) * sc, f0, f1);
        drow[c] = f0; drow[cols + c] = f1; drow[2 * cols + c] = f0;
    }
}

// activations: one wave per row of src fp32 [T, K] -> [f1 | f0 | f0] (split_map_a) + a_inv[t]; with w: the RMSNorm of the row
// first (x = embed[tok] if embed; y = (x * rsqrt(mean(x^2) + eps)) * w), i.e. rmsnorm_split_kernel on fp16 planes
__global__ __launch_bounds__(256) void rows_split_h_kernel(float* __restrict__ x, const float* __restrict__ embed,
                                                           const int* __restrict__ tok_id, const float* __restrict__ w,
                                                           bf16_t* __restrict__ xs, float* __restrict__ a_inv, int T, int K, float eps,
                                                           const float* __restrict__ gu_cmax, float* __restrict__ act_sc,
                                                           float* __restrict__ act_inv) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= T) return;
    float* xr = x + (int64_t)t * K;
    const float* src = embed ? embed + (int64_t)tok_id[t] * K : xr;
    float rs = 1.f;
    if (w) {
        float ss = 0.f;
        for (int i = lane * 4; i < K; i += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
            ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            if (embed) *reinterpret_cast<f32x4*>(xr + i) = v;
        }
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        rs = 1.0f / sqrtf(ss / (float)K + eps);
    }
    float mx = 0.f, s2 = 0.f;
    for (int i = lane * 4; i < K; i += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
        f32x4 g = {1.f, 1.f, 1.f, 1.f};
        if (w) g = *reinterpret_cast<const f32x4*>(w + i);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float y = w ? (v[c] * rs) * g[c] : v[c];
            mx = fmaxf(mx, fabsf(y));
            s2 += y * y;
        }
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    const float sc = row_scale_pow2(mx);
    if (lane == 0) a_inv[t] = 1.0f / sc;
    if (gu_cmax) {                              // see rows_split_h_reg_kernel
        for (int off = 32; off > 0; off >>= 1) s2 += __shfl_xor(s2, off);
        const float osc = row_scale_pow2(s2 * *gu_cmax * 1.02f);
        if (lane == 0) { act_sc[t] = osc; act_inv[t] = 1.0f / osc; }
    }
    bf16_t* orow = xs + (int64_t)t * K * 3;
    for (int i = lane * 4; i < K; i += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
        f32x4 g = {1.f, 1.f, 1.f, 1.f};
        if (w) g = *reinterpret_cast<const f32x4*>(w + i);
        bf16x4 p0, p1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned short f0, f1;
            split_f16x2((w ? (v[c] * rs) * g[c] : v[c]) * sc, f0, f1);
            p0[c] = (short)f0; p1[c] = (short)f1;
        }
        *reinterpret_cast<bf16x4*>(orow + i) = p1;
        *reinterpret_cast<bf16x4*>(orow + K + i) = p0;
        *reinterpret_cast<bf16x4*>(orow + 2 * K + i) = p0;
    }
}

// The same with the row held in registers between the passes (NV float4 per lane, K = 256 NV): one read of the row instead of
// up to three (sum of squares, maximum, split) - the row-split passes were 10 % of a fp32-regime query encode.
template <int NV>
__global__ __launch_bounds__(256) void rows_split_h_reg_kernel(float* __restrict__ x, const float* __restrict__ embed,
                                                               const int* __restrict__ tok_id, const float* __restrict__ w,
                                                               bf16_t* __restrict__ xs, float* __restrict__ a_inv, int T, float eps,
                                                               const float* __restrict__ gu_cmax, float* __restrict__ act_sc,
                                                               float* __restrict__ act_inv) {
    constexpr int K = 256 * NV;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (t >= T) return;
    float* xr = x + (int64_t)t * K;
    const float* src = embed ? embed + (int64_t)tok_id[t] * K : xr;
    f32x4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const f32x4*>(src + lane * 4 + 256 * j);
    if (w) {
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            ss += v[j][0] * v[j][0] + v[j][1] * v[j][1] + v[j][2] * v[j][2] + v[j][3] * v[j][3];
            if (embed) *reinterpret_cast<f32x4*>(xr + lane * 4 + 256 * j) = v[j];
        }
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        const float rs = 1.0f / sqrtf(ss / (float)K + eps);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(w + lane * 4 + 256 * j);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[j][c] = (v[j][c] * rs) * g[c];
        }
    }
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) mx = fmaxf(mx, fabsf(v[j][c]));
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    const float sc = row_scale_pow2(mx);
    if (lane == 0) a_inv[t] = 1.0f / sc;
    if (gu_cmax) {
        // scale of this row's SwiGLU output, known before the gate-up GEMM runs: |silu(g_j) u_j| <= |g_j||u_j| <=
        // |xn|^2 |wg_j||wu_j| <= |xn|^2 cmax (Cauchy-Schwarz; 1.02: the fp32 evaluation of the norms) - EPI_SWIGLU_SPLIT_H
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s2 += v[j][0] * v[j][0] + v[j][1] * v[j][1] + v[j][2] * v[j][2] + v[j][3] * v[j][3];
        for (int off = 32; off > 0; off >>= 1) s2 += __shfl_xor(s2, off);
        const float osc = row_scale_pow2(s2 * *gu_cmax * 1.02f);
        if (lane == 0) { act_sc[t] = osc; act_inv[t] = 1.0f / osc; }
    }
    bf16_t* orow = xs + (int64_t)t * K * 3;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        bf16x4 p0, p1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned short f0, f1;
            split_f16x2(v[j][c] * sc, f0, f1);
            p0[c] = (short)f0; p1[c] = (short)f1;
        }
        const int i = lane * 4 + 256 * j;
        *reinterpret_cast<bf16x4*>(orow + i) = p1;
        *reinterpret_cast<bf16x4*>(orow + K + i) = p0;
        *reinterpret_cast<bf16x4*>(orow + 2 * K + i) = p0;
    }
}

// gu_cmax (device float) / act_sc / act_inv: the row scales of the SwiGLU output this row will produce
static void launch_rows_split_h(float* x, const float* embed, const int* tok, const float* w, bf16_t* xs, float* inv, int T, int K,
                                float eps, hipStream_t s, const float* gu_cmax = nullptr, float* act_sc = nullptr,
                                float* act_inv = nullptr) {
    const dim3 grid((unsigned)ceil_div64(T, 4)), block(256);
    switch (K) {
        case 2048: hipLaunchKernelGGL(rows_split_h_reg_kernel<8>, grid, block, 0, s, x, embed, tok, w, xs, inv, T, eps, gu_cmax, act_sc, act_inv); break;
        case 4096: hipLaunchKernelGGL(rows_split_h_reg_kernel<16>, grid, block, 0, s, x, embed, tok, w, xs, inv, T, eps, gu_cmax, act_sc, act_inv); break;
        case 8192: hipLaunchKernelGGL(rows_split_h_reg_kernel<32>, grid, block, 0, s, x, embed, tok, w, xs, inv, T, eps, gu_cmax, act_sc, act_inv); break;
        default: hipLaunchKernelGGL(rows_split_h_kernel, grid, block, 0, s, x, embed, tok, w, xs, inv, T, K, eps, gu_cmax, act_sc, act_inv);
    }
}

// max over the MLP's feature pairs j of |w_gate_j| |w_up_j|, from the fp16 plane segments [w0 | w1 | w0] of the interleaved
// gate/up matrix (gate rows and up rows alternate in 16-row blocks) and the rows' inverse scales: one wave per pair
__global__ __launch_bounds__(256) void gu_cmax_kernel(const bf16_t* __restrict__ wgu_s, const float* __restrict__ wgu_i, int I, int K,
                                                      float* __restrict__ cmax) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= I) return;
    const int64_t rg = (int64_t)(j / 16) * 32 + (j % 16), ru = rg + 16;
    float n2[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const bf16_t* row = wgu_s + (h ? ru : rg) * 3 * (int64_t)K;
        float ss = 0.f;
        for (int i = lane; i < K; i += 64) {
            const float v = (float)__builtin_bit_cast(_Float16, row[i]) + (float)__builtin_bit_cast(_Float16, row[K + i]);
            ss += v * v;
        }
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        const float inv = wgu_i[h ? ru : rg];
        n2[h] = ss * inv * inv;
    }
    if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(cmax), __float_as_uint(sqrtf(n2[0]) * sqrtf(n2[1]) * 1.001f));
}

// ---- LoRA merge: W += scale * B @ A -----------------------------------------------------
__global__ void lora_merge_kernel(float* __restrict__ W, const float* __restrict__ A, const float* __restrict__ Bm,
                                  int64_t out_f, int64_t in_f, int r, float scale) {
    const int64_t o = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= in_f) return;
    float acc = 0.f;
    for (int k = 0; k < r; ++k) acc += Bm[o * r + k] * A[(int64_t)k * in_f + i];
    W[o * in_f + i] += scale * acc;
}

// ---- sparse reps -> CSR (torch.nonzero order: row-major, cols ascending) ----------------
__global__ __launch_bounds__(256) void nnz_count_kernel(const float* __restrict__ reps, int64_t V, int64_t* __restrict__ row_nnz) {
    __shared__ int red[4];
    const int64_t b = blockIdx.x;
    int c = 0;
    for (int64_t i = threadIdx.x; i < V; i += 256) c += reps[b * V + i] != 0.f ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) row_nnz[b] = red[0] + red[1] + red[2] + red[3];
}
__global__ void nnz_scan_kernel(const int64_t* __restrict__ row_nnz, int64_t B, int64_t* __restrict__ row_ptr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int64_t run = 0;
        for (int64_t b = 0; b < B; ++b) { row_ptr[b] = run; run += row_nnz[b]; }
        row_ptr[B] = run;
    }
}
__global__ __launch_bounds__(256) void nnz_fill_kernel(const float* __restrict__ reps, int64_t V, const int64_t* __restrict__ row_ptr,
                                                       int32_t* __restrict__ cols, float* __restrict__ vals, int64_t capacity) {
    __shared__ int wtot[4];
    __shared__ int64_t base;
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = row_ptr[b];
    __syncthreads();
    for (int64_t i0 = 0; i0 < V; i0 += 256) {
        const int64_t i = i0 + tid;
        const float v = i < V ? reps[b * V + i] : 0.f;
        const int nz = v != 0.f ? 1 : 0;
        int incl = nz;
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); if (lane >= off) incl += o; }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int wb = 0, tot = 0;
        for (int w = 0; w < 4; ++w) { if (w < wave) wb += wtot[w]; tot += wtot[w]; }
        const int64_t p = base + wb + incl - nz;
        if (nz && p < capacity) { cols[p] = (int32_t)i; vals[p] = v; }
        __syncthreads();
        if (tid == 0) base += tot;
        __syncthreads();
    }
}

// ==================================================================== model ====
struct LayerW {
    bf16_t* wqkv = nullptr;   // [(nh + 2 nkv) hd, H]
    bf16_t* wo = nullptr;     // [H, nh hd]
    bf16_t* wgu = nullptr;    // [2 I, H], gate/up interleaved in 16-row blocks
    bf16_t* wdown = nullptr;  // [H, I]
    float* ln1 = nullptr;     // [H]
    float* ln2 = nullptr;     // [H]
    unsigned have = 0;        // bit per tensor: q k v o gate up down ln1 ln2
    // fp32 regime (cfg.fp32_planes > 0): the same matrices as split-bf16 plane segments along K, [rows, n_seg * K]
    bf16_t *wqkv_s = nullptr, *wo_s = nullptr, *wgu_s = nullptr, *wdown_s = nullptr;
    // fp16-plane regime: inverse power-of-two scale of every weight row
    float *wqkv_i = nullptr, *wo_i = nullptr, *wgu_i = nullptr, *wdown_i = nullptr;
    float* gu_cmax = nullptr;   // fp16-plane regime: max_j |w_gate_j||w_up_j| (device float), see gu_cmax_kernel
};

struct sr_model {
    sr_model_config cfg;
    int Tm = 0, Bm = 0;        // workspace capacity (tokens rounded up to 128, sequences)
    int max_pos = 0;
    float* embed = nullptr;    // fp32 [V, H]
    bf16_t* lm_head = nullptr; // bf16 [V, H] (sparse head)
    bf16_t* lm_head_s = nullptr;   // fp32 regime: [V, n_seg * H] plane segments
    float* lm_head_i = nullptr;    // fp16-plane regime: [V] inverse row scales
    float *attn_f = nullptr, *act_f = nullptr;                  // fp16-plane regime: fp32 attention / SwiGLU outputs before the row split
    float *xs_i = nullptr, *attn_i = nullptr, *act_i = nullptr; // ... and the inverse row scales of xs / attn_s / act_s
    float* act_sc = nullptr;                                     // forward row scales of the fused SwiGLU split
    // fp32-regime workspace, allocated by the first fp32 encode call
    bf16_t* xs = nullptr;      // [Tm, n_seg * H]    normed hidden state, plane segments
    float* qkv_f = nullptr;    // [Tm, (nh + 2 nkv) hd] rotated q/k/v, fp32
    bf16_t* attn_s = nullptr;  // [Tm, n_seg * nh hd]
    bf16_t* act_s = nullptr;   // [Tm, n_seg * I]
    float* norm_w = nullptr;   // [H]
    std::vector<LayerW> layers;
    bool have_embed = false, have_norm = false, have_lm_head = false, finalized = false;
    float* rope_cos = nullptr; // [max_pos, hd/2]
    float* rope_sin = nullptr;
    // workspace
    float* x = nullptr;        // [Tm, H] fp32 residual stream
    bf16_t* xn = nullptr;      // [Tm, H]
    bf16_t* qkv = nullptr;     // [Tm, (nh + 2 nkv) hd]
    bf16_t* attn = nullptr;    // [Tm, nh hd]
    bf16_t* act = nullptr;     // [Tm, I]
    bf16_t* delta = nullptr;   // [Tm, H] bf16 output of o_proj / down_proj, added to x by the next norm kernel
    int *span_start = nullptr, *span_len = nullptr, *pool_start = nullptr, *row_len = nullptr, *cu = nullptr;
    int *tok_id = nullptr, *pos = nullptr, *seq_of = nullptr;
    unsigned char* key_valid = nullptr;
    int* h_cu = nullptr;       // pinned host copy of cu_seqlens
    int last_T = 0;
    std::mutex mu;
    StreamOrder order;     // chains calls that arrive on different streams (one set of activation buffers)
};

static void model_free(sr_model* m) {
    auto F = [](void* p) { if (p) (void)hipFree(p); };
    F(m->embed); F(m->lm_head); F(m->norm_w); F(m->rope_cos); F(m->rope_sin);
    F(m->lm_head_s); F(m->xs); F(m->qkv_f); F(m->attn_s); F(m->act_s);
    F(m->lm_head_i); F(m->attn_f); F(m->act_f); F(m->xs_i); F(m->attn_i); F(m->act_i); F(m->act_sc);
    for (auto& l : m->layers) {
        F(l.wqkv); F(l.wo); F(l.wgu); F(l.wdown); F(l.ln1); F(l.ln2); F(l.wqkv_s); F(l.wo_s); F(l.wgu_s); F(l.wdown_s);
        F(l.wqkv_i); F(l.wo_i); F(l.wgu_i); F(l.wdown_i); F(l.gu_cmax);
    }
    F(m->x); F(m->xn); F(m->qkv); F(m->attn); F(m->act); F(m->delta);
    F(m->span_start); F(m->span_len); F(m->pool_start); F(m->row_len); F(m->cu);
    F(m->tok_id); F(m->pos); F(m->seq_of); F(m->key_valid);
    if (m->h_cu) (void)hipHostFree(m->h_cu);
}

#define SR_ALLOC(ptr, bytes)                                                                      \
    do {                                                                                          \
        const hipError_t e_ = hipMalloc((void**)&(ptr), (size_t)(bytes));                          \
        if (e_ != hipSuccess) {                                                                   \
            sr_set_error("sr_model_create: hipMalloc of %zu bytes failed: %s", (size_t)(bytes), hipGetErrorString(e_)); \
            model_free(m);                                                                        \
            delete m;                                                                             \
            return SR_ERR_NOMEM;                                                                  \
        }                                                                                         \
    } while (0)

static void rope_tables(const sr_model_config& c, int max_pos, std::vector<float>& cosv, std::vector<float>& sinv) {
    // HF ROPE_INIT_FUNCTIONS["default" | "llama3"]; inv_freq kept in fp32, angles = fp32(pos) * inv_freq
    const int hd = c.head_dim, half = hd / 2;
    std::vector<float> inv(half);
    for (int i = 0; i < half; ++i) {
        double f = 1.0 / pow((double)c.rope_theta, (double)(2 * i) / (double)hd);
        if (c.rope_llama3) {
            const double factor = c.rope_factor, lo = c.rope_low_freq_factor, hi = c.rope_high_freq_factor;
            const double old = (double)c.rope_original_max_pos;
            const double low_wl = old / lo, high_wl = old / hi;
            const double wl = 2.0 * M_PI / f;
            const double f_l = wl > low_wl ? f / factor : f;
            if (!(wl < high_wl) && !(wl > low_wl)) {
                const double smooth = (old / wl - lo) / (hi - lo);
                f = (1.0 - smooth) * f_l / factor + smooth * f_l;
            } else {
                f = f_l;
            }
        }
        inv[i] = (float)f;
    }
    cosv.resize((size_t)max_pos * half);
    sinv.resize((size_t)max_pos * half);
    for (int p = 0; p < max_pos; ++p)
        for (int i = 0; i < half; ++i) {
            const float ang = (float)p * inv[i];
            cosv[(size_t)p * half + i] = (float)cos((double)ang);
            sinv[(size_t)p * half + i] = (float)sin((double)ang);
        }
}

extern "C" int sr_model_create(sr_model** out, const sr_model_config* cfg) {
    SR_REQUIRE(out && cfg, "sr_model_create: null argument");
    const sr_model_config& c = *cfg;
    SR_REQUIRE(c.vocab_size > 0 && c.hidden_size > 0 && c.intermediate_size > 0 && c.num_layers > 0,
               "sr_model_create: bad dimensions");
    SR_REQUIRE(c.num_heads > 0 && c.num_kv_heads > 0 && c.num_heads % c.num_kv_heads == 0,
               "sr_model_create: num_heads %d must be a multiple of num_kv_heads %d", c.num_heads, c.num_kv_heads);
    SR_REQUIRE(c.head_dim == 64 || c.head_dim == 128, "sr_model_create: head_dim %d not supported (64 or 128)", c.head_dim);
    SR_REQUIRE(c.hidden_size % 64 == 0 && c.intermediate_size % 64 == 0,
               "sr_model_create: hidden_size and intermediate_size must be multiples of 64");
    SR_REQUIRE(c.vocab_size % 16 == 0 || !c.has_lm_head, "sr_model_create: vocab_size must be a multiple of 16 for the sparse head");
    SR_REQUIRE(c.max_batch_tokens > 0 && c.max_batch_seqs > 0 && c.max_batch_seqs <= 65536, "sr_model_create: bad workspace sizes");
    SR_REQUIRE(c.fp32_planes == 0 || c.fp32_planes == 2 || c.fp32_planes == 3 || c.fp32_planes == SR_FP32_PLANES_F16,
               "sr_model_create: fp32_planes must be 0, 2, 3 or 16");
    sr_model* m = new sr_model();
    m->cfg = c;
    m->Tm = (int)(ceil_div64(c.max_batch_tokens, 128) * 128);
    m->Bm = c.max_batch_seqs;
    m->max_pos = 8192;
    m->layers.resize(c.num_layers);
    const int64_t H = c.hidden_size, I = c.intermediate_size, V = c.vocab_size;
    const int64_t nq = (int64_t)c.num_heads * c.head_dim, nkv = (int64_t)c.num_kv_heads * c.head_dim;
    SR_ALLOC(m->embed, V * H * 4);
    if (c.has_lm_head) SR_ALLOC(m->lm_head, V * H * 2);
    const int64_t nsg = c.fp32_planes ? split_map_w(c.fp32_planes).n_seg : 0;
    const bool f16p = c.fp32_planes == SR_FP32_PLANES_F16;
    if (c.has_lm_head && nsg) SR_ALLOC(m->lm_head_s, V * H * 2 * nsg);
    if (c.has_lm_head && f16p) SR_ALLOC(m->lm_head_i, V * 4);
    SR_ALLOC(m->norm_w, H * 4);
    for (auto& l : m->layers) {
        SR_ALLOC(l.wqkv, (nq + 2 * nkv) * H * 2);
        SR_ALLOC(l.wo, H * nq * 2);
        SR_ALLOC(l.wgu, 2 * I * H * 2);
        SR_ALLOC(l.wdown, H * I * 2);
        if (nsg) {
            SR_ALLOC(l.wqkv_s, (nq + 2 * nkv) * H * 2 * nsg);
            SR_ALLOC(l.wo_s, H * nq * 2 * nsg);
            SR_ALLOC(l.wgu_s, 2 * I * H * 2 * nsg);
            SR_ALLOC(l.wdown_s, H * I * 2 * nsg);
        }
        if (f16p) {
            SR_ALLOC(l.wqkv_i, (nq + 2 * nkv) * 4);
            SR_ALLOC(l.wo_i, H * 4);
            SR_ALLOC(l.wgu_i, 2 * I * 4);
            SR_ALLOC(l.wdown_i, H * 4);
        }
        SR_ALLOC(l.ln1, H * 4);
        SR_ALLOC(l.ln2, H * 4);
    }
    SR_ALLOC(m->rope_cos, (int64_t)m->max_pos * (c.head_dim / 2) * 4);
    SR_ALLOC(m->rope_sin, (int64_t)m->max_pos * (c.head_dim / 2) * 4);
    const int64_t Tm = m->Tm;
    SR_ALLOC(m->x, Tm * H * 4);
    SR_ALLOC(m->xn, Tm * H * 2);
    SR_ALLOC(m->qkv, Tm * (nq + 2 * nkv) * 2);
    SR_ALLOC(m->attn, Tm * nq * 2);
    SR_ALLOC(m->act, Tm * I * 2);
    SR_ALLOC(m->delta, Tm * H * 2);
    SR_ALLOC(m->span_start, m->Bm * 4);
    SR_ALLOC(m->span_len, m->Bm * 4);
    SR_ALLOC(m->pool_start, m->Bm * 4);
    SR_ALLOC(m->row_len, m->Bm * 4);
    SR_ALLOC(m->cu, (m->Bm + 1) * 4);
    SR_ALLOC(m->tok_id, Tm * 4);
    SR_ALLOC(m->pos, Tm * 4);
    SR_ALLOC(m->seq_of, Tm * 4);
    SR_ALLOC(m->key_valid, Tm);
    if (hipHostMalloc((void**)&m->h_cu, (size_t)(2 * m->Bm + 2) * 4) != hipSuccess) {
        sr_set_error("sr_model_create: hipHostMalloc failed");
        model_free(m);
        delete m;
        return SR_ERR_NOMEM;
    }
    // activations that feed GEMMs are read up to the tile edge: keep them finite
    (void)hipMemset(m->xn, 0, (size_t)(Tm * H * 2));
    (void)hipMemset(m->attn, 0, (size_t)(Tm * nq * 2));
    (void)hipMemset(m->act, 0, (size_t)(Tm * I * 2));
    std::vector<float> cs, sn;
    rope_tables(c, m->max_pos, cs, sn);
    if (hipMemcpy(m->rope_cos, cs.data(), cs.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->rope_sin, sn.data(), sn.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        sr_set_error("sr_model_create: rope table upload failed");
        model_free(m);
        delete m;
        return SR_ERR_HIP;
    }
    *out = m;
    return SR_OK;
}

extern "C" int sr_model_destroy(sr_model* m) {
    if (!m) return SR_OK;
    model_free(m);
    m->order.release();
    delete m;
    return SR_OK;
}

static int convert_rows(const void* src, int dtype, int64_t rows, int64_t cols, bf16_t* dbf, float* df32, int64_t base,
                        int interleave, hipStream_t s, bf16_t* dsplit = nullptr, int planes = 0, float* w_inv = nullptr) {
    hipLaunchKernelGGL(convert_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, src, dtype, rows, cols, dbf, df32, base,
                       interleave);
    if (dsplit && planes == SR_FP32_PLANES_F16)
        hipLaunchKernelGGL(convert_rows_split_h_kernel, dim3((unsigned)rows), dim3(256), 0, s, src, dtype, rows, cols, dsplit, w_inv, base,
                           interleave);
    else if (dsplit && planes)
        hipLaunchKernelGGL(convert_rows_split_kernel, dim3((unsigned)rows), dim3(256), 0, s, src, dtype, rows, cols, dsplit, base,
                           interleave, split_map_w(planes));
    SR_CHECK_LAUNCH();
    return SR_OK;
}

extern "C" int sr_model_set_weight(sr_model* m, const char* name, const void* d_ptr, int dtype, int64_t rows, int64_t cols,
                                   sr_stream stream) {
    SR_REQUIRE(m && name && d_ptr, "sr_model_set_weight: null argument");
    SR_REQUIRE(dtype == SR_DTYPE_F32 || dtype == SR_DTYPE_BF16, "sr_model_set_weight: dtype %d not supported", dtype);
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(m->mu);
    const sr_model_config& c = m->cfg;
    const int64_t H = c.hidden_size, I = c.intermediate_size, V = c.vocab_size;
    const int64_t nq = (int64_t)c.num_heads * c.head_dim, nkv = (int64_t)c.num_kv_heads * c.head_dim;
    std::string n(name);
    auto shape_is = [&](int64_t r, int64_t cc) { return rows == r && cols == cc; };
#define SHAPE_REQ(r, cc) SR_REQUIRE(shape_is((r), (cc)), "sr_model_set_weight: %s has shape [%lld, %lld], expected [%lld, %lld]", name, (long long)rows, (long long)cols, (long long)(r), (long long)(cc))
    if (n == "model.embed_tokens.weight") {
        SHAPE_REQ(V, H);
        SR_TRY(convert_rows(d_ptr, dtype, V, H, (c.has_lm_head && c.tie_word_embeddings) ? m->lm_head : nullptr, m->embed, 0, 0, s,
                            (c.has_lm_head && c.tie_word_embeddings) ? m->lm_head_s : nullptr, c.fp32_planes, m->lm_head_i));
        m->have_embed = true;
        if (c.has_lm_head && c.tie_word_embeddings) m->have_lm_head = true;
        return SR_OK;
    }
    if (n == "lm_head.weight") {
        SR_REQUIRE(c.has_lm_head, "sr_model_set_weight: model was created without an lm_head");
        SHAPE_REQ(V, H);
        SR_TRY(convert_rows(d_ptr, dtype, V, H, m->lm_head, nullptr, 0, 0, s, m->lm_head_s, c.fp32_planes, m->lm_head_i));
        m->have_lm_head = true;
        return SR_OK;
    }
    if (n == "model.norm.weight") {
        SHAPE_REQ(H, 1);
        SR_TRY(convert_rows(d_ptr, dtype, H, 1, nullptr, m->norm_w, 0, 0, s));
        m->have_norm = true;
        return SR_OK;
    }
    int li = -1;
    char rest[128] = {0};
    if (sscanf(name, "model.layers.%d.%127s", &li, rest) == 2 && li >= 0 && li < c.num_layers) {
        LayerW& l = m->layers[li];
        std::string r(rest);
        if (r == "self_attn.q_proj.weight") { SHAPE_REQ(nq, H); SR_TRY(convert_rows(d_ptr, dtype, nq, H, l.wqkv, nullptr, 0, 0, s, l.wqkv_s, c.fp32_planes, l.wqkv_i)); l.have |= 1; return SR_OK; }
        if (r == "self_attn.k_proj.weight") { SHAPE_REQ(nkv, H); SR_TRY(convert_rows(d_ptr, dtype, nkv, H, l.wqkv, nullptr, nq, 0, s, l.wqkv_s, c.fp32_planes, l.wqkv_i)); l.have |= 2; return SR_OK; }
        if (r == "self_attn.v_proj.weight") { SHAPE_REQ(nkv, H); SR_TRY(convert_rows(d_ptr, dtype, nkv, H, l.wqkv, nullptr, nq + nkv, 0, s, l.wqkv_s, c.fp32_planes, l.wqkv_i)); l.have |= 4; return SR_OK; }
        if (r == "self_attn.o_proj.weight") { SHAPE_REQ(H, nq); SR_TRY(convert_rows(d_ptr, dtype, H, nq, l.wo, nullptr, 0, 0, s, l.wo_s, c.fp32_planes, l.wo_i)); l.have |= 8; return SR_OK; }
        if (r == "mlp.gate_proj.weight") { SHAPE_REQ(I, H); SR_TRY(convert_rows(d_ptr, dtype, I, H, l.wgu, nullptr, 0, 1, s, l.wgu_s, c.fp32_planes, l.wgu_i)); l.have |= 16; return SR_OK; }
        if (r == "mlp.up_proj.weight") { SHAPE_REQ(I, H); SR_TRY(convert_rows(d_ptr, dtype, I, H, l.wgu, nullptr, 0, 2, s, l.wgu_s, c.fp32_planes, l.wgu_i)); l.have |= 32; return SR_OK; }
        if (r == "mlp.down_proj.weight") { SHAPE_REQ(H, I); SR_TRY(convert_rows(d_ptr, dtype, H, I, l.wdown, nullptr, 0, 0, s, l.wdown_s, c.fp32_planes, l.wdown_i)); l.have |= 64; return SR_OK; }
        if (r == "input_layernorm.weight") { SHAPE_REQ(H, 1); SR_TRY(convert_rows(d_ptr, dtype, H, 1, nullptr, l.ln1, 0, 0, s)); l.have |= 128; return SR_OK; }
        if (r == "post_attention_layernorm.weight") { SHAPE_REQ(H, 1); SR_TRY(convert_rows(d_ptr, dtype, H, 1, nullptr, l.ln2, 0, 0, s)); l.have |= 256; return SR_OK; }
    }
    sr_set_error("sr_model_set_weight: unknown tensor name '%s'", name);
    return SR_ERR_INVALID;
}

extern "C" int sr_model_finalize(sr_model* m) {
    SR_REQUIRE(m, "sr_model_finalize: null model");
    SR_REQUIRE(m->have_embed, "sr_model_finalize: model.embed_tokens.weight missing");
    SR_REQUIRE(m->have_norm, "sr_model_finalize: model.norm.weight missing");
    SR_REQUIRE(!m->cfg.has_lm_head || m->have_lm_head, "sr_model_finalize: lm_head.weight missing");
    for (int i = 0; i < m->cfg.num_layers; ++i)
        SR_REQUIRE(m->layers[i].have == 511, "sr_model_finalize: layer %d is missing tensors (mask 0x%x)", i, m->layers[i].have);
    // sr_model_set_weight converted the planes on the CALLER's streams; a non-blocking stream does not order with the null
    // stream the reduction below runs on, and a cmax read from half-written planes would let EPI_SWIGLU_SPLIT_H overflow fp16
    SR_CHECK_HIP(hipDeviceSynchronize());
    if (m->cfg.fp32_planes == SR_FP32_PLANES_F16) {      // row bound of every layer's SwiGLU output (EPI_SWIGLU_SPLIT_H)
        for (int i = 0; i < m->cfg.num_layers; ++i) {
            LayerW& l = m->layers[i];
            if (!l.gu_cmax) SR_CHECK_HIP(hipMalloc((void**)&l.gu_cmax, 4));
            SR_CHECK_HIP(hipMemsetAsync(l.gu_cmax, 0, 4, nullptr));
            hipLaunchKernelGGL(gu_cmax_kernel, dim3((unsigned)ceil_div64(m->cfg.intermediate_size, 4)), dim3(256), 0, nullptr, l.wgu_s, l.wgu_i,
                               m->cfg.intermediate_size, m->cfg.hidden_size, l.gu_cmax);
        }
        SR_CHECK_LAUNCH();
    }
    SR_CHECK_HIP(hipDeviceSynchronize());
    m->finalized = true;
    return SR_OK;
}

// --------------------------------------------------------------- forward ------
// Precision regimes:
//   PREC_BF16  torch.autocast(bf16): documents (indexer.py:46-52, :255-256) and sparse queries (:390-391)
//   PREC_FP32  no autocast: dense queries (eval_dense.py:94-106) and examples/quick_start.py - every nn.Linear is an fp32
//              GEMM there.  Here each GEMM runs on the bf16 MFMA pipe over split-bf16 planes of BOTH operands
//              (kernels.h: SplitMap; 3 planes / 6 products carry the full 24-bit significands, fp32 accumulate), the
//              activations between GEMMs stay fp32 (residual stream, rotated q/k/v, softmax, SwiGLU) and are split into
//              planes by the kernel that produces them.
enum { PREC_BF16 = 0, PREC_FP32 = 1 };

static int ensure_fp32_workspace(sr_model* m) {
    if (m->xs) return SR_OK;
    const sr_model_config& c = m->cfg;
    SR_REQUIRE(c.fp32_planes > 0, "encode(fp32): the model was created with fp32_planes = 0 (bf16 regime only)");
    const int64_t Tm = m->Tm, H = c.hidden_size, I = c.intermediate_size;
    const int64_t nq = (int64_t)c.num_heads * c.head_dim, nkv = (int64_t)c.num_kv_heads * c.head_dim;
    const int64_t nsg = split_map_a(c.fp32_planes).n_seg;
    auto A = [&](void** p, int64_t bytes) {
        if (hipMalloc(p, (size_t)bytes) != hipSuccess) { *p = nullptr; return false; }
        return hipMemset(*p, 0, (size_t)bytes) == hipSuccess;
    };
    const bool f16p = c.fp32_planes == SR_FP32_PLANES_F16;
    bool ok = A((void**)&m->xs, Tm * H * 2 * nsg) && A((void**)&m->qkv_f, Tm * (nq + 2 * nkv) * 4) &&
              A((void**)&m->attn_s, Tm * nq * 2 * nsg) && A((void**)&m->act_s, Tm * I * 2 * nsg);
    if (ok && f16p)
        ok = A((void**)&m->attn_f, Tm * nq * 4) && A((void**)&m->act_f, Tm * I * 4) && A((void**)&m->xs_i, Tm * 4) &&
             A((void**)&m->attn_i, Tm * 4) && A((void**)&m->act_i, Tm * 4) && A((void**)&m->act_sc, Tm * 4);
    if (!ok) {
        auto F = [](void* p) { if (p) (void)hipFree(p); };
        F(m->xs); F(m->qkv_f); F(m->attn_s); F(m->act_s); F(m->attn_f); F(m->act_f); F(m->xs_i); F(m->attn_i); F(m->act_i);
        m->xs = nullptr; m->qkv_f = nullptr; m->attn_s = nullptr; m->act_s = nullptr;
        m->attn_f = nullptr; m->act_f = nullptr; m->xs_i = nullptr; m->attn_i = nullptr; m->act_i = nullptr;
        sr_set_error("encode(fp32): hipMalloc of the fp32-regime workspace failed (%lld tokens)", (long long)Tm);
        return SR_ERR_NOMEM;
    }
    return SR_OK;
}

static int launch_rmsnorm_split(float* x, const float* embed, const int* tok_id, const float* w, bf16_t* xs, int T, int H,
                                float eps, const SplitMap& map, hipStream_t s) {
    if (T == 0) return SR_OK;
    hipLaunchKernelGGL(rmsnorm_split_kernel, dim3((unsigned)ceil_div64(T, 4)), dim3(256), 0, s, x, embed, tok_id, w, xs, T, H, eps,
                       map);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

static int model_forward(sr_model* m, const int64_t* d_ids, const int64_t* d_mask, int B, int L, int mode, int prec,
                         hipStream_t s, int* T_out, const int32_t* d_shift = nullptr) {
    const sr_model_config& c = m->cfg;
    SR_REQUIRE(m->finalized, "encode: sr_model_finalize was not called");
    SR_REQUIRE(B >= 1 && L >= 1, "encode: bad batch shape [%d, %d]", B, L);
    SR_REQUIRE(B <= m->Bm, "encode: batch of %d sequences exceeds max_batch_seqs %d", B, m->Bm);
    SR_REQUIRE(L <= m->max_pos, "encode: sequence length %d exceeds %d", L, m->max_pos);
    SR_REQUIRE(d_ids && d_mask, "encode: null input");
    if (prec == PREC_FP32) SR_TRY(ensure_fp32_workspace(m));
    const int H = c.hidden_size, I = c.intermediate_size;
    const int nq = c.num_heads * c.head_dim, nkv = c.num_kv_heads * c.head_dim;

    hipLaunchKernelGGL(plan_rows_kernel, dim3(B), dim3(64), 0, s, d_mask, B, L, mode, m->span_start, m->span_len,
                       m->pool_start, m->row_len, d_shift);
    hipLaunchKernelGGL(plan_scan_kernel, dim3(1), dim3(256), 0, s, m->span_len, B, m->cu);
    SR_CHECK_LAUNCH();
    SR_CHECK_HIP(hipMemcpyAsync(m->h_cu, m->cu, (size_t)(B + 1) * 4, hipMemcpyDeviceToHost, s));
    SR_CHECK_HIP(hipStreamSynchronize(s));
    const int T = m->h_cu[B];
    int max_len = 0;
    // a row whose attention_mask is all zero packs to zero tokens: every kernel skips it and both heads return the
    // zero vector for it (the reference's sparse head gives exactly that; its dense head would average garbage)
    for (int b = 0; b < B; ++b) max_len = (m->h_cu[b + 1] - m->h_cu[b]) > max_len ? (m->h_cu[b + 1] - m->h_cu[b]) : max_len;
    SR_REQUIRE(T <= m->Tm, "encode: batch packs to %d tokens, workspace holds %d (raise max_batch_tokens or split the batch)", T, m->Tm);
    *T_out = T;
    m->last_T = T;
    if (T == 0) return SR_OK;
    hipLaunchKernelGGL(plan_tokens_kernel, dim3(B), dim3(128), 0, s, d_ids, d_mask, L, m->span_start, m->cu, m->tok_id,
                       m->pos, m->key_valid, m->seq_of, c.vocab_size, mode, d_shift);
    SR_CHECK_LAUNCH();

    if (prec == PREC_FP32 && c.fp32_planes == SR_FP32_PLANES_F16) {
        // fp16 planes: every GEMM input is split by ONE kernel that sees whole rows (norm + split, or split of an fp32
        // buffer), because the power-of-two scale is per row; 3 plane products per GEMM
        auto split_rows = [&](float* src, const float* embed, const int* tok, const float* w, bf16_t* dst, float* inv, int K) {
            launch_rows_split_h(src, embed, tok, w, dst, inv, T, K, c.rms_norm_eps, s);
        };
        // dev switch SR_FP32_FUSED_ACT=0: SwiGLU output as fp32 + a separate row-split pass (A/B, and the reference of the test)
        const char* env_fa = sr_dev_getenv("SR_FP32_FUSED_ACT");
        const bool fused_act = m->layers[0].gu_cmax && !(env_fa && *env_fa == '0');
        for (int li = 0; li < c.num_layers; ++li) {
            LayerW& l = m->layers[li];
            split_rows(m->x, li == 0 ? m->embed : (const float*)nullptr, li == 0 ? m->tok_id : (const int*)nullptr, l.ln1, m->xs, m->xs_i, H);
            GemmArgs g{};
            g.A = m->xs; g.W = l.wqkv_s; g.M = T; g.N = nq + 2 * nkv; g.K = 3 * H; g.C = m->qkv_f; g.a_scale = m->xs_i; g.w_scale = l.wqkv_i;
            g.pos = m->pos; g.rope_cos = m->rope_cos; g.rope_sin = m->rope_sin; g.n_rope = nq + nkv; g.head_dim = c.head_dim;
            SR_TRY(launch_gemm_bf16(EPI_QKV_ROPE_F32_H, g, s));
            AttnF32Args a{};
            a.qkv = m->qkv_f; a.out_f32 = m->attn_f; a.out = nullptr; a.cu_seqlens = m->cu; a.key_valid = m->key_valid;
            a.B = B; a.nh = c.num_heads; a.nkv = c.num_kv_heads; a.hd = c.head_dim;
            a.scale = 1.0f / sqrtf((float)c.head_dim); a.max_seqlen = max_len;
            SR_TRY(launch_attention_f32(a, s));
            split_rows(m->attn_f, nullptr, nullptr, nullptr, m->attn_s, m->attn_i, nq);
            g = GemmArgs{};
            g.A = m->attn_s; g.W = l.wo_s; g.M = T; g.N = H; g.K = 3 * nq; g.C = m->x; g.a_scale = m->attn_i; g.w_scale = l.wo_i;
            SR_TRY(launch_gemm_bf16(EPI_RESID_F32_H, g, s));
            if (fused_act) {
                // the norm kernel also fixes the scale of the row's SwiGLU output (a rigorous bound, no overflow), so the
                // gate-up GEMM writes the down_proj's fp16 plane segments itself: no fp32 intermediate, no split pass
                launch_rows_split_h(m->x, nullptr, nullptr, l.ln2, m->xs, m->xs_i, T, H, c.rms_norm_eps, s, l.gu_cmax, m->act_sc, m->act_i);
                g = GemmArgs{};
                g.A = m->xs; g.W = l.wgu_s; g.M = T; g.N = 2 * I; g.K = 3 * H; g.C = m->act_s; g.a_scale = m->xs_i; g.w_scale = l.wgu_i;
                g.out_scale = m->act_sc;
                SR_TRY(launch_gemm_bf16(EPI_SWIGLU_SPLIT_H, g, s));
            } else {
                split_rows(m->x, nullptr, nullptr, l.ln2, m->xs, m->xs_i, H);
                g = GemmArgs{};
                g.A = m->xs; g.W = l.wgu_s; g.M = T; g.N = 2 * I; g.K = 3 * H; g.C = m->act_f; g.a_scale = m->xs_i; g.w_scale = l.wgu_i;
                SR_TRY(launch_gemm_bf16(EPI_SWIGLU_F32_H, g, s));
                split_rows(m->act_f, nullptr, nullptr, nullptr, m->act_s, m->act_i, I);
            }
            g = GemmArgs{};
            g.A = m->act_s; g.W = l.wdown_s; g.M = T; g.N = H; g.K = 3 * I; g.C = m->x; g.a_scale = m->act_i; g.w_scale = l.wdown_i;
            SR_TRY(launch_gemm_bf16(EPI_RESID_F32_H, g, s));
        }
        SR_CHECK_LAUNCH();
        return SR_OK;
    }
    if (prec == PREC_FP32) {
        const SplitMap ma = split_map_a(c.fp32_planes);
        const int nsg = ma.n_seg;
        for (int li = 0; li < c.num_layers; ++li) {
            LayerW& l = m->layers[li];
            // input_layernorm (the embedding gather is fused into the first one); o_proj / down_proj add straight into x
            SR_TRY(launch_rmsnorm_split(m->x, li == 0 ? m->embed : (const float*)nullptr, li == 0 ? m->tok_id : (const int*)nullptr,
                                        l.ln1, m->xs, T, H, c.rms_norm_eps, ma, s));
            GemmArgs g{};
            g.A = m->xs; g.W = l.wqkv_s; g.M = T; g.N = nq + 2 * nkv; g.K = nsg * H; g.C = m->qkv_f;
            g.pos = m->pos; g.rope_cos = m->rope_cos; g.rope_sin = m->rope_sin; g.n_rope = nq + nkv; g.head_dim = c.head_dim;
            SR_TRY(launch_gemm_bf16(EPI_QKV_ROPE_F32, g, s));
            AttnF32Args a{};
            a.qkv = m->qkv_f; a.out = m->attn_s; a.cu_seqlens = m->cu; a.key_valid = m->key_valid;
            a.B = B; a.nh = c.num_heads; a.nkv = c.num_kv_heads; a.hd = c.head_dim;
            a.scale = 1.0f / sqrtf((float)c.head_dim); a.max_seqlen = max_len; a.out_map = ma;
            SR_TRY(launch_attention_f32(a, s));
            g = GemmArgs{};
            g.A = m->attn_s; g.W = l.wo_s; g.M = T; g.N = H; g.K = nsg * nq; g.C = m->x;
            SR_TRY(launch_gemm_bf16(EPI_RESID_F32, g, s));
            SR_TRY(launch_rmsnorm_split(m->x, (const float*)nullptr, (const int*)nullptr, l.ln2, m->xs, T, H, c.rms_norm_eps, ma, s));
            g = GemmArgs{};
            g.A = m->xs; g.W = l.wgu_s; g.M = T; g.N = 2 * I; g.K = nsg * H; g.C = m->act_s; g.out_map = ma;
            SR_TRY(launch_gemm_bf16(EPI_SWIGLU_SPLIT, g, s));
            g = GemmArgs{};
            g.A = m->act_s; g.W = l.wdown_s; g.M = T; g.N = H; g.K = nsg * I; g.C = m->x;
            SR_TRY(launch_gemm_bf16(EPI_RESID_F32, g, s));
        }
        return SR_OK;
    }

    // embedding gather fused with the first input_layernorm
    SR_TRY(launch_rmsnorm(m->x, m->embed, m->tok_id, (const bf16_t*)nullptr, m->layers[0].ln1, m->xn,
                       (float*)nullptr, T, H, c.rms_norm_eps, s));
    for (int li = 0; li < c.num_layers; ++li) {
        LayerW& l = m->layers[li];
        if (li > 0) {
            // adds the previous layer's down_proj output, then input_layernorm
            SR_TRY(launch_rmsnorm(m->x, (const float*)nullptr, (const int*)nullptr,
                               (const bf16_t*)m->delta, l.ln1, m->xn, (float*)nullptr, T, H, c.rms_norm_eps, s));
        }
        GemmArgs g{};
        g.A = m->xn; g.W = l.wqkv; g.M = T; g.N = nq + 2 * nkv; g.K = H; g.C = m->qkv;
        g.pos = m->pos; g.rope_cos = m->rope_cos; g.rope_sin = m->rope_sin; g.n_rope = nq + nkv; g.head_dim = c.head_dim;
        SR_TRY(launch_gemm_bf16(EPI_QKV_ROPE, g, s));
        AttnArgs a{};
        a.qkv = m->qkv; a.out = m->attn; a.cu_seqlens = m->cu; a.pos = m->pos; a.key_valid = m->key_valid;
        a.rope_cos = m->rope_cos; a.rope_sin = m->rope_sin; a.B = B; a.nh = c.num_heads; a.nkv = c.num_kv_heads;
        a.hd = c.head_dim; a.scale = 1.0f / sqrtf((float)c.head_dim);
        a.apply_rope = 0; a.max_seqlen = max_len;
        SR_TRY(launch_attention(a, s));
        g = GemmArgs{};
        g.A = m->attn; g.W = l.wo; g.M = T; g.N = H; g.K = nq; g.C = m->delta;
        SR_TRY(launch_gemm_bf16(EPI_STORE_BF16, g, s));
        // x += o_proj output, then post_attention_layernorm
        SR_TRY(launch_rmsnorm(m->x, (const float*)nullptr, (const int*)nullptr,
                           (const bf16_t*)m->delta, l.ln2, m->xn, (float*)nullptr, T, H, c.rms_norm_eps, s));
        g = GemmArgs{};
        g.A = m->xn; g.W = l.wgu; g.M = T; g.N = 2 * I; g.K = H; g.C = m->act;
        SR_TRY(launch_gemm_bf16(EPI_SWIGLU, g, s));
        g = GemmArgs{};
        g.A = m->act; g.W = l.wdown; g.M = T; g.N = H; g.K = I; g.C = m->delta;
        SR_TRY(launch_gemm_bf16(EPI_STORE_BF16, g, s));   // added to x by the next norm kernel (or the head)
    }
    // fold the last down_proj output into the residual stream so that the heads see the complete hidden state
    SR_TRY(launch_rmsnorm(m->x, (const float*)nullptr, (const int*)nullptr,
                       (const bf16_t*)m->delta, (const float*)nullptr, (bf16_t*)nullptr, (float*)nullptr, T, H, c.rms_norm_eps, s));
    return SR_OK;
}

static int head_dense(sr_model* m, int B, int T, int prec, float* d_out, hipStream_t s) {
    const int H = m->cfg.hidden_size;
    if (T == 0) {
        SR_CHECK_HIP(hipMemsetAsync(d_out, 0, (size_t)B * H * 4, s));
        return SR_OK;
    }
    // a [T] float2 scratch: xn (bf16 [Tm, H]) is free after the last layer in the bf16 regime, xs in the fp32 regime
    float2* stats = reinterpret_cast<float2*>(prec == PREC_FP32 ? (void*)m->xs : (void*)m->xn);
    hipLaunchKernelGGL(dense_head_stats_kernel, dim3((unsigned)ceil_div64(T, 4)), dim3(256), 0, s, m->x, m->norm_w, stats, T, H,
                       m->cfg.rms_norm_eps);
    hipLaunchKernelGGL(dense_head_pool_kernel, dim3((unsigned)B, (unsigned)ceil_div64(H, 256)), dim3(256), 0, s, m->x, m->norm_w,
                       stats, m->cu, m->pos, m->pool_start, d_out, H);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

static int head_sparse(sr_model* m, int B, int T, int prec, float* d_out, hipStream_t s) {
    const int H = m->cfg.hidden_size, V = m->cfg.vocab_size;
    SR_CHECK_HIP(hipMemsetAsync(d_out, 0, (size_t)B * V * 4, s));
    if (T == 0) return SR_OK;       // log(1 + relu(max over no tokens)) = 0
    GemmArgs g{};
    GemmEpilogue epi = EPI_SEGMAX;
    if (prec == PREC_FP32 && m->cfg.fp32_planes == SR_FP32_PLANES_F16) {
        launch_rows_split_h(m->x, (const float*)nullptr, (const int*)nullptr, (const float*)m->norm_w, m->xs, m->xs_i, T, H,
                            m->cfg.rms_norm_eps, s);
        g.A = m->xs; g.W = m->lm_head_s; g.K = 3 * H; g.a_scale = m->xs_i; g.w_scale = m->lm_head_i;
        epi = EPI_SEGMAX_H;
    } else if (prec == PREC_FP32) {
        const SplitMap ma = split_map_a(m->cfg.fp32_planes);
        SR_TRY(launch_rmsnorm_split(m->x, (const float*)nullptr, (const int*)nullptr, m->norm_w, m->xs, T, H, m->cfg.rms_norm_eps,
                                    ma, s));
        g.A = m->xs; g.W = m->lm_head_s; g.K = ma.n_seg * H;
    } else {
        // final norm -> bf16 GEMM input
        SR_TRY(launch_rmsnorm(m->x, (const float*)nullptr,
                           (const int*)nullptr, (const bf16_t*)nullptr, m->norm_w, m->xn, (float*)nullptr, T, H, m->cfg.rms_norm_eps, s));
        g.A = m->xn; g.W = m->lm_head; g.K = H;
    }
    g.M = T; g.N = V; g.C = d_out; g.seq_of = m->seq_of; g.out_ld = V;
    // rows with seq_of == -2 (mask == 0 inside the span) are skipped by the segmented max
    SR_TRY(launch_gemm_bf16(epi, g, s));
    const int64_t n = (int64_t)B * V;
    // under autocast the lm_head output is bf16 (rounded before the fp32 upcast, llm_encoder.py:187-188); in fp32 it is not
    hipLaunchKernelGGL(sparse_finish_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, s, d_out, n,
                       powf((float)H, -0.25f), prec == PREC_FP32 ? 0 : 1);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

// mode 0: dense head, 1: sparse head, 2: both from ONE backbone pass (the hybrid model)
static int encode_any(sr_model* m, const int64_t* d_input_ids, const int64_t* d_attention_mask, int32_t B, int32_t L, int mode,
                      int prec, float* d_dense, float* d_sparse, hipStream_t s, const int32_t* d_shift = nullptr) {
    std::lock_guard<std::mutex> lock(m->mu);
    StreamOrder::Scope in_order(m->order, s);
    int T = 0;
    SR_TRY(model_forward(m, d_input_ids, d_attention_mask, B, L, mode, prec, s, &T, d_shift));
    if (mode != 1) SR_TRY(head_dense(m, B, T, prec, d_dense, s));     // first: its scratch is the sparse head's GEMM input
    if (mode != 0) SR_TRY(head_sparse(m, B, T, prec, d_sparse, s));
    return SR_OK;
}

static int encode_dense(sr_model* m, const int64_t* d_input_ids, const int64_t* d_attention_mask, int32_t B, int32_t L, int prec,
                        float* d_out, hipStream_t s) {
    return encode_any(m, d_input_ids, d_attention_mask, B, L, 0, prec, d_out, nullptr, s);
}

static int encode_sparse(sr_model* m, const int64_t* d_input_ids, const int64_t* d_attention_mask, int32_t B, int32_t L, int prec,
                         float* d_out, hipStream_t s) {
    return encode_any(m, d_input_ids, d_attention_mask, B, L, 1, prec, nullptr, d_out, s);
}

extern "C" int sr_encode_dense(sr_model* m, const int64_t* d_input_ids, const int64_t* d_attention_mask, int32_t B, int32_t L,
                               float* d_out, sr_stream stream) {
    SR_REQUIRE(m && d_out, "sr_encode_dense: null argument");
    return encode_dense(m, d_input_ids, d_attention_mask, B, L, PREC_BF16, d_out, (hipStream_t)stream);
}

extern "C" int sr_encode_dense_fp32(sr_model* m, const int64_t* d_input_ids, const int64_t* d_attention_mask, int32_t B, int32_t L,
                                    float* d_out, sr_stream stream) {
    SR_REQUIRE(m && d_out, "sr_encode_dense_fp32: null argument");
    return encode_dense(m, d_input_ids, d_attention_mask, B, L, PREC_FP32, d_out, (hipStream_t)stream);
}

extern "C" int sr_encode_sparse(sr_model* m, const int64_t* d_input_ids, const int64_t* d_attention_mask, int32_t B, int32_t L,
                                float* d_out, sr_stream stream) {
    SR_REQUIRE(m && d_out, "sr_encode_sparse: null argument");
    SR_REQUIRE(m->cfg.has_lm_head, "sr_encode_sparse: model was created without an lm_head (LlamaBiModel)");
    return encode_sparse(m, d_input_ids, d_attention_mask, B, L, PREC_BF16, d_out, (hipStream_t)stream);
}

extern "C" int sr_encode_sparse_fp32(sr_model* m, const int64_t* d_input_ids, const int64_t* d_attention_mask, int32_t B, int32_t L,
                                     float* d_out, sr_stream stream) {
    SR_REQUIRE(m && d_out, "sr_encode_sparse_fp32: null argument");
    SR_REQUIRE(m->cfg.has_lm_head, "sr_encode_sparse_fp32: model was created without an lm_head (LlamaBiModel)");
    return encode_sparse(m, d_input_ids, d_attention_mask, B, L, PREC_FP32, d_out, (hipStream_t)stream);
}

extern "C" int sr_encode_both(sr_model* m, const int64_t* d_input_ids, const int64_t* d_attention_mask, int32_t B, int32_t L,
                              int32_t fp32, float* d_out_sparse, float* d_out_dense, sr_stream stream) {
    SR_REQUIRE(m && d_out_sparse && d_out_dense, "sr_encode_both: null argument");
    SR_REQUIRE(m->cfg.has_lm_head, "sr_encode_both: model was created without an lm_head (LlamaBiModel)");
    return encode_any(m, d_input_ids, d_attention_mask, B, L, 2, fp32 ? PREC_FP32 : PREC_BF16, d_out_dense, d_out_sparse,
                      (hipStream_t)stream);
}

// Rows of SEVERAL left-padded batches in one call (the drop-in drivers hand the encoder eval_batch_size rows at a time,
// eval_dense.py:94-106 / indexer.py:382-403; 128 queries are ~1 100 tokens - far too few rows for 256-row GEMM tiles): the caller
// lays the batches into one [B, L] matrix (L = the widest batch, narrower batches get extra left padding) and passes, per row, how far
// it moved right.  Positions are counted from there, so every row is computed with the position_ids it had in its own batch and its
// output is bit-identical to encoding that batch alone (the kernels do not depend on batch composition).
extern "C" int sr_encode_rows(sr_model* m, const int64_t* d_input_ids, const int64_t* d_attention_mask, int32_t B, int32_t L,
                              const int32_t* d_row_shift, int32_t mode, int32_t fp32, float* d_out_sparse, float* d_out_dense,
                              sr_stream stream) {
    SR_REQUIRE(m, "sr_encode_rows: null model");
    SR_REQUIRE(mode >= 0 && mode <= 2, "sr_encode_rows: mode %d (0 dense, 1 sparse, 2 both)", mode);
    SR_REQUIRE(mode == 1 || d_out_dense, "sr_encode_rows: null dense output");
    SR_REQUIRE(mode == 0 || d_out_sparse, "sr_encode_rows: null sparse output");
    SR_REQUIRE(mode == 0 || m->cfg.has_lm_head, "sr_encode_rows: model was created without an lm_head (LlamaBiModel)");
    return encode_any(m, d_input_ids, d_attention_mask, B, L, mode, fp32 ? PREC_FP32 : PREC_BF16, d_out_dense, d_out_sparse,
                      (hipStream_t)stream, d_row_shift);
}

extern "C" int sr_model_last_hidden(sr_model* m, float* d_out, int64_t capacity_rows, int64_t* n_tokens, sr_stream stream) {
    SR_REQUIRE(m && d_out && n_tokens, "sr_model_last_hidden: null argument");
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(m->mu);
    StreamOrder::Scope in_order(m->order, s);
    const int T = m->last_T;
    *n_tokens = T;
    SR_REQUIRE(capacity_rows >= T, "sr_model_last_hidden: capacity %lld < %d tokens", (long long)capacity_rows, T);
    if (T == 0) return SR_OK;
    SR_TRY(launch_rmsnorm(m->x, (const float*)nullptr,
                       (const int*)nullptr, (const bf16_t*)nullptr, m->norm_w, (bf16_t*)nullptr, d_out, T, m->cfg.hidden_size, m->cfg.rms_norm_eps, s));
    return SR_OK;
}

extern "C" int sr_lora_merge(float* d_W, const float* d_A, const float* d_B, int64_t out_features, int64_t in_features,
                             int32_t r, float scale, sr_stream stream) {
    SR_REQUIRE(d_W && d_A && d_B && out_features > 0 && in_features > 0 && r > 0, "sr_lora_merge: bad argument");
    SR_REQUIRE(out_features < 65536 * 32, "sr_lora_merge: out_features too large");
    hipLaunchKernelGGL(lora_merge_kernel, dim3((unsigned)ceil_div64(in_features, 256), (unsigned)out_features), dim3(256), 0,
                       (hipStream_t)stream, d_W, d_A, d_B, out_features, in_features, r, scale);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

extern "C" int sr_sparse_compact(const float* d_reps, int64_t B, int64_t V, int64_t* d_row_ptr, int32_t* d_cols, float* d_vals,
                                 int64_t capacity, int64_t* h_nnz, sr_stream stream) {
    SR_REQUIRE(d_reps && d_row_ptr && h_nnz && B >= 0 && V > 0, "sr_sparse_compact: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) { *h_nnz = 0; return SR_OK; }
    int64_t* d_cnt = nullptr;
    SR_CHECK_HIP(hipMalloc((void**)&d_cnt, (size_t)B * 8));
    hipLaunchKernelGGL(nnz_count_kernel, dim3((unsigned)B), dim3(256), 0, s, d_reps, V, d_cnt);
    hipLaunchKernelGGL(nnz_scan_kernel, dim3(1), dim3(64), 0, s, d_cnt, B, d_row_ptr);
    int64_t total = 0;
    hipError_t e = hipMemcpyAsync(&total, d_row_ptr + B, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_cnt);
    SR_CHECK_HIP(e);
    *h_nnz = total;
    if (total > capacity || (total > 0 && (!d_cols || !d_vals))) {
        sr_set_error("sr_sparse_compact: %lld non-zeros, capacity %lld", (long long)total, (long long)capacity);
        return SR_ERR_NOMEM;
    }
    if (total > 0) {
        hipLaunchKernelGGL(nnz_fill_kernel, dim3((unsigned)B), dim3(256), 0, s, d_reps, V, d_row_ptr, d_cols, d_vals, capacity);
        SR_CHECK_LAUNCH();
    }
    return SR_OK;
}
