// Exact dense top-k at fp16 MFMA speed: a certified filter in front of an exact re-score (gfx950).
//
// sr_dense_search must return what faiss.IndexFlatIP.search returns (scaling_retriever/indexer.py:210-214): the k largest
// fp32 inner products.  dense_score_pipe_kernel computes every one of the nq x N products on the fp32 MFMA pipe (157 TF
// peak).  The 16-bit MFMA pipe is 16x faster, and ONE plane product q0 . d0 - the fp16 rounding of the (power-of-two scaled)
// query against the fp16 rounding of the (scaled) document - differs from the fp32 chain S_x by an amount that is bounded
// PER PAIR by quantities known exactly at index / query time, so it can decide which few documents need exact arithmetic:
//
//   1. upper-bound pass (dense_split_kernel<true>): U(q, j) = q0 . d0 + e(q, j) >= S_x(q, j) for every pair; the kp = 3k
//      documents with the largest U per query are the candidates (fused top-k, unchanged);
//   2. exact pass: S_x for the first k candidates, then for every later candidate whose U reaches the smallest exact score
//      found among those (a candidate with U below it cannot be in the top-k) - the same fp32 fmaf chain, in the same k
//      order, as dense_score_pipe_kernel (oracle: scoring.mfma_korder) - then the top-k by (S_x desc, doc index asc);
//   3. certificate: a document that is NOT a candidate has S_x <= U <= U_kp (the kp-th largest U).  If U_kp is strictly
//      below the k-th best exact score found among the candidates, no outsider can enter the top-k or tie with its last
//      member: the result IS the exact top-k.
// A query whose certificate fails (more than kp - k documents inside the margin: near-duplicates, a zero query) or for which
// any re-scored pair violates S_x in [U - 2e, U] (the bound, checked on every pair that is re-scored) is re-done by the exact
// kernel - that query alone (sr_dense_search gathers the flagged queries into one small exact batch).  Correctness never
// rests on the filter; tests/test_dense_filtered_gpu.py and every bench.py run compare with the exact kernel bit for bit.
//
// The bound.  Primes = the scaled domain (q' = q sq, d' = d sd, sq / sd powers of two: exact).  q0 = fp16(q'), d0 = fp16(d').
//   q'.d' - q0.d0 = q'.(d' - d0) + (q' - q0).d0, so by Cauchy-Schwarz   |q'.d' - q0.d0| <= |q'| |d' - d0| + |q' - q0| |d0|
//   with the ACTUAL residual norms |d' - d0| (per document, stored at index time) and |q' - q0| (per query): no worst-case
//   unit roundoff enters (round 2 used 2^-9 |x| per bf16 plane - but bf16 has 8 significand bits, its unit roundoff is 2^-8;
//   that constant was wrong by 2x.  fp16 planes: 11 bits, typical residual 2.3e-4 |x|, and nothing is assumed about it).
//   fp32 chain of H fmaf: |S_x - q.d| <= gamma_H |q||d|  (u = 2^-24, gamma_n = n u / (1 - n u));  the MFMA's fp32 summation
//   of the H exact products (fp16 x fp16 is exact in fp32), modelled as no better than a plain fp32 sum with truncation:
//   <= 2 H u |q0||d0|.   sigma(H) = 1.25 * 3 H 2^-24 + 1e-5 covers both and the epilogue's roundings.
//   e'(q, j) = A'[q] X'[j] + B'[q] Y'[j],  A' = |q'| 1.001, B' = |q' - q0| 1.001,
//              X' = (|d' - d0| + sigma |d'|) 1.001, Y' = |d0| 1.001          (1.001: the fp32 evaluation of the norms)
//   U' = acc + e' (two fmas in the kernel epilogue), compared with tau sq sd; a survivor's key carries U = U' / (sq sd).
//   At H = 2048 on unit-scale Gaussian data e ~ 8e-4 |q||d| (4.6e-4 of it sigma), against 4.4e-3 |q| max|d| in round 2.
#include "dense_filter.h"
#include <math.h>

double sr_filter_sigma(int H) { return 1.25 * (3.0 * (double)H * ldexp(1.0, -24)) + 1.0e-5; }

__global__ __launch_bounds__(256) void filter_absmax_kernel(const float* __restrict__ rows, int64_t n4, unsigned int* __restrict__ out) {
    unsigned int mx = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(rows)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned int b = __float_as_uint(v[e]) & 0x7fffffffu;     // |x| as bits: ordered like the magnitudes, NaN on top
            mx = b > mx ? b : mx;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { const unsigned int o = __shfl_xor(mx, off); mx = o > mx ? o : mx; }
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(out, mx);
}

int launch_filter_absmax(const float* rows, int64_t n, int H, unsigned int* d_absmax_bits, hipStream_t s) {
    const int64_t n4 = n * (int64_t)H / 4;
    if (n4 == 0) return SR_OK;
    int64_t blocks = ceil_div64(n4, 256 * 8);
    if (blocks > (1 << 16)) blocks = 1 << 16;
    hipLaunchKernelGGL(filter_absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, rows, n4, d_absmax_bits);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

bool sr_filter_scale_of(float absmax, float* scale, float* inv_scale) {
    *scale = *inv_scale = 1.0f;
    if (!(absmax < 3.0e38f)) return false;
    if (!(absmax > 0.f)) return true;                       // all zeros: any scale
    int e;
    (void)frexpf(absmax, &e);                               // absmax = m 2^e, m in [0.5, 1)
    int t = 15 - e;
    if (t < -SR_FILTER_MAX_SHIFT) return false;             // values beyond 2^55: the clamped scale would overflow fp16
    if (t > SR_FILTER_MAX_SHIFT) t = SR_FILTER_MAX_SHIFT;   // tiny values: less precise planes, honest residuals
    *scale = ldexpf(1.0f, t);
    *inv_scale = ldexpf(1.0f, -t);
    return true;
}

// one wave per row; a row of H <= 8192 floats passes through registers once
template <bool QUERY>
__global__ __launch_bounds__(256) void filter_plane_kernel(const float* __restrict__ rows, int64_t n, int H, float sd, float sigma,
                                                           unsigned short* __restrict__ plane, float* __restrict__ out, int* __restrict__ d_bad) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * 4) {
        const float* p = rows + r * H;
        float scale = sd, inv = 1.f;
        bool ok = true;
        if (QUERY) {                          // its own power-of-two scale
            unsigned int mx = 0;
            for (int i = lane * 4; i < H; i += 256) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const unsigned int b = __float_as_uint(v[e]) & 0x7fffffffu; mx = b > mx ? b : mx; }
            }
            for (int off = 32; off > 0; off >>= 1) { const unsigned int o = __shfl_xor(mx, off); mx = o > mx ? o : mx; }
            const float am = __uint_as_float(mx);
            scale = 1.f;
            if (!(am < 3.0e38f)) ok = false;
            else if (am > 0.f) {
                int e;
                (void)frexpf(am, &e);
                int t = 15 - e;
                if (t < -SR_FILTER_MAX_SHIFT) ok = false;
                t = t > SR_FILTER_MAX_SHIFT ? SR_FILTER_MAX_SHIFT : (t < -SR_FILTER_MAX_SHIFT ? -SR_FILTER_MAX_SHIFT : t);
                scale = ldexpf(1.0f, t);
                inv = ldexpf(1.0f, -t);
            }
        }
        float s_n = 0.f, s_0 = 0.f, s_r = 0.f;
        for (int i = lane * 4; i < H; i += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + i);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float vs = v[e] * scale;                 // exact (power of two), |vs| < 2^15 unless the clamp is active
                const _Float16 h = (_Float16)vs;               // round to nearest even
                const float hf = (float)h;
                const float res = vs - hf;                     // exact in fp32
                s_n += vs * vs;
                s_0 += hf * hf;
                s_r += res * res;
                o[e] = (short)__builtin_bit_cast(unsigned short, h);
            }
            *reinterpret_cast<bf16x4*>(plane + r * H + i) = o;
        }
        for (int off = 32; off > 0; off >>= 1) {
            s_n += __shfl_xor(s_n, off);
            s_0 += __shfl_xor(s_0, off);
            s_r += __shfl_xor(s_r, off);
        }
        if (lane == 0) {
            const float nn = sqrtf(s_n), n0 = sqrtf(s_0), nr = sqrtf(s_r);
            if (QUERY) {
                float A = nn * 1.001f, B = nr * 1.001f;
                if (!ok || !(A < INFINITY) || !(B < INFINITY) || !(n0 < INFINITY)) A = INFINITY;     // not filterable: exact kernel
                reinterpret_cast<f32x4*>(out)[r] = f32x4{A, B, scale, inv};
            } else {
                const float X = (nr + sigma * nn) * 1.001f, Y = n0 * 1.001f;
                out[r * 2] = X;
                out[r * 2 + 1] = Y;
                if (!(X < INFINITY) || !(Y < INFINITY)) atomicOr(d_bad, 1);
            }
        }
    }
}

int launch_filter_plane(const float* rows, int64_t n, int H, float sd, double sigma, unsigned short* plane, float* xy, int* d_bad,
                        hipStream_t s) {
    if (n == 0) return SR_OK;
    int64_t blocks = ceil_div64(n, 4);
    if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL(filter_plane_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, rows, n, H, sd, (float)sigma, plane, xy, d_bad);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

// one wave per group of 128 documents (2 per lane)
__global__ __launch_bounds__(256) void filter_group_max_kernel(const float* __restrict__ xy, int64_t n, float* __restrict__ gmax) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g * 128 >= n) return;
    float mx = 0.f, my = 0.f;                     // x, y are norms: >= 0
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int64_t r = g * 128 + t * 64 + lane;
        if (r < n) { mx = fmaxf(mx, xy[r * 2]); my = fmaxf(my, xy[r * 2 + 1]); }
    }
    for (int off = 32; off > 0; off >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, off)); my = fmaxf(my, __shfl_xor(my, off)); }
    if (lane == 0) { gmax[g * 2] = mx; gmax[g * 2 + 1] = my; }
}

int launch_filter_group_max(const float* xy, int64_t n, float* gmax, hipStream_t s) {
    if (n == 0) return SR_OK;
    hipLaunchKernelGGL(filter_group_max_kernel, dim3((unsigned)ceil_div64(ceil_div64(n, 128), 4)), dim3(256), 0, s, xy, n, gmax);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

__global__ void filter_slack_kernel(const float* __restrict__ qa, int64_t nq, FilterSegMax m, float* __restrict__ slack, float* __restrict__ tau2) {
#pragma clang fp contract(off)
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const float A = qa[q * 4], B = qa[q * 4 + 1], isq = qa[q * 4 + 3];
    float e = 0.f;
    for (int i = 0; i < m.count; ++i) {
        const float v = (A * m.x[i] + B * m.y[i]) * m.isd[i];      // isd, isq: powers of two
        e = v > e || !(v == v) ? v : e;                           // a NaN (inf * 0) sticks: the threshold is then never used
    }
    slack[q] = 2.f * (e * isq) * 1.001953125f;                    // (1 + 2^-9): over the three roundings of e and of the subtraction
    tau2[q] = -INFINITY;
}

int launch_filter_slack(const float* qa, int64_t nq, const FilterSegMax& m, float* slack, float* tau2, hipStream_t s) {
    if (nq == 0) return SR_OK;
    hipLaunchKernelGGL(filter_slack_kernel, dim3((unsigned)ceil_div64(nq, 256)), dim3(256), 0, s, qa, nq, m, slack, tau2);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

__global__ void filter_tau_kernel(const float* __restrict__ tau, const float* __restrict__ tau2, const float* __restrict__ slack,
                                  float* __restrict__ tau_eff, int64_t nq) {
#pragma clang fp contract(off)
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const float t = tau[q], u = tau2[q];
    const float t2 = (u - slack[q]) - fabsf(u) * 2.384185791015625e-07f;      // -inf while fewer than k keys were seen; NaN never wins below
    tau_eff[q] = t2 > t ? t2 : t;
}

int launch_filter_tau(const float* tau, const float* tau2, const float* slack, float* tau_eff, int64_t nq, hipStream_t s) {
    if (nq == 0) return SR_OK;
    hipLaunchKernelGGL(filter_tau_kernel, dim3((unsigned)ceil_div64(nq, 256)), dim3(256), 0, s, tau, tau2, slack, tau_eff, nq);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

int launch_filter_queries(const float* Q, int64_t nq, int H, unsigned short* plane, float* qa, hipStream_t s) {
    if (nq == 0) return SR_OK;
    hipLaunchKernelGGL(filter_plane_kernel<true>, dim3((unsigned)ceil_div64(nq, 4)), dim3(256), 0, s, Q, nq, H, 1.0f, 0.0f, plane, qa,
                       (int*)nullptr);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

__global__ void filter_certify_kernel(const float* __restrict__ u_scores, const float* __restrict__ x_scores,
                                      const float* __restrict__ qa, int64_t nq, int k, int kp, int* __restrict__ flags,
                                      const float* __restrict__ thr) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const float A = qa[q * 4], B = qa[q * 4 + 1];
    const float ukp = u_scores[q * kp + (kp - 1)];   // kp-th largest upper bound; -FLT_MAX pad when fewer documents exist
    const float xk = x_scores[q * k + (k - 1)];      // k-th best EXACT score among the candidates (pad when fewer than k)
    // fewer than kp documents: every document was re-scored.  Otherwise every outsider's exact score is <= ukp, which must
    // stay strictly below the k-th exact score.  NaN / inf anywhere -> not certified.
    const bool finite = A < INFINITY && B < INFINITY;
    const bool all_docs = ukp <= -3.0e38f;
    // doc-sharded search: an outsider is also out if it lies strictly below thr <= the GLOBAL k-th exact score
    const bool ok = finite && (all_docs || (xk > -3.0e38f && ukp < xk) || (thr && ukp < thr[q]));
    if (!ok) atomicOr(&flags[q], 1);
}

int launch_filter_certify(const float* u_scores, const float* x_scores, const float* qa, int64_t nq, int k, int kp, int* flags,
                          const float* thr, hipStream_t s) {
    hipLaunchKernelGGL(filter_certify_kernel, dim3((unsigned)ceil_div64(nq, 256)), dim3(256), 0, s, u_scores, x_scores, qa, nq, k, kp,
                       flags, thr);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

__global__ __launch_bounds__(256) void filter_gather_rows_kernel(const uint32_t* __restrict__ src, const int64_t* __restrict__ idx, int64_t width,
                                                                 uint32_t* __restrict__ dst, int scatter) {
    const int64_t i = blockIdx.x;
    const int64_t r = idx[i];
    const uint32_t* sp = src + (scatter ? i : r) * width;
    uint32_t* dp = dst + (scatter ? r : i) * width;
    for (int64_t c = threadIdx.x; c < width; c += blockDim.x) dp[c] = sp[c];
}

int launch_filter_gather_rows(const void* src, const int64_t* idx, int64_t n, int64_t width, void* dst, bool scatter, hipStream_t s) {
    if (n == 0 || width == 0) return SR_OK;
    hipLaunchKernelGGL(filter_gather_rows_kernel, dim3((unsigned)n), dim3(256), 0, s, (const uint32_t*)src, idx, width, (uint32_t*)dst,
                       scatter ? 1 : 0);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

// Doc-sharded search, first half: lower[q] = the smallest EXACT score (the rescore kernel's fmaf chain) among the j candidates with
// the largest upper bounds - at least j documents of THIS index reach lower[q] exactly, whatever the filter's error model is worth
// (round 3 published min(U - 2e), which is a bound only while the model holds: a violation on one shard was caught by that
// shard's own flag, but the other shards had already pruned with it - ADVICE r03).  With j = ceil(k / W) on each of W shards, the
// minimum of the W values is not above the global k-th exact score.  -inf when the index holds fewer than j documents, when the
// query is not filterable, or when a re-scored pair lies outside [U - 2e, U] (flag 2: the query is then re-done exactly by _finish).
__global__ void filter_lower_exact_kernel(const unsigned int* __restrict__ xmin, const int64_t* __restrict__ u_ids,
                                          const float* __restrict__ qa, const int* __restrict__ flags, int64_t nq, int kp, int j,
                                          float* __restrict__ lower) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const bool have_j = u_ids[q * kp + (j - 1)] >= 0;          // the list is sorted by U, pads (id -1) at its end
    const float x = sr_ord2f(xmin[q]);
    const bool ok = have_j && qa[q * 4] < INFINITY && flags[q] == 0 && x > -INFINITY && x < INFINITY;
    lower[q] = ok ? x : -INFINITY;
}

// One wave per (query, 64 candidates).  The candidates' rows are gathered 64 columns at a time through LDS (coalesced
// 256-byte pieces of each row; lane = candidate reads its row conflict-free from the padded tile), the query chunk is a
// broadcast read, and every lane runs the fp32 fmaf chain of ITS candidate in dense_score_pipe_kernel's k order: per
// group of 8 columns, k = 8s + j then 8s + 4 + j for j = 0..3.
#define RS_KC 64
__global__ __launch_bounds__(64) void filter_rescore_kernel(FilterSegs segs, const float* __restrict__ Q, const float* __restrict__ u_scores,
                                                            const int64_t* __restrict__ u_ids, const float* __restrict__ qa,
                                                            int k, int kp, int H, uint64_t* __restrict__ cand_keys,
                                                            int* __restrict__ cand_count, int64_t cand_cap, int* __restrict__ flags,
                                                            int j_begin, unsigned int* __restrict__ xmin, const float* __restrict__ thr,
                                                            int lb_j) {
#pragma clang fp contract(off)
    __shared__ float tile[64][RS_KC + 1];
    __shared__ float qs[RS_KC];
    __shared__ const float* rowp[64];
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    const int j0 = j_begin + blockIdx.y * 64;
    const int j = j0 + lane;
    // stage 1: the k largest upper bounds; stage 2: the rest; lb_j > 0 (sr_dense_search_begin): only the first lb_j, nothing is
    // appended - the smallest exact score among them goes to xmin
    const int j_end = lb_j > 0 ? lb_j : (j_begin == 0 ? (k < kp ? k : kp) : kp);
    if (j0 >= j_end) return;
    // stage 2: the smallest EXACT score among the k candidates of stage 1 is a lower bound of the exact k-th score; a
    // candidate whose upper bound lies strictly below it cannot reach the top-k, nor tie with its last member.  The list is
    // sorted by U, so what has to be re-scored is a prefix of it.
    // thr (doc-sharded search): a value that is provably not above the GLOBAL k-th exact score; a candidate whose upper bound lies
    // strictly below it cannot be in the global top-k.  With fewer than k candidates at or above thr the stage-1 minimum is no
    // bound of anything - but then every later candidate is below thr anyway.
    float need = thr ? thr[q] : -INFINITY;
    if (j_begin > 0) need = fmaxf(need, sr_ord2f(xmin[q]));
    if (u_scores[q * kp + j0] < need) return;           // the whole wave
    int64_t gid = j < j_end ? u_ids[q * kp + j] : -1;
    if (gid >= 0 && u_scores[q * kp + j] < need) gid = -1;
    const float* row = nullptr;
    double e2 = 0.0;                                     // 2 e(q, j) in the true domain
    if (gid >= 0) {
        for (int sgi = 0; sgi < segs.count; ++sgi) {
            const int64_t off = gid - (int64_t)segs.id_base[sgi];
            if (off >= 0 && off % segs.id_stride[sgi] == 0 && off / segs.id_stride[sgi] < segs.n[sgi]) {
                const int64_t r = off / segs.id_stride[sgi];
                row = segs.rows[sgi] + r * (int64_t)H;
                e2 = 2.0 * ((double)qa[q * 4] * (double)segs.xy[sgi][r * 2] + (double)qa[q * 4 + 1] * (double)segs.xy[sgi][r * 2 + 1]) *
                     (double)qa[q * 4 + 3] * (double)segs.isd[sgi];
                break;
            }
        }
    }
    rowp[lane] = row;
    __syncthreads();
    float acc = 0.f;
    // 64 rows x 256 B per chunk: 16 lanes per row, 4 rows per instruction.  All 16 loads of a chunk are issued together
    // (interleaved with the LDS stores each one was waited for before the next went out: one memory latency per 4 rows),
    // and the loads of chunk k + 1 fly while chunk k goes through LDS and the fmaf chain.
    const float* myp[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) myp[i] = rowp[i * 4 + (lane >> 4)];
    auto fetch = [&](int k0, f32x4 (&v)[16], float& qv) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (myp[i]) v[i] = *reinterpret_cast<const f32x4*>(myp[i] + k0 + (lane & 15) * 4);
        }
        qv = Q[q * H + k0 + lane];
    };
    auto chunk = [&](const f32x4 (&v)[16], float qv) {
        __syncthreads();                              // the previous chunk is consumed
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float* t = &tile[i * 4 + (lane >> 4)][(lane & 15) * 4];
            t[0] = v[i][0]; t[1] = v[i][1]; t[2] = v[i][2]; t[3] = v[i][3];
        }
        qs[lane] = qv;
        __syncthreads();
#pragma unroll
        for (int s8 = 0; s8 < RS_KC; s8 += 8)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                acc = __builtin_fmaf(qs[s8 + jj], tile[lane][s8 + jj], acc);
                acc = __builtin_fmaf(qs[s8 + 4 + jj], tile[lane][s8 + 4 + jj], acc);
            }
    };
    f32x4 va[16], vb[16];
    float qva, qvb;
    fetch(0, va, qva);
    for (int k0 = 0; k0 < H; k0 += 2 * RS_KC) {       // H is a multiple of RS_KC; an odd chunk count ends in the first half
        const bool has_b = k0 + RS_KC < H;
        if (has_b) fetch(k0 + RS_KC, vb, qvb);
        chunk(va, qva);
        if (!has_b) break;
        if (k0 + 2 * RS_KC < H) fetch(k0 + 2 * RS_KC, va, qva);
        chunk(vb, qvb);
    }
    if (row) {
        // the bound, checked on every pair that is re-scored: S_x in [U - 2e, U]
        const double U = (double)u_scores[q * kp + j];
        if (!((double)acc <= U && (double)acc >= U - e2 * 1.001)) atomicOr(&flags[q], 2);
        if (lb_j == 0) {
            const int pos = atomicAdd(&cand_count[q], 1);
            if (pos < cand_cap) cand_keys[q * cand_cap + pos] = sr_make_key(acc, (uint32_t)gid);
        }
        if (j_begin == 0) atomicMin(&xmin[q], sr_f2ord(acc));
    }
}

int launch_filter_rescore(const FilterSegs& segs, const float* Q, const float* u_scores, const int64_t* u_ids, const float* qa,
                          int64_t nq, int k, int kp, int H, uint64_t* cand_keys, int* cand_count, int64_t cand_cap, int* flags,
                          unsigned int* xmin, const float* thr, hipStream_t s) {
    SR_REQUIRE(H % RS_KC == 0, "filter(rescore): dim %d must be a multiple of %d", H, RS_KC);
    SR_REQUIRE(nq <= 0x7fffffff && ceil_div64(kp, 64) <= 65535, "filter(rescore): grid too large");
    // stage 1: the k candidates with the largest upper bounds (and the smallest exact score among them, xmin); stage 2: the
    // rest, pruned against xmin.  When stage 1 saw fewer than k documents there is nothing left for stage 2.
    SR_CHECK_HIP(hipMemsetAsync(xmin, 0xff, (size_t)nq * 4, s));
    const int k1 = k < kp ? k : kp;
    hipLaunchKernelGGL(filter_rescore_kernel, dim3((unsigned)nq, (unsigned)ceil_div64(k1, 64)), dim3(64), 0, s, segs, Q, u_scores, u_ids,
                       qa, k, kp, H, cand_keys, cand_count, cand_cap, flags, 0, xmin, thr, 0);
    SR_CHECK_LAUNCH();
    if (kp > k1) {
        hipLaunchKernelGGL(filter_rescore_kernel, dim3((unsigned)nq, (unsigned)ceil_div64(kp - k1, 64)), dim3(64), 0, s, segs, Q, u_scores,
                           u_ids, qa, k, kp, H, cand_keys, cand_count, cand_cap, flags, k1, xmin, thr, 0);
        SR_CHECK_LAUNCH();
    }
    return SR_OK;
}

int launch_filter_lower_bound(const FilterSegs& segs, const float* Q, const float* u_scores, const int64_t* u_ids, const float* qa,
                              int64_t nq, int kp, int j, int H, int* flags, unsigned int* xmin, float* lower, hipStream_t s) {
    SR_REQUIRE(j >= 1 && j <= kp, "filter(lower bound): j = %d outside [1, %d]", j, kp);
    SR_REQUIRE(H % RS_KC == 0, "filter(lower bound): dim %d must be a multiple of %d", H, RS_KC);
    SR_CHECK_HIP(hipMemsetAsync(xmin, 0xff, (size_t)nq * 4, s));
    hipLaunchKernelGGL(filter_rescore_kernel, dim3((unsigned)nq, (unsigned)ceil_div64(j, 64)), dim3(64), 0, s, segs, Q, u_scores, u_ids,
                       qa, j, kp, H, (uint64_t*)nullptr, (int*)nullptr, (int64_t)0, flags, 0, xmin, (const float*)nullptr, j);
    hipLaunchKernelGGL(filter_lower_exact_kernel, dim3((unsigned)ceil_div64(nq, 256)), dim3(256), 0, s, xmin, u_ids, qa, flags, nq, kp, j,
                       lower);
    SR_CHECK_LAUNCH();
    return SR_OK;
}
