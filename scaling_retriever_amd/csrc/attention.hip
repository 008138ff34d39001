// Bidirectional GQA attention over packed var-len sequences, RoPE fused (gfx950).
//
// Replaces SDPA inside HF's LlamaAttention as driven by LlamaBiModel: full (NON-causal)
// attention with only padded KEYS masked
// (scaling_retriever/modeling/bidirectional_llama.py:138-161, is_causal=False :26-41).
// Sequences are packed (no pad rows are computed unless the caller needs them), so the
// key-padding mask is the per-token key_valid flag.
//
// One workgroup = (sequence, kv head); its 4 waves walk the (q head of the group, 32-row
// q tile) items.  Keys/values of the sequence are staged through LDS in chunks of 256
// keys: K row-major (16-B chunks XOR-swizzled) with RoPE applied while staging, V
// TRANSPOSED ([d][key]) so that the P.V MFMA's B operand is two 8-byte reads.
// Scores use the swapped product S^T = K.Q^T (v_mfma_f32_32x32x16_bf16): a lane owns
// one q row, so the softmax statistics are lane-local (+ one cross-half shuffle), and the
// S^T accumulator is fed straight back as the A operand of O += P^T-as-A . V (no LDS
// round trip for P).  Two passes over the keys: pass 1 = row max and normaliser (online),
// pass 2 = recompute scores, p = exp(s - m) / l, accumulate O.  Attention is ~1 % of the
// encoder FLOPs at S <= 192, so the second QK^T is cheaper than rescaling O.
#include "kernels.h"
#include <stdlib.h>

typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));

#define AT_KC 256          // keys per LDS chunk
#define AT_VT_LD (AT_KC + 4)  // padded row of the transposed V image (bf16 elements)
#ifndef AT_IPW
#define AT_IPW 2           // work items per wave of the fast path (8 per workgroup: 52 us per layer at 9.6 k tokens, 54 with 4, 67 with all)
#endif

union Frag8 {
    mfma_bf16x8 v;
    bf16_t u[8];
    uint32_t w[4];
    uint2 d2[2];
    uint4 q;
};

__device__ inline uint32_t pack_bf16x2(float a, float b) {
    return (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
}

template <int HD, bool ROPE>
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
    constexpr int NCH = HD / 8;        // 16-B chunks per head row
    constexpr int NKK = HD / 16;       // MFMA k-steps over the head dim
    constexpr int NDB = HD / 32;       // 32-wide output blocks over the head dim
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);                       // [AT_KC][HD], chunk-swizzled
    bf16_t* Vt = Ks + AT_KC * HD;                                        // [HD][AT_VT_LD]
    unsigned char* kval = reinterpret_cast<unsigned char*>(Vt + HD * AT_VT_LD);  // [AT_KC]

    const int b = blockIdx.x, kvh = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = a.cu_seqlens[b];
    const int S = a.cu_seqlens[b + 1] - t0;
    if (S <= 0) return;
    const int G = a.nh / a.nkv;
    const int ldq = (a.nh + 2 * a.nkv) * HD;
    const int koff = a.nh * HD + kvh * HD;
    const int voff = (a.nh + a.nkv) * HD + kvh * HD;
    const int n_qt = (S + 31) / 32;
    const int n_items = G * n_qt;
    const int n_chunks = (S + AT_KC - 1) / AT_KC;
    const float sc_log2 = a.scale * 1.4426950408889634f;
    const int r = lane & 31, h = lane >> 5;

    // per-item state carried across key chunks: items are wave-strided, MAXI per wave per round
    // (a round re-stages K/V; MS MARCO passages (S ~ 75, G = 4) need one round)
    constexpr int MAXI = HD <= 64 ? 4 : 2;
    for (int item_base = 0; item_base < n_items; item_base += 4 * MAXI) {
        Frag8 qf[MAXI][NKK];
        float m_run[MAXI], l_run[MAXI];
        f32x16 o[MAXI][NDB];
        // ---- load + rotate Q fragments (B operand: lane = q row r, elements d = 16kk + 8h + j)
#pragma unroll
        for (int it = 0; it < MAXI; ++it) {
            const int item = item_base + it * 4 + wave;
            m_run[it] = -INFINITY;
            l_run[it] = 0.f;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int x = 0; x < 16; ++x) o[it][db][x] = 0.f;
            if (item < n_items) {
                const int qh = kvh * G + item / n_qt;
                int qrow = (item % n_qt) * 32 + r;
                qrow = qrow < S ? qrow : S - 1;
                const bf16_t* qp = a.qkv + (int64_t)(t0 + qrow) * ldq + qh * HD;
                Frag8 raw[NKK];
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) raw[kk].q = *reinterpret_cast<const uint4*>(qp + 16 * kk + 8 * h);
                if constexpr (!ROPE) {
#pragma unroll
                    for (int kk = 0; kk < NKK; ++kk) qf[it][kk].q = raw[kk].q;
                } else {
                const int p = a.pos[t0 + qrow];
                const float* cs = a.rope_cos + (int64_t)p * (HD / 2);
                const float* sn = a.rope_sin + (int64_t)p * (HD / 2);
#pragma unroll
                for (int kk = 0; kk < NKK / 2; ++kk) {
                    // d = 16kk + 8h + j (first half of the head), partner d + HD/2 lives in raw[kk + NKK/2]
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int d = 16 * kk + 8 * h + j;
                        const float c = cs[d], s_ = sn[d];
                        const float x1 = bf16_to_f32(raw[kk].u[j]), x2 = bf16_to_f32(raw[kk + NKK / 2].u[j]);
                        qf[it][kk].u[j] = f32_to_bf16(x1 * c - x2 * s_);
                        qf[it][kk + NKK / 2].u[j] = f32_to_bf16(x2 * c + x1 * s_);
                    }
                }
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) qf[it][kk].q = uint4{0, 0, 0, 0};
            }
        }

        for (int pass = 0; pass < 2; ++pass) {
            for (int ch = 0; ch < n_chunks; ++ch) {
                const int key0 = ch * AT_KC;
                const int nkeys = (S - key0) < AT_KC ? (S - key0) : AT_KC;
                const int nkb = (nkeys + 31) / 32;
                __syncthreads();  // previous chunk fully consumed
                // ---- stage K (rope applied) : thread handles a chunk pair (c, c + NCH/2) of one key
                for (int idx = tid; idx < nkb * 32 * (NCH / 2); idx += 256) {
                    const int key = idx / (NCH / 2), c = idx % (NCH / 2);
                    Frag8 lo, hi;
                    if (key < nkeys) {
                        const int tok = t0 + key0 + key;
                        const bf16_t* kp = a.qkv + (int64_t)tok * ldq + koff;
                        Frag8 x1, x2;
                        x1.q = *reinterpret_cast<const uint4*>(kp + c * 8);
                        x2.q = *reinterpret_cast<const uint4*>(kp + c * 8 + HD / 2);
                        if constexpr (!ROPE) {
                            lo.q = x1.q;
                            hi.q = x2.q;
                        } else {
                        const int p = a.pos[tok];
                        const float* cs = a.rope_cos + (int64_t)p * (HD / 2) + c * 8;
                        const float* sn = a.rope_sin + (int64_t)p * (HD / 2) + c * 8;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float v1 = bf16_to_f32(x1.u[j]), v2 = bf16_to_f32(x2.u[j]);
                            lo.u[j] = f32_to_bf16(v1 * cs[j] - v2 * sn[j]);
                            hi.u[j] = f32_to_bf16(v2 * cs[j] + v1 * sn[j]);
                        }
                        }
                    } else {
                        lo.q = uint4{0, 0, 0, 0};
                        hi.q = uint4{0, 0, 0, 0};
                    }
                    const int sw = key & 7;
                    *reinterpret_cast<uint4*>(Ks + key * HD + ((c ^ sw) * 8)) = lo.q;
                    *reinterpret_cast<uint4*>(Ks + key * HD + (((c + NCH / 2) ^ sw) * 8)) = hi.q;
                }
                for (int key = tid; key < nkb * 32; key += 256)
                    kval[key] = (key < nkeys) ? a.key_valid[t0 + key0 + key] : 0;
                // ---- stage V transposed (pass 2 only)
                if (pass == 1) {
                    for (int idx = tid; idx < nkb * 32 * NCH; idx += 256) {
                        const int key = idx / NCH, c = idx % NCH;
                        Frag8 x;
                        if (key < nkeys)
                            x.q = *reinterpret_cast<const uint4*>(a.qkv + (int64_t)(t0 + key0 + key) * ldq + voff + c * 8);
                        else
                            x.q = uint4{0, 0, 0, 0};
#pragma unroll
                        for (int j = 0; j < 8; ++j) Vt[(c * 8 + j) * AT_VT_LD + key] = x.u[j];
                    }
                }
                __syncthreads();

#pragma unroll
                for (int it = 0; it < MAXI; ++it) {
                    const int item = item_base + it * 4 + wave;
                    if (item >= n_items) continue;  // wave-uniform
                    for (int kb = 0; kb < nkb; ++kb) {
                        // S^T block = K[32 keys] . Q^T : A = K rows (lane r = key), B = Q (lane r = q row)
                        f32x16 st;
#pragma unroll
                        for (int x = 0; x < 16; ++x) st[x] = 0.f;
                        const int key = kb * 32 + r;
#pragma unroll
                        for (int kk = 0; kk < NKK; ++kk) {
                            Frag8 kf;
                            kf.q = *reinterpret_cast<const uint4*>(Ks + key * HD + (((2 * kk + h) ^ (key & 7)) * 8));
                            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf.v, qf[it][kk].v, st, 0, 0, 0);
                        }
                        // lane holds q row r; register x = key kb*32 + (x&3) + 8*(x>>2) + 4h
                        float sv[16];
#pragma unroll
                        for (int x = 0; x < 16; ++x) {
                            const int kl = kb * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;
                            sv[x] = kval[kl] ? st[x] * sc_log2 : -INFINITY;
                        }
                        if (pass == 0) {
                            float bm = sv[0];
#pragma unroll
                            for (int x = 1; x < 16; ++x) bm = fmaxf(bm, sv[x]);
                            bm = fmaxf(bm, __shfl_xor(bm, 32));
                            const float mn = fmaxf(m_run[it], bm);
                            if (mn > -INFINITY) {
                                float ps = 0.f;
#pragma unroll
                                for (int x = 0; x < 16; ++x) ps += exp2f(sv[x] - mn);
                                ps += __shfl_xor(ps, 32);
                                l_run[it] = l_run[it] * exp2f(m_run[it] - mn) + ps;
                                m_run[it] = mn;
                            }
                        } else {
                            const float mf = m_run[it];
                            const float inv_l = l_run[it] > 0.f ? 1.f / l_run[it] : 0.f;
                            float pv[16];
#pragma unroll
                            for (int x = 0; x < 16; ++x) pv[x] = (mf > -INFINITY) ? exp2f(sv[x] - mf) * inv_l : 0.f;
                            // P^T block as the A operand of O[q][d] += sum_key P[key][q] V[key][d]:
                            // k-step s uses registers 8s..8s+7; element j of lane half h is key 16s + 8(j>>2) + 4h + (j&3)
#pragma unroll
                            for (int s2 = 0; s2 < 2; ++s2) {
                                Frag8 pf;
#pragma unroll
                                for (int w = 0; w < 4; ++w) pf.w[w] = pack_bf16x2(pv[8 * s2 + 2 * w], pv[8 * s2 + 2 * w + 1]);
#pragma unroll
                                for (int db = 0; db < NDB; ++db) {
                                    Frag8 vf;  // B operand: lane r = column d, elements = the same key order
                                    const bf16_t* vp = Vt + (db * 32 + r) * AT_VT_LD + kb * 32 + 16 * s2 + 4 * h;
                                    vf.d2[0] = *reinterpret_cast<const uint2*>(vp);
                                    vf.d2[1] = *reinterpret_cast<const uint2*>(vp + 8);
                                    o[it][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf.v, vf.v, o[it][db], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
            }
        }

        // ---- write O: lane = column d (r), registers = q rows
#pragma unroll
        for (int it = 0; it < MAXI; ++it) {
            const int item = item_base + it * 4 + wave;
            if (item >= n_items) continue;
            const int qh = kvh * G + item / n_qt;
            const int q0 = (item % n_qt) * 32;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int x = 0; x < 16; ++x) {
                    const int qrow = q0 + (x & 3) + 8 * (x >> 2) + 4 * h;
                    if (qrow < S)
                        a.out[(int64_t)(t0 + qrow) * (a.nh * HD) + qh * HD + db * 32 + r] = f32_to_bf16(o[it][db][x]);
                }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fast path for sequences of at most 256 tokens (every MS MARCO batch: doc_max_length 192,
// query_max_length 64) with q/k already rotated by the QKV GEMM epilogue.  A workgroup = (sequence, kv head, group of
// 4 * AT_IPW work items); a work item = (q head of the group, 32-row q tile), AT_IPW per wave, with ALL score blocks of the row held
// in registers - one QK^T pass, exact softmax, no rescaling.  Work per sequence grows with S^2, so the items of a long
// sequence are spread over several workgroups (each stages K and V^T of the sequence again - a few KB from L2) instead
// of one workgroup walking them while the rest of the chip has finished.  MAXKB = key blocks (of 32) the batch needs:
// it sizes the LDS image and the score registers, so short batches run at higher occupancy.
// Workgroups per CU the register allocation is sized for.  Heads of 128 (Lion-DS-8B) hold twice the output accumulators and twice the K
// fragments of heads of 64: at the occupancy of the 64-wide kernels (4 or 3 workgroups: 128 / 168 VGPRs) the 4- and 6-key-block
// instantiations spilled 1 300 registers per lane to scratch (kernel metadata, round 6); two workgroups (256 VGPRs) hold them.
constexpr int at_small_blocks(int hd, int maxkb) { return hd > 64 && maxkb > 2 ? 2 : (maxkb <= 4 ? 4 : (maxkb <= 6 ? 3 : 2)); }
template <int HD, int MAXKB>
__global__ __launch_bounds__(256, at_small_blocks(HD, MAXKB)) void attention_small_kernel(AttnArgs a) {
    constexpr int NCH = HD / 8, NKK = HD / 16, NDB = HD / 32, KC = MAXKB * 32, VT_LD = KC + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);       // [KC][HD], chunk-swizzled
    bf16_t* Vt = Ks + KC * HD;                          // [HD][VT_LD]
    unsigned char* kval = reinterpret_cast<unsigned char*>(Vt + HD * VT_LD);

    const int b = blockIdx.x, kvh = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = a.cu_seqlens[b];
    const int S = a.cu_seqlens[b + 1] - t0;
    if (S <= 0) return;
    const int G = a.nh / a.nkv;
    const int n_qt = (S + 31) / 32, n_items = G * n_qt, nkb = n_qt;
    if ((int)blockIdx.z * 4 * AT_IPW >= n_items) return;          // this sequence has fewer item groups than the longest one
    const int ldq = (a.nh + 2 * a.nkv) * HD;
    const int koff = a.nh * HD + kvh * HD;
    const int voff = (a.nh + a.nkv) * HD + kvh * HD;
    const float sc_log2 = a.scale * 1.4426950408889634f;
    const int r = lane & 31, h = lane >> 5;

    for (int idx = tid; idx < nkb * 32 * NCH; idx += 256) {
        const int key = idx / NCH, c = idx % NCH;
        uint4 kx = uint4{0, 0, 0, 0};
        Frag8 vx;
        vx.q = uint4{0, 0, 0, 0};
        if (key < S) {
            const bf16_t* base = a.qkv + (int64_t)(t0 + key) * ldq;
            kx = *reinterpret_cast<const uint4*>(base + koff + c * 8);
            vx.q = *reinterpret_cast<const uint4*>(base + voff + c * 8);
        }
        *reinterpret_cast<uint4*>(Ks + key * HD + ((c ^ (key & 7)) * 8)) = kx;
#pragma unroll
        for (int j = 0; j < 8; ++j) Vt[(c * 8 + j) * VT_LD + key] = vx.u[j];
    }
    for (int key = tid; key < nkb * 32; key += 256) kval[key] = (key < S) ? a.key_valid[t0 + key] : 0;
    __syncthreads();

    for (int it = 0; it < AT_IPW; ++it) {
        const int item = ((int)blockIdx.z * AT_IPW + it) * 4 + wave;
        if (item >= n_items) break;
        const int qh = kvh * G + item / n_qt;
        const int q0 = (item % n_qt) * 32;
        int qrow = q0 + r;
        qrow = qrow < S ? qrow : S - 1;
        const bf16_t* qp = a.qkv + (int64_t)(t0 + qrow) * ldq + qh * HD;
        Frag8 qf[NKK];
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) qf[kk].q = *reinterpret_cast<const uint4*>(qp + 16 * kk + 8 * h);

        f32x16 st[MAXKB];
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < MAXKB; ++kb) {
            if (kb < nkb) {
                f32x16 acc;
#pragma unroll
                for (int x = 0; x < 16; ++x) acc[x] = 0.f;
                const int key = kb * 32 + r;
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    Frag8 kf;
                    kf.q = *reinterpret_cast<const uint4*>(Ks + key * HD + (((2 * kk + h) ^ (key & 7)) * 8));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf.v, qf[kk].v, acc, 0, 0, 0);
                }
#pragma unroll
                for (int x = 0; x < 16; ++x) {
                    const int kl = kb * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;
                    const float v = kval[kl] ? acc[x] * sc_log2 : -INFINITY;
                    acc[x] = v;
                    mx = fmaxf(mx, v);
                }
                st[kb] = acc;
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < MAXKB; ++kb) {
            if (kb < nkb) {
#pragma unroll
                for (int x = 0; x < 16; ++x) {
                    const float p = (mx > -INFINITY) ? __builtin_amdgcn_exp2f(st[kb][x] - mx) : 0.f;
                    st[kb][x] = p;
                    sum += p;
                }
            }
        }
        sum += __shfl_xor(sum, 32);
        const float inv_l = sum > 0.f ? 1.f / sum : 0.f;

        f32x16 o[NDB];
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int x = 0; x < 16; ++x) o[db][x] = 0.f;
#pragma unroll
        for (int kb = 0; kb < MAXKB; ++kb) {
            if (kb < nkb) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    Frag8 pf;
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        pf.w[w] = pack_bf16x2(st[kb][8 * s2 + 2 * w] * inv_l, st[kb][8 * s2 + 2 * w + 1] * inv_l);
#pragma unroll
                    for (int db = 0; db < NDB; ++db) {
                        Frag8 vf;
                        const bf16_t* vp = Vt + (db * 32 + r) * VT_LD + kb * 32 + 16 * s2 + 4 * h;
                        vf.d2[0] = *reinterpret_cast<const uint2*>(vp);
                        vf.d2[1] = *reinterpret_cast<const uint2*>(vp + 8);
                        // V^T as the A operand: the result is O^T (rows = head dims, column = this lane's q row), so a lane
                        // ends up with 4 consecutive head dims per register group and stores 8 B at a time
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, o[db], 0, 0, 0);
                    }
                }
            }
        }
        if (q0 + r < S) {
            bf16_t* orow = a.out + (int64_t)(t0 + q0 + r) * (a.nh * HD) + qh * HD + 4 * h;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    bf16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (short)f32_to_bf16(o[db][4 * gq + e]);
                    *reinterpret_cast<bf16x4*>(orow + db * 32 + 8 * gq) = v;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Sequences longer than 256 tokens (BEIR passages up to 512, anything up to the position table) with pre-rotated q/k:
// the same work-item grid as the fast path, with the keys walked in chunks of 256 and the softmax kept online
// (running row maximum m and sum l per q row; the O^T accumulators of a lane all belong to its q row, so a rescale is
// one multiply per register).  P stays unnormalised (<= 1) when it is rounded to bf16 for the PV MFMA; the division by l
// happens once at the end.  One work item per wave: the chunk loop is outermost because the 4 waves share the staged K/V^T.
// CKB = key blocks (of 32) per chunk: it sizes the LDS image (4 * HD * 32 * CKB bytes) and the score registers, i.e. how
// many workgroups share a CU: 8 for head dim 64; 2 for head dim 128 (33 KB), where a 256-key image (131 KB) would leave one
// workgroup per CU - there this kernel also serves the short sequences (128 us per layer at 8B dims and 9.6 k tokens against
// 745 us for the all-keys-in-registers kernel at one workgroup per CU).
template <int HD, int CKB>
// (second launch bound = waves per SIMD; head dim 128 needs the 256-register budget of two: at three the 2- and 3-block plans spilled 9-25
// registers - same rate either way at 8B dims, tools/quick_encode_8b.py in a same-box A/B, but no scratch)
__global__ __launch_bounds__(256, (CKB <= 3 && HD < 128) ? 3 : 2) void attention_long_kernel(AttnArgs a) {
    constexpr int NCH = HD / 8, NKK = HD / 16, NDB = HD / 32, MAXKB = CKB, KC = CKB * 32, VT_LD = KC + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);       // [KC][HD], chunk-swizzled
    bf16_t* Vt = Ks + KC * HD;                          // [HD][VT_LD]
    unsigned char* kval = reinterpret_cast<unsigned char*>(Vt + HD * VT_LD);

    const int b = blockIdx.x, kvh = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = a.cu_seqlens[b];
    const int S = a.cu_seqlens[b + 1] - t0;
    if (S <= 0) return;
    const int G = a.nh / a.nkv;
    const int n_qt = (S + 31) / 32, n_items = G * n_qt;
    if ((int)blockIdx.z * 4 >= n_items) return;          // uniform for the workgroup
    const int ldq = (a.nh + 2 * a.nkv) * HD;
    const int koff = a.nh * HD + kvh * HD;
    const int voff = (a.nh + a.nkv) * HD + kvh * HD;
    const float sc_log2 = a.scale * 1.4426950408889634f;
    const int r = lane & 31, h = lane >> 5;

    const int item = (int)blockIdx.z * 4 + wave;
    const bool live = item < n_items;                    // a wave without an item still takes part in staging and barriers
    const int qh = kvh * G + (live ? item / n_qt : 0);
    const int q0 = (live ? item % n_qt : 0) * 32;
    int qrow = q0 + r;
    qrow = qrow < S ? qrow : S - 1;
    const bf16_t* qp = a.qkv + (int64_t)(t0 + qrow) * ldq + qh * HD;
    Frag8 qf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) qf[kk].q = *reinterpret_cast<const uint4*>(qp + 16 * kk + 8 * h);
    f32x16 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int x = 0; x < 16; ++x) o[db][x] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int n_chunks = (S + KC - 1) / KC;
    for (int ch = 0; ch < n_chunks; ++ch) {
        const int key0 = ch * KC;
        const int nkeys = (S - key0) < KC ? (S - key0) : KC;
        const int nkb = (nkeys + 31) / 32;
        __syncthreads();                                 // previous chunk fully consumed
        for (int idx = tid; idx < nkb * 32 * NCH; idx += 256) {
            const int key = idx / NCH, c = idx % NCH;
            uint4 kx = uint4{0, 0, 0, 0};
            Frag8 vx;
            vx.q = uint4{0, 0, 0, 0};
            if (key < nkeys) {
                const bf16_t* base = a.qkv + (int64_t)(t0 + key0 + key) * ldq;
                kx = *reinterpret_cast<const uint4*>(base + koff + c * 8);
                vx.q = *reinterpret_cast<const uint4*>(base + voff + c * 8);
            }
            *reinterpret_cast<uint4*>(Ks + key * HD + ((c ^ (key & 7)) * 8)) = kx;
#pragma unroll
            for (int j = 0; j < 8; ++j) Vt[(c * 8 + j) * VT_LD + key] = vx.u[j];
        }
        for (int key = tid; key < nkb * 32; key += 256) kval[key] = (key < nkeys) ? a.key_valid[t0 + key0 + key] : 0;
        __syncthreads();
        if (!live) continue;

        f32x16 st[MAXKB];
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < MAXKB; ++kb) {
            if (kb < nkb) {
                f32x16 acc;
#pragma unroll
                for (int x = 0; x < 16; ++x) acc[x] = 0.f;
                const int key = kb * 32 + r;
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    Frag8 kf;
                    kf.q = *reinterpret_cast<const uint4*>(Ks + key * HD + (((2 * kk + h) ^ (key & 7)) * 8));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf.v, qf[kk].v, acc, 0, 0, 0);
                }
#pragma unroll
                for (int x = 0; x < 16; ++x) {
                    const int kl = kb * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;
                    const float v = kval[kl] ? acc[x] * sc_log2 : -INFINITY;
                    acc[x] = v;
                    mx = fmaxf(mx, v);
                }
                st[kb] = acc;
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = (m_run > -INFINITY) ? __builtin_amdgcn_exp2f(m_run - m_new) : 0.f;   // o and l are 0 while m_run = -inf
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < MAXKB; ++kb) {
            if (kb < nkb) {
#pragma unroll
                for (int x = 0; x < 16; ++x) {
                    const float p = (m_new > -INFINITY) ? __builtin_amdgcn_exp2f(st[kb][x] - m_new) : 0.f;
                    st[kb][x] = p;
                    sum += p;
                }
            }
        }
        sum += __shfl_xor(sum, 32);
        l_run = l_run * alpha + sum;
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int x = 0; x < 16; ++x) o[db][x] *= alpha;
#pragma unroll
        for (int kb = 0; kb < MAXKB; ++kb) {
            if (kb < nkb) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    Frag8 pf;
#pragma unroll
                    for (int w = 0; w < 4; ++w) pf.w[w] = pack_bf16x2(st[kb][8 * s2 + 2 * w], st[kb][8 * s2 + 2 * w + 1]);
#pragma unroll
                    for (int db = 0; db < NDB; ++db) {
                        Frag8 vf;
                        const bf16_t* vp = Vt + (db * 32 + r) * VT_LD + kb * 32 + 16 * s2 + 4 * h;
                        vf.d2[0] = *reinterpret_cast<const uint2*>(vp);
                        vf.d2[1] = *reinterpret_cast<const uint2*>(vp + 8);
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, o[db], 0, 0, 0);   // O^T: see the fast path
                    }
                }
            }
        }
    }
    if (live && q0 + r < S) {
        const float inv_l = l_run > 0.f ? 1.f / l_run : 0.f;
        bf16_t* orow = a.out + (int64_t)(t0 + q0 + r) * (a.nh * HD) + qh * HD + 4 * h;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (short)f32_to_bf16(o[db][4 * gq + e] * inv_l);
                *reinterpret_cast<bf16x4*>(orow + db * 32 + 8 * gq) = v;
            }
    }
}

template <int HD, int CKB>
static int launch_long(const AttnArgs& a, hipStream_t s) {
    constexpr int KC = CKB * 32;
    constexpr size_t lds = (size_t)KC * HD * 2 + (size_t)HD * (KC + 4) * 2 + KC;
    static DeviceOnce attr_once;
    bool* attr_slot = attr_once.pending();
    if (attr_slot) {
        SR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_long_kernel<HD, CKB>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        *attr_slot = true;
    }
    const int max_items = (a.nh / a.nkv) * ((a.max_seqlen + 31) / 32);
    const dim3 grid((unsigned)a.B, (unsigned)a.nkv, (unsigned)((max_items + 3) / 4));
    hipLaunchKernelGGL((attention_long_kernel<HD, CKB>), grid, dim3(256), lds, s, a);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

template <int HD, int MAXKB>
static int launch_small(const AttnArgs& a, hipStream_t s) {
    constexpr int KC = MAXKB * 32;
    constexpr size_t lds = (size_t)KC * HD * 2 + (size_t)HD * (KC + 4) * 2 + KC;
    static DeviceOnce attr_once;
    bool* attr_slot = attr_once.pending();
    if (attr_slot) {
        SR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_small_kernel<HD, MAXKB>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        *attr_slot = true;
    }
    const int max_items = (a.nh / a.nkv) * ((a.max_seqlen + 31) / 32);
    const dim3 grid((unsigned)a.B, (unsigned)a.nkv, (unsigned)((max_items + 4 * AT_IPW - 1) / (4 * AT_IPW)));
    hipLaunchKernelGGL((attention_small_kernel<HD, MAXKB>), grid, dim3(256), lds, s, a);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

template <int HD>
static int launch_hd(const AttnArgs& a, hipStream_t s) {
    constexpr size_t lds = (size_t)AT_KC * HD * 2 + (size_t)HD * AT_VT_LD * 2 + AT_KC;
    static DeviceOnce attr_once;
    bool* attr_slot = attr_once.pending();
    if (attr_slot) {
        SR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<HD, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<HD, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        *attr_slot = true;
    }
    if constexpr (HD == 128) {
        if (!a.apply_rope && a.max_seqlen > 64) {
            const char* e = sr_dev_getenv("SR_ATTN_CKB");      // A/B switch: key blocks per chunk for head dim 128 (2 | 3 | 8)
            const int ckb = e ? atoi(e) : 2;
            if (ckb == 2) return launch_long<HD, 2>(a, s);
            if (ckb == 3) return launch_long<HD, 3>(a, s);
            if (a.max_seqlen > AT_KC) return launch_long<HD, 8>(a, s);
        }
    }
    if (!a.apply_rope && a.max_seqlen > 0 && a.max_seqlen <= AT_KC) {
        if (a.max_seqlen <= 64) return launch_small<HD, 2>(a, s);
        if (a.max_seqlen <= 128) return launch_small<HD, 4>(a, s);
        if (a.max_seqlen <= 192) return launch_small<HD, 6>(a, s);
        return launch_small<HD, 8>(a, s);
    }
    if (!a.apply_rope && a.max_seqlen > AT_KC) return launch_long<HD, 8>(a, s);
    const dim3 grid((unsigned)a.B, (unsigned)a.nkv);
    if (a.apply_rope) hipLaunchKernelGGL((attention_kernel<HD, true>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((attention_kernel<HD, false>), grid, dim3(256), lds, s, a);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

int launch_attention(const AttnArgs& a, hipStream_t s) {
    SR_REQUIRE(a.nh % a.nkv == 0, "attention: num_heads %d not a multiple of num_kv_heads %d", a.nh, a.nkv);
    if (a.B == 0) return SR_OK;
    switch (a.hd) {
        case 64: return launch_hd<64>(a, s);
        case 128: return launch_hd<128>(a, s);
    }
    sr_set_error("attention: head_dim %d not supported (64 or 128)", a.hd);
    return SR_ERR_UNSUPPORTED;
}
