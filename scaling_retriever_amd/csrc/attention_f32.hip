// fp32 bidirectional GQA attention over packed var-len sequences (gfx950) - the fp32 regime of the encoder.
//
// The reference encodes dense queries WITHOUT autocast (eval_dense.py:94-106): SDPA then runs on fp32 q/k/v with an
// fp32 softmax (math path; the float mask of scaling_retriever/modeling/bidirectional_llama.py:138-161 rules out the
// flash kernel).  Attention is ~1 % of the encoder's FLOPs and the fp32 regime is the query side (sequences of ~9
// tokens, 64 at most), so this kernel is plain fp32 FMA work, not MFMA:
//   workgroup = (sequence, q head, block of 16 q rows); its 4 waves own 4 q rows each.
//   keys are walked in chunks of 64: the K chunk is staged once in LDS ([64][hd + 1] floats: a lane = one key reads
//   its row conflict-free, q is a broadcast read), scores / softmax are lane-local + wave reductions, the softmax is
//   carried online across chunks (running max and sum per q row), and P.V reads V rows straight from global memory
//   (lane = head dim: coalesced), p broadcast from LDS.
// q/k arrive rotated (RoPE is fused into the QKV GEMM epilogue).  The output is stored as the split-bf16 plane
// segments the o_proj GEMM consumes (kernels.h: SplitMap).
#include "kernels.h"
#include <math.h>

#define AF_KC 64      // keys per chunk
#define AF_RPW 4      // q rows per wave
#define AF_ROWS 16    // q rows per workgroup

template <int HD>
__global__ __launch_bounds__(256) void attention_f32_kernel(AttnF32Args a) {
    constexpr int DPL = HD / 64;                     // head dims per lane
    __shared__ float Ks[AF_KC][HD + 1];
    __shared__ float qs[AF_ROWS][HD];
    __shared__ float ps[AF_ROWS][AF_KC];
    __shared__ unsigned char kval[AF_KC];
    const int b = blockIdx.x, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = a.cu_seqlens[b];
    const int S = a.cu_seqlens[b + 1] - t0;
    const int r0 = blockIdx.z * AF_ROWS;
    if (r0 >= S || S <= a.only_gt) return;
    const int G = a.nh / a.nkv, kvh = h / G;
    const int ld = (a.nh + 2 * a.nkv) * HD;
    const int koff = a.nh * HD + kvh * HD, voff = (a.nh + a.nkv) * HD + kvh * HD;
    const int nrows = (S - r0) < AF_ROWS ? (S - r0) : AF_ROWS;

    for (int i = tid; i < nrows * HD; i += 256) {
        const int r = i / HD, d = i % HD;
        qs[r][d] = a.qkv[(int64_t)(t0 + r0 + r) * ld + h * HD + d];
    }
    float m[AF_RPW], l[AF_RPW], o[AF_RPW][DPL];
#pragma unroll
    for (int r = 0; r < AF_RPW; ++r) {
        m[r] = -INFINITY;
        l[r] = 0.f;
#pragma unroll
        for (int e = 0; e < DPL; ++e) o[r][e] = 0.f;
    }
    for (int k0 = 0; k0 < S; k0 += AF_KC) {
        const int nk = (S - k0) < AF_KC ? (S - k0) : AF_KC;
        __syncthreads();                              // previous chunk fully consumed (and qs visible)
        for (int i = tid; i < nk * HD; i += 256) {
            const int k = i / HD, d = i % HD;
            Ks[k][d] = a.qkv[(int64_t)(t0 + k0 + k) * ld + koff + d];
        }
        if (tid < AF_KC) kval[tid] = (tid < nk) ? a.key_valid[t0 + k0 + tid] : 0;
        __syncthreads();
        const bool valid = kval[lane] != 0;
#pragma unroll
        for (int r = 0; r < AF_RPW; ++r) {
            const int row = wave * AF_RPW + r;      // wave-uniform
            if (row >= nrows) break;
            float s = 0.f;
            for (int d = 0; d < HD; ++d) s += qs[row][d] * Ks[lane][d];
            s = valid ? s * a.scale : -INFINITY;
            float mx = s;
            for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            if (mx == -INFINITY) continue;          // every key of this chunk is masked
            const float m_new = fmaxf(m[r], mx);
            const float corr = expf(m[r] - m_new);  // 0 on the first chunk (m = -inf)
            const float p = valid ? expf(s - m_new) : 0.f;
            float sum = p;
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
            l[r] = l[r] * corr + sum;
            m[r] = m_new;
            ps[row][lane] = p;                      // same wave writes and reads this row: no barrier needed
            __builtin_amdgcn_wave_barrier();
            float acc[DPL];
#pragma unroll
            for (int e = 0; e < DPL; ++e) acc[e] = 0.f;
            for (int k = 0; k < nk; ++k) {
                const float pk = ps[row][k];
                const float* vrow = a.qkv + (int64_t)(t0 + k0 + k) * ld + voff;
#pragma unroll
                for (int e = 0; e < DPL; ++e) acc[e] += pk * vrow[lane + 64 * e];
            }
#pragma unroll
            for (int e = 0; e < DPL; ++e) o[r][e] = o[r][e] * corr + acc[e];
        }
    }
    const int64_t ldo = (int64_t)a.out_map.n_seg * a.nh * HD;
#pragma unroll
    for (int r = 0; r < AF_RPW; ++r) {
        const int row = wave * AF_RPW + r;
        if (row >= nrows) break;
        const float inv = l[r] > 0.f ? 1.0f / l[r] : 0.f;
        if (a.out_f32) {
#pragma unroll
            for (int e = 0; e < DPL; ++e) a.out_f32[(int64_t)(t0 + r0 + row) * a.nh * HD + h * HD + lane + 64 * e] = o[r][e] * inv;
            continue;
        }
        bf16_t* orow = a.out + (int64_t)(t0 + r0 + row) * ldo;
#pragma unroll
        for (int e = 0; e < DPL; ++e) {
            unsigned short p[3];
            split_bf16x3(o[r][e] * inv, p[0], p[1], p[2]);
            const int col = h * HD + lane + 64 * e;
            for (int sg = 0; sg < a.out_map.n_seg; ++sg) orow[(int64_t)sg * a.nh * HD + col] = p[a.out_map.plane[sg]];
        }
    }
}

// The same attention with ONE workgroup per (sequence, KV head, block of 16 q rows): wave w serves q head kvh * G + w, all
// 16 rows of the block four at a time, and the G waves share the staged K chunk; with head_dim 64 the V chunk is staged next
// to it, so that P.V no longer waits for one dependent global load per key (the critical path of the kernel above on the
// ~9-token sequences of the query side: query encode 362 -> 355 ms; grouping the heads alone changed nothing).  Same
// arithmetic per (row, head) as the kernel above (same chunking, same summation order): bit-identical outputs.
template <int HD, int G>
__global__ __launch_bounds__(64 * G) void attention_f32_gqa_kernel(AttnF32Args a) {
    constexpr int DPL = HD / 64;
    constexpr bool V_LDS = HD <= 64;                 // the V chunk too, while both fit the 64 KB of static LDS
    __shared__ float Ks[AF_KC][HD + 1];
    __shared__ float Vs[V_LDS ? AF_KC : 1][V_LDS ? HD : 1];
    __shared__ float qs[G][HD];
    __shared__ float ps[G][AF_KC];
    __shared__ unsigned char kval[AF_KC];
    const int b = blockIdx.x, kvh = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = kvh * G + wave;
    const int t0 = a.cu_seqlens[b];
    const int S = a.cu_seqlens[b + 1] - t0;
    const int r0 = blockIdx.z * AF_ROWS;
    if (r0 >= S || S <= a.only_gt) return;
    const int ld = (a.nh + 2 * a.nkv) * HD;
    const int koff = a.nh * HD + kvh * HD, voff = (a.nh + a.nkv) * HD + kvh * HD;
    const int nrows = (S - r0) < AF_ROWS ? (S - r0) : AF_ROWS;
    const int64_t ldo = (int64_t)a.out_map.n_seg * a.nh * HD;
    // four q rows at a time (their running max / sum / output stay in registers); the K chunk is staged again for every
    // group of four - a few hundred floats for the sequences this kernel is for
    for (int rq = 0; rq < nrows; rq += AF_RPW) {
        float m[AF_RPW], l[AF_RPW], o[AF_RPW][DPL];
#pragma unroll
        for (int r = 0; r < AF_RPW; ++r) {
            m[r] = -INFINITY;
            l[r] = 0.f;
#pragma unroll
            for (int e = 0; e < DPL; ++e) o[r][e] = 0.f;
        }
        for (int k0 = 0; k0 < S; k0 += AF_KC) {
            const int nk = (S - k0) < AF_KC ? (S - k0) : AF_KC;
            __syncthreads();                              // previous chunk fully consumed
            for (int i = tid; i < nk * HD; i += 64 * G) {
                const int k = i / HD, d = i % HD;
                Ks[k][d] = a.qkv[(int64_t)(t0 + k0 + k) * ld + koff + d];
                if constexpr (V_LDS) Vs[k][d] = a.qkv[(int64_t)(t0 + k0 + k) * ld + voff + d];
            }
            if (tid < AF_KC) kval[tid] = (tid < nk) ? a.key_valid[t0 + k0 + tid] : 0;
            __syncthreads();
            const bool valid = kval[lane] != 0;
#pragma unroll
            for (int r = 0; r < AF_RPW; ++r) {
                const int row = rq + r;                 // workgroup-uniform
                if (row >= nrows) break;
#pragma unroll
                for (int e = 0; e < DPL; ++e) qs[wave][lane + 64 * e] = a.qkv[(int64_t)(t0 + r0 + row) * ld + h * HD + lane + 64 * e];
                __builtin_amdgcn_wave_barrier();        // the wave's own row: LDS operations of one wave execute in order
                float s = 0.f;
                for (int d = 0; d < HD; ++d) s += qs[wave][d] * Ks[lane][d];
                s = valid ? s * a.scale : -INFINITY;
                float mx = s;
                for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
                if (mx == -INFINITY) continue;          // every key of this chunk is masked
                const float m_new = fmaxf(m[r], mx);
                const float corr = expf(m[r] - m_new);
                const float p = valid ? expf(s - m_new) : 0.f;
                float sum = p;
                for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
                l[r] = l[r] * corr + sum;
                m[r] = m_new;
                ps[wave][lane] = p;
                __builtin_amdgcn_wave_barrier();
                float acc[DPL];
#pragma unroll
                for (int e = 0; e < DPL; ++e) acc[e] = 0.f;
                for (int k = 0; k < nk; ++k) {
                    const float pk = ps[wave][k];
                    if constexpr (V_LDS) {               // one dependent global load per key was the kernel's critical path
                        acc[0] += pk * Vs[k][lane];
                    } else {
                        const float* vrow = a.qkv + (int64_t)(t0 + k0 + k) * ld + voff;
#pragma unroll
                        for (int e = 0; e < DPL; ++e) acc[e] += pk * vrow[lane + 64 * e];
                    }
                }
#pragma unroll
                for (int e = 0; e < DPL; ++e) o[r][e] = o[r][e] * corr + acc[e];
            }
        }
#pragma unroll
        for (int r = 0; r < AF_RPW; ++r) {
            const int row = rq + r;
            if (row >= nrows) break;
            const float inv = l[r] > 0.f ? 1.0f / l[r] : 0.f;
            if (a.out_f32) {
#pragma unroll
                for (int e = 0; e < DPL; ++e) a.out_f32[(int64_t)(t0 + r0 + row) * a.nh * HD + h * HD + lane + 64 * e] = o[r][e] * inv;
                continue;
            }
            bf16_t* orow = a.out + (int64_t)(t0 + r0 + row) * ldo;
#pragma unroll
            for (int e = 0; e < DPL; ++e) {
                unsigned short p[3];
                split_bf16x3(o[r][e] * inv, p[0], p[1], p[2]);
                const int col = h * HD + lane + 64 * e;
                for (int sg = 0; sg < a.out_map.n_seg; ++sg) orow[(int64_t)sg * a.nh * HD + col] = p[a.out_map.plane[sg]];
            }
        }
    }
}

// ---- sequences of at most 64 tokens (the query side: MS MARCO Dev queries are ~9 tokens): fp32 MFMA -----------------------
// v_mfma_f32_16x16x4_f32 multiplies and accumulates in fp32 (bitwise a chain of fmaf, MI355X_MICROARCH.md), so this is still
// exact-fp32 attention; what changes against the kernels above is the shape of the work.  They give a key to every lane and
// a q row to a wave step, which leaves 55 of 64 lanes idle on a 9-token sequence and walks the rows one after another
// (405 us per layer for 16 384 query tokens against a ~75 us floor for reading q/k/v and writing the output once).  Here:
//   workgroup = (sequence, KV head), its 4 waves = the 4 q heads of the group; K and V of the head are staged once in LDS
//   ([keys][hd + 1] floats: conflict-free operand reads), each wave stages its own q rows;
//   S^T = K Q^T per (16 keys x 16 q rows) block - 16 MFMAs for head dim 64 - so that a lane owns ONE q row (its column of
//   S^T): softmax statistics are lane-local + two cross-group shuffles, and the S^T accumulators ARE the A operand of
//   P V (keys contracted in the order 4 g + m: lane group g supplies its register m to MFMA m; V rows are read to match).
// The contraction order is fixed per (row, head) and independent of the batch: outputs do not depend on batch composition.
template <int HD>
__global__ __launch_bounds__(256) void attention_f32_mfma_kernel(AttnF32Args a) {
    constexpr int LDK = HD + 1, CB = HD / 16;
    extern __shared__ float af_smem[];
    const int b = blockIdx.x, kvh = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = a.cu_seqlens[b];
    const int S = a.cu_seqlens[b + 1] - t0;
    if (S <= 0 || S > a.only_le) return;
    const int nkb = (S + 15) >> 4;                  // key blocks = q blocks (<= 4)
    const int SP = nkb * 16;
    float* Ks = af_smem;
    float* Vs = Ks + SP * LDK;
    float* Qs = Vs + SP * LDK + wave * 16 * LDK;
    const int G = a.nh / a.nkv;                     // 4 (checked at launch)
    const int h = kvh * G + wave;
    const int ld = (a.nh + 2 * a.nkv) * HD;
    const int koff = a.nh * HD + kvh * HD, voff = (a.nh + a.nkv) * HD + kvh * HD;
    for (int i = tid; i < SP * (HD / 4); i += 256) {          // rows beyond the sequence are zeros (0 * p = 0, never NaN)
        const int k = i / (HD / 4), d = (i % (HD / 4)) * 4;
        f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
        if (k < S) {
            kv = *reinterpret_cast<const f32x4*>(a.qkv + (int64_t)(t0 + k) * ld + koff + d);
            vv = *reinterpret_cast<const f32x4*>(a.qkv + (int64_t)(t0 + k) * ld + voff + d);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { Ks[k * LDK + d + e] = kv[e]; Vs[k * LDK + d + e] = vv[e]; }
    }
    const int g = lane >> 4, c = lane & 15;
    // validity of the keys this lane's S^T registers hold: key = 16 kb + 4 g + r
    unsigned vmask = 0;
    for (int kb = 0; kb < nkb; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * kb + 4 * g + r;
            if (key < S && a.key_valid[t0 + key]) vmask |= 1u << (4 * kb + r);
        }
    __syncthreads();
    const int64_t ldo = (int64_t)a.out_map.n_seg * a.nh * HD;
    for (int qb = 0; qb < nkb; ++qb) {
        // this wave's q rows 16 qb .. + 16 of head h (its own LDS region: LDS operations of one wave execute in order)
        for (int i = lane; i < 16 * (HD / 4); i += 64) {
            const int r = i / (HD / 4), d = (i % (HD / 4)) * 4;
            f32x4 qv = {0.f, 0.f, 0.f, 0.f};
            if (16 * qb + r < S) qv = *reinterpret_cast<const f32x4*>(a.qkv + (int64_t)(t0 + 16 * qb + r) * ld + h * HD + d);
#pragma unroll
            for (int e = 0; e < 4; ++e) Qs[r * LDK + d + e] = qv[e];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        f32x4 st[4];
        float qf[HD / 4];
#pragma unroll
        for (int kk = 0; kk < HD / 4; ++kk) qf[kk] = Qs[c * LDK + 4 * kk + g];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            st[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kb < nkb) {
#pragma unroll
                for (int kk = 0; kk < HD / 4; ++kk)
                    st[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[(16 * kb + c) * LDK + 4 * kk + g], qf[kk], st[kb], 0, 0, 0);
            }
        }
        // softmax over the keys of q row c (this lane's column of S^T): registers, then the 4 lane groups
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool v = (vmask >> (4 * kb + r)) & 1;
                st[kb][r] = v ? st[kb][r] * a.scale : -INFINITY;
                m = fmaxf(m, st[kb][r]);
            }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = (m > -INFINITY && st[kb][r] > -INFINITY) ? expf(st[kb][r] - m) : 0.f;
                st[kb][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;       // every key masked: the row is zero
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[kb][r] *= inv;
        // O = P V: output block cb = head dims 16 cb .. + 16; lane holds O[q row 4 g + r][16 cb + c]
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                if (kb < nkb) {
#pragma unroll
                    for (int mm = 0; mm < 4; ++mm)
                        o = __builtin_amdgcn_mfma_f32_16x16x4f32(st[kb][mm], Vs[(16 * kb + 4 * g + mm) * LDK + 16 * cb + c], o, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * qb + 4 * g + r;
                if (row >= S) continue;
                const int col = h * HD + 16 * cb + c;
                if (a.out_f32) {
                    a.out_f32[(int64_t)(t0 + row) * a.nh * HD + col] = o[r];
                } else {
                    unsigned short p[3];
                    split_bf16x3(o[r], p[0], p[1], p[2]);
                    bf16_t* orow = a.out + (int64_t)(t0 + row) * ldo;
                    for (int sg = 0; sg < a.out_map.n_seg; ++sg) orow[(int64_t)sg * a.nh * HD + col] = p[a.out_map.plane[sg]];
                }
            }
        }
    }
}

template <int HD>
static int launch_attention_f32_mfma(const AttnF32Args& a, hipStream_t s) {
    const int sp = (int)ceil_div64(a.max_seqlen < 64 ? a.max_seqlen : 64, 16) * 16;
    const size_t lds = sizeof(float) * (size_t)(2 * sp + 4 * 16) * (HD + 1);
    static DeviceOnce attr_once;
    if (bool* slot = attr_once.pending()) {
        SR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f32_mfma_kernel<HD>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * (2 * 64 + 64) * (HD + 1))));
        *slot = true;
    }
    const dim3 grid((unsigned)a.B, (unsigned)a.nkv), block(256);
    SR_REQUIRE(grid.y <= 65535, "attention(fp32): grid too large");
    hipLaunchKernelGGL(attention_f32_mfma_kernel<HD>, grid, block, lds, s, a);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

int launch_attention_f32(const AttnF32Args& a_in, hipStream_t s) {
    AttnF32Args a = a_in;
    SR_REQUIRE(a.nh % a.nkv == 0, "attention(fp32): num_heads %d not a multiple of num_kv_heads %d", a.nh, a.nkv);
    SR_REQUIRE(a.out_f32 || (a.out_map.n_seg >= 1 && a.out_map.n_seg <= SR_MAX_SEG), "attention(fp32): bad segment map");
    if (a.B == 0 || a.max_seqlen <= 0) return SR_OK;
    const int G = a.nh / a.nkv;
    a.only_le = 0; a.only_gt = 0;
    const char* env_m = sr_dev_getenv("SR_ATTN_F32_MFMA");    // A/B switch: 0 = the FMA kernels below for every length
    if (G == 4 && (a.hd == 64 || a.hd == 128) && !(env_m && *env_m == '0')) {
        // The kernel is chosen PER SEQUENCE: <= 64 tokens -> the fp32-MFMA kernel, longer -> the FMA kernels.  A mixed batch runs
        // both launches, each skipping the other's sequences (choosing by the batch's longest sequence made the bits
        // of a short sequence depend on its neighbours).
        a.only_le = 64;
        SR_TRY(a.hd == 64 ? launch_attention_f32_mfma<64>(a, s) : launch_attention_f32_mfma<128>(a, s));
        if (a.max_seqlen <= 64) return SR_OK;
        a.only_le = 0; a.only_gt = 64;
    }
    const char* env = sr_dev_getenv("SR_ATTN_F32_GQA");       // A/B switch: 0 = one workgroup per q head
    if (G == 4 && !(env && *env == '0')) {
        const dim3 grid((unsigned)a.B, (unsigned)a.nkv, (unsigned)ceil_div64(a.max_seqlen, AF_ROWS)), block(256);
        SR_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "attention(fp32): grid too large");
        switch (a.hd) {
            case 64: hipLaunchKernelGGL((attention_f32_gqa_kernel<64, 4>), grid, block, 0, s, a); break;
            case 128: hipLaunchKernelGGL((attention_f32_gqa_kernel<128, 4>), grid, block, 0, s, a); break;
            default: sr_set_error("attention(fp32): head_dim %d not supported (64 or 128)", a.hd); return SR_ERR_UNSUPPORTED;
        }
        SR_CHECK_LAUNCH();
        return SR_OK;
    }
    const dim3 grid((unsigned)a.B, (unsigned)a.nh, (unsigned)ceil_div64(a.max_seqlen, AF_ROWS)), block(256);
    SR_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "attention(fp32): grid too large");
    switch (a.hd) {
        case 64: hipLaunchKernelGGL(attention_f32_kernel<64>, grid, block, 0, s, a); break;
        case 128: hipLaunchKernelGGL(attention_f32_kernel<128>, grid, block, 0, s, a); break;
        default: sr_set_error("attention(fp32): head_dim %d not supported (64 or 128)", a.hd); return SR_ERR_UNSUPPORTED;
    }
    SR_CHECK_LAUNCH();
    return SR_OK;
}
