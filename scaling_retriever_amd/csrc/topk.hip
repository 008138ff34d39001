// Fused top-k machinery shared by the dense and sparse scorers (gfx950).
//
// The score kernels never materialise [nq, N] scores.  Each one filters its
// tile against a per-query threshold tau (score of the current k-th best) and
// appends survivors as 64-bit keys (ordered score bits << 32 | ~doc index) to a
// per-query candidate buffer.  After every doc chunk, topk_compact_kernel folds
// the candidates into the running (unsorted) set.  The set has room for 2k keys:
// once tau is finite, candidates are only APPENDED (they all beat tau, so the set
// stays a superset of the top-k) and the LDS radix select that cuts it back to k
// and raises tau runs only when the 2k slots overflow - about every 30th chunk of
// a long scan instead of every chunk.  topk_sort_kernel sorts what is left once at
// the end and emits the best k.
//
// Result = top-k by (score desc, doc index asc): deterministic, independent of
// tile scheduling.  Stands in for faiss' heap top-k (indexer.py:211) and for
// np.argpartition in select_topk (indexer.py:315-322).
#include "common.h"
#include <mutex>

// ------------------------------------------------------------------ reset ---
__global__ void topk_reset_kernel(int* run_count, int* cand_count, float* tau, int64_t nq) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq) {
        run_count[i] = 0;
        cand_count[i] = 0;
        tau[i] = -INFINITY;
    }
}

// ---------------------------------------------------------- segment gather ---
// Segmented candidate slots (common.h TopkWS): every segment's survivors are appended behind the nc atomically appended
// candidates of the query, cand_count is updated and the segment counts are cleared for the next launch.  Sources (the
// buffer's tail, from seg_off) and destinations (its head) are disjoint.  Launched in front of topk_compact_kernel.
// One WAVE per query (4 queries per workgroup, no barriers, no LDS): a workgroup per query spent its time in two barriers
// and 3.4 rounds of workgroups over the chip for ~30 keys of work (18-25 us per launch, now one round).
__global__ __launch_bounds__(256) void topk_gather_segments_kernel(uint64_t* __restrict__ cand_keys, int* __restrict__ cand_count,
                                                                   int64_t cand_cap, unsigned char* __restrict__ seg_cnt, int seg_n,
                                                                   int64_t seg_off, int64_t nq) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;                                  // wave-uniform
    int nc = cand_count[q];                               // read and rewritten by this wave only
    uint64_t* cbuf = cand_keys + q * cand_cap;
    uint32_t* cw = reinterpret_cast<uint32_t*>(seg_cnt + q * seg_n);      // seg_n is a multiple of 4
    const int n_words = seg_n / 4;
    uint32_t next = lane < n_words ? cw[lane] : 0u;
    for (int w0 = 0; w0 < n_words; w0 += 64) {
        const int wi = w0 + lane;
        const uint32_t word = next;
        next = wi + 64 < n_words ? cw[wi + 64] : 0u;      // the next round's counts are on their way during this round's copies
        const int c[4] = {(int)(word & 255u), (int)((word >> 8) & 255u), (int)((word >> 16) & 255u), (int)(word >> 24)};
        const int mine = c[0] + c[1] + c[2] + c[3];
        int v = mine;                                     // inclusive prefix inside the wave
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(v, off);
            if (lane >= off) v += o;
        }
        const int total = __shfl(v, 63);
        int base = nc + v - mine;
        if (mine) {
            for (int e = 0; e < 4; ++e)
                for (int j = 0; j < c[e]; ++j) {
                    if (base < seg_off) cbuf[base] = cbuf[seg_off + ((int64_t)wi * 4 + e) * SR_SEG_P + j];
                    ++base;
                }
            cw[wi] = 0u;
        }
        nc += total;
    }
    if (lane == 0) cand_count[q] = nc < seg_off ? nc : (int)seg_off;
}

// ---------------------------------------------------------------- compact ---
// One workgroup (256 threads) per query.  run_keys has 2k slots per query.
// TOPK_SEL_R = keys per thread the select keeps in registers: 28 (unions of up to 7 168 keys; 120 VGPRs = 4 workgroups per CU) by default,
// 20 (5 120 keys) for callers whose unions are known to be smaller (the certified sparse scorer: at most 2 (k + 1 024) keys + a launch's survivors)
template <int TOPK_SEL_R>
__global__ __launch_bounds__(256) void topk_compact_kernel(uint64_t* __restrict__ run_keys, int* __restrict__ run_count,
                                                           float* __restrict__ tau, uint64_t* __restrict__ cand_keys,
                                                           int* __restrict__ cand_count, int k, int64_t cand_cap,
                                                           int k2, float* __restrict__ tau2, int select_over) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int* hist = reinterpret_cast<int*>(smem_raw);          // [256]
    int* scan = hist + 256;                                 // [256]
    int* ctrl = scan + 256;                                 // [8]
    uint32_t* hole_pos = reinterpret_cast<uint32_t*>(ctrl + 8);  // [k]
    uint32_t* filler_idx = hole_pos + k;                          // [k]

    const int q = blockIdx.x, tid = threadIdx.x;
    int nc = cand_count[q];
    if (nc == 0) return;
    if ((int64_t)nc > cand_cap) nc = (int)cand_cap;  // unreachable by construction (cap >= docs per chunk)
    const int nr = run_count[q];
    const int cap = 2 * k;
    uint64_t* run = run_keys + (int64_t)q * cap;
    const uint64_t* cand = cand_keys + (int64_t)q * cand_cap;
    const int n = nr + nc;

    // append only: below k keys there is nothing to cut; with k or more already held tau is finite and every candidate
    // beats it, so the set stays a superset of the top-k until its 2k slots overflow
    if (n <= k || (nr >= k && n <= select_over)) {      // select_over = 2 k unless the caller wants fresher thresholds (topk_compact2)
        for (int i = tid; i < nc; i += 256) run[nr + i] = cand[i];
        if (n == k) {
            // tau = smallest kept score
            uint64_t mn = ~0ull;
            for (int i = tid; i < n; i += 256) {
                uint64_t key = i < nr ? run[i] : cand[i - nr];
                mn = key < mn ? key : mn;
            }
            for (int off = 32; off > 0; off >>= 1) {
                uint64_t o = __shfl_xor(mn, off);
                mn = o < mn ? o : mn;
            }
            uint64_t* red = reinterpret_cast<uint64_t*>(hist);
            if ((tid & 63) == 0) red[tid >> 6] = mn;
            __syncthreads();
            if (tid == 0) {
                uint64_t m = red[0];
                for (int w = 1; w < 4; ++w) m = red[w] < m ? red[w] : m;
                tau[q] = sr_key_score(m);
            }
        }
        // every wave has read cand_count[q] and run_count[q] before one thread publishes the new values: the waves of a
        // workgroup do not start together, and without this barrier a late wave could read the counts thread 0 had already
        // rewritten (round 3: one in ~10^7 workgroups, first seen as wild writes when the run_count load became a slower
        // vector load; the torn wave ran the select alone on LDS nobody had initialised - DESIGN.md section 0)
        __syncthreads();
        if (tid == 0) {
            run_count[q] = n;
            cand_count[q] = 0;
        }
        return;
    }

    // radix select (MSB first, 8 bits per pass) of the k-th largest key of the union.  One selecting workgroup holds up the whole
    // launch, and its time is memory latency: the union (up to 2k + a launch's survivors, ~50 KB) read from L2 / HBM once per
    // pass, a key per trip, took ~60 us; a union of at most 256 x TOPK_SEL_R keys is now read ONCE into registers (independent
    // loads in flight together) and every pass works from there (~15 us); a larger one re-reads it per pass, 8 loads in flight.
    const bool in_regs = n <= 256 * TOPK_SEL_R;
    uint64_t kreg[TOPK_SEL_R];
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < TOPK_SEL_R; ++j) {
            const int i = tid + 256 * j;
            kreg[j] = i < n ? (i < nr ? run[i] : cand[i - nr]) : 0ull;
        }
    }
    // the leading bytes of a query's keys (sign, exponent, first mantissa bits of scores that lie close together) are mostly ONE
    // value: 64 adds to one LDS word would run one after another, so a wave whose keys agree adds once
#define TOPK_HIST_ADD(key)                                                                            \
    do {                                                                                              \
        const int bin_ = (int)(((key) >> shift) & 255);                                               \
        const int b0_ = __builtin_amdgcn_readfirstlane(bin_);                                         \
        const unsigned long long act_ = __ballot(1), same_ = __ballot(bin_ == b0_);                   \
        if (same_ == act_) {                                                                          \
            if ((tid & 63) == __ffsll((long long)act_) - 1) atomicAdd(&hist[b0_], __popcll(act_));    \
        } else {                                                                                      \
            atomicAdd(&hist[bin_], 1);                                                                \
        }                                                                                             \
    } while (0)
    uint64_t prefix = 0;
    int remaining = k;
    for (int pass = 7; pass >= 0; --pass) {
        const int shift = pass * 8;
        hist[tid] = 0;
        __syncthreads();
        if (in_regs) {
#pragma unroll
            for (int j = 0; j < TOPK_SEL_R; ++j)
                if (tid + 256 * j < n && (pass == 7 || (kreg[j] >> (shift + 8)) == prefix)) TOPK_HIST_ADD(kreg[j]);
        } else {
            for (int i0 = tid; i0 < n; i0 += 256 * 8) {
                uint64_t keys[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * 256;
                    keys[u] = i < n ? (i < nr ? run[i] : cand[i - nr]) : 0ull;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + u * 256 < n && (pass == 7 || (keys[u] >> (shift + 8)) == prefix)) TOPK_HIST_ADD(keys[u]);
            }
        }
        __syncthreads();
        // suffix sums of the 256 bins, s = sum_{b >= tid} hist[b]: inside a wave by shuffles, across the four waves through four
        // LDS words (4 barriers per pass; the 8-step LDS scan this replaces took 21)
        const int c = hist[tid];
        int s = c;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_down(s, off);
            if ((tid & 63) + off < 64) s += o;
        }
        if ((tid & 63) == 0) scan[tid >> 6] = s;
        __syncthreads();
        for (int w = (tid >> 6) + 1; w < 4; ++w) s += scan[w];
        const int above = s - c;
        if (s >= remaining && above < remaining) {       // exactly one bin: an empty bin has s == above
            ctrl[0] = tid;
            ctrl[1] = remaining - above;
        }
        __syncthreads();
        prefix = (prefix << 8) | (uint64_t)ctrl[0];
        remaining = ctrl[1];
        // no barrier here: the next pass rewrites hist after these reads of it (two barriers back), scan[] two barriers and
        // ctrl[] three barriers further on
    }
    const uint64_t T = prefix;  // the k-th largest key (keys are unique)
    // Optional second rank k2 < k (sparse_cert.hip: the k2-th best SCORE so far gives a tighter, still valid, filter threshold than the k-th
    // best key): four more passes over the score half of the keys, from the registers only (a larger union leaves tau2 as it was - a lower
    // bound of the k2-th best score stays one).
    if (k2 > 0 && in_regs) {
        uint64_t prefix2 = 0;
        int remaining2 = k2;
        for (int pass = 7; pass >= 4; --pass) {
            const int shift = pass * 8;
            hist[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < TOPK_SEL_R; ++j)
                if (tid + 256 * j < n && (pass == 7 || (kreg[j] >> (shift + 8)) == prefix2)) TOPK_HIST_ADD(kreg[j]);
            __syncthreads();
            const int c = hist[tid];
            int s2 = c;
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_down(s2, off);
                if ((tid & 63) + off < 64) s2 += o;
            }
            if ((tid & 63) == 0) scan[tid >> 6] = s2;
            __syncthreads();
            for (int w = (tid >> 6) + 1; w < 4; ++w) s2 += scan[w];
            const int above = s2 - c;
            if (s2 >= remaining2 && above < remaining2) {
                ctrl[0] = tid;
                ctrl[1] = remaining2 - above;
            }
            __syncthreads();
            prefix2 = (prefix2 << 8) | (uint64_t)ctrl[0];
            remaining2 = ctrl[1];
        }
        if (tid == 0) tau2[q] = sr_key_score(prefix2 << 32);
    }
#undef TOPK_HIST_ADD

    if (tid == 0) {
        ctrl[2] = 0;
        ctrl[3] = 0;
    }
    __syncthreads();
    // the k keys >= T end up in run[0, k): slots there that are empty or hold a smaller key are holes, filled from the
    // kept keys of run[k, nr) and of the candidates
    const int extra = nr > k ? nr - k : 0;
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < TOPK_SEL_R; ++j) {
            const int i = tid + 256 * j;
            if (i < n) {
                const bool keep = kreg[j] >= T;
                if (i < nr) {
                    if (i < k) {
                        if (!keep) hole_pos[atomicAdd(&ctrl[2], 1)] = (uint32_t)i;
                    } else if (keep) {
                        filler_idx[atomicAdd(&ctrl[3], 1)] = (uint32_t)(i - k);
                    }
                } else if (keep) {
                    filler_idx[atomicAdd(&ctrl[3], 1)] = (uint32_t)(extra + i - nr);
                }
            }
        }
        for (int i = nr + tid; i < k; i += 256) hole_pos[atomicAdd(&ctrl[2], 1)] = (uint32_t)i;      // empty slots (nr < k)
    } else {
        for (int i0 = tid; i0 < k; i0 += 256 * 4) {
            uint64_t keys[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256;
                keys[u] = i < k && i < nr ? run[i] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256;
                if (i < k && (i >= nr || keys[u] < T)) hole_pos[atomicAdd(&ctrl[2], 1)] = (uint32_t)i;
            }
        }
        for (int i0 = tid; i0 < extra; i0 += 256 * 4) {
            uint64_t keys[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) keys[u] = i0 + u * 256 < extra ? run[k + i0 + u * 256] : 0ull;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * 256 < extra && keys[u] >= T) filler_idx[atomicAdd(&ctrl[3], 1)] = (uint32_t)(i0 + u * 256);
        }
        for (int i0 = tid; i0 < nc; i0 += 256 * 4) {
            uint64_t keys[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) keys[u] = i0 + u * 256 < nc ? cand[i0 + u * 256] : 0ull;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * 256 < nc && keys[u] >= T) filler_idx[atomicAdd(&ctrl[3], 1)] = (uint32_t)(extra + i0 + u * 256);
        }
    }
    __syncthreads();
    const int nf = ctrl[3] < ctrl[2] ? ctrl[3] : ctrl[2];  // equal by construction
    for (int j0 = tid; j0 < nf; j0 += 256 * 4) {             // sources lie at or above slot k, holes below it; 4 loads in flight
        uint64_t v[4];
        uint32_t hp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * 256;
            v[u] = 0ull;
            hp[u] = 0u;
            if (j < nf) {
                const uint32_t f = filler_idx[j];
                hp[u] = hole_pos[j];
                v[u] = f < (uint32_t)extra ? run[k + f] : cand[f - (uint32_t)extra];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j0 + u * 256 < nf) run[hp[u]] = v[u];
    }
    if (tid == 0) {
        run_count[q] = k;
        cand_count[q] = 0;
        tau[q] = sr_key_score(T);
    }
}

// ------------------------------------------------------------------- sort ---
// Bitonic sort (descending) of the running set (up to 2k keys) in LDS; P = next pow2 >= 2k; the best k are written.
__global__ __launch_bounds__(256) void topk_sort_kernel(const uint64_t* __restrict__ run_keys, const int* __restrict__ run_count,
                                                        int k, int P, float pad_score, float* __restrict__ out_scores,
                                                        int64_t* __restrict__ out_ids, int32_t* __restrict__ out_counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw);
    const int q = blockIdx.x, tid = threadIdx.x;
    const int held = run_count[q];
    const int cnt = held < k ? held : k;
    const uint64_t* run = run_keys + (int64_t)q * (2 * k);
    // sort only as many slots as this query holds (a power of two, at most the P the launch reserved LDS for)
    int Pq = 2;
    while (Pq < held && Pq < P) Pq <<= 1;
    P = Pq;
    for (int i = tid; i < P; i += 256) keys[i] = i < held ? run[i] : 0ull;
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) {
        for (int j = size >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool desc = (i & size) == 0;
                    const uint64_t a = keys[i], b = keys[ixj];
                    if (desc ? (a < b) : (a > b)) {
                        keys[i] = b;
                        keys[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k; i += 256) {
        const int64_t o = (int64_t)q * k + i;
        if (i < cnt) {
            out_scores[o] = sr_key_score(keys[i]);
            out_ids[o] = (int64_t)sr_key_gid(keys[i]);
        } else {
            out_scores[o] = pad_score;
            out_ids[o] = -1;
        }
    }
    if (out_counts && tid == 0) out_counts[q] = cnt;
}

// ------------------------------------------------------------ list packing ---
__global__ void topk_pack_lists_kernel(const float* __restrict__ scores, const int64_t* __restrict__ ids, int n_lists,
                                       int64_t nq, int k, uint64_t* __restrict__ cand_keys, int* __restrict__ cand_count,
                                       int64_t cand_cap) {
    // one workgroup per query; keeps only valid (id >= 0) entries
    const int64_t q = blockIdx.x;
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const int total = n_lists * k;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int l = i / k, j = i - l * k;
        const int64_t src = ((int64_t)l * nq + q) * k + j;
        const int64_t id = ids[src];
        if (id >= 0) {
            const int pos = atomicAdd(&cnt, 1);
            cand_keys[q * cand_cap + pos] = sr_make_key(scores[src], (uint32_t)id);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) cand_count[q] = cnt;
}

// ------------------------------------------------------------------- host ---
static int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

int TopkWS::ensure_segments(int64_t nq, int kk, int64_t dense_cap, int sn) {
    // (re)allocated whenever the shape differs: the segment geometry is part of the kernels' addressing
    if (nq <= nq_cap && kk <= k && seg_n == sn && seg_off == dense_cap && seg_cnt) return SR_OK;
    release();
    SR_TRY(ensure(nq, kk, dense_cap + (int64_t)sn * SR_SEG_P));
    if (hipMalloc(&seg_cnt, (size_t)nq * (size_t)sn) != hipSuccess) {
        (void)hipGetLastError();
        release();
        sr_set_error("top-k workspace: out of device memory for the segment counts");
        return SR_ERR_NOMEM;
    }
    SR_CHECK_HIP(hipMemset(seg_cnt, 0, (size_t)nq * (size_t)sn));
    seg_n = sn;
    seg_off = dense_cap;
    return SR_OK;
}

int TopkWS::ensure(int64_t nq, int kk, int64_t cc) {
    if (nq <= nq_cap && kk <= k && cc <= cand_cap && seg_n == 0) return SR_OK;
    release();
    // the shape is recorded only once every buffer exists: after a failed allocation the next call with the same shape must not take the
    // early-out above with null buffers behind it
    if (hipMalloc(&run_keys, sizeof(uint64_t) * (size_t)nq * 2 * (size_t)kk) != hipSuccess || hipMalloc(&run_count, sizeof(int) * (size_t)nq) != hipSuccess ||
        hipMalloc(&tau, sizeof(float) * (size_t)nq) != hipSuccess || hipMalloc(&cand_keys, sizeof(uint64_t) * (size_t)nq * (size_t)cc) != hipSuccess ||
        hipMalloc(&cand_count, sizeof(int) * (size_t)nq) != hipSuccess) {
        (void)hipGetLastError();
        release();
        sr_set_error("top-k workspace: out of device memory (%lld queries, k = %d, %lld candidate slots each)", (long long)nq, kk, (long long)cc);
        return SR_ERR_NOMEM;
    }
    nq_cap = nq;
    k = kk;
    cand_cap = cc;
    return SR_OK;
}

void TopkWS::release() {
    if (run_keys) (void)hipFree(run_keys);
    if (run_count) (void)hipFree(run_count);
    if (tau) (void)hipFree(tau);
    if (cand_keys) (void)hipFree(cand_keys);
    if (cand_count) (void)hipFree(cand_count);
    if (seg_cnt) (void)hipFree(seg_cnt);
    seg_cnt = nullptr;
    seg_n = 0;
    seg_off = 0;
    run_keys = cand_keys = nullptr;
    run_count = cand_count = nullptr;
    tau = nullptr;
    nq_cap = 0;
    k = 0;
    cand_cap = 0;
}

int topk_reset(TopkWS& ws, int64_t nq, hipStream_t s) {
    if (nq == 0) return SR_OK;
    hipLaunchKernelGGL(topk_reset_kernel, dim3((unsigned)ceil_div64(nq, 256)), dim3(256), 0, s, ws.run_count,
                       ws.cand_count, ws.tau, nq);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

// NOTE: kernels index run_keys with stride 2 * k (k = the logical k of this search, at most the allocated ws.k) and
// cand_keys with stride ws.cand_cap.
int topk_compact(TopkWS& ws, int64_t nq, int k, hipStream_t s) { return topk_compact2(ws, nq, k, 0, nullptr, 2 * k, s); }

int topk_compact2(TopkWS& ws, int64_t nq, int k, int k2, float* d_tau2, int select_over, hipStream_t s) {
    select_over = select_over < k ? k : (select_over > 2 * k ? 2 * k : select_over);
    if (nq == 0) return SR_OK;
    const size_t lds = sizeof(int) * (256 + 256 + 8) + sizeof(uint32_t) * 2 * (size_t)k;
    if (ws.seg_n > 0) {
        hipLaunchKernelGGL(topk_gather_segments_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, ws.cand_keys, ws.cand_count,
                           ws.cand_cap, ws.seg_cnt, ws.seg_n, ws.seg_off, nq);
        SR_CHECK_LAUNCH();
    }
    if (k2 > 0 && 2 * k + 1024 <= 256 * 20)
        hipLaunchKernelGGL(topk_compact_kernel<20>, dim3((unsigned)nq), dim3(256), lds, s, ws.run_keys, ws.run_count, ws.tau,
                           ws.cand_keys, ws.cand_count, k, ws.cand_cap, k2, d_tau2, select_over);
    else
        hipLaunchKernelGGL(topk_compact_kernel<28>, dim3((unsigned)nq), dim3(256), lds, s, ws.run_keys, ws.run_count, ws.tau,
                           ws.cand_keys, ws.cand_count, k, ws.cand_cap, k2, d_tau2, select_over);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

int topk_finalize(TopkWS& ws, int64_t nq, int k, float pad_score, float* d_out_scores, int64_t* d_out_ids,
                  int32_t* d_out_counts, hipStream_t s) {
    if (nq == 0) return SR_OK;
    const int P = next_pow2(2 * k);
    static DeviceOnce lds_set;      // k = 4096: 8192 keys = the whole 64 KB
    if (bool* slot = lds_set.pending()) {
        SR_CHECK_HIP(hipFuncSetAttribute((const void*)topk_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(sizeof(uint64_t) * 2 * SR_MAX_TOPK)));
        *slot = true;
    }
    hipLaunchKernelGGL(topk_sort_kernel, dim3((unsigned)nq), dim3(256), sizeof(uint64_t) * (size_t)P, s, ws.run_keys,
                       ws.run_count, k, P, pad_score, d_out_scores, d_out_ids, d_out_counts);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

static std::mutex g_merge_mu;
static TopkWS g_merge_ws;

extern "C" int sr_topk_merge(const float* d_scores, const int64_t* d_ids, int n_lists, int64_t nq, int k,
                             float pad_score, float* d_out_scores, int64_t* d_out_ids, sr_stream stream) {
    SR_REQUIRE(n_lists >= 1 && nq >= 0 && k >= 1 && k <= SR_MAX_TOPK, "sr_topk_merge: bad sizes (n_lists=%d nq=%lld k=%d)",
               n_lists, (long long)nq, k);
    SR_REQUIRE(d_scores && d_ids && d_out_scores && d_out_ids, "sr_topk_merge: null pointer");
    if (nq == 0) return SR_OK;
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(g_merge_mu);
    // exact-size workspace: strides must equal the logical sizes
    if (g_merge_ws.k != k || g_merge_ws.cand_cap != (int64_t)n_lists * k || g_merge_ws.nq_cap < nq) {
        g_merge_ws.release();
        SR_TRY(g_merge_ws.ensure(nq, k, (int64_t)n_lists * k));
    }
    SR_TRY(topk_reset(g_merge_ws, nq, s));
    hipLaunchKernelGGL(topk_pack_lists_kernel, dim3((unsigned)nq), dim3(256), 0, s, d_scores, d_ids, n_lists, nq, k,
                       g_merge_ws.cand_keys, g_merge_ws.cand_count, g_merge_ws.cand_cap);
    SR_CHECK_LAUNCH();
    SR_TRY(topk_compact(g_merge_ws, nq, k, s));
    SR_TRY(topk_finalize(g_merge_ws, nq, k, pad_score, d_out_scores, d_out_ids, nullptr, s));
    return SR_OK;
}
