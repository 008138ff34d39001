// On-device inverted-index build (gfx950): doc-major COO postings -> CSR by term.
//
// Replaces IndexDictOfArray.add_batch_document's per-posting Python append (scaling_retriever/utils/inverted_index.py:67-76) and
// the rank-major concatenation of merge_indexes (:108-170).  SparseIndexer.index (indexer.py:239-308) hands over the postings of
// its encoded batches as (global doc row, term, value) triples in insertion order; the reference's posting list of a term holds
// that term's postings in insertion order.  That is a STABLE sort of the triples by term: here a least-significant-digit radix
// sort with digits of up to 9 bits, one wave per 8 192 consecutive postings:
//   pass = histogram kernel (digit counts per wave, digit-major) -> exclusive scan -> scatter kernel (a wave walks its postings
//   in rounds of 64; a lane's rank among the round's lanes with the same digit comes from one ballot per digit bit, the wave's
//   running digit counters live in 2 KB of LDS, so equal digits keep their input order).
// With sort_docs the triples are first sorted by doc row the same way (posting lists ascending by doc id whatever the input
// order: a merged multi-rank index is rank-major inside a term, inverted_index.py:139-146).  indptr = lower bound of every term
// in the sorted term array.  Traffic per pass: 4 B (histogram) + 24 B (scatter) per posting, HBM-bound.
#include "sparse_index.h"
#include <type_traits>

#define SBW_ELEMS 8192        // postings per wave
#define SBW_WAVES 4           // waves per workgroup
#define SB_MAXBITS 9

// ------------------------------------------------------------------------------------------------ exclusive scan ---
// Three launches: per-block sums (2 048 items per block), one workgroup scans the block sums in place (loop with carry), every
// block rescans its items on top of its base.  T = int64_t or uint32_t.
#define SCAN_ITEMS 8
#define SCAN_BLOCK (256 * SCAN_ITEMS)

template <typename T>
__device__ inline T block_exclusive_scan_256(T v, T* lds_wave /* [4] */, T& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) lds_wave[wave] = incl;
    __syncthreads();
    T base = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wave) base += lds_wave[w];
        total += lds_wave[w];
    }
    __syncthreads();
    return base + incl - v;
}

template <typename T>
__global__ __launch_bounds__(256) void scan_block_sums_kernel(const T* __restrict__ in, int64_t n, T* __restrict__ sums) {
    __shared__ T red[4];
    const int64_t b0 = (int64_t)blockIdx.x * SCAN_BLOCK;
    T acc = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t j = b0 + (int64_t)i * 256 + threadIdx.x;
        if (j < n) acc += in[j];
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

template <typename T>
__global__ __launch_bounds__(256) void scan_sums_kernel(T* __restrict__ sums, int64_t nb, T* __restrict__ total_out) {
    __shared__ T lw[4];
    T carry = 0;
    for (int64_t c0 = 0; c0 < nb; c0 += 256) {
        const int64_t j = c0 + threadIdx.x;
        const T v = j < nb ? sums[j] : (T)0;
        T tot;
        const T ex = block_exclusive_scan_256<T>(v, lw, tot);
        if (j < nb) sums[j] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

template <typename T>
__global__ __launch_bounds__(256) void scan_apply_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t n,
                                                         const T* __restrict__ sums) {
    __shared__ T lw[4];
    const int64_t b0 = (int64_t)blockIdx.x * SCAN_BLOCK;
    // thread t owns items [t * SCAN_ITEMS, (t + 1) * SCAN_ITEMS) of the block: a blocked arrangement keeps the order
    T v[SCAN_ITEMS];
    T mine = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t j = b0 + (int64_t)threadIdx.x * SCAN_ITEMS + i;
        v[i] = j < n ? in[j] : (T)0;
        mine += v[i];
    }
    T tot;
    T run = sums[blockIdx.x] + block_exclusive_scan_256<T>(mine, lw, tot);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t j = b0 + (int64_t)threadIdx.x * SCAN_ITEMS + i;
        if (j < n) out[j] = run;
        run += v[i];
    }
}

// out[i] = sum_{j < i} in[j] for i < n; out[n] = total when write_total (out then has n + 1 slots).  in == out allowed.
template <typename T>
static int device_exclusive_scan(const T* d_in, T* d_out, int64_t n, bool write_total, hipStream_t s) {
    if (n <= 0) {
        if (write_total) SR_CHECK_HIP(hipMemsetAsync(d_out, 0, sizeof(T), s));
        return SR_OK;
    }
    const int64_t nb = ceil_div64(n, SCAN_BLOCK);
    T* sums = nullptr;
    SR_CHECK_HIP(hipMalloc((void**)&sums, sizeof(T) * (size_t)nb));
    hipLaunchKernelGGL(scan_block_sums_kernel<T>, dim3((unsigned)nb), dim3(256), 0, s, d_in, n, sums);
    hipLaunchKernelGGL(scan_sums_kernel<T>, dim3(1), dim3(256), 0, s, sums, nb, write_total ? d_out + n : (T*)nullptr);
    hipLaunchKernelGGL(scan_apply_kernel<T>, dim3((unsigned)nb), dim3(256), 0, s, d_in, d_out, n, sums);
    const hipError_t e = hipGetLastError();
    const hipError_t e2 = hipStreamSynchronize(s);      // sums is freed below
    (void)hipFree(sums);
    SR_CHECK_HIP(e);
    SR_CHECK_HIP(e2);
    return SR_OK;
}

int sr_device_exclusive_scan_i64(const int64_t* d_in, int64_t* d_out, int64_t n, hipStream_t s) {
    return device_exclusive_scan<int64_t>(d_in, d_out, n, true, s);
}

// ------------------------------------------------------------------------------------------- stable radix passes ---
struct RadixArgs {
    const int32_t* term_in;
    const int32_t* row_in;
    const float* val_in;
    int32_t* term_out;
    int32_t* row_out;
    float* val_out;
    int64_t n;
    int by_row;          // digit taken from the doc row instead of the term
    int shift, bits;
    uint32_t* hist;      // [1 << bits][n_waves]
    int64_t n_waves;
    int64_t n_terms;     // range check of the terms (first histogram only)
    int* flags;          // |= 1: a term outside [0, n_terms), |= 2: a negative doc row
    int check;
    int64_t row_limit;       // sort_docs: n_docs (a row at or beyond it would leave the lists not ascending by doc); 0 = rows are payload only
};

__global__ __launch_bounds__(64 * SBW_WAVES) void radix_hist_kernel(RadixArgs a) {
    __shared__ uint32_t cnt[SBW_WAVES][1 << SB_MAXBITS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wid = (int64_t)blockIdx.x * SBW_WAVES + wave;
    const int nbins = 1 << a.bits;
    for (int i = lane; i < nbins; i += 64) cnt[wave][i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (wid >= a.n_waves) return;
    const int32_t* key = a.by_row ? a.row_in : a.term_in;
    const int64_t e0 = wid * SBW_ELEMS;
    const uint32_t mask = (uint32_t)nbins - 1u;
    int bad = 0;
    for (int r = 0; r < SBW_ELEMS / 64; r += 4) {
        int32_t kv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = e0 + (int64_t)(r + u) * 64 + lane;
            kv[u] = j < a.n ? key[j] : -1;
            if (a.check && j < a.n) {
                const int32_t t = a.term_in[j], rw = a.row_in[j];
                if (t < 0 || (int64_t)t >= a.n_terms) bad |= 1;
                if (rw < 0) bad |= 2;
                if (a.row_limit > 0 && (int64_t)rw >= a.row_limit) bad |= 4;      // sort_docs: the doc digits cover [0, n_docs) only
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = e0 + (int64_t)(r + u) * 64 + lane;
            if (j < a.n) atomicAdd(&cnt[wave][((uint32_t)kv[u] >> a.shift) & mask], 1u);
        }
    }
    if (a.check && bad) atomicOr(a.flags, bad);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int i = lane; i < nbins; i += 64) a.hist[(int64_t)i * a.n_waves + wid] = cnt[wave][i];
}

__global__ __launch_bounds__(64 * SBW_WAVES) void radix_scatter_kernel(RadixArgs a) {
    __shared__ uint32_t run[SBW_WAVES][1 << SB_MAXBITS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wid = (int64_t)blockIdx.x * SBW_WAVES + wave;
    if (wid >= a.n_waves) return;
    const int nbins = 1 << a.bits;
    for (int i = lane; i < nbins; i += 64) run[wave][i] = a.hist[(int64_t)i * a.n_waves + wid];     // scanned: first output slot
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int64_t e0 = wid * SBW_ELEMS;
    const uint32_t mask = (uint32_t)nbins - 1u;
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int r = 0; r < SBW_ELEMS / 64; ++r) {
        const int64_t j = e0 + (int64_t)r * 64 + lane;
        if (e0 + (int64_t)r * 64 >= a.n) break;                  // wave-uniform
        const bool live = j < a.n;
        int32_t t = 0, rw = 0;
        float v = 0.f;
        if (live) { t = a.term_in[j]; rw = a.row_in[j]; v = a.val_in[j]; }
        const uint32_t d = live ? (((uint32_t)(a.by_row ? rw : t) >> a.shift) & mask) : 0u;
        // lanes of this round with my digit
        uint64_t same = __ballot(live);
        for (int b = 0; b < a.bits; ++b) {
            const uint64_t bb = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bb : ~bb;
        }
        if (live) {
            const uint32_t first = run[wave][d];                 // every lane of the group reads before its leader writes
            const uint32_t pos = first + (uint32_t)__popcll(same & lt);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if ((same & lt) == 0) run[wave][d] = first + (uint32_t)__popcll(same);
            a.term_out[pos] = t;
            a.row_out[pos] = rw;
            a.val_out[pos] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

__global__ void csr_indptr_kernel(const int32_t* __restrict__ sorted_terms, int64_t n, int64_t n_terms, int64_t* __restrict__ indptr) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_terms) return;
    int64_t lo = 0, hi = n;             // first posting with term >= t
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)sorted_terms[mid] < t) lo = mid + 1; else hi = mid;
    }
    indptr[t] = lo;
}

// term of posting p: the last t with indptr[t] <= p (binary search per posting; empty lists are skipped by the "last")
__global__ void csr_expand_terms_kernel(const int64_t* __restrict__ indptr, int64_t n_terms, int64_t nnz, int32_t* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    int64_t lo = 0, hi = n_terms;       // first t with indptr[t] > p
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (indptr[mid] <= p) lo = mid + 1; else hi = mid;
    }
    out[p] = (int32_t)(lo - 1);
}

extern "C" int sr_sparse_csr_expand_terms(const int64_t* d_indptr, int64_t n_terms, int64_t nnz, int32_t* d_out_terms, sr_stream stream) {
    SR_REQUIRE(n_terms >= 1 && nnz >= 0, "sr_sparse_csr_expand_terms: bad sizes");
    if (nnz == 0) return SR_OK;
    SR_REQUIRE(d_indptr && d_out_terms, "sr_sparse_csr_expand_terms: null pointer");
    hipLaunchKernelGGL(csr_expand_terms_kernel, dim3((unsigned)ceil_div64(nnz, 256)), dim3(256), 0, (hipStream_t)stream, d_indptr, n_terms, nnz, d_out_terms);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

static int bits_for(int64_t n_values) {
    int b = 1;
    while (b < 31 && (1ll << b) < n_values) ++b;
    return b;
}

extern "C" int sr_sparse_csr_build(const int32_t* d_rows, const int32_t* d_cols, const float* d_vals, int64_t nnz, int64_t n_terms,
                                   int64_t n_docs, int sort_docs, int64_t* d_indptr, int32_t* d_out_rows, float* d_out_vals,
                                   sr_stream stream) {
    SR_REQUIRE(nnz >= 0 && nnz < 0xffffffffll, "sr_sparse_csr_build: nnz=%lld outside [0, 2^32)", (long long)nnz);
    SR_REQUIRE(n_terms >= 1 && n_terms < (1ll << 31), "sr_sparse_csr_build: bad n_terms");
    SR_REQUIRE(d_indptr, "sr_sparse_csr_build: null indptr");
    hipStream_t s = (hipStream_t)stream;
    if (nnz == 0) {
        SR_CHECK_HIP(hipMemsetAsync(d_indptr, 0, sizeof(int64_t) * (size_t)(n_terms + 1), s));
        return SR_OK;
    }
    SR_REQUIRE(d_rows && d_cols && d_vals && d_out_rows && d_out_vals, "sr_sparse_csr_build: null pointer");
    SR_REQUIRE(!sort_docs || (n_docs >= 1 && n_docs < (1ll << 31)), "sr_sparse_csr_build: sort_docs needs n_docs in [1, 2^31)");

    // the passes: doc-row digits first (only with sort_docs), then term digits, least significant first
    struct Pass { int by_row, shift, bits; };
    Pass passes[16];
    int np = 0;
    auto add_passes = [&](int by_row, int total_bits) {
        const int n = (total_bits + SB_MAXBITS - 1) / SB_MAXBITS;
        const int per = (total_bits + n - 1) / n;
        for (int i = 0; i < n; ++i) passes[np++] = Pass{by_row, i * per, (i + 1) * per <= total_bits ? per : total_bits - i * per};
    };
    if (sort_docs) add_passes(1, bits_for(n_docs));
    add_passes(0, bits_for(n_terms));

    const int64_t n_waves = ceil_div64(nnz, SBW_ELEMS);
    int32_t *tA = nullptr, *rA = nullptr, *tB = nullptr, *rB = nullptr;
    float *vA = nullptr, *vB = nullptr;
    uint32_t* hist = nullptr;
    int* flags = nullptr;
    int rc = SR_OK;
    auto cleanup = [&]() {
        void* ptrs[] = {tA, rA, vA, tB, rB, vB, hist, flags};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
    };
    // buffer A: output of the odd passes from the end ... simplest ping-pong: the LAST pass writes rows / vals straight into the
    // caller's arrays (terms into tA or tB), the passes before it alternate between A and B
    const size_t e4 = sizeof(int32_t) * (size_t)nnz;
    bool ok = hipMalloc((void**)&tA, e4) == hipSuccess && hipMalloc((void**)&hist, sizeof(uint32_t) * (size_t)n_waves << SB_MAXBITS) == hipSuccess &&
              hipMalloc((void**)&flags, sizeof(int)) == hipSuccess;
    if (ok && np >= 2) ok = hipMalloc((void**)&rA, e4) == hipSuccess && hipMalloc((void**)&vA, e4) == hipSuccess && hipMalloc((void**)&tB, e4) == hipSuccess;
    if (ok && np >= 3) ok = hipMalloc((void**)&rB, e4) == hipSuccess && hipMalloc((void**)&vB, e4) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        cleanup();
        sr_set_error("sr_sparse_csr_build: out of device memory (%lld postings, %d passes)", (long long)nnz, np);
        return SR_ERR_NOMEM;
    }
    do {
        if (hipMemsetAsync(flags, 0, sizeof(int), s) != hipSuccess) { rc = SR_ERR_HIP; break; }
        const int32_t *tin = d_cols, *rin = d_rows;
        const float* vin = d_vals;
        const int32_t* final_terms = nullptr;
        for (int p = 0; p < np && rc == SR_OK; ++p) {
            const bool last = p == np - 1;
            // outputs: the last pass -> (terms: the buffer not being read, rows / vals: the caller's arrays)
            int32_t* tout = (tin == tA) ? tB : tA;
            int32_t* rout = last ? d_out_rows : ((rin == rA) ? rB : rA);
            float* vout = last ? d_out_vals : ((vin == vA) ? vB : vA);
            RadixArgs a;
            a.term_in = tin; a.row_in = rin; a.val_in = vin;
            a.term_out = tout; a.row_out = rout; a.val_out = vout;
            a.n = nnz; a.by_row = passes[p].by_row; a.shift = passes[p].shift; a.bits = passes[p].bits;
            a.hist = hist; a.n_waves = n_waves; a.n_terms = n_terms; a.flags = flags; a.check = p == 0; a.row_limit = sort_docs ? n_docs : 0;
            const unsigned grid = (unsigned)ceil_div64(n_waves, SBW_WAVES);
            hipLaunchKernelGGL(radix_hist_kernel, dim3(grid), dim3(64 * SBW_WAVES), 0, s, a);
            if (hipGetLastError() != hipSuccess) { rc = SR_ERR_HIP; break; }
            rc = device_exclusive_scan<uint32_t>(hist, hist, n_waves << a.bits, false, s);
            if (rc != SR_OK) break;
            if (p == 0) {
                int h = 0;
                if (hipMemcpyAsync(&h, flags, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rc = SR_ERR_HIP; break; }
                if (h) {
                    sr_set_error("sr_sparse_csr_build: invalid postings (%s%s%s)", (h & 1) ? "term outside [0, n_terms); " : "",
                                 (h & 2) ? "negative doc row; " : "", (h & 4) ? "sort_docs with a doc row >= n_docs" : "");
                    rc = SR_ERR_INVALID;
                    break;
                }
            }
            hipLaunchKernelGGL(radix_scatter_kernel, dim3(grid), dim3(64 * SBW_WAVES), 0, s, a);
            if (hipGetLastError() != hipSuccess) { rc = SR_ERR_HIP; break; }
            tin = tout; rin = rout; vin = vout;
            final_terms = tout;
        }
        if (rc != SR_OK) break;
        hipLaunchKernelGGL(csr_indptr_kernel, dim3((unsigned)ceil_div64(n_terms + 1, 256)), dim3(256), 0, s, final_terms, nnz, n_terms, d_indptr);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = SR_ERR_HIP;
    } while (0);
    if (rc == SR_ERR_HIP) sr_set_error("sr_sparse_csr_build: %s", hipGetErrorString(hipGetLastError()));
    cleanup();
    return rc;
}
