// On-device inverted-index build (gfx950): doc-major COO postings -> CSR by term.
//
// Replaces IndexDictOfArray.add_batch_document's per-posting Python append (scaling_retriever/utils/inverted_index.py:67-76) and
// the rank-major concatenation of merge_indexes (:108-170).  SparseIndexer.index (indexer.py:239-308) hands over the postings of
// its encoded batches as (global doc row, term, value) triples in insertion order; the reference's posting list of a term holds
// that term's postings in insertion order.  That is a STABLE sort of the triples by term: here a least-significant-digit radix
// sort with digits of up to 9 bits over tiles of 8 192 consecutive postings:
//   pass = histogram kernel (digit counts per tile, digit-major; one wave per tile) -> exclusive scan -> scatter kernel (one workgroup
//   per tile: every wave ranks its 1 024 postings in rounds of 64 - the lanes of a round with the same digit find each other through
//   a 64-bit word per digit in LDS - the waves' counts are scanned in wave order, the tile is put in digit order in LDS and leaves as
//   runs of neighbouring slots; equal digits keep their input order).  radix_scatter_tile_kernel's comment has the measurements that
//   shaped it (round 6: 78 -> 17 ms for the 1.12 G postings of the MSMARCO-sized collection).
// With sort_docs the triples are first sorted by doc row the same way (posting lists ascending by doc id whatever the input
// order: a merged multi-rank index is rank-major inside a term, inverted_index.py:139-146).  indptr = lower bound of every term
// in the sorted term array.  Traffic per pass: 4 B (histogram) + 24 B (scatter) per posting, HBM-bound.
#include "sparse_index.h"
#include <type_traits>

#define SBW_ELEMS 8192        // postings per tile (the unit of the histogram)
#define SBW_WAVES 4           // tiles (one wave each) per workgroup of the histogram kernel
#define SB_MAXBITS 9
#define SBH_COPIES 4

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

// Workgroup b runs on XCD b % 8: every XCD gets one contiguous eighth of the work, so that what is in flight on an XCD is neighbouring
// tiles - their 4-byte histogram entries of a digit (and, in the scatter, the ends of the runs they write) share lines in that XCD's L2.
__device__ __forceinline__ int64_t xcd_contiguous(unsigned b, int64_t n_items) {
    const int64_t per_xcd = (n_items + 7) / 8;
    const int64_t i = (int64_t)(b & 7u) * per_xcd + (b >> 3);
    return (int64_t)(b >> 3) < per_xcd && i < n_items ? i : -1;
}
static unsigned xcd_contiguous_grid(int64_t n_items) { return (unsigned)(((n_items + 7) / 8) * 8); }

// The range check of the first pass (a.check): the histogram kernel checks the array it reads (the pass's key), the scatter kernel the other
// one, which it reads anyway - reading it in the histogram kernel as well was 4.5 GB more at the MSMARCO size.
__device__ __forceinline__ int check_term(const RadixArgs& a, int32_t t) { return (uint32_t)t >= (uint64_t)a.n_terms ? 1 : 0; }      // negative or beyond the vocabulary
__device__ __forceinline__ int check_row(const RadixArgs& a, int32_t rw) {
    return (rw < 0 ? 2 : 0) | (a.row_limit > 0 && (int64_t)rw >= a.row_limit ? 4 : 0);      // 4: sort_docs, the doc digits cover [0, n_docs) only
}
__device__ __forceinline__ int check_key(const RadixArgs& a, int32_t key) { return a.by_row ? check_row(a, key) : check_term(a, key); }
__device__ __forceinline__ int check_other(const RadixArgs& a, int32_t t, int32_t rw) { return a.by_row ? check_term(a, t) : check_row(a, rw); }

__global__ __launch_bounds__(64 * SBW_WAVES) void radix_hist_kernel(RadixArgs a) {
    // SBH_COPIES counters per digit and wave, lane l counts in copy l % SBH_COPIES: the high digits of the terms are skewed (half the postings
    // of a Zipf collection in one digit), and LDS atomics of one instruction on one address are taken one after the other
    __shared__ uint32_t cnt[SBW_WAVES][SBH_COPIES][1 << SB_MAXBITS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t group = xcd_contiguous(blockIdx.x, (a.n_waves + SBW_WAVES - 1) / SBW_WAVES);
    const int64_t wid = group * SBW_WAVES + wave;
    const int nbins = 1 << a.bits;
    for (int i = lane; i < SBH_COPIES << SB_MAXBITS; i += 64) (&cnt[wave][0][0])[i] = 0;
    uint32_t* const mine = cnt[wave][lane % SBH_COPIES];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (group < 0 || wid >= a.n_waves) return;
    const int32_t* key = a.by_row ? a.row_in : a.term_in;
    const int64_t e0 = wid * SBW_ELEMS;
    const uint32_t mask = (uint32_t)nbins - 1u;
    int bad = 0;
    const bool aligned = (reinterpret_cast<uintptr_t>(key) & 15u) == 0;      // a caller's view may start anywhere
    if (e0 + SBW_ELEMS <= a.n && aligned) {
        // a whole tile: 16 bytes per lane and load, four loads in flight (4 KB per wave - with one dword per lane the kernel waited on
        // 1 KB per wave at a time and read at 2 TB/s)
        const uint4* k4 = reinterpret_cast<const uint4*>(key + e0);
        for (int it = 0; it < SBW_ELEMS / (4 * 64 * 4); ++it) {
            uint4 kv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) kv[u] = k4[(it * 4 + u) * 64 + lane];
            if (a.check) {
#pragma unroll
                for (int u = 0; u < 4; ++u) bad |= check_key(a, (int32_t)kv[u].x) | check_key(a, (int32_t)kv[u].y) | check_key(a, (int32_t)kv[u].z) | check_key(a, (int32_t)kv[u].w);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                atomicAdd(&mine[(kv[u].x >> a.shift) & mask], 1u);
                atomicAdd(&mine[(kv[u].y >> a.shift) & mask], 1u);
                atomicAdd(&mine[(kv[u].z >> a.shift) & mask], 1u);
                atomicAdd(&mine[(kv[u].w >> a.shift) & mask], 1u);
            }
        }
    } else {
        for (int r = 0; r < SBW_ELEMS / 64; ++r) {               // the last, partial tile (or input that is not 16-byte aligned)
            const int64_t j = e0 + (int64_t)r * 64 + lane;
            if (j < a.n) {
                const int32_t kv = key[j];
                if (a.check) bad |= check_key(a, kv);
                atomicAdd(&mine[((uint32_t)kv >> a.shift) & mask], 1u);
            }
        }
    }
    if (a.check && bad) atomicOr(a.flags, bad);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int i = lane; i < nbins; i += 64) {
        uint32_t c = 0;
#pragma unroll
        for (int u = 0; u < SBH_COPIES; ++u) c += cnt[wave][u][i];
        a.hist[(int64_t)i * a.n_waves + wid] = c;
    }
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
        if (a.check && live) { const int bad = check_other(a, t, rw); if (bad) atomicOr(a.flags, bad); }
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

// The scatter of a pass with the tile's postings put in digit order in LDS first (round 6).  radix_scatter_kernel above stores a round's 64
// postings straight to their slots: three 4-byte stores per posting into up to 64 different lines per instruction, each output line
// completed by 16 separate stores rounds apart - far more lines in flight than the L2s hold, so lines reach HBM half written again and
// again: 33 ms a pass of 1.12 G postings.  Here one workgroup of 8 waves takes the tile (the 8 192 postings the histogram counted):
//   1. wave w ranks its 1 024 postings among themselves, 16 rounds of 64 (see the loop), the wave's running count per digit in LDS - the
//      stable order inside a wave;
//   2. per digit: the waves' counts are summed in wave order and scanned over the digits: where a (wave, digit) group starts in the tile;
//   3. the output slot of every posting, then its term, row and value go to their place in digit order in ONE 32 KB LDS array and are read
//      back in that order, thread i taking places i, i + 512, ...: neighbouring lanes store to neighbouring slots, a digit's run of
//      16-32 postings leaves as one 64-128-byte piece.
// Same result, posting for posting (a stable sort has only one); tests/test_sparse_csr_build_gpu.py compares both kernels with the stable
// sort of the triples.  Dev switch SR_SPARSE_BUILD_TILE=0: the per-wave kernel.  What the measurements said, per pass of 1.12 G postings
// (tools/micro/csr_build_ab.py, rocprofv3 kernel trace):
//   three LDS arrays at once (96 KB, one workgroup per CU)                        12.7 ms
//   one array at a time (56 KB, two workgroups per CU)                            ~11.5 ms
//   ranking by digit words instead of a ballot per digit bit                      ~10 ms   (skipping all global stores: 3.8 ms - the stores
//                                                                                           were the bound, not the arithmetic)
//   tiles of an XCD contiguous (xcd_contiguous)                                    5.9 ms   = 4.5 TB/s of reads + writes
#define SBT_THREADS 512
// (hipcc reads the second launch bound as waves per SIMD: 4 = two workgroups of 8 waves per CU, at most 128 registers per lane)
__global__ __launch_bounds__(SBT_THREADS, 4) void radix_scatter_tile_kernel(RadixArgs a) {
    constexpr int NW = SBT_THREADS / 64, PER_WAVE = SBW_ELEMS / NW, ROUNDS = PER_WAVE / 64;      // 8 waves x 1 024 postings, 16 rounds
    extern __shared__ __attribute__((aligned(16))) unsigned char sbt_smem[];
    uint32_t* cnt = reinterpret_cast<uint32_t*>(sbt_smem);                  // [NW][nbins]: running counts, then group starts inside the tile
    uint32_t* lstart = cnt + NW * (1 << SB_MAXBITS);                         // [nbins]: where a digit starts inside the tile
    uint32_t* gbase = lstart + (1 << SB_MAXBITS);                            // [nbins]: first output slot of (digit, this tile)
    uint32_t* st = gbase + (1 << SB_MAXBITS);                                // [SBW_ELEMS]: one array of the tile in digit order
    __shared__ uint32_t wtot[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the tiles in flight on an XCD are neighbours: the runs they write for a digit adjoin, and the partly written lines at their ends
    // meet in that XCD's L2 (with tile = blockIdx.x the neighbours sat on 8 different XCDs: about 10 ms a pass of 1.12 G postings, now 5.9)
    const int64_t tile = xcd_contiguous(blockIdx.x, a.n_waves);
    if (tile < 0) return;
    const int nbins = 1 << a.bits;
    const uint32_t mask = (uint32_t)nbins - 1u;
    for (int i = tid; i < NW * nbins; i += SBT_THREADS) cnt[i] = 0;
    for (int i = tid; i < SBW_ELEMS / 4; i += SBT_THREADS) reinterpret_cast<uint4*>(st)[i] = make_uint4(0u, 0u, 0u, 0u);      // the waves' digit words
    for (int i = tid; i < nbins; i += SBT_THREADS) gbase[i] = a.hist[(int64_t)i * a.n_waves + tile];
    __syncthreads();
    const int64_t e0 = tile * SBW_ELEMS + (int64_t)wave * PER_WAVE;
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    int32_t t[ROUNDS], rw[ROUNDS];
    float v[ROUNDS];
    uint32_t dr[ROUNDS];                     // digit | rank inside (wave, digit) << 9; 0xffffffff = past the end
    uint32_t* const mycnt = cnt + wave * nbins;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int64_t j = e0 + (int64_t)r * 64 + lane;
        const bool live = j < a.n;
        t[r] = 0; rw[r] = 0; v[r] = 0.f;
        if (live) { t[r] = a.term_in[j]; rw[r] = a.row_in[j]; v[r] = a.val_in[j]; }
    }
    if (a.check) {
        int bad = 0;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
            if (e0 + (int64_t)r * 64 + lane < a.n) bad |= check_other(a, t[r], rw[r]);
        if (bad) atomicOr(a.flags, bad);
    }
    // the lanes of a round with my digit: every lane ORs its bit into the wave's 64-bit word of that digit (LDS, in the buffer the
    // exchanges use later; an OR does not depend on the order the hardware takes the lanes in), reads the word back, and the group's
    // first lane advances the running count and clears the word.  A wave's LDS operations complete in program order, so the reads see
    // every lane's OR and happen before the clear.  (Until round 6 this was a ballot per digit bit: ~9 VALU instructions per bit and
    // round, which made the kernel instruction-bound - without any global load or store it took as long as with them.)
    static_assert(sizeof(uint32_t) * SBW_ELEMS >= sizeof(unsigned long long) * NW * (1 << SB_MAXBITS), "the digit words of all waves fit the exchange buffer");
    unsigned long long* const mymask = reinterpret_cast<unsigned long long*>(st) + wave * nbins;
    const unsigned long long mybit = 1ull << lane;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const bool live = e0 + (int64_t)r * 64 + lane < a.n;
        const uint32_t d = live ? (((uint32_t)(a.by_row ? rw[r] : t[r]) >> a.shift) & mask) : 0u;
        if (live) atomicOr(&mymask[d], mybit);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        dr[r] = 0xffffffffu;
        if (live) {
            const uint64_t same = mymask[d];
            const uint32_t first = mycnt[d];                     // every lane of the group reads before its leader writes
            dr[r] = d | ((first + (uint32_t)__popcll(same & lt)) << SB_MAXBITS);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if ((same & lt) == 0) { mycnt[d] = first + (uint32_t)__popcll(same); mymask[d] = 0ull; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    __syncthreads();
    // per digit: counts of the waves -> exclusive sums in wave order (kept in cnt), the digit's total -> exclusive scan over the digits
    uint32_t tot = 0;
    if (tid < nbins) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const uint32_t c = cnt[w * nbins + tid];
            cnt[w * nbins + tid] = tot;
            tot += c;
        }
    }
    uint32_t incl = tot;                     // SBT_THREADS >= nbins: thread d owns digit d
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) wbase += w < wave ? wtot[w] : 0u;
    if (tid < nbins) lstart[tid] = wbase + incl - tot;
    __syncthreads();
    // every posting to its place in the tile's digit order - one array at a time through ONE 32 KB buffer (56 KB of LDS with the counters:
    // two workgroups per CU, so one's barriers and loads overlap the other's stores; three buffers at once was one workgroup per CU).
    // First the output slots themselves: the owner of a posting knows its digit, so it computes the slot; the thread that stores place p
    // reads it from there (the slots of a digit's run are consecutive: that is what makes the stores coalesce).
    uint32_t lp[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint32_t d = dr[r] & ((1u << SB_MAXBITS) - 1u);
        lp[r] = dr[r] == 0xffffffffu ? 0xffffffffu : lstart[d] + mycnt[d] + (dr[r] >> SB_MAXBITS);
        if (lp[r] != 0xffffffffu) st[lp[r]] = gbase[d] + mycnt[d] + (dr[r] >> SB_MAXBITS);
    }
    __syncthreads();
    const int64_t left = a.n - tile * SBW_ELEMS;
    const int n_tile = left < SBW_ELEMS ? (int)left : SBW_ELEMS;
    uint32_t slot[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) slot[r] = st[tid + r * SBT_THREADS];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) if (lp[r] != 0xffffffffu) st[lp[r]] = (uint32_t)t[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) if (tid + r * SBT_THREADS < n_tile) a.term_out[slot[r]] = (int32_t)st[tid + r * SBT_THREADS];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) if (lp[r] != 0xffffffffu) st[lp[r]] = (uint32_t)rw[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) if (tid + r * SBT_THREADS < n_tile) a.row_out[slot[r]] = (int32_t)st[tid + r * SBT_THREADS];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) if (lp[r] != 0xffffffffu) st[lp[r]] = __float_as_uint(v[r]);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) if (tid + r * SBT_THREADS < n_tile) a.val_out[slot[r]] = __uint_as_float(st[tid + r * SBT_THREADS]);
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
        // the narrower digits first: the low bits of a key are spread evenly, so a tile's run per digit is shortest there (8 192 / 256 =
        // 32 postings = one 128-byte line with 8 bits, half a line with 9); the high bits are skewed (frequent terms have small ids)
        const int base = total_bits / n, extra = total_bits % n;
        for (int i = 0, sh = 0; i < n; ++i) {
            const int b = base + (i >= n - extra ? 1 : 0);
            passes[np++] = Pass{by_row, sh, b};
            sh += b;
        }
    };
    if (sort_docs) add_passes(1, bits_for(n_docs));
    add_passes(0, bits_for(n_terms));

    const int64_t n_waves = ceil_div64(nnz, SBW_ELEMS);
    // ONE allocation for the ping-pong buffers, the histogram and the flags (six hipMalloc + hipFree pairs of 4.5 GB each were 2 ms of
    // the 18 ms the MSMARCO-sized build takes).  The LAST pass writes rows / vals straight into the caller's arrays (terms into tA or
    // tB), the passes before it alternate between A and B.
    const size_t e4 = (sizeof(int32_t) * (size_t)nnz + 255) & ~(size_t)255;
    const size_t hist_bytes = ((sizeof(uint32_t) * (size_t)n_waves << SB_MAXBITS) + 255) & ~(size_t)255;
    const int n_arrays = np >= 3 ? 6 : np == 2 ? 4 : 1;
    unsigned char* arena = nullptr;
    int rc = SR_OK;
    auto cleanup = [&]() {
        if (arena) (void)hipFree(arena);
    };
    if (hipMalloc((void**)&arena, e4 * n_arrays + hist_bytes + 256) != hipSuccess) {
        (void)hipGetLastError();
        sr_set_error("sr_sparse_csr_build: out of device memory (%lld postings, %d passes)", (long long)nnz, np);
        return SR_ERR_NOMEM;
    }
    int32_t* tA = reinterpret_cast<int32_t*>(arena);
    int32_t* rA = np >= 2 ? reinterpret_cast<int32_t*>(arena + e4) : nullptr;
    float* vA = np >= 2 ? reinterpret_cast<float*>(arena + 2 * e4) : nullptr;
    int32_t* tB = np >= 2 ? reinterpret_cast<int32_t*>(arena + 3 * e4) : nullptr;
    int32_t* rB = np >= 3 ? reinterpret_cast<int32_t*>(arena + 4 * e4) : nullptr;
    float* vB = np >= 3 ? reinterpret_cast<float*>(arena + 5 * e4) : nullptr;
    uint32_t* hist = reinterpret_cast<uint32_t*>(arena + e4 * n_arrays);
    int* flags = reinterpret_cast<int*>(arena + e4 * n_arrays + hist_bytes);
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
            hipLaunchKernelGGL(radix_hist_kernel, dim3(xcd_contiguous_grid(ceil_div64(n_waves, SBW_WAVES))), dim3(64 * SBW_WAVES), 0, s, a);
            if (hipGetLastError() != hipSuccess) { rc = SR_ERR_HIP; break; }
            rc = device_exclusive_scan<uint32_t>(hist, hist, n_waves << a.bits, false, s);
            if (rc != SR_OK) break;
            bool by_tile = true;
            if (const char* e = sr_dev_getenv("SR_SPARSE_BUILD_TILE")) by_tile = atoi(e) != 0;       // A/B switch: 0 = the per-wave scatter
            if (by_tile) {
                constexpr size_t lds = sizeof(uint32_t) * ((SBT_THREADS / 64) + 2) * (1 << SB_MAXBITS) + 4 * (size_t)SBW_ELEMS;
                static DeviceOnce attr_once;
                if (bool* slot = attr_once.pending()) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&radix_scatter_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                        rc = SR_ERR_HIP;
                        break;
                    }
                    *slot = true;
                }
                hipLaunchKernelGGL(radix_scatter_tile_kernel, dim3(xcd_contiguous_grid(n_waves)), dim3(SBT_THREADS), lds, s, a);
            } else {
                hipLaunchKernelGGL(radix_scatter_kernel, dim3(grid), dim3(64 * SBW_WAVES), 0, s, a);
            }
            if (hipGetLastError() != hipSuccess) { rc = SR_ERR_HIP; break; }
            tin = tout; rin = rout; vin = vout;
            final_terms = tout;
        }
        if (rc != SR_OK) break;
        hipLaunchKernelGGL(csr_indptr_kernel, dim3((unsigned)ceil_div64(n_terms + 1, 256)), dim3(256), 0, s, final_terms, nnz, n_terms, d_indptr);
        // the range check of the first histogram is read at the end, with everything else: the digits are masked, so postings that fail
        // it were moved around inside the arrays like any others (no wild write) - the call fails and the outputs are to be discarded
        int h = 0;
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&h, flags, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rc = SR_ERR_HIP; break; }
        if (h) {
            sr_set_error("sr_sparse_csr_build: invalid postings (%s%s%s)", (h & 1) ? "term outside [0, n_terms); " : "",
                         (h & 2) ? "negative doc row; " : "", (h & 4) ? "sort_docs with a doc row >= n_docs" : "");
            rc = SR_ERR_INVALID;
        }
    } while (0);
    if (rc == SR_ERR_HIP) sr_set_error("sr_sparse_csr_build: %s", hipGetErrorString(hipGetLastError()));
    cleanup();
    return rc;
}
