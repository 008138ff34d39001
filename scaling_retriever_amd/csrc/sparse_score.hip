// Inverted-index scoring with LDS-resident score tiles and fused top-k (gfx950).
//
// Replaces SparseRetrieval.numba_score_float + select_topk
// (scaling_retriever/indexer.py:315-344).  The reference zeroes an N-sized fp32
// array per query, scatter-adds q_t * v over each query term's posting list in
// term order, then scans it for score > threshold.  Here the doc space is cut
// into tiles of SP_TILE docs; one workgroup owns (query, tile): its slice of the
// score array lives in LDS, each query term contributes the contiguous run of its
// (doc-sorted) posting list that falls in the tile - found through a per-term
// skip table - and the tile is filtered against max(threshold, tau[q]) straight
// from LDS.  No N-sized array ever touches HBM: traffic = 8 B per touched
// posting.  Terms are applied in the query's term order with a barrier between
// them and an unfused multiply-add, so every per-doc sum is bit-identical to the
// reference's term-serial fp32 accumulation.
#include "common.h"
#include <mutex>

#define SP_TILE 8192
#define SP_TERMS 64   // query terms staged per batch (one wave builds the batch's work list)
#define SP_U 4        // postings per thread and group
#define SP_GROUP (SP_U * 256)                              // postings per group
#ifndef SP_DIAG
#define SP_DIAG 0
#endif
#ifndef SP_RING
#define SP_RING 3     // register sets of the group walk: SP_RING - 1 groups of loads in flight per workgroup
#endif

struct SparseArgs {
    const int64_t* indptr;
    const int32_t* doc_ids;
    const float* vals;
    const int32_t* skip;  // [n_terms, n_tiles + 1] offsets relative to indptr[t]
    int n_tiles;
    int64_t n_docs;
    int64_t n_terms;
    const int64_t* q_indptr;
    const int32_t* q_cols;
    const float* q_vals;
    int64_t q_base;   // first query of this batch
    int tile_begin;   // first tile of this launch
    float threshold;
    const float* tau;     // indexed by (q - q_base)
    uint64_t* cand_keys;
    int* cand_count;
    int64_t cand_cap;
    uint32_t id_base, id_stride;
};

// The postings a (query, tile) workgroup has to apply are cut into groups of SP_GROUP postings of ONE term.  The run of
// every query term inside this tile (skip table) is held in REGISTERS, lane j of every wave = term j of the current batch
// of 64 terms, and read back with v_readlane (wave-uniform j): the group walk needs no LDS bookkeeping at all, so the
// LDS pipe carries nothing but the score read-modify-writes.  The groups are walked in term order with the loads of
// group i + 1 (8 per thread, clamped so that they are always issued) in flight while group i is applied to the LDS score
// tile, and the barrier - an s_barrier behind lgkmcnt(0) only, so that it does not drain those loads - is taken only
// after the last group of a term: postings of one term never share a doc, terms do.  Same per-doc addition order as the
// reference's term-serial loop.
typedef int i32x2_u __attribute__((ext_vector_type(2), aligned(4)));      // 8-byte loads from 4-byte aligned runs
typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));

__device__ inline int64_t readlane64(int64_t v, int j) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v & 0xffffffffll), j);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), j);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

__global__ __launch_bounds__(256) void sparse_score_kernel(SparseArgs a) {
#pragma clang fp contract(off)
    __shared__ float sc[SP_TILE + 64];         // + one dummy slot per lane for the postings beyond a run's end
    __shared__ int wave_tot[4];
    __shared__ int s_base;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int ql = blockIdx.x;                 // query within batch
    const int64_t q = a.q_base + ql;
    const int tile = a.tile_begin + blockIdx.y;
    const int64_t doc0 = (int64_t)tile * SP_TILE;
    const int n_here = (int)((a.n_docs - doc0) < SP_TILE ? (a.n_docs - doc0) : SP_TILE);

    for (int d = tid; d < SP_TILE; d += 256) sc[d] = 0.f;

    const int64_t tb = a.q_indptr[q], te = a.q_indptr[q + 1];
    for (int64_t t0 = tb; t0 < te; t0 += SP_TERMS) {
        const int nt = (int)((te - t0) < SP_TERMS ? (te - t0) : SP_TERMS);
        // lane j: run of term j inside this tile (every wave holds the same 64 entries)
        int64_t seg_b = 0;
        int seg_n = 0;
        float seg_w = 0.f;
        if (lane < nt) {
            // a query term the index does not know has an empty posting list (the reference fills its numba dict with
            // an empty array for every vocabulary id, indexer.py:364-370)
            const int term = a.q_cols[t0 + lane];
            const bool known = term >= 0 && (int64_t)term < a.n_terms;
            const int32_t* sk = a.skip + (int64_t)(known ? term : 0) * (a.n_tiles + 1) + tile;
            const int b = sk[0], e = sk[1];
            seg_n = known ? e - b : 0;
            seg_b = a.indptr[known ? term : 0] + b;
            seg_w = a.q_vals[t0 + lane];
        }
        uint64_t todo = __ballot(seg_n > 0);          // terms with postings here, walked in ascending lane = query order
        if (t0 == tb) __syncthreads();                // zero fill done (later batches: the last term's barrier covers it)
        if (todo == 0) continue;

        // wave-uniform cursor over (term j, group g of that term)
        struct Cur { int j, g, ngr, n; int64_t b; float w; };
        auto first_of = [&](uint64_t& m, Cur& c) {
            c.j = __builtin_ctzll(m);
            m &= m - 1;
            c.g = 0;
            c.n = __builtin_amdgcn_readlane(seg_n, c.j);
            c.ngr = (c.n + SP_GROUP - 1) / SP_GROUP;
            c.b = readlane64(seg_b, c.j);
            c.w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(seg_w), c.j));
        };
        auto advance = [&](uint64_t& m, Cur& c) -> bool {     // false: c was the last group of the batch
            if (c.g + 1 < c.ngr) { ++c.g; return true; }
            if (m == 0) return false;
            first_of(m, c);
            return true;
        };
        // A group is SP_GROUP = 1024 consecutive postings of one term; thread t takes postings t, t + 256, t + 512, t + 768
        // (coalesced 4-byte loads from a wave-uniform base + one 32-bit lane offset, the 1 KB steps in the instruction's
        // immediate; consecutive docs land in consecutive LDS banks).  FULL groups - all of a heavy term's run but its
        // tail - need no bounds handling: per posting one subtract, one shift, the LDS read, the multiply, the add and
        // the LDS write.  The tail group clamps its loads to the run and sends the lanes beyond it to a dummy slot past
        // the tile (no exec-mask branches either way).
        // SP_RING register sets keep SP_RING - 1 groups of loads in flight, also across term boundaries.  Loads are ALWAYS
        // issued (past the last group they re-read it), which keeps hipcc's wait counts static (vmcnt(16..23) in the loop).
        // What bounds the walk (tools/micro/sparse_diag.sh, full MSMARCO shape, one pass of 6 980 queries): the complete
        // kernel 352 ms; posting loads alone (tile untouched) 281 ms; LDS read-modify-writes alone (no loads) 305 ms - the
        // two sides overlap almost completely and each is within 15-20 % of the whole, so neither fewer VALU instructions
        // (40 -> 9 per posting slot), nor a deeper ring (2 / 3 / 4 / 6 sets: 344 - 355 ms), nor dropping the LDS bookkeeping
        // moved it by more than 5 %; skipping the LDS instructions of all-dummy wave steps with wave-uniform branches made it
        // 24 % SLOWER (the reads of a group no longer fly together).
        auto load_group = [&](const Cur& c, int (&dd)[SP_U], float (&vv)[SP_U]) {
            const int32_t* ib = a.doc_ids + c.b;       // wave-uniform
            const float* vb = a.vals + c.b;
            const uint32_t p0 = (uint32_t)c.g * SP_GROUP + (uint32_t)tid;
            // one branch-free form for full and tail groups (a uniform branch around the loads makes hipcc merge its wait
            // counts pessimistically and drain the ring): clamp to the run, lanes beyond it go to their dummy slot
            const uint32_t last = (uint32_t)c.n - 1u;
#pragma unroll
            for (int u = 0; u < SP_U; ++u) {
                const uint32_t p = p0 + 256u * u;
                const uint32_t pc = p < last ? p : last;
#if SP_DIAG == 2      // diagnostic build (tools/micro/sparse_diag.sh): no posting loads, synthetic in-tile doc ids
                dd[u] = (int)doc0 + (int)((pc * 2654435761u) >> 19);
                vv[u] = 1.0f;
#else
                dd[u] = ib[pc];       // raw: nothing may depend on the loaded values before apply_group (a use here would
                vv[u] = vb[pc];       // make the compiler wait for the loads right away and drain the ring)
#endif
            }
        };
        auto apply_group = [&](const int (&dd)[SP_U], const float (&vv)[SP_U], float w, bool last_of_term, uint32_t left) {
            // doc ids are unique inside one posting list: the group's reads can all be in flight before its writes;
            // `left` = postings of the run from this group's first one on: lanes at or beyond it use their dummy slot
            int d[SP_U];
            float cur[SP_U];
#if SP_DIAG == 1          // diagnostic build: loads only, the score tile is not touched (one dummy-slot update per group)
            {
                float acc = 0.f;
#pragma unroll
                for (int u = 0; u < SP_U; ++u) acc += (float)dd[u] * vv[u];
                sc[SP_TILE + lane] += acc * w;
                if (last_of_term) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                return;
            }
#endif
#pragma unroll
            for (int u = 0; u < SP_U; ++u) {
                d[u] = ((uint32_t)tid + 256u * u < left) ? dd[u] - (int)doc0 : SP_TILE + lane;
                cur[u] = sc[d[u]];
            }
#pragma unroll
            for (int u = 0; u < SP_U; ++u) {
                const float prod = w * vv[u];
                sc[d[u]] = cur[u] + prod;
            }
            if (last_of_term)   // term-serial: the next term may touch the same docs
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        // ring of SP_RING register sets: set s holds group i = s (mod SP_RING); while group i is applied, groups
        // i + 1 .. i + SP_RING - 1 are in flight
        int dR[SP_RING][SP_U];
        float vR[SP_RING][SP_U];
        float wR[SP_RING];
        uint32_t leftR[SP_RING];
        bool lastR[SP_RING], liveR[SP_RING];
        Cur head;                                  // cursor of the most recently LOADED group
        first_of(todo, head);
        bool more = true;
#pragma unroll
        for (int s2 = 0; s2 < SP_RING - 1; ++s2) {
            load_group(head, dR[s2], vR[s2]);
            wR[s2] = head.w; lastR[s2] = head.g == head.ngr - 1; liveR[s2] = more; leftR[s2] = (uint32_t)(head.n - head.g * SP_GROUP);
            if (more) { Cur nx = head; more = advance(todo, nx); if (more) head = nx; }
        }
        for (bool done = false; !done;) {
#pragma unroll
            for (int s2 = 0; s2 < SP_RING; ++s2) {
                const int ld = (s2 + SP_RING - 1) % SP_RING;
                load_group(head, dR[ld], vR[ld]);
                wR[ld] = head.w; lastR[ld] = head.g == head.ngr - 1; liveR[ld] = more; leftR[ld] = (uint32_t)(head.n - head.g * SP_GROUP);
                if (more) { Cur nx = head; more = advance(todo, nx); if (more) head = nx; }
                if (!liveR[s2]) { done = true; break; }
                apply_group(dR[s2], vR[s2], wR[s2], lastR[s2], leftR[s2]);
            }
        }
    }
    __syncthreads();

    // ---- filter the tile: score > threshold and score >= tau -----------------
    const float tq = a.tau[ql];
    const float thr = a.threshold;
    int cnt = 0;
    for (int d = tid; d < n_here; d += 256) {
        const float s = sc[d];
        cnt += (s > thr && s >= tq) ? 1 : 0;
    }
    // block exclusive scan of cnt
    int incl = cnt;
    const int wave = tid >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
    for (int w = 0; w < 4; ++w) {
        if (w < wave) wbase += wave_tot[w];
        total += wave_tot[w];
    }
    if (total == 0) return;
    if (tid == 0) s_base = atomicAdd(&a.cand_count[ql], total);
    __syncthreads();
    int pos = s_base + wbase + incl - cnt;
    uint64_t* dst = a.cand_keys + (int64_t)ql * a.cand_cap;
    for (int d = tid; d < n_here; d += 256) {
        const float s = sc[d];
        if (s > thr && s >= tq) {
            if (pos < a.cand_cap) dst[pos] = sr_make_key(s, a.id_base + (uint32_t)(doc0 + d) * a.id_stride);
            ++pos;
        }
    }
}

// ---- index build: skip table + validation -----------------------------------
// skip[t][b] = number of postings of term t with doc id < b * SP_TILE  (lower bound)
__global__ void sparse_skip_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ doc_ids,
                                   int64_t n_terms, int n_tiles, int32_t* __restrict__ skip) {
    const int64_t t = blockIdx.x;
    const int64_t b = indptr[t], e = indptr[t + 1];
    const int64_t len = e - b;
    for (int tile = threadIdx.x; tile <= n_tiles; tile += blockDim.x) {
        const int64_t bound = (int64_t)tile * SP_TILE;
        int64_t lo = 0, hi = len;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)doc_ids[b + mid] < bound) lo = mid + 1; else hi = mid;
        }
        skip[t * (n_tiles + 1) + tile] = (int32_t)lo;
    }
}

// flags[0] |= 1 if a posting list is not strictly ascending, |= 2 if a doc id is out of range,
// |= 4 if indptr is not monotone
__global__ void sparse_validate_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ doc_ids,
                                       int64_t n_terms, int64_t n_docs, int* __restrict__ flags) {
    const int64_t t = blockIdx.x;
    const int64_t b = indptr[t], e = indptr[t + 1];
    if (e < b) { if (threadIdx.x == 0) atomicOr(flags, 4); return; }
    int bad = 0;
    for (int64_t p = b + threadIdx.x; p < e; p += blockDim.x) {
        const int32_t d = doc_ids[p];
        if (d < 0 || (int64_t)d >= n_docs) bad |= 2;
        if (p > b && doc_ids[p - 1] >= d) bad |= 1;
    }
    if (bad) atomicOr(flags, bad);
}

struct sr_sparse_index {
    const int64_t* indptr = nullptr;
    const int32_t* doc_ids = nullptr;
    const float* vals = nullptr;
    int64_t n_terms = 0, n_docs = 0;
    int n_tiles = 0;
    int32_t* skip = nullptr;
    int64_t ws_limit = 4ll << 30;
    TopkWS ws;
    StreamOrder order;
    LaunchProfile prof;
    unsigned long long* d_postings = nullptr;  // device counter of postings touched (profiling only)
    std::mutex mu;
};

// postings of the batch's query terms inside tiles [tile_begin, tile_begin + n_t): one thread per query term
__global__ void sparse_count_postings_kernel(const int32_t* __restrict__ skip, int n_tiles, int64_t n_terms,
                                             const int64_t* __restrict__ q_indptr,
                                             const int32_t* __restrict__ q_cols, int64_t q_begin, int64_t q_end, int tile_begin,
                                             int n_t, unsigned long long* __restrict__ total) {
    const int64_t t = q_indptr[q_begin] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n = 0;
    if (t < q_indptr[q_end] && q_cols[t] >= 0 && (int64_t)q_cols[t] < n_terms) {
        const int32_t* sk = skip + (int64_t)q_cols[t] * (n_tiles + 1);
        n = (unsigned long long)(sk[tile_begin + n_t] - sk[tile_begin]);
    }
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(total, n);
}

extern "C" int sr_sparse_index_create(sr_sparse_index** out, const int64_t* d_indptr, const int32_t* d_doc_ids,
                                      const float* d_vals, int64_t n_terms, int64_t n_docs, sr_stream stream) {
    SR_REQUIRE(out, "sr_sparse_index_create: null out");
    SR_REQUIRE(n_terms >= 1 && n_docs >= 1, "sr_sparse_index_create: need n_terms >= 1 and n_docs >= 1");
    SR_REQUIRE(n_docs < 0xffffffffll, "sr_sparse_index_create: n_docs exceeds 32 bits");
    SR_REQUIRE(d_indptr, "sr_sparse_index_create: null indptr");
    hipStream_t s = (hipStream_t)stream;
    sr_sparse_index* idx = new sr_sparse_index();
    idx->indptr = d_indptr;
    idx->doc_ids = d_doc_ids;
    idx->vals = d_vals;
    idx->n_terms = n_terms;
    idx->n_docs = n_docs;
    idx->n_tiles = (int)ceil_div64(n_docs, SP_TILE);
    int* d_flags = nullptr;
    int h_flags = 0;
    int rc = SR_OK;
    do {
        if (hipMalloc(&idx->skip, sizeof(int32_t) * (size_t)n_terms * (size_t)(idx->n_tiles + 1)) != hipSuccess ||
            hipMalloc(&d_flags, sizeof(int)) != hipSuccess) {
            sr_set_error("sr_sparse_index_create: out of device memory for the skip table (%lld x %d)",
                         (long long)n_terms, idx->n_tiles + 1);
            rc = SR_ERR_NOMEM;
            break;
        }
        if (hipMemsetAsync(d_flags, 0, sizeof(int), s) != hipSuccess) { rc = SR_ERR_HIP; break; }
        hipLaunchKernelGGL(sparse_validate_kernel, dim3((unsigned)n_terms), dim3(256), 0, s, d_indptr, d_doc_ids,
                           n_terms, n_docs, d_flags);
        hipLaunchKernelGGL(sparse_skip_kernel, dim3((unsigned)n_terms), dim3(64), 0, s, d_indptr, d_doc_ids, n_terms,
                           idx->n_tiles, idx->skip);
        if (hipMemcpyAsync(&h_flags, d_flags, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            sr_set_error("sr_sparse_index_create: %s", hipGetErrorString(hipGetLastError()));
            rc = SR_ERR_HIP;
            break;
        }
        if (h_flags) {
            sr_set_error("sr_sparse_index_create: invalid index (%s%s%s)",
                         (h_flags & 1) ? "posting list not strictly ascending by doc id; " : "",
                         (h_flags & 2) ? "doc id out of [0, n_docs); " : "", (h_flags & 4) ? "indptr not monotone" : "");
            rc = SR_ERR_INVALID;
        }
    } while (0);
    if (d_flags) (void)hipFree(d_flags);
    if (rc != SR_OK) {
        if (idx->skip) (void)hipFree(idx->skip);
        delete idx;
        return rc;
    }
    *out = idx;
    return SR_OK;
}

extern "C" int sr_sparse_index_set_workspace_limit(sr_sparse_index* idx, int64_t bytes) {
    SR_REQUIRE(idx && bytes >= (1 << 20), "sr_sparse_index_set_workspace_limit: bad argument");
    idx->ws_limit = bytes;
    return SR_OK;
}

extern "C" int sr_sparse_index_destroy(sr_sparse_index* idx) {
    if (!idx) return SR_OK;
    idx->ws.release();
    idx->order.release();
    if (idx->skip) (void)hipFree(idx->skip);
    if (idx->d_postings) (void)hipFree(idx->d_postings);
    delete idx;
    return SR_OK;
}

extern "C" int sr_sparse_index_profile(sr_sparse_index* idx, int enable) {
    SR_REQUIRE(idx, "sr_sparse_index_profile: null index");
    std::lock_guard<std::mutex> lock(idx->mu);
    if (enable && !idx->d_postings) {
        SR_CHECK_HIP(hipMalloc((void**)&idx->d_postings, 8));
        SR_CHECK_HIP(hipMemset(idx->d_postings, 0, 8));
    }
    idx->prof.enabled = enable != 0;
    return SR_OK;
}

extern "C" int sr_sparse_index_profile_read(sr_sparse_index* idx, int64_t* n_launches, double* total_ms,
                                            double* total_posting_bytes) {
    SR_REQUIRE(idx && n_launches && total_ms && total_posting_bytes, "sr_sparse_index_profile_read: null argument");
    std::lock_guard<std::mutex> lock(idx->mu);
    *n_launches = idx->prof.read(total_ms);
    unsigned long long n = 0;
    if (idx->d_postings) {
        SR_CHECK_HIP(hipMemcpy(&n, idx->d_postings, 8, hipMemcpyDeviceToHost));
        SR_CHECK_HIP(hipMemset(idx->d_postings, 0, 8));
    }
    *total_posting_bytes = 8.0 * (double)n;
    return SR_OK;
}

extern "C" int sr_sparse_search(sr_sparse_index* idx, const int64_t* d_q_indptr, const int32_t* d_q_cols,
                                const float* d_q_vals, int64_t nq, int k, float threshold, int64_t id_base,
                                int64_t id_stride, float* d_out_scores, int64_t* d_out_ids, int32_t* d_out_counts,
                                sr_stream stream) {
    SR_REQUIRE(idx, "sr_sparse_search: null index");
    SR_REQUIRE(nq >= 0 && nq < (1ll << 30), "sr_sparse_search: bad nq");
    SR_REQUIRE(k >= 1 && k <= SR_MAX_TOPK, "sr_sparse_search: k=%d outside [1, %d]", k, SR_MAX_TOPK);
    SR_REQUIRE(id_stride >= 1 && id_base >= 0 && id_base + (idx->n_docs - 1) * id_stride < 0xffffffffll,
               "sr_sparse_search: global doc index exceeds 32 bits");
    if (nq == 0) return SR_OK;
    SR_REQUIRE(d_q_indptr && d_out_scores && d_out_ids, "sr_sparse_search: null pointer");
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(idx->mu);
    StreamOrder::Scope in_order(idx->order, s);

    // query batches bound the candidate workspace: cap (slots per query) = docs per launch
    const int64_t q_batch = nq < 1024 ? nq : 1024;
    int64_t max_tiles = idx->ws_limit / (8 * q_batch * SP_TILE);
    if (max_tiles < 1) max_tiles = 1;
    if (max_tiles > 64) max_tiles = 64;
    if (max_tiles > idx->n_tiles) max_tiles = idx->n_tiles;
    SR_TRY(idx->ws.ensure(q_batch, k, max_tiles * SP_TILE));

    for (int64_t qb = 0; qb < nq; qb += q_batch) {
        const int64_t nqb = (nq - qb) < q_batch ? (nq - qb) : q_batch;
        SR_TRY(topk_reset(idx->ws, nqb, s));
        int64_t step = 1;  // tiles per launch grow geometrically: tau tightens early
        for (int64_t t0 = 0; t0 < idx->n_tiles;) {
            int64_t nt = step < max_tiles ? step : max_tiles;
            if (t0 + nt > idx->n_tiles) nt = idx->n_tiles - t0;
            SparseArgs a;
            a.indptr = idx->indptr;
            a.doc_ids = idx->doc_ids;
            a.vals = idx->vals;
            a.skip = idx->skip;
            a.n_tiles = idx->n_tiles;
            a.n_docs = idx->n_docs;
            a.n_terms = idx->n_terms;
            a.q_indptr = d_q_indptr;
            a.q_cols = d_q_cols;
            a.q_vals = d_q_vals;
            a.q_base = qb;
            a.tile_begin = (int)t0;
            a.threshold = threshold;
            a.tau = idx->ws.tau;
            a.cand_keys = idx->ws.cand_keys;
            a.cand_count = idx->ws.cand_count;
            a.cand_cap = idx->ws.cand_cap;
            a.id_base = (uint32_t)id_base;
            a.id_stride = (uint32_t)id_stride;
            idx->prof.begin(s);
            hipLaunchKernelGGL(sparse_score_kernel, dim3((unsigned)nqb, (unsigned)nt), dim3(256), 0, s, a);
            SR_CHECK_LAUNCH();
            idx->prof.end(s, 0, 0);
            if (idx->prof.enabled && idx->d_postings) {
                int64_t h[2];
                SR_CHECK_HIP(hipMemcpyAsync(h, d_q_indptr + qb, 8, hipMemcpyDeviceToHost, s));
                SR_CHECK_HIP(hipMemcpyAsync(h + 1, d_q_indptr + qb + nqb, 8, hipMemcpyDeviceToHost, s));
                SR_CHECK_HIP(hipStreamSynchronize(s));
                const int64_t nterms = h[1] - h[0];
                if (nterms > 0)
                    hipLaunchKernelGGL(sparse_count_postings_kernel, dim3((unsigned)ceil_div64(nterms, 256)), dim3(256), 0, s,
                                       idx->skip, idx->n_tiles, idx->n_terms, d_q_indptr, d_q_cols, qb, qb + nqb, (int)t0, (int)nt,
                                       idx->d_postings);
            }
            SR_TRY(topk_compact(idx->ws, nqb, k, s));
            t0 += nt;
            step *= 2;
        }
        SR_TRY(topk_finalize(idx->ws, nqb, k, 0.f, d_out_scores + qb * k, d_out_ids + qb * k,
                             d_out_counts ? d_out_counts + qb : nullptr, s));
    }
    return SR_OK;
}
