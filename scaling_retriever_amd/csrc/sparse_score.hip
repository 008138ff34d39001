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
#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>

#define SP_TILE 8192
#ifndef SPB_Q
#define SPB_Q 4         // queries per workgroup of the query-block kernel (4 or 8), one wave each
#endif
#ifndef SP_SUB
#define SP_SUB 4096     // skip-table granularity (the query-block kernel's tile); SP_TILE is a multiple of it
#endif
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
    const int32_t* skip;  // [n_terms, n_tiles * (SP_TILE / SP_SUB) + 1] offsets relative to indptr[t], one per SP_SUB docs
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
    const uint8_t* q_done;     // optional: queries the query-block kernel handles (skipped here)
    // query-block kernel, wave-owned candidate regions: wave w of the workgroup of (block, sub-tile s) keeps the survivors among ITS
    // 1024 docs of query q in slots [(s - first sub-tile of the launch) * 4096 + 1024 w, + 1024) of q's candidate buffer - as
    // many slots as docs, so no reservation is needed - and their number in seg_cnt[q][4 (s - first) + w]; sparse_gather_kernel
    // packs the regions before the compaction.  Null: one atomic reservation per (wave, query), the round trip of which was
    // half of the filter phase
    int* seg_cnt;
    int seg_n;
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
    if (a.q_done && a.q_done[q]) return;
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
            const int32_t* sk = a.skip + (int64_t)(known ? term : 0) * (a.n_tiles * (SP_TILE / SP_SUB) + 1) + tile * (SP_TILE / SP_SUB);
            const int b = sk[0], e = sk[SP_TILE / SP_SUB];
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

// ---- query-block kernel: 4 queries per workgroup, heavy terms as dense columns in registers ----------------------------
// What bounds sparse_score_kernel is one LDS read-modify-write and 8 B of loads per posting, and two thirds of the postings a
// query touches belong to a few dozen terms that occur in a quarter or more of all docs.  Those terms are ALSO stored as dense
// columns (value per doc, 0 where the doc lacks the term: 4 B per doc, no doc ids), and a workgroup of 4 waves owns
// (4 consecutive queries, sub-tile of SPB_TILE docs); two workgroups share a CU.  A run of consecutive dense terms is applied
// in REGISTERS - thread t owns 16 docs of each query's slice - from coalesced 16-byte column loads that serve all 4 queries.
// The other terms go through the LDS scatter, wave q walking query q's terms on slice q: no barrier between terms at all,
// because the LDS operations of ONE wave execute in program order (term t's writes precede term t + 1's reads) and a slice is
// touched by one wave only.  The slices move registers <-> LDS (with a workgroup barrier) only where the term order switches
// between the two kinds.  Exactness: the plan lists the union of the 4 queries' terms in ascending term id with a weight per
// query (0 = the query lacks the term = skipped), which is each query's own term order when its terms ascend strictly (the plan
// kernel checks; other blocks go to sparse_score_kernel); s + w * 0 == s, so a dense column's zeros change nothing; products and
// sums are unfused.  Per-doc sums are therefore the reference's term-serial fp32 sums, bit for bit.
#define SPB_THREADS (64 * SPB_Q)
#define SPB_TILE SP_SUB
#ifndef SPB_U
#define SPB_U 4       // postings per lane and group
#endif
#ifndef SPB_RING
#define SPB_RING 5    // register sets of a wave's group walk: SPB_RING - 1 groups of loads in flight per wave.  Round 4: the kernel is
#endif                // indifferent to the ring depths (3 .. 6 sets: 222-230 ms per pass) - it is not the loads it waits for; the
                      // shallower rings leave the kernel without scratch (245 VGPRs, no VGPR spill) and are 3 % faster
#define SPB_GROUP (SPB_U * 64)                       // postings per wave and group of the longer runs
#define SPB_TSTRIDE (SPB_TILE + 64)
#define SPB_DESC 256
#define SPB_LIGHT 64      // runs up to this many postings take the one-step path
#ifndef SPB_DRING
#define SPB_DRING 2     // register sets of the dense-column walk
#endif
#ifndef SPB_LRING
#define SPB_LRING 4
#endif
#ifndef SPB_SUBS
#define SPB_SUBS 1     // consecutive sub-tiles per workgroup: the plan fetch and the launch cost amortise, the next sub-tile's
#endif                 // runs are looked up while this one is scored
#define SPB_DV (SPB_TILE / (4 * SPB_THREADS))        // float4 column loads per thread and dense term (4)

struct SparseBlockArgs {
    SparseArgs a;
    const float* dense;          // [n_dense][dense_stride]
    const int32_t* dense_slot;   // [n_terms]: column of a dense term, -1 otherwise
    int64_t dense_stride;        // n_tiles * SP_TILE
    const int32_t* plan_term;    // union term lists, block b at plan_off[b]
    const float* plan_w;         // [.][SPB_Q]
    const int32_t* plan_n;       // entries of block b
    const uint8_t* plan_ok;      // 0: block goes to sparse_score_kernel
    const int64_t* plan_off;
    const int32_t* blk_q;        // this batch's queries in block order (batch-local indices, -1 = none): block j holds blk_q[SPB_Q j ..]
    int64_t blk_base;            // global index of this batch's first block
    int n_sub;                   // sub-tiles of this launch
    unsigned long long* stamps;  // dev switch SR_SPARSE_STAMPS: [7] sums of s_memrealtime ticks (10 ns) per phase over the sampled waves + [7] = waves
    int diag;                    // dev switch SR_SPARSE_DIAG (timing only, wrong results): bit mask of skipped run kinds, 2 = dense, 4 = light scatter, 8 = big scatter
    // work counters (sr_sparse_index_work_counters; off unless enabled): [0] dense columns loaded (one = SPB_TILE floats),
    // [1] (dense column, query) applications (one = SPB_TILE unfused multiply-adds), [2] postings loaded by the one-step runs,
    // [3] postings loaded by the grouped runs, [4] plan entries fetched, [5] (block, sub-tile) workgroups
    unsigned long long* counters;
};

#ifndef SPB_WAVES_PER_SIMD
#define SPB_WAVES_PER_SIMD 2
#endif
__global__ __launch_bounds__(SPB_THREADS, SPB_WAVES_PER_SIMD) void sparse_block_kernel(SparseBlockArgs b) {
#pragma clang fp contract(off)
    extern __shared__ float sc[];              // [SPB_Q][SPB_TSTRIDE]: score slices + one dummy slot per lane
    __shared__ int desc_all[SPB_THREADS / 64][SPB_DESC];    // per wave: (term lane << 8 | group) of the scatter walk
    const SparseArgs& a = b.a;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    const int64_t gblk = b.blk_base + bx;
    // the block's flag, plan offset, entry count, its queries and their tau values are independent loads: issue them together
    // (a return on the flag first would put one more memory latency in front of every workgroup)
    const uint8_t blk_ok = b.plan_ok[gblk];
    const int64_t pe0 = b.plan_off[gblk];
    const int ne = b.plan_n[gblk];
    const int skip_stride = a.n_tiles * (SP_TILE / SP_SUB) + 1;
    const float thr = a.threshold;
    int qls[SPB_Q];            // batch-local query of every slice (the blocks are cut from a work-sorted order), -1 = none
    float tq[SPB_Q];
#pragma unroll
    for (int qi = 0; qi < SPB_Q; ++qi) {
        qls[qi] = b.blk_q[(int64_t)bx * SPB_Q + qi];
        tq[qi] = qls[qi] >= 0 ? a.tau[qls[qi]] : 0.f;
    }
    if (!blk_ok) return;
    // diagnostic: time per phase of every 256th workgroup's waves (0 fetch, 1 dense runs, 2 short runs, 3 longer runs, 4 filter, 5 all)
    const bool st = b.stamps != nullptr && (blockIdx.x & 15) == 0 && (blockIdx.y & 15) == 0;
    unsigned long long st_t[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = st ? __builtin_amdgcn_s_memrealtime() : 0, st_first = st_last;
    auto stamp = [&](int kind) {
        if (st) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            st_t[kind] += now - st_last;
            st_last = now;
        }
    };
    float* const my_slice = sc + wave * SPB_TSTRIDE;           // scatter phase: wave q owns slice q

    // lane j: plan entry j of a batch of 64 (every wave holds the same entries); the next batch's entries - of this sub-tile
    // or the first of the next one - are fetched while this one is applied
    struct Ent { int64_t seg_b; int seg_n, slot; f32x4 w[SPB_Q / 4]; };
    auto fetch = [&](int sub, int e0) -> Ent {
        Ent en;
        en.seg_b = 0; en.seg_n = 0; en.slot = -1;
#pragma unroll
        for (int h = 0; h < SPB_Q / 4; ++h) en.w[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (e0 + lane < ne) {
            const int term = b.plan_term[pe0 + e0 + lane];     // known terms only (plan kernel)
            en.slot = b.dense_slot[term];
#pragma unroll
            for (int h = 0; h < SPB_Q / 4; ++h)
                en.w[h] = *reinterpret_cast<const f32x4*>(b.plan_w + (pe0 + e0 + lane) * SPB_Q + 4 * h);
            const int32_t* sk = a.skip + (int64_t)term * skip_stride + sub;
            const int sb = sk[0], se = sk[1];
            en.seg_n = se - sb;
            en.seg_b = a.indptr[term] + sb;
        }
        return en;
    };
    // SPB_SUBS consecutive sub-tiles of SPB_TILE docs per workgroup
    const int sub_first = a.tile_begin * (SP_TILE / SPB_TILE) + by * SPB_SUBS;
    int sub_end = a.tile_begin * (SP_TILE / SPB_TILE) + b.n_sub;
    if (sub_first + SPB_SUBS < sub_end) sub_end = sub_first + SPB_SUBS;
    if ((int64_t)(sub_end - 1) * SPB_TILE >= a.n_docs) sub_end = (int)((a.n_docs + SPB_TILE - 1) / SPB_TILE);
    if (sub_first >= sub_end) return;
    Ent nxt = fetch(sub_first, 0);
  for (int sub = sub_first; sub < sub_end; ++sub) {
    const int64_t doc0 = (int64_t)sub * SPB_TILE;
    const int n_here = (int)((a.n_docs - doc0) < SPB_TILE ? (a.n_docs - doc0) : SPB_TILE);

    // thread t owns docs 4 t + 1024 i + e (i < SPB_DV, e < 4) of every query while the slices are in registers
    f32x4 acc[SPB_Q][SPB_DV];
#pragma unroll
    for (int q = 0; q < SPB_Q; ++q)
#pragma unroll
        for (int i = 0; i < SPB_DV; ++i) acc[q][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bool in_regs = true;
    // dense columns are addressed as (wave-uniform column base in SGPRs) + (32-bit byte offset of the thread, the same for every
    // column): no per-column 64-bit vector address arithmetic
    const char* const dense_tile = reinterpret_cast<const char*>(b.dense + doc0);
    unsigned dvoff[SPB_DV];
#pragma unroll
    for (int i = 0; i < SPB_DV; ++i) dvoff[i] = 16u * (unsigned)tid + 16u * SPB_THREADS * (unsigned)i;
    auto to_lds = [&]() {
#pragma unroll
        for (int q = 0; q < SPB_Q; ++q)
#pragma unroll
            for (int i = 0; i < SPB_DV; ++i)
                *reinterpret_cast<f32x4*>(sc + q * SPB_TSTRIDE + 4 * SPB_THREADS * i + 4 * tid) = acc[q][i];
        __syncthreads();
    };
    auto to_regs = [&]() {
        __syncthreads();         // every wave is through with its slice
#pragma unroll
        for (int q = 0; q < SPB_Q; ++q)
#pragma unroll
            for (int i = 0; i < SPB_DV; ++i)
                acc[q][i] = *reinterpret_cast<const f32x4*>(sc + q * SPB_TSTRIDE + 4 * SPB_THREADS * i + 4 * tid);
    };

    for (int e0 = 0; e0 < ne; e0 += 64) {
        const Ent en = nxt;
        if (e0 + 64 < ne) nxt = fetch(sub, e0 + 64);
        else if (sub + 1 < sub_end) nxt = fetch(sub + 1, 0);
        const int64_t seg_b = en.seg_b;
        const int seg_n = en.seg_n, slot = en.slot;
        float seg_w[SPB_Q];
#pragma unroll
        for (int q = 0; q < SPB_Q; ++q) seg_w[q] = en.w[q / 4][q % 4];
        float my_w = seg_w[0];
#pragma unroll
        for (int q = 1; q < SPB_Q; ++q) my_w = wave == q ? seg_w[q] : my_w;
        int wmask = 0;
#pragma unroll
        for (int q = 0; q < SPB_Q; ++q) wmask |= seg_w[q] != 0.f ? (1 << q) : 0;
        const uint64_t dmask = __ballot(slot >= 0 && seg_n > 0);
        const uint64_t bmask = __ballot(slot < 0 && seg_n > SPB_LIGHT);                    // scatter runs walked in groups
        const uint64_t lmask = __ballot(slot < 0 && seg_n > 0 && seg_n <= SPB_LIGHT);      // one wave step per run
        uint64_t rem = dmask | bmask | lmask;
        if (b.counters && wave == 0) {
            int nqw = 0;
#pragma unroll
            for (int q = 0; q < SPB_Q; ++q) nqw += seg_w[q] != 0.f ? 1 : 0;
            const bool is_d = slot >= 0 && seg_n > 0, is_l = slot < 0 && seg_n > 0 && seg_n <= SPB_LIGHT, is_b = slot < 0 && seg_n > SPB_LIGHT;
            unsigned long long c1 = is_d ? nqw : 0, c2 = is_l ? (unsigned long long)seg_n * nqw : 0, c3 = is_b ? (unsigned long long)seg_n * nqw : 0;
            for (int off = 32; off > 0; off >>= 1) {
                c1 += __shfl_xor(c1, off); c2 += __shfl_xor(c2, off); c3 += __shfl_xor(c3, off);
            }
            if (lane == 0) {
                atomicAdd(&b.counters[0], (unsigned long long)__builtin_popcountll(dmask));
                atomicAdd(&b.counters[1], c1);
                atomicAdd(&b.counters[2], c2);
                atomicAdd(&b.counters[3], c3);
                atomicAdd(&b.counters[4], (unsigned long long)((ne - e0) < 64 ? (ne - e0) : 64));
                if (e0 == 0) atomicAdd(&b.counters[5], 1ull);
            }
        }
        stamp(0);
        int st_prev = -1;
        while (rem) {
            if (st_prev >= 0) stamp(st_prev);
            const int j_first = __builtin_ctzll(rem);
            const bool dense_run = (dmask >> j_first) & 1;
            const bool light_run = (lmask >> j_first) & 1;
            const uint64_t same = dense_run ? dmask : light_run ? lmask : bmask;
            const uint64_t other = (dmask | bmask | lmask) & ~same & rem;       // all above j_first
            const uint64_t below = other ? ((1ull << __builtin_ctzll(other)) - 1ull) : ~0ull;
            uint64_t run = same & rem & below;
            rem &= ~run;
            if (b.diag & (dense_run ? 2 : light_run ? 4 : 8)) continue;      // 2: dense, 4: light, 8: big runs skipped
            st_prev = dense_run ? 1 : light_run ? 2 : 3;
            if (dense_run) {
                if (!in_regs) { to_regs(); in_regs = true; }
                auto dload = [&](int j, f32x4 (&v)[SPB_DV]) {
                    const char* p = dense_tile + (int64_t)__builtin_amdgcn_readlane(slot, j) * b.dense_stride * 4;
#pragma unroll
                    for (int i = 0; i < SPB_DV; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + dvoff[i]);
                };
                auto dapply = [&](int j, const f32x4 (&v)[SPB_DV]) {
                    const int qm = __builtin_amdgcn_readlane(wmask, j);          // which of the block's queries carry the term
#pragma unroll
                    for (int q = 0; q < SPB_Q; ++q) {
                        if (qm & (1 << q)) {         // wave-uniform: one scalar bit test per (column, query)
                            const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(seg_w[q]), j));
#pragma unroll
                            for (int i = 0; i < SPB_DV; ++i) {
                                const f32x4 prod = v[i] * w;
                                acc[q][i] = acc[q][i] + prod;
                            }
                        }
                    }
                };
                // SPB_DRING register sets: the column loads of the next SPB_DRING - 1 terms fly while one is applied (always
                // issued: past the run's end they re-read the last term)
                f32x4 vD[SPB_DRING][SPB_DV];
                int jD[SPB_DRING];
                bool liveD[SPB_DRING];
                int jn = __builtin_ctzll(run);
                run &= run - 1;
                bool more = true;
#pragma unroll
                for (int s2 = 0; s2 < SPB_DRING - 1; ++s2) {
                    dload(jn, vD[s2]);
                    asm volatile("" ::: "memory");
                    jD[s2] = jn; liveD[s2] = more;
                    more = more && run != 0;
                    if (more) { jn = __builtin_ctzll(run); run &= run - 1; }
                }
                for (bool done = false; !done;) {
#pragma unroll
                    for (int s2 = 0; s2 < SPB_DRING; ++s2) {
                        const int ld = (s2 + SPB_DRING - 1) % SPB_DRING;
                        dload(jn, vD[ld]);
                        asm volatile("" ::: "memory");
                        jD[ld] = jn; liveD[ld] = more;
                        more = more && run != 0;
                        if (more) { jn = __builtin_ctzll(run); run &= run - 1; }
                        if (!liveD[s2]) { done = true; break; }
                        dapply(jD[s2], vD[s2]);
                    }
                }
            } else if (light_run) {
                if (in_regs) { to_lds(); in_regs = false; }
                // This wave's query, runs of at most 64 postings (most of a query's terms): one posting per lane, one load
                // pair and one LDS read-modify-write per term, SPB_LRING - 1 terms of loads in flight.  No barriers.
                uint64_t todo = run & __ballot(my_w != 0.f);
                if (todo == 0) continue;
                auto lload = [&](int j, int& dd, float& vv) {
                    const int64_t sb = readlane64(seg_b, j);
                    const uint32_t last = (uint32_t)__builtin_amdgcn_readlane(seg_n, j) - 1u;
                    const uint32_t pc = (uint32_t)lane < last ? (uint32_t)lane : last;
                    dd = a.doc_ids[sb + pc];
                    vv = a.vals[sb + pc];
                    asm volatile("" ::: "memory");      // see gload below
                };
                auto lapply = [&](int j, int dd, float vv) {
                    const int n = __builtin_amdgcn_readlane(seg_n, j);
                    const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), j));
                    const int d = lane < n ? dd - (int)doc0 : SPB_TILE + lane;
                    const float prod = w * vv;
                    my_slice[d] = my_slice[d] + prod;
                };
                int jL[SPB_LRING], dL[SPB_LRING];
                float vL[SPB_LRING];
                bool liveL[SPB_LRING];
                int jn = __builtin_ctzll(todo);
                todo &= todo - 1;
                bool more = true;
#pragma unroll
                for (int s2 = 0; s2 < SPB_LRING - 1; ++s2) {
                    lload(jn, dL[s2], vL[s2]);
                    jL[s2] = jn; liveL[s2] = more;
                    more = more && todo != 0;
                    if (more) { jn = __builtin_ctzll(todo); todo &= todo - 1; }
                }
                for (bool done = false; !done;) {
#pragma unroll
                    for (int s2 = 0; s2 < SPB_LRING; ++s2) {
                        const int ld = (s2 + SPB_LRING - 1) % SPB_LRING;
                        lload(jn, dL[ld], vL[ld]);
                        jL[ld] = jn; liveL[ld] = more;
                        more = more && todo != 0;
                        if (more) { jn = __builtin_ctzll(todo); todo &= todo - 1; }
                        if (!liveL[s2]) { done = true; break; }
                        lapply(jL[s2], dL[s2], vL[s2]);
                    }
                }
            } else {
                if (in_regs) { to_lds(); in_regs = false; }
                // This wave's query.  Its terms' runs are cut into groups of SPB_GROUP postings; the (term, group) pairs are
                // flattened into per-lane descriptors (lane l = l-th group of the run: first posting, postings left, weight),
                // so that the walk itself is a counted loop over lanes - no per-term cursor, no branches on run lengths:
                // per group 3 v_readlane, SPB_U clamped load pairs (always issued, SPB_RING - 1 groups in flight) and
                // SPB_U LDS read-modify-writes; lanes beyond a run's end use their dummy slot.  No barriers: one wave, one slice.
                const bool has = ((run >> lane) & 1) && my_w != 0.f;
                const int ngr = has ? (seg_n + SPB_GROUP - 1) / SPB_GROUP : 0;
                int incl_g = ngr;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int o = __shfl_up(incl_g, off);
                    if (lane >= off) incl_g += o;
                }
                const int off_j = incl_g - ngr;
                const int G = __builtin_amdgcn_readlane(incl_g, 63);
                int* const desc = desc_all[wave];
                for (int c0 = 0; c0 < G; c0 += SPB_DESC) {
                    for (int i = 0; i < ngr; ++i) {
                        const int pos = off_j + i - c0;
                        if (pos >= 0 && pos < SPB_DESC) desc[pos] = (lane << 8) | i;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const int Gc = (G - c0) < SPB_DESC ? (G - c0) : SPB_DESC;
                    for (int g0 = 0; g0 < Gc; g0 += 64) {
                        const int ng = (Gc - g0) < 64 ? (Gc - g0) : 64;
                        const int e = lane < ng ? desc[g0 + lane] : 0;
                        const int j = e >> 8, gi = e & 255;
                        const int n_j = __shfl(seg_n, j);
                        const int64_t fb = (((int64_t)__shfl((int)(seg_b >> 32), j) << 32) | (uint32_t)__shfl((int)(seg_b & 0xffffffffll), j)) +
                                           (int64_t)gi * SPB_GROUP;
                        const float w_l = __shfl(my_w, j);
                        const int left_l = n_j - gi * SPB_GROUP;
                        auto gload = [&](int l, int (&dd)[SPB_U], float (&vv)[SPB_U]) {
                            const int64_t sb = readlane64(fb, l);
                            const uint32_t last = (uint32_t)__builtin_amdgcn_readlane(left_l, l) - 1u;
                            const int32_t* ib = a.doc_ids + sb;
                            const float* vb = a.vals + sb;
#pragma unroll
                            for (int u = 0; u < SPB_U; ++u) {
                                const uint32_t pp = (uint32_t)lane + 64u * u;
                                const uint32_t pc = pp < last ? pp : last;
                                dd[u] = ib[pc];
                                vv[u] = vb[pc];
                            }
                            // compiler barrier: without it hipcc recognises the re-read of the last group past the end of the
                            // walk, replaces those loads by copies of the earlier registers - and waits for them right there
                            asm volatile("" ::: "memory");
                        };
                        auto gapply = [&](int l, const int (&dd)[SPB_U], const float (&vv)[SPB_U]) {
                            const uint32_t lf = (uint32_t)__builtin_amdgcn_readlane(left_l, l);
                            const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w_l), l));
                            int d[SPB_U];
                            float cur[SPB_U];
#pragma unroll
                            for (int u = 0; u < SPB_U; ++u) {
                                d[u] = ((uint32_t)lane + 64u * u < lf) ? dd[u] - (int)doc0 : SPB_TILE + lane;
                                cur[u] = my_slice[d[u]];
                            }
#pragma unroll
                            for (int u = 0; u < SPB_U; ++u) {
                                const float prod = w * vv[u];
                                my_slice[d[u]] = cur[u] + prod;
                            }
                        };
                        int dR[SPB_RING][SPB_U];
                        float vR[SPB_RING][SPB_U];
#pragma unroll
                        for (int s2 = 0; s2 < SPB_RING - 1; ++s2) gload(s2 < ng ? s2 : ng - 1, dR[s2], vR[s2]);
                        for (int l0 = 0; l0 < ng; l0 += SPB_RING) {
#pragma unroll
                            for (int s2 = 0; s2 < SPB_RING; ++s2) {
                                const int ln = l0 + s2 + SPB_RING - 1;
                                gload(ln < ng ? ln : ng - 1, dR[(s2 + SPB_RING - 1) % SPB_RING], vR[(s2 + SPB_RING - 1) % SPB_RING]);
                                if (l0 + s2 >= ng) break;
                                gapply(l0 + s2, dR[s2], vR[s2]);
                            }
                        }
                    }
                }
            }
        }
        if (st_prev >= 0) stamp(st_prev);
    }
    // ---- filter the 4 slices, from registers: score > threshold and score >= tau ---------------
    if (!in_regs) to_regs();
    stamp(6);            // the barrier in front of the filter: waiting for the slowest wave's scatter walk
    // Wave-level: every wave counts and places the survivors among ITS lanes' docs (the order inside a query's candidate list is
    // free).  Per query one wave scan; the lanes 63 of the waves that keep something reserve their slots with one atomic each -
    // all queries' atomics are in flight together - and nothing waits on a workgroup barrier or on another wave.  (Almost every
    // (block, sub-tile) keeps a few docs per query for most of the scan: the workgroup-wide scan + serial atomics this replaces
    // cost 6 of a workgroup's 30 us.)
    int cnt[SPB_Q], incl[SPB_Q], base[SPB_Q];
    const bool full_tile = n_here == SPB_TILE;
#pragma unroll
    for (int qi = 0; qi < SPB_Q; ++qi) {
        cnt[qi] = 0;
        if (qls[qi] >= 0) {
            // the lane's largest score first: a wave keeps nothing of a query in most (block, sub-tile) pairs once tau has risen,
            // and then the 16 three-way comparisons below are skipped wave-wide
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < SPB_DV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (full_tile || 4 * SPB_THREADS * i + 4 * tid + e < n_here) mx = fmaxf(mx, acc[qi][i][e]);
            if (__ballot(mx > thr && mx >= tq[qi])) {          // wave-uniform
#pragma unroll
                for (int i = 0; i < SPB_DV; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float sv = acc[qi][i][e];
                        cnt[qi] += (4 * SPB_THREADS * i + 4 * tid + e < n_here && sv > thr && sv >= tq[qi]) ? 1 : 0;
                    }
            }
        }
        incl[qi] = cnt[qi];
        if (__ballot(cnt[qi] != 0)) {          // wave-uniform
            for (int off = 1; off < 64; off <<= 1) {
                int o = __shfl_up(incl[qi], off);
                if (lane >= off) incl[qi] += o;
            }
        }
        base[qi] = 0;
        if (a.seg_cnt) {
            const int seg = (sub - a.tile_begin * (SP_TILE / SPB_TILE)) * 4 + wave;
            base[qi] = seg * (SPB_TILE / 4);
            if (lane == 63 && incl[qi] > 0) a.seg_cnt[(int64_t)qls[qi] * a.seg_n + seg] = incl[qi];
        } else if (lane == 63 && incl[qi] > 0) {
            base[qi] = atomicAdd(&a.cand_count[qls[qi]], incl[qi]);
        }
    }
#pragma unroll
    for (int qi = 0; qi < SPB_Q; ++qi) {
        if (__ballot(cnt[qi] != 0) == 0) continue;          // wave-uniform: nothing of this query among this wave's docs
        int pos = (a.seg_cnt ? base[qi] : __builtin_amdgcn_readlane(base[qi], 63)) + incl[qi] - cnt[qi];
        uint64_t* dst = a.cand_keys + (int64_t)qls[qi] * a.cand_cap;
        // per block of 4 docs a wave-uniform skip: the per-element `if (keep) store` code is 16 exec-masked micro-branches per
        // query, run by the whole wave for the sake of the one or two lanes that keep something
#pragma unroll
        for (int i = 0; i < SPB_DV; ++i) {
            const float m4 = fmaxf(fmaxf(acc[qi][i][0], acc[qi][i][1]), fmaxf(acc[qi][i][2], acc[qi][i][3]));
            if (__ballot(cnt[qi] != 0 && m4 > thr && m4 >= tq[qi]) == 0) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sv = acc[qi][i][e];
                const int d = 4 * SPB_THREADS * i + 4 * tid + e;
                if (cnt[qi] != 0 && d < n_here && sv > thr && sv >= tq[qi]) {
                    if (pos < a.cand_cap) dst[pos] = sr_make_key(sv, a.id_base + (uint32_t)(doc0 + d) * a.id_stride);
                    ++pos;
                }
            }
        }
    }
    stamp(4);
    __syncthreads();       // the slices are re-used by the next sub-tile
  }
    if (st && lane == 0) {
        st_t[5] = __builtin_amdgcn_s_memrealtime() - st_first;
        for (int i = 0; i < 7; ++i) atomicAdd(&b.stamps[i], st_t[i]);
        atomicAdd(&b.stamps[7], 1ull);
    }
}

// Packs the wave-owned candidate regions of one launch (SparseArgs::seg_cnt) to the head of every query's candidate buffer, in
// place, and sets cand_count: what topk_compact_kernel expects.  One workgroup per query.  A region's keys move DOWN (the packed
// position of a key is never above its region's start), outputs are produced in ascending rounds of 4096 keys - every round reads
// all its sources, then a barrier, then writes - so a write can only land on sources that were already read.
#define SPG_ROUND 4096
__global__ __launch_bounds__(256) void sparse_gather_kernel(uint64_t* __restrict__ cand_keys, int* __restrict__ cand_count, int64_t cand_cap,
                                                            int* __restrict__ seg_cnt, int seg_n, int region) {
    __shared__ int pre[513];                 // exclusive prefix of the region counts (seg_n <= 512)
    __shared__ int wave_tot[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    int* cnt = seg_cnt + (int64_t)q * seg_n;
    uint64_t* buf = cand_keys + (int64_t)q * cand_cap;
    int c0 = 0, c1 = 0;
    if (2 * tid < seg_n) c0 = cnt[2 * tid];
    if (2 * tid + 1 < seg_n) c1 = cnt[2 * tid + 1];
    const int mine = c0 + c1;
    int v = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off);
        if ((tid & 63) >= off) v += o;
    }
    if ((tid & 63) == 63) wave_tot[tid >> 6] = v;
    __syncthreads();
    int ex = v - mine;
    for (int w = 0; w < (tid >> 6); ++w) ex += wave_tot[w];
    const int total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    pre[2 * tid] = ex;
    pre[2 * tid + 1] = ex + c0;
    if (tid == 255) pre[512] = total;
    if (mine) {
        if (c0) cnt[2 * tid] = 0;
        if (c1) cnt[2 * tid + 1] = 0;
    }
    __syncthreads();
    if (total == 0) return;                  // cand_count stays 0 (reset by the compaction / topk_reset)
    for (int p0 = 0; p0 < total; p0 += SPG_ROUND) {
        uint64_t keep[SPG_ROUND / 256];
#pragma unroll
        for (int i = 0; i < SPG_ROUND / 256; ++i) {
            const int p = p0 + i * 256 + tid;
            keep[i] = 0;
            if (p < total) {
                int lo = 0, hi = 511;        // the region r with pre[r] <= p < pre[r + 1]
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (pre[mid] <= p) lo = mid; else hi = mid - 1;
                }
                keep[i] = buf[(int64_t)lo * region + (p - pre[lo])];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SPG_ROUND / 256; ++i) {
            const int p = p0 + i * 256 + tid;
            if (p < total) buf[p] = keep[i];
        }
        __syncthreads();
    }
    if (tid == 0) cand_count[q] = total;
}

// Order of a batch's queries before it is cut into blocks of SPB_Q.  Wave q of a workgroup walks query q's scatter terms
// alone, so a block takes as long as its heaviest query (consecutive Dev-shaped queries: heaviest / mean = 1.4 for the longer
// runs), and a dense column is loaded for a block if ANY of its queries carries the term (union of 4 consecutive queries: 23
// columns for 9 per query).  Queries are therefore sorted by (scatter-work bucket, set of their rarer dense terms): a block's
// queries then have about the same work and mostly the same columns: 242 -> 235 ms per pass at the MSMARCO shape (work only or
// columns only: 238).  One workgroup per batch (<= SPB_SORT_MAX queries, bitonic sort in LDS; larger batches keep the caller's
// order).  Also lays out the blocks' plan regions (plan_off).  Which queries share a block changes nothing in any query's result.
#define SPB_SORT_MAX 1024
__global__ __launch_bounds__(256) void sparse_block_order_kernel(const int64_t* __restrict__ q_indptr, const int32_t* __restrict__ q_cols,
                                                                 int64_t nq, int64_t q_batch, int pad, int64_t n_terms,
                                                                 const int64_t* __restrict__ indptr, const int32_t* __restrict__ dense_slot,
                                                                 int n_sub, int do_sort, int32_t* __restrict__ perm,
                                                                 int64_t* __restrict__ plan_off, int64_t blocks_per_batch) {
    __shared__ unsigned long long keys[SPB_SORT_MAX];
    __shared__ int red[4];
    const int tid = threadIdx.x;
    const int64_t qb = (int64_t)blockIdx.x * q_batch;
    const int nqb = (int)((nq - qb) < q_batch ? (nq - qb) : q_batch);
    int32_t* pm = perm + (int64_t)blockIdx.x * pad;
    if (do_sort && pad <= SPB_SORT_MAX) {
        unsigned long long mask[SPB_SORT_MAX / 256];
        int work[SPB_SORT_MAX / 256];
        int wmax = 0;
#pragma unroll
        for (int i = 0; i < SPB_SORT_MAX / 256; ++i) {
            const int ql = tid + 256 * i;
            mask[i] = 0;
            work[i] = 0;
            if (ql < nqb) {
                for (int64_t j = q_indptr[qb + ql]; j < q_indptr[qb + ql + 1]; ++j) {
                    const int64_t t = q_cols[j];
                    if (t < 0 || t >= n_terms) continue;
                    const int slot = dense_slot[t];
                    if (slot >= 0) {
                        mask[i] |= 1ull << (slot & 63);
                    } else {                            // expected run per sub-tile: one step when short, groups otherwise
                        const int64_t per = (indptr[t + 1] - indptr[t]) / n_sub;
                        work[i] += per <= SPB_LIGHT ? 1 : 2 * (int)((per + SPB_GROUP - 1) / SPB_GROUP);
                    }
                }
            }
            wmax = work[i] > wmax ? work[i] : wmax;
        }
        for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(wmax, off); wmax = o > wmax ? o : wmax; }
        if ((tid & 63) == 0) red[tid >> 6] = wmax;
        __syncthreads();
        wmax = max(max(red[0], red[1]), max(red[2], red[3]));
#pragma unroll
        for (int i = 0; i < SPB_SORT_MAX / 256; ++i) {
            const int ql = tid + 256 * i;
            // [work bucket: 5 bits][dense slots 8..55, the rarer ones in the high bits: 48 bits][query: 10 bits]; pads sort last
            const unsigned long long bucket = (unsigned long long)((int64_t)work[i] * 31 / (wmax > 0 ? wmax : 1));
            keys[ql] = ql < nqb ? (bucket << 58) | (((mask[i] >> 8) & ((1ull << 48) - 1)) << 10) | (unsigned long long)ql : ~0ull;
        }
        __syncthreads();
        for (int size = 2; size <= SPB_SORT_MAX; size <<= 1)
            for (int j = size >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < SPB_SORT_MAX; i += 256) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const bool up = (i & size) == 0;
                        const unsigned long long x = keys[i], y = keys[ixj];
                        if (up ? (x > y) : (x < y)) { keys[i] = y; keys[ixj] = x; }
                    }
                }
                __syncthreads();
            }
        for (int i = tid; i < pad; i += 256) pm[i] = keys[i] == ~0ull ? -1 : (int32_t)(keys[i] & 1023);
    } else {
        for (int i = tid; i < pad; i += 256) pm[i] = i < nqb ? i : -1;
    }
    __syncthreads();
    if (tid == 0) {       // plan regions in block order: a block's union list is at most the sum of its queries' term counts
        int64_t off = q_indptr[qb] - q_indptr[0];
        for (int64_t j = 0; j < blocks_per_batch; ++j) {
            plan_off[(int64_t)blockIdx.x * blocks_per_batch + j] = off;
            for (int i = 0; i < SPB_Q; ++i) {
                const int ql = pm[j * SPB_Q + i];
                if (ql >= 0) off += q_indptr[qb + ql + 1] - q_indptr[qb + ql];
            }
        }
    }
}

// Plan of a block of SPB_Q queries: the union of their known terms in ascending term id, one weight per query (0 where the
// query lacks the term).  ok = every query of the block lists its terms in strictly ascending id, i.e. the union's order is
// each query's own accumulation order.  One thread per block.
__global__ void sparse_block_plan_kernel(const int64_t* __restrict__ q_indptr, const int32_t* __restrict__ q_cols,
                                         const float* __restrict__ q_vals, int64_t nq, int64_t n_terms, int64_t q_batch, int pad,
                                         int64_t blocks_per_batch, int64_t n_blocks, const int32_t* __restrict__ perm,
                                         const int64_t* __restrict__ plan_off, int32_t* __restrict__ plan_term,
                                         float* __restrict__ plan_w, int32_t* __restrict__ plan_n, uint8_t* __restrict__ plan_ok,
                                         uint8_t* __restrict__ q_done, int* __restrict__ any_bad) {
    const int64_t blk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= n_blocks) return;
    const int64_t bat = blk / blocks_per_batch, j = blk - bat * blocks_per_batch;
    const int64_t qb = bat * q_batch;
    int64_t q[SPB_Q], p[SPB_Q], e[SPB_Q];
    bool ok = true, any = false;
    for (int i = 0; i < SPB_Q; ++i) {
        const int ql = perm[bat * pad + j * SPB_Q + i];
        q[i] = ql >= 0 ? qb + ql : -1;
        if (q[i] >= 0) { p[i] = q_indptr[q[i]]; e[i] = q_indptr[q[i] + 1]; any = true; } else { p[i] = e[i] = 0; }
        for (int64_t c = p[i] + 1; c < e[i]; ++c) ok = ok && q_cols[c] > q_cols[c - 1];
    }
    int n = 0;
    if (ok && any) {
        const int64_t out = plan_off[blk];
        for (;;) {
            int64_t t = INT64_MAX;
            for (int i = 0; i < SPB_Q; ++i)
                if (p[i] < e[i] && (int64_t)q_cols[p[i]] < t) t = q_cols[p[i]];
            if (t == INT64_MAX) break;
            float w[SPB_Q];
            for (int i = 0; i < SPB_Q; ++i) {
                w[i] = 0.f;
                if (p[i] < e[i] && (int64_t)q_cols[p[i]] == t) w[i] = q_vals[p[i]++];
            }
            if (t < 0 || t >= n_terms) continue;   // unknown term: empty posting list
            plan_term[out + n] = (int32_t)t;
            for (int i = 0; i < SPB_Q; ++i) plan_w[(out + n) * SPB_Q + i] = w[i];
            ++n;
        }
    } else if (!ok) {
        atomicOr(any_bad, 1);
    }
    plan_n[blk] = n;
    plan_ok[blk] = (ok && any) ? 1 : 0;
    for (int i = 0; i < SPB_Q; ++i)
        if (q[i] >= 0) q_done[q[i]] = ok ? 1 : 0;
}

// dense column of a heavy term: column[slot][doc] = value (the buffer is zero-filled first)
__global__ void sparse_dense_fill_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ doc_ids,
                                         const float* __restrict__ vals, const int32_t* __restrict__ slot_term,
                                         int64_t stride, float* __restrict__ dense) {
    const int slot = blockIdx.y;
    const int64_t t = slot_term[slot];
    const int64_t b = indptr[t], e = indptr[t + 1];
    float* col = dense + (int64_t)slot * stride;
    for (int64_t pp = b + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pp < e; pp += (int64_t)gridDim.x * blockDim.x)
        col[doc_ids[pp]] = vals[pp];
}

// ---- index build: skip table + validation -----------------------------------
// skip[t][b] = number of postings of term t with doc id < b * SP_SUB  (lower bound); n_sub entries + 1 per term
__global__ void sparse_skip_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ doc_ids,
                                   int64_t n_terms, int n_sub, int32_t* __restrict__ skip) {
    const int64_t t = blockIdx.x;
    const int64_t b = indptr[t], e = indptr[t + 1];
    const int64_t len = e - b;
    const int n_tiles = n_sub;
    for (int tile = threadIdx.x; tile <= n_tiles; tile += blockDim.x) {
        const int64_t bound = (int64_t)tile * SP_SUB;
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

#include "sparse_index.h"

// postings of the batch's query terms inside tiles [tile_begin, tile_begin + n_t): one thread per query term
__global__ void sparse_count_postings_kernel(const int32_t* __restrict__ skip, int n_tiles, int64_t n_terms,
                                             const int64_t* __restrict__ q_indptr,
                                             const int32_t* __restrict__ q_cols, int64_t q_begin, int64_t q_end, int tile_begin,
                                             int n_t, unsigned long long* __restrict__ total) {
    const int64_t t = q_indptr[q_begin] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n = 0;
    if (t < q_indptr[q_end] && q_cols[t] >= 0 && (int64_t)q_cols[t] < n_terms) {
        const int32_t* sk = skip + (int64_t)q_cols[t] * (n_tiles * (SP_TILE / SP_SUB) + 1);
        n = (unsigned long long)(sk[(tile_begin + n_t) * (SP_TILE / SP_SUB)] - sk[tile_begin * (SP_TILE / SP_SUB)]);
    }
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(total, n);
}

static void sparse_free_device(sr_sparse_index* idx) {
    void* ptrs[] = {idx->skip, idx->dense, idx->dense_slot, idx->plan_term, idx->plan_w, idx->plan_n, idx->plan_ok, idx->plan_bad,
                    idx->plan_off, idx->plan_perm, idx->q_done};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    idx->skip = nullptr;
    idx->dense = nullptr;
    idx->dense_slot = nullptr;
    idx->plan_term = nullptr;
    idx->plan_w = nullptr;
    idx->plan_n = nullptr;
    idx->plan_ok = nullptr;
    idx->plan_bad = nullptr;
    idx->plan_off = nullptr;
    idx->plan_perm = nullptr;
    idx->q_done = nullptr;
}

// Terms present in at least 1 / SP_DENSE_DIV of the docs get a dense column: the longest lists first, at most SP_DENSE_MAX of
// them and never more than a quarter of the free device memory (a column costs 4 B per doc, the list it shadows 8 B per
// posting; measured at MSMARCO shape, thresholds of 1/3 .. 1/8 of the docs are within 2 % of each other, 1/2 is 5 % slower).
// No such term: the query-block kernel is not used.
#define SP_DENSE_DIV 4
#define SP_DENSE_MAX 64
static int sparse_build_dense(sr_sparse_index* idx, hipStream_t s) {
    int div = SP_DENSE_DIV, max_slots = SP_DENSE_MAX;
    if (const char* e = sr_dev_getenv("SR_SPARSE_DENSE_DIV")) div = atoi(e);
    if (const char* e = sr_dev_getenv("SR_SPARSE_DENSE_MAX")) max_slots = atoi(e);
    if (div <= 0 || max_slots <= 0) return SR_OK;
    std::vector<int64_t> h_indptr((size_t)idx->n_terms + 1);
    SR_CHECK_HIP(hipMemcpyAsync(h_indptr.data(), idx->indptr, sizeof(int64_t) * h_indptr.size(), hipMemcpyDeviceToHost, s));
    SR_CHECK_HIP(hipStreamSynchronize(s));
    std::vector<std::pair<int64_t, int32_t>> heavy;   // (-length, term)
    for (int64_t t = 0; t < idx->n_terms; ++t) {
        const int64_t len = h_indptr[(size_t)t + 1] - h_indptr[(size_t)t];
        if (len > 0 && len * div >= idx->n_docs) heavy.emplace_back(-len, (int32_t)t);
    }
    if (heavy.empty()) return SR_OK;
    std::sort(heavy.begin(), heavy.end());
    const int64_t stride = (int64_t)idx->n_tiles * SP_TILE;
    size_t free_b = 0, total_b = 0;
    SR_CHECK_HIP(hipMemGetInfo(&free_b, &total_b));
    int64_t fit = (int64_t)(free_b / 4) / (stride * 4);       // at most a quarter of what is free
    int n = (int)heavy.size();
    if (n > max_slots) n = max_slots;
    if (n > fit) n = (int)fit;
    if (n <= 0) return SR_OK;
    std::vector<int32_t> h_slot((size_t)idx->n_terms, -1), h_terms((size_t)n);
    for (int i = 0; i < n; ++i) {
        h_slot[(size_t)heavy[(size_t)i].second] = i;
        h_terms[(size_t)i] = heavy[(size_t)i].second;
    }
    int32_t* d_terms = nullptr;
    if (hipMalloc(&idx->dense, sizeof(float) * (size_t)stride * (size_t)n) != hipSuccess ||
        hipMalloc(&idx->dense_slot, sizeof(int32_t) * (size_t)idx->n_terms) != hipSuccess ||
        hipMalloc(&d_terms, sizeof(int32_t) * (size_t)n) != hipSuccess) {
        (void)hipGetLastError();      // columns are an optimisation: without them the per-query kernel serves every block
        if (idx->dense) (void)hipFree(idx->dense);
        if (idx->dense_slot) (void)hipFree(idx->dense_slot);
        if (d_terms) (void)hipFree(d_terms);
        idx->dense = nullptr;
        idx->dense_slot = nullptr;
        return SR_OK;
    }
    int rc = SR_OK;
    if (hipMemsetAsync(idx->dense, 0, sizeof(float) * (size_t)stride * (size_t)n, s) != hipSuccess ||
        hipMemcpyAsync(idx->dense_slot, h_slot.data(), sizeof(int32_t) * h_slot.size(), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_terms, h_terms.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, s) != hipSuccess) {
        rc = SR_ERR_HIP;
    } else {
        hipLaunchKernelGGL(sparse_dense_fill_kernel, dim3(256, (unsigned)n), dim3(256), 0, s, idx->indptr, idx->doc_ids, idx->vals,
                           d_terms, stride, idx->dense);
        if (hipStreamSynchronize(s) != hipSuccess) rc = SR_ERR_HIP;
    }
    (void)hipFree(d_terms);
    if (rc != SR_OK) {
        sr_set_error("sr_sparse_index_create: building the dense columns failed: %s", hipGetErrorString(hipGetLastError()));
        return rc;
    }
    idx->n_dense = n;
    idx->dense_stride = stride;
    return SR_OK;
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
        const int n_sub = idx->n_tiles * (SP_TILE / SP_SUB);
        if (hipMalloc(&idx->skip, sizeof(int32_t) * (size_t)n_terms * (size_t)(n_sub + 1)) != hipSuccess ||
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
                           n_sub, idx->skip);
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
    if (rc == SR_OK) rc = sparse_build_dense(idx, s);
    if (rc == SR_OK) rc = sparse_cert_build(idx, s);
    if (rc != SR_OK) {
        sparse_free_device(idx);
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
    if (idx->d_stamps) {       // diagnostic: mean microseconds per phase of a sampled wave
        unsigned long long h[8] = {0};
        if (hipMemcpy(h, idx->d_stamps, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h[7])
            fprintf(stderr, "[sparse stamps] waves %llu: fetch %.2f dense %.2f short %.2f long %.2f wait-for-slowest %.2f filter %.2f total %.2f us\n",
                    h[7], h[0] * 0.01 / h[7], h[1] * 0.01 / h[7], h[2] * 0.01 / h[7], h[3] * 0.01 / h[7], h[6] * 0.01 / h[7], h[4] * 0.01 / h[7],
                    h[5] * 0.01 / h[7]);
        (void)hipFree(idx->d_stamps);
    }
    idx->ws.release();
    idx->order.release();
    sparse_cert_destroy(idx->cert);
    idx->cert = nullptr;
    sparse_free_device(idx);
    if (idx->d_postings) (void)hipFree(idx->d_postings);
    if (idx->d_counters) (void)hipFree(idx->d_counters);
    if (idx->seg_cnt) (void)hipFree(idx->seg_cnt);
    delete idx;
    return SR_OK;
}

extern "C" int sr_sparse_index_block_stats(sr_sparse_index* idx, int64_t* n_dense_terms, int64_t* n_block_calls,
                                           int64_t* n_fallback_calls) {
    SR_REQUIRE(idx && n_dense_terms && n_block_calls && n_fallback_calls, "sr_sparse_index_block_stats: null argument");
    std::lock_guard<std::mutex> lock(idx->mu);
    *n_dense_terms = idx->n_dense;
    *n_block_calls = idx->n_block_calls;
    *n_fallback_calls = idx->n_fallback_calls;
    return SR_OK;
}

// Work counters of the query-block kernel (measurement hook, like sr_sparse_index_profile): while enabled every workgroup adds
// what it loads and applies to 6 device counters; the call returns them and resets them.  out[0] dense columns loaded (one =
// 4096 floats), [1] (column, query) applications (one = 4096 unfused multiply-adds), [2] / [3] postings loaded by the one-step /
// grouped scatter runs (8 bytes each, one LDS read-modify-write each), [4] plan entries fetched, [5] workgroup-tiles.
extern "C" int sr_sparse_index_work_counters(sr_sparse_index* idx, int enable, uint64_t* out6) {
    SR_REQUIRE(idx, "sr_sparse_index_work_counters: null index");
    std::lock_guard<std::mutex> lock(idx->mu);
    if (!idx->d_counters) {
        SR_CHECK_HIP(hipMalloc((void**)&idx->d_counters, 6 * 8));
        SR_CHECK_HIP(hipMemset(idx->d_counters, 0, 6 * 8));
    }
    SR_CHECK_HIP(hipDeviceSynchronize());
    if (out6) SR_CHECK_HIP(hipMemcpy(out6, idx->d_counters, 6 * 8, hipMemcpyDeviceToHost));
    SR_CHECK_HIP(hipMemset(idx->d_counters, 0, 6 * 8));
    idx->count_work = enable != 0;
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

// The exact kernels over all docs for the queries of one CSR (the caller holds idx->mu and the stream order).
static int sparse_exact_search(sr_sparse_index* idx, const int64_t* d_q_indptr, const int32_t* d_q_cols,
                               const float* d_q_vals, int64_t nq, int k, float threshold, int64_t id_base,
                               int64_t id_stride, float* d_out_scores, int64_t* d_out_ids, int32_t* d_out_counts,
                               hipStream_t s) {

    // query batches bound the candidate workspace: cap (slots per query) = docs per launch
    int64_t q_batch_max = 1024;
    if (const char* e = sr_dev_getenv("SR_SPARSE_QBATCH")) q_batch_max = atoll(e) >= 4 ? atoll(e) / 4 * 4 : 4;
    const int64_t q_batch = nq < q_batch_max ? nq : q_batch_max;
    int64_t max_tiles = idx->ws_limit / (8 * q_batch * SP_TILE);
    if (max_tiles < 1) max_tiles = 1;
    if (max_tiles > 64) max_tiles = 64;
    if (max_tiles > idx->n_tiles) max_tiles = idx->n_tiles;
    SR_TRY(idx->ws.ensure(q_batch, k, max_tiles * SP_TILE));
    const int seg_cap = (int)(max_tiles * (SP_TILE / SPB_TILE) * 4);          // regions per query and launch (<= 512)
    if (idx->seg_q_cap < q_batch || idx->seg_cap < seg_cap) {
        if (idx->seg_cnt) (void)hipFree(idx->seg_cnt);
        idx->seg_cnt = nullptr;
        SR_CHECK_HIP(hipMalloc((void**)&idx->seg_cnt, sizeof(int) * (size_t)q_batch * (size_t)seg_cap));
        SR_CHECK_HIP(hipMemsetAsync(idx->seg_cnt, 0, sizeof(int) * (size_t)q_batch * (size_t)seg_cap, s));
        idx->seg_q_cap = q_batch;
        idx->seg_cap = seg_cap;
    }

    // query-block path: order every batch's queries, cut them into blocks of SPB_Q and plan the blocks, once per call
    bool use_blocks = idx->n_dense > 0;
    if (const char* e = sr_dev_getenv("SR_SPARSE_BLOCKS")) use_blocks = use_blocks && atoi(e) != 0;
    bool any_fallback = !use_blocks;
    const int64_t n_batches = ceil_div64(nq, q_batch);
    const int64_t blocks_per_batch = ceil_div64(q_batch, SPB_Q);
    const int pad = (int)(blocks_per_batch * SPB_Q);
    const int64_t n_blocks = n_batches * blocks_per_batch;
    if (use_blocks) {
        int64_t h[2];
        SR_CHECK_HIP(hipMemcpyAsync(h, d_q_indptr, 8, hipMemcpyDeviceToHost, s));
        SR_CHECK_HIP(hipMemcpyAsync(h + 1, d_q_indptr + nq, 8, hipMemcpyDeviceToHost, s));
        SR_CHECK_HIP(hipStreamSynchronize(s));
        const int64_t nnz = h[1] - h[0];
        SR_REQUIRE(nnz >= 0, "sr_sparse_search: query indptr not monotone");
        if (nnz > idx->plan_cap || n_blocks > idx->plan_blocks_cap || nq > idx->plan_q_cap) {
            void* old[] = {idx->plan_term, idx->plan_w, idx->plan_n, idx->plan_ok, idx->plan_off, idx->plan_perm, idx->q_done};
            for (void* p : old)
                if (p) (void)hipFree(p);
            idx->plan_term = nullptr; idx->plan_w = nullptr; idx->plan_n = nullptr; idx->plan_ok = nullptr;
            idx->plan_off = nullptr; idx->plan_perm = nullptr; idx->q_done = nullptr;
            idx->plan_cap = idx->plan_blocks_cap = idx->plan_q_cap = 0;
            const int64_t cap = nnz > 0 ? nnz : 1;
            SR_CHECK_HIP(hipMalloc(&idx->plan_term, sizeof(int32_t) * (size_t)cap));
            SR_CHECK_HIP(hipMalloc(&idx->plan_w, sizeof(float) * SPB_Q * (size_t)cap));
            SR_CHECK_HIP(hipMalloc(&idx->plan_n, sizeof(int32_t) * (size_t)n_blocks));
            SR_CHECK_HIP(hipMalloc(&idx->plan_ok, (size_t)n_blocks));
            SR_CHECK_HIP(hipMalloc(&idx->plan_off, sizeof(int64_t) * (size_t)n_blocks));
            SR_CHECK_HIP(hipMalloc(&idx->plan_perm, sizeof(int32_t) * (size_t)n_blocks * SPB_Q));
            SR_CHECK_HIP(hipMalloc(&idx->q_done, (size_t)nq));
            idx->plan_cap = cap;
            idx->plan_blocks_cap = n_blocks;
            idx->plan_q_cap = nq;
        }
        if (!idx->plan_bad) SR_CHECK_HIP(hipMalloc(&idx->plan_bad, sizeof(int)));
        SR_CHECK_HIP(hipMemsetAsync(idx->plan_bad, 0, sizeof(int), s));
        int do_sort = 1;
        if (const char* e = sr_dev_getenv("SR_SPARSE_SORT")) do_sort = atoi(e);       // A/B switch: 0 = blocks of consecutive queries
        hipLaunchKernelGGL(sparse_block_order_kernel, dim3((unsigned)n_batches), dim3(256), 0, s, d_q_indptr, d_q_cols, nq, q_batch, pad,
                           idx->n_terms, idx->indptr, idx->dense_slot, idx->n_tiles * (SP_TILE / SP_SUB), do_sort, idx->plan_perm,
                           idx->plan_off, blocks_per_batch);
        SR_CHECK_LAUNCH();
        hipLaunchKernelGGL(sparse_block_plan_kernel, dim3((unsigned)ceil_div64(n_blocks, 64)), dim3(64), 0, s, d_q_indptr, d_q_cols,
                           d_q_vals, nq, idx->n_terms, q_batch, pad, blocks_per_batch, n_blocks, idx->plan_perm, idx->plan_off,
                           idx->plan_term, idx->plan_w, idx->plan_n, idx->plan_ok, idx->q_done, idx->plan_bad);
        SR_CHECK_LAUNCH();
        int h_bad = 0;
        SR_CHECK_HIP(hipMemcpyAsync(&h_bad, idx->plan_bad, sizeof(int), hipMemcpyDeviceToHost, s));
        SR_CHECK_HIP(hipStreamSynchronize(s));
        any_fallback = h_bad != 0;        // a query whose terms do not ascend: its block goes through the per-query kernel
        static DeviceOnce lds_set;
        if (bool* slot = lds_set.pending()) {
            SR_CHECK_HIP(hipFuncSetAttribute((const void*)sparse_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)(sizeof(float) * SPB_Q * SPB_TSTRIDE)));
            *slot = true;
        }
        if (sr_dev_getenv("SR_SPARSE_STAMPS") && !idx->d_stamps) {
            SR_CHECK_HIP(hipMalloc((void**)&idx->d_stamps, 8 * 8));
            SR_CHECK_HIP(hipMemsetAsync(idx->d_stamps, 0, 8 * 8, s));
        }
        ++idx->n_block_calls;
        if (any_fallback) ++idx->n_fallback_calls;
    }

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
            a.q_done = use_blocks ? idx->q_done : nullptr;
            // wave-owned regions only when every block runs the query-block kernel (the per-query kernel appends through the
            // atomic counter into the same buffer); SR_SPARSE_SEG=0: atomic reservations (A/B)
            bool use_seg = use_blocks && !any_fallback && SPB_Q == 4 && idx->seg_cap <= 512;
            if (const char* e = sr_dev_getenv("SR_SPARSE_SEG")) use_seg = use_seg && atoi(e) != 0;
            a.seg_cnt = use_seg ? idx->seg_cnt : nullptr;
            a.seg_n = idx->seg_cap;
            idx->prof.begin(s);
            if (use_blocks) {
                SparseBlockArgs b;
                b.a = a;
                b.dense = idx->dense;
                b.dense_slot = idx->dense_slot;
                b.dense_stride = idx->dense_stride;
                b.plan_term = idx->plan_term;
                b.plan_w = idx->plan_w;
                b.plan_n = idx->plan_n;
                b.plan_ok = idx->plan_ok;
                b.plan_off = idx->plan_off;
                b.blk_q = idx->plan_perm + (qb / q_batch) * pad;
                b.blk_base = (qb / q_batch) * blocks_per_batch;
                b.diag = 0;
                b.stamps = idx->d_stamps;
                b.counters = idx->count_work ? idx->d_counters : nullptr;
#ifdef SR_DIAG_BUILD        // wrong results by design: read only by a diagnostic build, never by the product .so
                if (const char* e = sr_dev_getenv("SR_SPARSE_DIAG")) b.diag = atoi(e);
#endif
                b.n_sub = (int)(nt * (SP_TILE / SPB_TILE));
                const dim3 grid((unsigned)ceil_div64(nqb, SPB_Q), (unsigned)ceil_div64(b.n_sub, SPB_SUBS));
                hipLaunchKernelGGL(sparse_block_kernel, grid, dim3(SPB_THREADS),
                                   sizeof(float) * SPB_Q * SPB_TSTRIDE, s, b);
                SR_CHECK_LAUNCH();
            }
            if (any_fallback) {
                hipLaunchKernelGGL(sparse_score_kernel, dim3((unsigned)nqb, (unsigned)nt), dim3(256), 0, s, a);
                SR_CHECK_LAUNCH();
            }
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
            if (a.seg_cnt) {
                hipLaunchKernelGGL(sparse_gather_kernel, dim3((unsigned)nqb), dim3(256), 0, s, idx->ws.cand_keys, idx->ws.cand_count,
                                   idx->ws.cand_cap, idx->seg_cnt, idx->seg_cap, SPB_TILE / 4);
                SR_CHECK_LAUNCH();
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

// ---- queries the certified scorer hands back: their rows are gathered into a CSR of their own, scored by the exact kernels and
// scattered into the result rows
__global__ void sparse_sub_gather_kernel(const int64_t* __restrict__ q_indptr, const int32_t* __restrict__ q_cols, const float* __restrict__ q_vals,
                                         const int64_t* __restrict__ sel, const int64_t* __restrict__ sub_indptr, int32_t* __restrict__ sub_cols,
                                         float* __restrict__ sub_vals) {
    const int64_t j = blockIdx.x;
    const int64_t q = sel[j];
    const int64_t b = q_indptr[q], n = q_indptr[q + 1] - b, o = sub_indptr[j];
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        sub_cols[o + i] = q_cols[b + i];
        sub_vals[o + i] = q_vals[b + i];
    }
}
__global__ void sparse_sub_scatter_kernel(const int64_t* __restrict__ sel, int k, const float* __restrict__ s_in, const int64_t* __restrict__ i_in,
                                          const int32_t* __restrict__ c_in, float* __restrict__ s_out, int64_t* __restrict__ i_out,
                                          int32_t* __restrict__ c_out) {
    const int64_t j = blockIdx.x;
    const int64_t q = sel[j];
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        s_out[q * k + i] = s_in[j * k + i];
        i_out[q * k + i] = i_in[j * k + i];
    }
    if (c_out && threadIdx.x == 0) c_out[q] = c_in[j];
}

// wide_band > 0 (and at least SR_CERT_RETRY_MIN queries handed back): the sub-batch first goes through the certified scorer once more with
// that many keys beyond k - a certificate that failed for want of room is then usually given - and only what it hands back again reaches
// the exact kernels.
#define SR_CERT_RETRY_MIN 256
static int sparse_redo_exact(sr_sparse_index* idx, const int64_t* d_q_indptr, const int32_t* d_q_cols, const float* d_q_vals, int64_t nq,
                             int k, float threshold, int64_t id_base, int64_t id_stride, float* d_out_scores, int64_t* d_out_ids,
                             int32_t* d_out_counts, const uint8_t* d_uncert, hipStream_t s, int wide_band = 0) {
    std::vector<uint8_t> h_un((size_t)nq);
    std::vector<int64_t> h_ip((size_t)nq + 1);
    SR_CHECK_HIP(hipMemcpyAsync(h_un.data(), d_uncert, (size_t)nq, hipMemcpyDeviceToHost, s));
    SR_CHECK_HIP(hipMemcpyAsync(h_ip.data(), d_q_indptr, sizeof(int64_t) * ((size_t)nq + 1), hipMemcpyDeviceToHost, s));
    SR_CHECK_HIP(hipStreamSynchronize(s));
    std::vector<int64_t> sel, sub_ip(1, 0);
    for (int64_t q = 0; q < nq; ++q)
        if (h_un[(size_t)q]) {
            sel.push_back(q);
            sub_ip.push_back(sub_ip.back() + (h_ip[(size_t)q + 1] - h_ip[(size_t)q]));
        }
    const int64_t ns = (int64_t)sel.size();
    if (ns == 0) return SR_OK;
    const int64_t nnz = sub_ip.back();
    int64_t *d_sel = nullptr, *d_sip = nullptr, *d_ids = nullptr;
    int32_t *d_cols = nullptr, *d_cnt = nullptr;
    float *d_vals = nullptr, *d_sc = nullptr;
    int rc = SR_OK;
    if (hipMalloc((void**)&d_sel, 8 * (size_t)ns) != hipSuccess || hipMalloc((void**)&d_sip, 8 * ((size_t)ns + 1)) != hipSuccess ||
        hipMalloc((void**)&d_cols, 4 * (size_t)(nnz > 0 ? nnz : 1)) != hipSuccess || hipMalloc((void**)&d_vals, 4 * (size_t)(nnz > 0 ? nnz : 1)) != hipSuccess ||
        hipMalloc((void**)&d_sc, 4 * (size_t)ns * (size_t)k) != hipSuccess || hipMalloc((void**)&d_ids, 8 * (size_t)ns * (size_t)k) != hipSuccess ||
        hipMalloc((void**)&d_cnt, 4 * (size_t)ns) != hipSuccess) {
        sr_set_error("sr_sparse_search: out of device memory for %lld re-done queries", (long long)ns);
        rc = SR_ERR_NOMEM;
    }
    if (rc == SR_OK && (hipMemcpyAsync(d_sel, sel.data(), 8 * (size_t)ns, hipMemcpyHostToDevice, s) != hipSuccess ||
                        hipMemcpyAsync(d_sip, sub_ip.data(), 8 * ((size_t)ns + 1), hipMemcpyHostToDevice, s) != hipSuccess))
        rc = SR_ERR_HIP;
    if (rc == SR_OK) {
        hipLaunchKernelGGL(sparse_sub_gather_kernel, dim3((unsigned)ns), dim3(64), 0, s, d_q_indptr, d_q_cols, d_q_vals, d_sel, d_sip, d_cols, d_vals);
        bool done = false;
        if (wide_band > 0 && ns >= SR_CERT_RETRY_MIN && idx->cert) {
            uint8_t* d_un2 = nullptr;
            if (hipMalloc((void**)&d_un2, (size_t)ns) == hipSuccess) {
                int64_t n_un2 = 0;
                bool no_memory = false;
                int band_used = 0;
                rc = sparse_cert_search(idx, d_sip, d_cols, d_vals, ns, k, threshold, id_base, id_stride, d_sc, d_ids, d_cnt, d_un2, &n_un2, &no_memory,
                                        wide_band, &band_used, s);
                if (rc == SR_OK && !no_memory) {
                    ++idx->n_cert_retries;
                    sparse_cert_count_retry(idx->cert, ns);
                    if (n_un2 > 0)
                        rc = sparse_redo_exact(idx, d_sip, d_cols, d_vals, ns, k, threshold, id_base, id_stride, d_sc, d_ids, d_cnt, d_un2, s);
                    done = true;
                }
                if (hipStreamSynchronize(s) != hipSuccess && rc == SR_OK) rc = SR_ERR_HIP;
                (void)hipFree(d_un2);
            } else {
                (void)hipGetLastError();
            }
        }
        if (rc == SR_OK && !done)
            rc = sparse_exact_search(idx, d_sip, d_cols, d_vals, ns, k, threshold, id_base, id_stride, d_sc, d_ids, d_cnt, s);
    }
    if (rc == SR_OK) {
        hipLaunchKernelGGL(sparse_sub_scatter_kernel, dim3((unsigned)ns), dim3(256), 0, s, d_sel, k, d_sc, d_ids, d_cnt, d_out_scores, d_out_ids,
                           d_out_counts);
        if (hipGetLastError() != hipSuccess) rc = SR_ERR_HIP;
    }
    if (hipStreamSynchronize(s) != hipSuccess && rc == SR_OK) rc = SR_ERR_HIP;      // the temporaries are freed below
    if (rc == SR_ERR_HIP) sr_set_error("sr_sparse_search: re-doing uncertified queries failed: %s", hipGetErrorString(hipGetLastError()));
    void* ptrs[] = {d_sel, d_sip, d_cols, d_vals, d_sc, d_ids, d_cnt};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    return rc;
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
    // certified two-stage scorer (sparse_cert.hip) where the index has one and k leaves room for its band of extra keys; the
    // queries it cannot certify are re-done by the exact kernels.  Dev switch SR_SPARSE_CERT_SEARCH=0: exact kernels only (A/B)
    bool use_cert = idx->cert != nullptr && k + 1024 <= SR_MAX_TOPK;
    {
        const char* forced = sr_dev_getenv("SR_SPARSE_CERT");        // 1: also on collections too small for it to pay (tests)
        if (!(forced && atoi(forced) == 1)) use_cert = use_cert && idx->n_docs >= 8ll * (k + 1024);
    }
    if (const char* e = sr_dev_getenv("SR_SPARSE_CERT_SEARCH")) use_cert = use_cert && atoi(e) != 0;
    if (use_cert) {
        // The scorer's workspace is ~200 KB per query: the query set goes through in batches (a whole MSMARCO-Dev set is one batch), and
        // a batch whose buffers do not fit in device memory is served by the exact kernels instead of failing the call.
        int64_t batch = SR_CERT_QUERY_BATCH;
        if (const char* e = sr_dev_getenv("SR_SPARSE_CERT_BATCH")) batch = std::max<int64_t>(32, atoll(e) / 32 * 32);   // tests: several batches on small inputs
        uint8_t* d_uncert = sparse_cert_uncert_buffer(idx->cert, nq < batch ? nq : batch);
        for (int64_t qb = 0; qb < nq; qb += batch) {
            const int64_t nqb = nq - qb < batch ? nq - qb : batch;
            float* o_s = d_out_scores + qb * k;
            int64_t* o_i = d_out_ids + qb * k;
            int32_t* o_c = d_out_counts ? d_out_counts + qb : nullptr;
            int64_t n_un = 0;
            bool no_memory = d_uncert == nullptr;
            if (sr_dev_getenv("SR_SPARSE_CERT_FAKE_OOM")) no_memory = true;          // tests: the fallback below
            int band_used = 0;
            if (!no_memory)
                SR_TRY(sparse_cert_search(idx, d_q_indptr + qb, d_q_cols, d_q_vals, nqb, k, threshold, id_base, id_stride, o_s, o_i, o_c, d_uncert,
                                          &n_un, &no_memory, 0, &band_used, s));
            if (no_memory) {
                ++idx->n_cert_no_memory;
                SR_TRY(sparse_exact_search(idx, d_q_indptr + qb, d_q_cols, d_q_vals, nqb, k, threshold, id_base, id_stride, o_s, o_i, o_c, s));
            } else if (n_un > 0) {
                // many queries handed back under a band that could still grow: once more through the scorer with the widest band, then the exact kernels
                const int widest = SR_MAX_TOPK - k;                // worth a second pass only if it at least doubles the band
                SR_TRY(sparse_redo_exact(idx, d_q_indptr + qb, d_q_cols, d_q_vals, nqb, k, threshold, id_base, id_stride, o_s, o_i, o_c, d_uncert, s,
                                         2 * band_used <= widest ? widest : 0));
            }
        }
        return SR_OK;
    }
    return sparse_exact_search(idx, d_q_indptr, d_q_cols, d_q_vals, nq, k, threshold, id_base, id_stride, d_out_scores, d_out_ids,
                               d_out_counts, s);
}
