// Shared helpers for libsr_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/sr_hip.h"

void sr_set_error(const char* fmt, ...);

// Development switches (A/B tile plans, kernel variants, diagnostics) are read from the environment ONLY when
// SR_DEV_SWITCHES=1 is set as well: a stray SR_* variable in a production environment cannot change which kernel runs.
// tests/ and tools/ that force a path set both.
#include <stdlib.h>
#include <string.h>
static inline const char* sr_dev_getenv(const char* name) {
    const char* on = getenv("SR_DEV_SWITCHES");
    if (!on || strcmp(on, "1") != 0) return nullptr;
    return getenv(name);
}

#define SR_CHECK_HIP(expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            sr_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return SR_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

// dev switch SR_DEBUG_SYNC=1: every launch is named on stderr and waited for, so that a GPU memory fault (which aborts the
// process without saying where) is attributed to the last line printed
static inline bool sr_debug_sync() {
    static const bool on = [] { const char* e = sr_dev_getenv("SR_DEBUG_SYNC"); return e && atoi(e) != 0; }();
    return on;
}
#define SR_CHECK_LAUNCH()                                                        \
    do {                                                                         \
        SR_CHECK_HIP(hipGetLastError());                                         \
        if (sr_debug_sync()) {                                                   \
            fprintf(stderr, "[launch] %s:%d\n", __FILE__, __LINE__);             \
            fflush(stderr);                                                      \
            SR_CHECK_HIP(hipDeviceSynchronize());                                \
        }                                                                        \
    } while (0)

#define SR_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            sr_set_error(__VA_ARGS__);   \
            return SR_ERR_INVALID;       \
        }                                \
    } while (0)

#define SR_TRY(expr)                 \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != SR_OK) return rc_; \
    } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
// compute units of the current device (256 on MI355X); api.hip
int sr_cu_count();

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: a per-kernel flag array indexed by the CURRENT device, so a
// process that drives several GPUs sets it on each (set twice by racing threads is harmless: same value).
#define SR_MAX_DEVICES 64
struct DeviceOnce {
    bool done[SR_MAX_DEVICES] = {};
    // returns the slot to mark after a successful set, or nullptr if already set on the current device
    bool* pending() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= SR_MAX_DEVICES) d = 0;
        return done[d] ? nullptr : &done[d];
    }
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

// ---- order-preserving float <-> uint32 (larger float => larger uint) ----
__host__ __device__ inline uint32_t sr_f2ord(float f) {
    union { float f; uint32_t u; } c; c.f = f;
    return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}
__host__ __device__ inline float sr_ord2f(uint32_t o) {
    union { float f; uint32_t u; } c;
    c.u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return c.f;
}
// 64-bit candidate key: larger key = better (higher score, then LOWER doc index).
__host__ __device__ inline uint64_t sr_make_key(float score, uint32_t gid) {
    return ((uint64_t)sr_f2ord(score) << 32) | (uint64_t)(~gid);
}
__host__ __device__ inline float sr_key_score(uint64_t k) { return sr_ord2f((uint32_t)(k >> 32)); }
__host__ __device__ inline uint32_t sr_key_gid(uint64_t k) { return ~(uint32_t)(k & 0xffffffffu); }

// ---- bf16 helpers (round-to-nearest-even via hardware cvt at -O3) ----
__device__ inline float bf16_to_f32(unsigned short h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ inline unsigned short f32_to_bf16(float f) {
    // plain cast path keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

// fp32 -> three bf16 planes, x = p0 + p1 + p2 up to 2^-27 |x| (each residual subtraction is exact in fp32)
__device__ inline void split_bf16x3(float v, unsigned short& p0, unsigned short& p1, unsigned short& p2) {
    p0 = f32_to_bf16(v);
    const float r1 = v - bf16_to_f32(p0);
    p1 = f32_to_bf16(r1);
    const float r2 = r1 - bf16_to_f32(p1);
    p2 = f32_to_bf16(r2);
}

// fp32 -> two fp16 planes of v * scale (scale a power of two chosen per row so that the row maximum lands in [2^14, 2^15)):
// v * scale = f0 + f1 up to 2^-22 |v * scale|; values that far below the row maximum lose bits only below 2^-24 of it
__device__ inline void split_f16x2(float vs, unsigned short& f0, unsigned short& f1) {
    const _Float16 h0 = (_Float16)vs;
    const float r = vs - (float)h0;               // exact in fp32
    const _Float16 h1 = (_Float16)r;
    f0 = __builtin_bit_cast(unsigned short, h0);
    f1 = __builtin_bit_cast(unsigned short, h1);
}
// power-of-two scale for a row whose largest magnitude is mx: mx * scale in [2^14, 2^15); 1 for an all-zero row
__device__ inline float row_scale_pow2(float mx) {
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.0f;
    int e;
    (void)frexpf(mx, &e);                          // mx = m * 2^e, m in [0.5, 1)
    int t = 15 - e;
    t = t > 100 ? 100 : (t < -100 ? -100 : t);
    return ldexpf(1.0f, t);
}

// ---- fused top-k workspace shared by the dense and sparse scorers ----
struct TopkWS {
    int64_t nq_cap = 0;      // allocated queries
    int k = 0;               // allocated k
    int64_t cand_cap = 0;    // allocated candidate slots per query
    uint64_t* run_keys = nullptr;   // [nq_cap, 2k]  running set, unsorted: a superset of the top-k, cut back to k when full
    int* run_count = nullptr;       // [nq_cap]
    float* tau = nullptr;           // [nq_cap]     score of the current k-th best (-inf until k kept)
    uint64_t* cand_keys = nullptr;  // [nq_cap, cand_cap]
    int* cand_count = nullptr;      // [nq_cap]
    // Segmented candidate slots (dense_split.hip): the tail [seg_off, seg_off + seg_n * SR_SEG_P) of a query's candidate
    // buffer is cut into seg_n segments of SR_SEG_P slots, one per (doc tile of the launch, producer lane group); a producer
    // with at most SR_SEG_P survivors stores them in ITS segment and its count in seg_cnt - no atomic, no memory round trip -
    // and topk_compact_kernel gathers the segments behind the atomically appended candidates.  seg_n = 0: not in use.
    unsigned char* seg_cnt = nullptr;   // [nq_cap, seg_n]
    int64_t seg_off = 0;
    int seg_n = 0;
    int ensure(int64_t nq, int k, int64_t cand_cap);
    int ensure_segments(int64_t nq, int k, int64_t dense_cap, int seg_n);     // cand_cap = dense_cap + seg_n * SR_SEG_P
    void release();
};
#define SR_SEG_P 4          // slots per segment
#define SR_SEG_PROD 8       // producer lane groups per (query, 256-doc tile) of dense_split_kernel
// tau := -inf, counts := 0 for the first nq queries
int topk_reset(TopkWS& ws, int64_t nq, hipStream_t s);
// merge candidates into the running top-k (per query), update tau, clear candidates
int topk_compact(TopkWS& ws, int64_t nq, int k, hipStream_t s);
// the same, and where a select runs also d_tau2[q] := the score of the k2-th best key of the union (k2 < k; 0 / nullptr: off); d_tau2 is never lowered
// by the caller's protocol: a query whose compaction only appends keeps its previous value
// select_over in [k, 2 k]: the running set is cut back to k (and tau, tau2 refreshed) once it holds more keys than this - 2 k fills the
// slots before it pays for a select, k selects in every call that brought a candidate
int topk_compact2(TopkWS& ws, int64_t nq, int k, int k2, float* d_tau2, int select_over, hipStream_t s);
// sort the running top-k descending and write [nq, k] outputs (+ optional counts)
int topk_finalize(TopkWS& ws, int64_t nq, int k, float pad_score, float* d_out_scores,
                  int64_t* d_out_ids, int32_t* d_out_counts, hipStream_t s);

#define SR_MAX_TOPK 4096

// ---- per-launch HIP event log (measurement hook of the search handles) ----
#include <vector>
// A search handle owns ONE top-k workspace.  The handle's mutex serialises the host side of concurrent calls; this
// chains their device side when they arrive on different streams: the next call's stream waits for an event the
// previous call recorded after its last kernel.
struct StreamOrder {
    hipEvent_t ev = nullptr;
    hipStream_t last = nullptr;
    bool used = false;
    struct Scope {
        StreamOrder& o;
        hipStream_t s;
        Scope(StreamOrder& o_, hipStream_t s_) : o(o_), s(s_) {
            if (o.used && o.last != s && o.ev) (void)hipStreamWaitEvent(s, o.ev, 0);
        }
        ~Scope() {
            if (!o.ev && hipEventCreateWithFlags(&o.ev, hipEventDisableTiming) != hipSuccess) { o.ev = nullptr; return; }
            if (hipEventRecord(o.ev, s) == hipSuccess) { o.last = s; o.used = true; }
        }
    };
    void release() { if (ev) { (void)hipEventDestroy(ev); ev = nullptr; } }
};

struct LaunchProfile {
    bool enabled = false;
    std::vector<hipEvent_t> ev;   // pairs: start, stop
    double flop = 0, bytes = 0;
    void begin(hipStream_t s) {
        if (!enabled) return;
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        ev.push_back(a); ev.push_back(b);
        (void)hipEventRecord(a, s);
    }
    void end(hipStream_t s, double f, double by) {
        if (!enabled || ev.empty()) return;
        (void)hipEventRecord(ev.back(), s);
        flop += f; bytes += by;
    }
    // returns launches; synchronises
    int64_t read(double* total_ms) {
        double ms = 0;
        for (size_t i = 0; i + 1 < ev.size(); i += 2) {
            float t = 0;
            (void)hipEventSynchronize(ev[i + 1]);
            if (hipEventElapsedTime(&t, ev[i], ev[i + 1]) == hipSuccess) ms += t;
            (void)hipEventDestroy(ev[i]); (void)hipEventDestroy(ev[i + 1]);
        }
        const int64_t n = (int64_t)(ev.size() / 2);
        ev.clear();
        *total_ms = ms;
        return n;
    }
};
