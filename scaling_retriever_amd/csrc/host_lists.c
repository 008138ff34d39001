// _host_lists: the list-of-lists of db ids that DenseFlatIndexer.search_knn returns (/root/reference/scaling_retriever/indexer.py:
// 210-214: `[[self.index_id_to_db_id[i] for i in query_top_idxs] for query_top_idxs in indexes]`), built from the [nq, k] array of
// index positions without walking the id objects in hit order.
//
// A Dev-sized result holds 7 M references to id objects scattered over a 500 MB heap of 4 KB pages: taking them in hit order costs
// one TLB miss + one cache miss per reference (65-110 ns each on the GPU boxes' hosts: 0.45-0.8 s, more than the GPU needs for the
// search).  Here the hits are radix-sorted by index position first (no GIL held), the objects are then visited in index = allocation
// order (every page once, hardware prefetch), each reference is counted there and written to a stream per block of 256 queries, and
// the streams are finally scattered into the lists' item arrays, 2 MB at a time.
//
// CPython extension (the lists are Python objects); plain C, no numpy C API: arrays arrive as addresses (numpy `.ctypes.data`), kept
// alive by the caller for the duration of the call.
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <pthread.h>
#include <time.h>
#include <unistd.h>
typedef struct { uint32_t key, pos; } Pair;
typedef struct { uint32_t pos; uint32_t pad; PyObject* obj; } Hit;

// Scratch (two pair arrays + the hit streams: 32 B per hit) is kept between calls and only grows: a fresh 220 MB allocation per
// Dev-sized call costs more in first-touch page faults than the sort.  Guarded by the GIL (taken and released with it held).
static void* g_scratch = NULL;
static size_t g_scratch_bytes = 0;
static int g_scratch_busy = 0;
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

#define RELEASE_SCRATCH() do { if (cached) g_scratch_busy = 0; else free(scratch); } while (0)
#define RADIX_BITS 11
#define RADIX (1 << RADIX_BITS)
#define BLOCK_Q 256          // queries per final scatter block: 256 x k x 8 B of item arrays stay cache- and TLB-resident

static Pair* radix_sort_pairs_serial(Pair* a, Pair* b, size_t n, int bits) {
    size_t* count = (size_t*)malloc(sizeof(size_t) * RADIX);
    if (!count) return NULL;
    for (int shift = 0; shift < bits; shift += RADIX_BITS) {
        memset(count, 0, sizeof(size_t) * RADIX);
        for (size_t i = 0; i < n; ++i) ++count[(a[i].key >> shift) & (RADIX - 1)];
        size_t run = 0;
        for (int d = 0; d < RADIX; ++d) { const size_t c = count[d]; count[d] = run; run += c; }
        for (size_t i = 0; i < n; ++i) b[count[(a[i].key >> shift) & (RADIX - 1)]++] = a[i];
        Pair* t = a; a = b; b = t;
    }
    free(count);
    return a;
}

// LSD radix sort of `n` pairs by key (< 2^bits), stable, on up to SORT_THREADS threads (the GIL is not held here): per pass every
// thread counts the digits of its slice, the (digit, thread) counts are prefix-summed into write positions, every thread scatters its
// slice.  The sorted pairs end in the returned buffer (a or b).
#define SORT_THREADS 8
typedef struct {
    const Pair* src; Pair* dst; size_t lo, hi; int shift; size_t* count;          // count: this thread's RADIX counters / positions
    pthread_barrier_t* bar; int phase_count;
} SortJob;
static void sort_count(SortJob* j) {
    memset(j->count, 0, sizeof(size_t) * RADIX);
    for (size_t i = j->lo; i < j->hi; ++i) ++j->count[(j->src[i].key >> j->shift) & (RADIX - 1)];
}
static void sort_scatter(SortJob* j) {
    for (size_t i = j->lo; i < j->hi; ++i) j->dst[j->count[(j->src[i].key >> j->shift) & (RADIX - 1)]++] = j->src[i];
}
typedef struct { pthread_mutex_t mu; pthread_cond_t cv; int go; } SortStart;          // go: 0 wait, 1 run, -1 leave (a thread failed to start)
typedef struct { SortJob* jobs; int t, nt, bits; Pair* a; Pair* b; size_t n; pthread_barrier_t* bar; SortStart* start; } SortThread;
static void* sort_thread(void* arg) {
    SortThread* st = (SortThread*)arg;
    if (st->t != 0) {
        pthread_mutex_lock(&st->start->mu);
        while (st->start->go == 0) pthread_cond_wait(&st->start->cv, &st->start->mu);
        const int go = st->start->go;
        pthread_mutex_unlock(&st->start->mu);
        if (go < 0) return NULL;
    }
    SortJob* j = &st->jobs[st->t];
    Pair* a = st->a; Pair* b = st->b;
    for (int shift = 0; shift < st->bits; shift += RADIX_BITS) {
        j->src = a; j->dst = b; j->shift = shift;
        sort_count(j);
        pthread_barrier_wait(st->bar);
        if (st->t == 0) {          // positions: digit-major, thread-minor (stable)
            size_t run = 0;
            for (int d = 0; d < RADIX; ++d)
                for (int t = 0; t < st->nt; ++t) { const size_t c = st->jobs[t].count[d]; st->jobs[t].count[d] = run; run += c; }
        }
        pthread_barrier_wait(st->bar);
        sort_scatter(j);
        pthread_barrier_wait(st->bar);
        Pair* t = a; a = b; b = t;
    }
    return NULL;
}
static Pair* radix_sort_pairs(Pair* a, Pair* b, size_t n, int bits) {
    int nt = n < ((size_t)1 << 18) ? 1 : SORT_THREADS;
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    if (ncpu > 0 && nt > ncpu) nt = (int)ncpu;
    size_t* counts = (size_t*)malloc(sizeof(size_t) * RADIX * (size_t)nt);
    SortJob* jobs = (SortJob*)calloc((size_t)nt, sizeof(SortJob));
    SortThread* sts = (SortThread*)calloc((size_t)nt, sizeof(SortThread));
    pthread_t* th = (pthread_t*)calloc((size_t)nt, sizeof(pthread_t));
    pthread_barrier_t bar;
    if (!counts || !jobs || !sts || !th || pthread_barrier_init(&bar, NULL, (unsigned)nt)) {
        free(counts); free(jobs); free(sts); free(th);
        return NULL;
    }
    SortStart start;
    pthread_mutex_init(&start.mu, NULL);
    pthread_cond_init(&start.cv, NULL);
    start.go = 0;
    int started = 0;
    for (int t = 0; t < nt; ++t) {
        jobs[t].lo = n * (size_t)t / (size_t)nt; jobs[t].hi = n * (size_t)(t + 1) / (size_t)nt; jobs[t].count = counts + (size_t)t * RADIX;
        sts[t].jobs = jobs; sts[t].t = t; sts[t].nt = nt; sts[t].bits = bits; sts[t].a = a; sts[t].b = b; sts[t].n = n; sts[t].bar = &bar;
        sts[t].start = &start;
    }
    for (int t = 1; t < nt; ++t) {
        if (pthread_create(&th[t], NULL, sort_thread, &sts[t])) break;
        ++started;
    }
    const int all = started == nt - 1;          // the helpers wait for `go`: if one could not be started the others leave again
    pthread_mutex_lock(&start.mu);
    start.go = all ? 1 : -1;
    pthread_cond_broadcast(&start.cv);
    pthread_mutex_unlock(&start.mu);
    if (all) sort_thread(&sts[0]);
    for (int t = 1; t <= started; ++t) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&bar);
    pthread_cond_destroy(&start.cv);
    pthread_mutex_destroy(&start.mu);
    free(counts); free(jobs); free(sts); free(th);
    if (!all) return nt > 1 ? radix_sort_pairs_serial(a, b, n, bits) : NULL;
    int passes = 0;
    for (int shift = 0; shift < bits; shift += RADIX_BITS) ++passes;
    return (passes & 1) ? b : a;
}

// take_rows(table_addr, n_table, idx_addr, nq, k) -> list of nq lists of k objects
//   table_addr: address of PyObject*[n_table + 1] (a numpy object array's data; entry n_table is what position -1 maps to)
//   idx_addr:   address of int64[nq * k], C order; values in [-1, n_table)
static PyObject* take_rows(PyObject* self, PyObject* args) {
    Py_ssize_t table_addr, n_table, idx_addr, nq, k;
    if (!PyArg_ParseTuple(args, "nnnnn", &table_addr, &n_table, &idx_addr, &nq, &k)) return NULL;
    if (n_table < 0 || nq < 0 || k < 0 || n_table >= ((Py_ssize_t)1 << 32) - 1 || (nq > 0 && k > 0 && nq > (((Py_ssize_t)1 << 32) - 1) / k)) {
        PyErr_SetString(PyExc_ValueError, "take_rows: sizes out of range (n_table and nq * k must stay below 2^32)");
        return NULL;
    }
    PyObject** table = (PyObject**)table_addr;
    const int64_t* idx = (const int64_t*)idx_addr;
    const size_t n = (size_t)nq * (size_t)k;
    PyObject* out = PyList_New(nq);
    if (!out) return NULL;
    for (Py_ssize_t q = 0; q < nq; ++q) {
        PyObject* row = PyList_New(k);          // k NULL slots, every one of them filled below
        if (!row) { Py_DECREF(out); return NULL; }
        PyList_SET_ITEM(out, q, row);
    }
    if (n == 0) return out;
    // scratch: the cached block unless another thread's call holds it (then a private one)
    const size_t need = (2 * sizeof(Pair) + sizeof(Hit)) * n;
    void* scratch = NULL;
    int cached = 0;
    if (!g_scratch_busy) {
        if (g_scratch_bytes < need) {
            free(g_scratch);
            g_scratch = malloc(need);
            g_scratch_bytes = g_scratch ? need : 0;
        }
        scratch = g_scratch;
        cached = scratch != NULL;
        g_scratch_busy = cached;
    }
    if (!scratch) scratch = malloc(need);
    Pair* a = (Pair*)scratch;
    Pair* b = a ? a + n : NULL;
    Hit* hits = a ? (Hit*)(b + n) : NULL;
    const int timing = getenv("SR_HOST_LISTS_TIMING") != NULL;
    const double t0 = now_s();
    double t1 = t0, t2 = t0;
    const size_t n_blocks = ((size_t)nq + BLOCK_Q - 1) / BLOCK_Q;
    size_t* fill = (size_t*)calloc(n_blocks + 1, sizeof(size_t));
    int bad = 0;
    Pair* sorted = NULL;
    if (a && b && hits && fill) {
        Py_BEGIN_ALLOW_THREADS
        for (size_t i = 0; i < n; ++i) {
            const int64_t v = idx[i];
            if (v < -1 || v >= (int64_t)n_table) { bad = 1; break; }
            a[i].key = v < 0 ? (uint32_t)n_table : (uint32_t)v;
            a[i].pos = (uint32_t)i;
        }
        if (!bad) {
            int bits = 1;
            while (((uint64_t)1 << bits) <= (uint64_t)n_table) ++bits;
            sorted = radix_sort_pairs(a, b, n, bits);
            // stream offsets: block c of BLOCK_Q queries receives exactly its rows' hits
            for (size_t c = 0; c < n_blocks; ++c) {
                const size_t q0 = c * BLOCK_Q, q1 = q0 + BLOCK_Q < (size_t)nq ? q0 + BLOCK_Q : (size_t)nq;
                fill[c + 1] = fill[c] + (q1 - q0) * (size_t)k;
            }
        }
        Py_END_ALLOW_THREADS
    }
    t1 = now_s();
    if (!a || !b || !hits || !fill || (!bad && !sorted)) {
        RELEASE_SCRATCH(); free(fill);
        Py_DECREF(out);
        return PyErr_NoMemory();
    }
    if (bad) {
        RELEASE_SCRATCH(); free(fill);
        Py_DECREF(out);
        PyErr_SetString(PyExc_IndexError, "take_rows: index position outside [-1, n_table)");
        return NULL;
    }
    // objects in index order: one reference counted per hit (the GIL is held: reference counts are not atomic)
    {
        const size_t block_elems = (size_t)BLOCK_Q * (size_t)k;
        size_t* cur = (size_t*)malloc(sizeof(size_t) * n_blocks);
        if (!cur) { RELEASE_SCRATCH(); free(fill); Py_DECREF(out); return PyErr_NoMemory(); }
        memcpy(cur, fill, sizeof(size_t) * n_blocks);
        for (size_t i = 0; i < n; ++i) {
            PyObject* o = table[sorted[i].key];
            Py_INCREF(o);
            Hit* h = &hits[cur[sorted[i].pos / block_elems]++];
            h->pos = sorted[i].pos;
            h->obj = o;
        }
        free(cur);
    }
    t2 = now_s();
    // scatter: one block of lists at a time
    for (size_t i = 0; i < n; ++i) {
        const size_t p = hits[i].pos, q = p / (size_t)k, j = p - q * (size_t)k;
        PyList_SET_ITEM(PyList_GET_ITEM(out, (Py_ssize_t)q), (Py_ssize_t)j, hits[i].obj);
    }
    RELEASE_SCRATCH(); free(fill);
    if (timing) fprintf(stderr, "[take_rows] %zu hits: sort %.1f ms, objects in index order %.1f ms, scatter %.1f ms\n", n, (t1 - t0) * 1e3,
                        (t2 - t1) * 1e3, (now_s() - t2) * 1e3);
    return out;
}

static PyMethodDef methods[] = {
    {"take_rows", take_rows, METH_VARARGS,
     "take_rows(table_addr, n_table, idx_addr, nq, k) -> [[table[i] for i in row] for row in idx] (position -1 -> table[n_table])"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_host_lists", "id lists of a dense search result (host side)", -1, methods};

PyMODINIT_FUNC PyInit__host_lists(void) { return PyModule_Create(&module); }
