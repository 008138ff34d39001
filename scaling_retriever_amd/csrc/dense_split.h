// Split-plane dense scorer (dense_split.hip): bf16 plane products (bf16x3 / bf16x6 score modes) and the certified
// filter's one-product fp16 upper-bound pass.
#pragma once
#include "common.h"
struct DenseSplitArgs {
    const unsigned short* D[3];  // [rows, H] planes of the segment: bf16 (hi, mid, lo), or D[0] = the fp16 filter plane (upper_bound)
    const unsigned short* Q[3];  // [nq, H] planes of the queries, same types
    int n_pairs;                 // 1 (upper_bound), 3 (bf16x3) or 6 (bf16x6)
    int pair_d[6], pair_q[6];    // plane pair of every product, accumulated in this order (smallest terms first)
    int64_t row_begin, row_end;
    int H, nq;
    const float* tau;
    uint64_t* cand_keys;
    int* cand_count;
    int64_t cand_cap;
    uint32_t id_base, id_stride;
    // segmented candidate slots (common.h TopkWS): null = every survivor is appended through the atomic counter
    unsigned char* seg_cnt;
    int seg_n;
    int64_t seg_off;
    // upper_bound = 1 (dense_filter.hip): fp16 planes of power-of-two scaled operands; the key of (q, j) is not the plane
    // product but U = acc / (sq sd) + e(q, j), an upper bound of the exact fp32 score:
    //   U' = acc + qa[q].x * dxy[j].x + qa[q].y * dxy[j].y   (scaled domain),   U = U' * qa[q].w * isd
    int upper_bound;
    const float* dxy;            // [rows, 2]: per document (x, y) of the scaled plane, see filter_plane_kernel
    const float* qa;             // [nq, 4]: per query (A', B', sq, 1 / sq)
    const float* dxy_gmax;       // [ceil(rows / 128), 2] or null: (max x, max y) per group of 128 documents - the epilogue's block test
    float sd, isd;               // the segment's power-of-two scale and its inverse
    // filled by launch_dense_split: workgroup -> tile mapping (dense_split.hip split_tile_of)
    int xcd_order, grid_qt, grid_dt, grid_bq, grid_bd, grid_nbq, grid_total;
    int diag;                    // dev switch SR_SPLIT_DIAG (timing only)
    unsigned long long* stamps;  // dev switch SR_SPLIT_STAMPS
};
// p1 and p2 may be null (fewer planes)
int launch_split_bf16(const float* src, unsigned short* p0, unsigned short* p1, unsigned short* p2, int64_t n_elems, hipStream_t s);
int launch_dense_split(const DenseSplitArgs& a, hipStream_t s);
