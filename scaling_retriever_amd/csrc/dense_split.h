// Split-bf16 dense scorer (dense_split.hip).
#pragma once
#include "common.h"
struct DenseSplitArgs {
    const unsigned short* D[3];  // [rows, H] bf16 planes of the segment (hi, mid, lo); D[2] unused in bf16x3
    const unsigned short* Q[3];  // [nq, H] bf16 planes of the queries
    int n_pairs;                 // 2 (the certified filter: doc plane 0 x query planes 1, 0), 3 (bf16x3) or 6 (bf16x6)
    int pair_d[6], pair_q[6];    // plane pair of every product, accumulated in this order (smallest terms first)
    int64_t row_begin, row_end;
    int H, nq;
    const float* tau;
    uint64_t* cand_keys;
    int* cand_count;
    int64_t cand_cap;
    uint32_t id_base, id_stride;
    // filled by launch_dense_split: workgroup -> tile mapping (dense_split.hip split_tile_of)
    int xcd_order, grid_qt, grid_dt, grid_bq, grid_bd, grid_nbq;
};
// p1 and p2 may be null (fewer planes)
int launch_split_bf16(const float* src, unsigned short* p0, unsigned short* p1, unsigned short* p2, int64_t n_elems, hipStream_t s);
int launch_dense_split(const DenseSplitArgs& a, hipStream_t s);
