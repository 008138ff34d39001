// bf16x3 dense scorer (dense_split.hip).
#pragma once
#include "common.h"
struct DenseSplitArgs {
    const unsigned short* Dhi;   // [rows, H] bf16 planes of the segment
    const unsigned short* Dlo;
    const unsigned short* Qhi;   // [nq, H]
    const unsigned short* Qlo;
    int64_t row_begin, row_end;
    int H, nq;
    const float* tau;
    uint64_t* cand_keys;
    int* cand_count;
    int64_t cand_cap;
    uint32_t id_base, id_stride;
};
int launch_split_bf16(const float* src, unsigned short* hi, unsigned short* lo, int64_t n_elems, hipStream_t s);
int launch_dense_split(const DenseSplitArgs& a, hipStream_t s);
