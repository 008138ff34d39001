// Streaming dense scorer for small query batches (dense_stream.hip).
#pragma once
#include "common.h"
struct DenseStreamArgs {
    const float* D;
    const float* Q;
    int64_t row_begin, row_end;
    int H, nq;
    const float* tau;
    uint64_t* cand_keys;
    int* cand_count;
    int64_t cand_cap;
    uint32_t id_base, id_stride;
};
int launch_dense_stream(const DenseStreamArgs& a, hipStream_t s);
bool dense_stream_supports(int nq, int H);
