// placeholder: encoder entry points (implemented next)
#include "common.h"
#define NOT_YET(name) do { sr_set_error(name ": not implemented in this build"); return SR_ERR_UNSUPPORTED; } while (0)
extern "C" int sr_model_create(sr_model** out, const sr_model_config* cfg) { NOT_YET("sr_model_create"); }
extern "C" int sr_model_set_weight(sr_model* m, const char* name, const void* d_ptr, int dtype, int64_t rows, int64_t cols, sr_stream stream) { NOT_YET("sr_model_set_weight"); }
extern "C" int sr_model_finalize(sr_model* m) { NOT_YET("sr_model_finalize"); }
extern "C" int sr_encode_dense(sr_model* m, const int64_t* a, const int64_t* b, int32_t B, int32_t L, float* o, sr_stream s) { NOT_YET("sr_encode_dense"); }
extern "C" int sr_encode_sparse(sr_model* m, const int64_t* a, const int64_t* b, int32_t B, int32_t L, float* o, sr_stream s) { NOT_YET("sr_encode_sparse"); }
extern "C" int sr_model_last_hidden(sr_model* m, float* d_out, int64_t cap, int64_t* n, sr_stream s) { NOT_YET("sr_model_last_hidden"); }
extern "C" int sr_model_destroy(sr_model* m) { return SR_OK; }
extern "C" int sr_lora_merge(float* W, const float* A, const float* B, int64_t o, int64_t i, int32_t r, float sc, sr_stream s) { NOT_YET("sr_lora_merge"); }
extern "C" int sr_sparse_compact(const float* d, int64_t B, int64_t V, int64_t* rp, int32_t* c, float* v, int64_t cap, int64_t* n, sr_stream s) { NOT_YET("sr_sparse_compact"); }
