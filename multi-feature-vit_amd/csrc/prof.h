#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mfvit {
enum { PROF_GEMM_TILE = 0, PROF_GEMM_ROW_FWD = 1, PROF_GEMM_ROW_BWD = 2, PROF_GEMM_TN = 3, PROF_ATTN_FWD = 4, PROF_ATTN_BWD = 5,
       PROF_XATTN_FWD = 6, PROF_XATTN_BWD = 7, PROF_INFONCE = 8, PROF_OTHER = 9, PROF_NCLS = 10 };
bool prof_enabled(int cls);
// sub-attribution inside a class: launches between prof_set_tag(t) and prof_set_tag(0) of this thread are ALSO counted under tag t
// (1 = qkv projection of the forward, 2 = output projection + residual + LayerNorm of the forward: with class attention_fwd the parts of
// the fused-MHSA figure of SURVEY.md 7 / 8d; mfvit_prof_collect_tags)
enum { PROF_TAG_NONE = 0, PROF_TAG_MHSA_QKV = 1, PROF_TAG_MHSA_PROJ = 2, PROF_NTAG = 3 };
void prof_set_tag(int tag);
struct ProfTag {
    ProfTag(int t) { prof_set_tag(t); }
    ~ProfTag() { prof_set_tag(0); }
};
void prof_begin(int cls, double flops, double bytes, hipStream_t st, void** token);
void prof_end(void* token, hipStream_t st);
struct ProfScope {
    void* tok = nullptr;
    hipStream_t st;
    ProfScope(int cls, double flops, double bytes, hipStream_t s) : st(s) { if (prof_enabled(cls)) prof_begin(cls, flops, bytes, s, &tok); }
    ~ProfScope() { if (tok) prof_end(tok, st); }
};
}  // namespace mfvit
