// Opt-in per-kernel-class timing with HIP events on the launch stream (used by bench.py for the `roofline` object):
// every launcher brackets its kernel with a ProfScope; when profiling is off this is one predictable branch.
#include "prof.h"

#include <mutex>
#include <vector>

namespace mfvit {

namespace {
struct Rec { hipEvent_t a, b; int cls; double flops; double bytes; int tag; };
thread_local int t_tag = 0;
double g_tags[PROF_NTAG * 3] = {};      // per tag: launches, ms, flops (filled by mfvit_prof_collect)
std::mutex g_mu;
unsigned g_mask = 0;
std::vector<Rec> g_recs;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pool;
}  // namespace

bool prof_enabled(int cls) { return (g_mask >> cls) & 1u; }
void prof_set_tag(int tag) { t_tag = tag; }

void prof_begin(int cls, double flops, double bytes, hipStream_t st, void** token) {
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r;
    if (!g_pool.empty()) { r.a = g_pool.back().first; r.b = g_pool.back().second; g_pool.pop_back(); }
    else { (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b); }
    r.cls = cls; r.flops = flops; r.bytes = bytes; r.tag = t_tag;
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    *token = (void*)(uintptr_t)g_recs.size();
}
void prof_end(void* token, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t i = (size_t)(uintptr_t)token;
    if (i == 0 || i > g_recs.size()) return;
    (void)hipEventRecord(g_recs[i - 1].b, st);
}

}  // namespace mfvit

extern "C" {
int mfvit_prof_enable(int class_mask) {
    std::lock_guard<std::mutex> lk(mfvit::g_mu);
    mfvit::g_mask = (unsigned)class_mask;
    return 0;
}
// Synchronises the recorded events, accumulates per class into out[cls*4 + {0: launches, 1: ms, 2: flops, 3: bytes}] and clears.
int mfvit_prof_collect(double* out, int ncls) {
    std::lock_guard<std::mutex> lk(mfvit::g_mu);
    for (int i = 0; i < ncls * 4; ++i) out[i] = 0.0;
    for (double& v : mfvit::g_tags) v = 0.0;
    for (auto& r : mfvit::g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && r.cls >= 0 && r.cls < ncls) {
            out[r.cls * 4 + 0] += 1.0;
            out[r.cls * 4 + 1] += ms;
            out[r.cls * 4 + 2] += r.flops;
            out[r.cls * 4 + 3] += r.bytes;
            if (r.tag > 0 && r.tag < mfvit::PROF_NTAG) {
                mfvit::g_tags[r.tag * 3 + 0] += 1.0;
                mfvit::g_tags[r.tag * 3 + 1] += ms;
                mfvit::g_tags[r.tag * 3 + 2] += r.flops;
            }
        }
        mfvit::g_pool.emplace_back(r.a, r.b);
    }
    mfvit::g_recs.clear();
    return 0;
}
// The tagged launches of the records the LAST mfvit_prof_collect call consumed: out[tag*3 + {0 launches, 1 ms, 2 flops}], tags 1 (qkv projection of the
// encoder forward) and 2 (output projection + residual + LayerNorm of the forward)
int mfvit_prof_collect_tags(double* out, int ntags) {
    std::lock_guard<std::mutex> lk(mfvit::g_mu);
    for (int t = 0; t < ntags; ++t)
        for (int i = 0; i < 3; ++i) out[t * 3 + i] = t < mfvit::PROF_NTAG ? mfvit::g_tags[t * 3 + i] : 0.0;
    return 0;
}
const char* mfvit_prof_class_name(int cls) {
    static const char* names[] = {"gemm_nt_tile", "gemm_nt_row_res_ln", "gemm_nt_row_lnbwd", "gemm_tn_wgrad", "attention_fwd",
                                  "attention_bwd", "xattn_stream_fwd", "xattn_stream_bwd", "infonce", "other"};
    return cls >= 0 && cls < 10 ? names[cls] : "?";
}
}
