// Kernel-level bench for the GlobalStage attention kernels (be_attn.hip), without Python: the source file is included, so the
// kernels in its anonymous namespace can be launched and timed one by one with hipEvents.
//   make -C lab        (then: lab/bin/attn_lab [B] [reps])
// B x 8 heads x 4096 tokens.  This is where the round-2 rewrite of the three kernels was developed (variants lived here and were
// compared with the then-product kernels before they replaced them; numbers in profiles/r02_attention_rewrite.md).
#include "../blurry-edges_amd/csrc/be_attn.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

namespace be { char* last_error_buf() { static thread_local char buf[512]; return buf; } }

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float frand(uint32_t& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f * 2.f - 1.f; }

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, reps = argc > 2 ? atoi(argv[2]) : 20;
    const int L = 4096, H = 8;
    const size_t T = (size_t)B * L, D = (size_t)H * DH;
    uint32_t s = 12345u;
    std::vector<float> hqkv(T * 3 * D), hdo(T * D);
    for (auto& x : hqkv) x = frand(s) * 1.5f;
    for (auto& x : hdo) x = frand(s);
    float *qkv, *out, *lse, *dout, *dqkv, *ws, *wsi;
    CK(hipMalloc(&qkv, T * 3 * D * 4)); CK(hipMalloc(&out, T * D * 4)); CK(hipMalloc(&lse, (size_t)B * H * L * 4));
    CK(hipMalloc(&dout, T * D * 4)); CK(hipMalloc(&dqkv, T * 3 * D * 4));
    CK(hipMalloc(&ws, be_attention_train_workspace_floats(B, L, H) * 4));
    CK(hipMalloc(&wsi, be_attention_workspace_floats(B, L, H) * 4));
    CK(hipMemcpy(qkv, hqkv.data(), T * 3 * D * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dout, hdo.data(), T * D * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid(L / 128, B * H), blk(256);
    const double fl = 4.0 * L * L * 16 * B * H;          // one forward: two products of 2 L^2 d flops per head
    auto tk = [&](const char* name, double flops, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-34s %.3f ms  %6.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
    };
    if (be_attention_f32(qkv, out, wsi, B, L, L, H, nullptr)) { printf("inference call failed\n"); return 1; }
    {
        const size_t n = (size_t)B * H * L * DH;
        float *Q = wsi, *K = wsi + n, *Vt = wsi + 2 * n;
        tk("k_attention<infer>", fl, [&] { hipLaunchKernelGGL((k_attention<false, false>), grid, blk, 0, 0, Q, K, Vt, out, (float*)nullptr, L, L, H, 0u, 0u, 1.0f, (float*)nullptr, (uint16_t*)nullptr); });
    }
    for (int pi = 0; pi < 2; ++pi) {
        const float p = pi ? 0.0f : 0.1f;
        const uint32_t th = drop_threshold(p);
        const float ik = 1.0f / (1.0f - p);
        if (be_attention_train_fwd_f32(qkv, out, lse, ws, B, L, L, H, p, 7u, nullptr)) { printf("fwd failed\n"); return 1; }
        {
            float* sc0;
            CK(hipMalloc(&sc0, be_attention_bwd_scratch_floats(B, L, H) * 4));
            if (be_attention_bwd_f32(qkv, out, lse, dout, dqkv, ws, sc0, 1, B, L, L, H, p, 7u, nullptr)) { printf("bwd failed\n"); return 1; }
            CK(hipDeviceSynchronize());
            CK(hipFree(sc0));
        }
        CK(hipDeviceSynchronize());
        const TrainWs w = train_ws(ws, B, L, H);
        char nm[64];
        snprintf(nm, sizeof nm, "k_attention<train> p=%.1f", p);
        tk(nm, fl, [&] { hipLaunchKernelGGL((k_attention<true, false>), grid, blk, 0, 0, w.Q, w.K, w.Vt, out, lse, L, L, H, 7u, th, ik, (float*)nullptr, w.keep); });
        {
            float* part;
            CK(hipMalloc(&part, be_attention_bwd_scratch_floats(B, L, H) * 4));
            const int ngroups = L / (KB * FW);
            const dim3 gridf(ngroups, B * H), blkf(64 * FW);
            snprintf(nm, sizeof nm, "k_attn_bwd_fused p=%.1f", p);
            if (th) tk(nm, 2.5 * fl, [&] { hipLaunchKernelGGL((k_attn_bwd_fused<false, true>), gridf, blkf, 0, 0, w.Q, w.K, w.V, w.Qt, w.Kt, w.dOh, w.dOt, w.nlse, w.nD, dqkv, part, L, L, H, w.keep, ik); });
            else tk(nm, 2.5 * fl, [&] { hipLaunchKernelGGL((k_attn_bwd_fused<false, false>), gridf, blkf, 0, 0, w.Q, w.K, w.V, w.Qt, w.Kt, w.dOh, w.dOt, w.nlse, w.nD, dqkv, part, L, L, H, w.keep, ik); });
            snprintf(nm, sizeof nm, "k_attn_dq_finish p=%.1f", p);
            tk(nm, 0.0, [&] { hipLaunchKernelGGL(k_attn_dq_finish, dim3(4096), dim3(256), 0, 0, part, dqkv, ngroups, B * H, L, H, L, 0.25f * ik); });
            CK(hipFree(part));
        }
    }
    return 0;
}
