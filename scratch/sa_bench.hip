// Diagnostic harness for the set-abstraction kernel (not part of the product): hipcc -DSA_DEBUG ...
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <random>
#include <vector>
#include "../deepclr_amd/csrc/fps.hip"
#include "../deepclr_amd/csrc/sa.hip"

int main(int argc, char **argv) {
    // usage: sa_bench [groups 0|1] [modelnet 0|1] [clouds.bin b n]   (clouds.bin: raw f32 (b, n, 4), scratch/make_clouds.py)
    const bool use_groups = argc < 2 || atoi(argv[1]) != 0;
    const bool modelnet = argc > 2 && atoi(argv[2]) != 0;          // 128 clouds of 2048 points on a sphere shell, c = 3
    const char *file = argc > 5 ? argv[3] : nullptr;
    const int b = file ? atoi(argv[4]) : modelnet ? 128 : 16, n = file ? atoi(argv[5]) : modelnet ? 2048 : 16384,
              m = modelnet ? 512 : 1024, c = modelnet ? 3 : 4;
    std::mt19937 rng(1);
    std::normal_distribution<float> g(0.f, 1.f);
    std::uniform_real_distribution<float> u(-0.3f, 0.3f);
    std::vector<float> h((size_t)b * n * c);
    for (size_t i = 0; i < (size_t)b * n; ++i) {
        if (modelnet) {
            float x = g(rng), y = g(rng), z = g(rng);
            const float r = 0.75f / sqrtf(x * x + y * y + z * z + 1e-12f);
            h[i * c + 0] = x * r + 0.02f * g(rng); h[i * c + 1] = y * r + 0.02f * g(rng); h[i * c + 2] = z * r + 0.02f * g(rng);
        } else {
            h[i * c + 0] = 20.f * g(rng); h[i * c + 1] = 20.f * g(rng); h[i * c + 2] = -1.f + 0.5f * g(rng); h[i * c + 3] = 0.5f;
        }
    }
    if (file) {
        FILE *f = fopen(file, "rb");
        if (!f || fread(h.data(), 4, h.size(), f) != h.size()) { printf("cannot read %s\n", file); return 1; }
        fclose(f);
    }
    std::vector<float> w(2 * 896);
    for (auto &v : w) v = u(rng);
    float *d, *gp, *gb, *rows, *wd, *sb = nullptr; int32_t *idx;
    int ng = 64, gs = 256;
    dclr_fps_group_layout(n, &ng, &gs);
    hipMalloc(&d, h.size() * 4); hipMalloc(&idx, (size_t)b * m * 4); hipMalloc(&gp, (size_t)b * ng * gs * 16);
    hipMalloc(&gb, (size_t)b * ng * 32); hipMalloc(&rows, (size_t)b * m * 68 * 4); hipMalloc(&wd, w.size() * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(wd, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    int rc;
    if (n > 16384) {
        void *ws; const long long need = dclr_fps_workspace_bytes(b, n);
        hipMalloc(&ws, need);
        rc = dclr_fps_clouds_grouped_ws(b, n, c, m, d, idx, gp, gb, ws, need, nullptr);
    } else if (getenv("SA_SLICES") && gs > 64 && b % 2 == 0) {
        hipMalloc(&sb, (size_t)b * ng * (gs / 64) * 32);
        rc = dclr_fps_clouds_grouped_batched(b, n, c, m, d, b / 2, 1, 0, idx, gp, gb, sb, nullptr, 0, nullptr);
    } else {
        rc = dclr_fps_clouds_grouped(b, n, c, m, d, idx, gp, gb, nullptr);
    }
    hipDeviceSynchronize();
    const float radii[2] = {modelnet ? 0.1f : 0.5f, modelnet ? 0.2f : 1.0f}; const int ns[2] = {modelnet ? 256 : 512, modelnet ? 512 : 1024};
    const float *mlps[2] = {wd, wd + 896};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    { int one = getenv("SA_DBG2") ? 1 : 0; hipMemcpyToSymbol(HIP_SYMBOL(getenv_dbg2), &one, sizeof(one)); }
    for (int rep = 0; rep < 3; ++rep) {
        unsigned long long zero[8] = {0};
        { static std::vector<unsigned long long> z(16384 * 8, 0); hipMemcpyToSymbol(HIP_SYMBOL(sa_dbg_w), z.data(), z.size() * 8); }
        hipEventRecord(e0);
        const bool f16 = getenv("SA_F32") == nullptr;
        int rc2 = sb ? dclr_sa_msg_fused_batched(f16, b, n, c, m, d, b / 2, 1, 0, idx, 2, radii, ns, mlps, rows, nullptr, gp, gb, sb, nullptr)
                     : (f16 ? dclr_sa_msg_fused_f16 : dclr_sa_msg_fused)(b, n, c, m, d, idx, 2, radii, ns, mlps, rows, nullptr, use_groups ? gp : nullptr,
                                    use_groups ? gb : nullptr, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long dbg[8];
        { static std::vector<unsigned long long> w(16384 * 8); hipMemcpyFromSymbol(w.data(), HIP_SYMBOL(sa_dbg_w), w.size() * 8);
          for (int j = 0; j < 8; ++j) { dbg[j] = 0; for (int i = 0; i < 16384; ++i) dbg[j] += w[(size_t)i * 8 + j]; }
          std::vector<unsigned long long> tot;
          for (int i = 0; i < 16384; ++i) if (w[(size_t)i * 8 + 5]) tot.push_back(w[(size_t)i * 8]);
          std::sort(tot.begin(), tot.end());
          if (!tot.empty()) printf("   per-wave total cycles: median %llu  p90 %llu  p99 %llu  max %llu\n", tot[tot.size() / 2],
                                   tot[tot.size() * 9 / 10], tot[tot.size() * 99 / 100], tot.back()); }
        const double nw = dbg[5] ? (double)dbg[5] : 1;
        printf("rc=%d/%d groups=%d  %.1f us | per wave (cycles): total %.0f  fast-path(incl drains) %.0f  drains %.0f (%.2f drains)  sweep %.0f | waves %.0f | inside drains: load %.0f mlp+fold %.0f\n",
               rc, rc2, (int)use_groups, ms * 1e3, dbg[0] / nw, dbg[1] / nw, dbg[2] / nw, dbg[3] / nw, dbg[4] / nw, nw, dbg[6] / nw, dbg[7] / nw);
    }
    return 0;
}
