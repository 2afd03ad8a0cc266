// Diagnostic harness for the FPS kernels (not part of the product): hipcc -DFPS_DEBUG ...
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include <vector>
#include "../deepclr_amd/csrc/fps.hip"

int main(int argc, char **argv) {
    // usage: fps_bench [n] [m] [clouds.bin b]   (clouds.bin: raw f32 (b, n, 4), scratch/make_clouds.py)
    const char *file = argc > 4 ? argv[3] : nullptr;
    const int b = file ? atoi(argv[4]) : 16, n = argc > 1 ? atoi(argv[1]) : 16384, m = argc > 2 ? atoi(argv[2]) : 1024, c = 4;
    std::mt19937 rng(1);
    std::normal_distribution<float> g(0.f, 1.f);
    std::vector<float> h((size_t)b * n * c);
    for (size_t i = 0; i < (size_t)b * n; ++i) {
        h[i * c + 0] = 20.f * g(rng); h[i * c + 1] = 20.f * g(rng); h[i * c + 2] = -1.f + 0.5f * g(rng); h[i * c + 3] = 0.5f;
    }
    if (file) {
        FILE *f = fopen(file, "rb");
        if (!f || fread(h.data(), 4, h.size(), f) != h.size()) { printf("cannot read %s\n", file); return 1; }
        fclose(f);
    }
    float *d; int32_t *idx;
    hipMalloc(&d, h.size() * 4); hipMalloc(&idx, (size_t)b * m * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        unsigned long long zero[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(fps_dbg), zero, sizeof(zero));
        hipEventRecord(e0);
        int rc = dclr_fps_clouds(b, n, c, m, d, idx, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long dbg[16];
        hipMemcpyFromSymbol(dbg, HIP_SYMBOL(fps_dbg), sizeof(dbg));
        { unsigned long long st[8]; hipMemcpyFromSymbol(st, HIP_SYMBOL(fps_dbg_setup), sizeof(st));
          printf("   setup (cycles, cloud 0 thread 0): bounding box %llu | cells + sort %llu | rank groups + slice boxes %llu | load groups %llu\n",
                 st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3]); }
        const double rounds = m - 1;
        printf("rc=%d  %.1f us | active groups/round %.2f of 64 | cycles/round wave0/cloud0: update+publish %.0f barrier-wait %.0f combine %.0f total %.0f\n",
               rc, ms * 1e3, dbg[0] / (rounds * b), dbg[1] / rounds, dbg[2] / rounds, dbg[3] / rounds, dbg[4] / rounds);
        printf("   super-rounds: %llu for %d samples | wave 3 per super-round (cycles): updates %.0f select %.0f publish+barrier %.0f combine %.0f | touched in %llu\n", dbg[11], m - 1, (double)dbg[12] / dbg[11], (double)dbg[13] / dbg[11], (double)dbg[14] / dbg[11], (double)dbg[15] / dbg[11], dbg[10]);
        if (!getenv("DCLR_FPS_WAVECAND") && !getenv("DCLR_FPS_SINGLE") && dbg[11])
            printf("   table mode: %llu rounds (%.2f samples/round) | per round (cycles): wave 3: work %.0f wait-A %.0f leader+B %.0f, touched in %llu rounds, %llu group updates | wave 0: work %.0f wait-A %.0f leader+B %.0f, touched in %llu, %llu group updates\n",
                   dbg[11], (double)(m - 1) / dbg[11], (double)dbg[12] / dbg[11], (double)dbg[14] / dbg[11], (double)dbg[15] / dbg[11], dbg[10], dbg[13],
                   (double)dbg[1] / dbg[11], (double)dbg[2] / dbg[11], (double)dbg[3] / dbg[11], dbg[5], dbg[6]);
        if (!getenv("DCLR_FPS_WAVECAND") && !getenv("DCLR_FPS_SINGLE") && dbg[11])
            printf("   leader per round (cycles): table read + J maxima + box tests %.0f | pair tests %.0f | accept + publish %.0f\n",
                   (double)dbg[0] / dbg[11], (double)dbg[4] / dbg[11], (double)dbg[9] / dbg[11]);
        if (!getenv("DCLR_FPS_WAVECAND") && !getenv("DCLR_FPS_SINGLE") && dbg[11])
            printf("   entries rewritten: wave 3 %llu of %llu group updates, wave 0 %llu of %llu\n", dbg[7], dbg[13], dbg[8], dbg[6]);
        if (rep == 2 && !getenv("DCLR_FPS_WAVECAND") && !getenv("DCLR_FPS_SINGLE")) {
            unsigned int grp[64];
            hipMemcpyFromSymbol(grp, HIP_SYMBOL(fps_grp), sizeof(grp));
            unsigned long long bk[16][6][3];
            hipMemcpyFromSymbol(bk, HIP_SYMBOL(fps_bucket), sizeof(bk));
            for (int nm = 0; nm <= 4; ++nm) {
                unsigned long long rounds = 0, cyc = 0, rw = 0;
                for (int w = 0; w < 16; ++w) { rounds += bk[w][nm][0]; cyc += bk[w][nm][1]; rw += bk[w][nm][2]; }
                if (rounds) printf("   (wave, round) pairs with %d%s marks: %llu, mean work %.0f cycles, %.2f entry rewrites\n", nm, nm == 4 ? "+" : "", rounds, (double)cyc / rounds, (double)rw / rounds);
            }
            printf("   marks per group (3 reps):");
            for (int q = 0; q < 64; ++q) printf(" %u", grp[q]);
            printf("\n");
        }
        for (int w = 0; w < 2; ++w) {
            const unsigned long long *d = dbg + 8 * w;
            if (d[4]) printf("   chains mode, wave %d: rounds %llu picks %llu | per round (cycles): apply %.0f chain %.0f publish+barrier %.0f merge(+barrier) %.0f\n",
                             w ? 3 : 0, d[4], d[5], (double)d[0] / d[4], (double)d[1] / d[4], (double)d[2] / d[4], (double)d[3] / d[4]);
        }
        const double na = dbg[10] ? (double)dbg[10] : 1.0, ni = rounds - dbg[10] > 0 ? rounds - dbg[10] : 1.0;
        printf("   wave0/cloud0: active in %.0f of %.0f rounds; per ACTIVE round: box %.0f update %.0f select %.0f publish %.0f | per IDLE round: pre-barrier %.0f\n",
               (double)dbg[10], rounds, dbg[5] / na, dbg[6] / na, dbg[7] / na, dbg[8] / na, dbg[9] / ni);
    }
    return 0;
}
