// Stand-alone timing + spot-check harness for the 2-D Winograd convolution kernel (development tool, not shipped):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I sbgm_danra_amd/csrc [-DKFILE='"path/to/variant.hip"'] tools/micro/w2d_bench.hip -o /tmp/w2d_bench
//   w2d_bench B H W Cin Cout fco minw db in_mode [reps]
// Includes the kernel translation unit directly, so experimental variants of conv_w2d.hip can be timed without touching the library.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#ifndef KFILE
#define KFILE "../../sbgm_danra_amd/csrc/conv_w2d.hip"
#endif
#include KFILE

void sbgm_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);
}
int sbgm_zero_async(void* p, size_t bytes, hipStream_t st) { return hipMemsetAsync(p, 0, bytes, st) != hipSuccess; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 10) { fprintf(stderr, "usage: B H W Cin Cout fco minw db in_mode [reps]\n"); return 2; }
    const int B = atoi(argv[1]), H = atoi(argv[2]), W = atoi(argv[3]), Cin = atoi(argv[4]), Cout = atoi(argv[5]);
    const int fco = atoi(argv[6]), minw = atoi(argv[7]), db = atoi(argv[8]), in_mode = atoi(argv[9]);
    const int reps = argc > 10 ? atoi(argv[10]) : 20;
    const int h = in_mode == 2 ? H / 2 : H, w = in_mode == 2 ? W / 2 : W;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hx((size_t)B * h * w * Cin), hw((size_t)Cout * Cin * 9), hb(Cout);
    for (auto& v : hx) v = nd(rng);
    for (auto& v : hw) v = nd(rng) / sqrtf(9.f * Cin);
    for (auto& v : hb) v = nd(rng);
    float *dx, *dw, *dwp, *dout, *dbias;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&dbias, Cout * 4));
    const size_t pf = sbgm_w2d_packed_floats(Cout, Cin);
    CK(hipMalloc(&dwp, pf * 4)); CK(hipMalloc(&dout, (size_t)B * H * W * Cout * 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hb.data(), Cout * 4, hipMemcpyHostToDevice));
    if (sbgm_launch_pack_w2d_weight(dw, dwp, Cout, Cin, Cin, nullptr)) return 1;
    ConvParams p{};
    p.x = dx; p.wp = dwp; p.out = dout; p.bias = dbias; p.B = B; p.H = H; p.W = W; p.Cs = Cin; p.Cout = Cout; p.in_mode = in_mode;
    // optional epilogue modes (environment): W2D_PROJ=1 -> tap projection of the final block (partial planes per co tile), W2D_STATS=G ->
    // GroupNorm statistics with G groups; a checksum of what they wrote is printed so two builds can be compared
    const bool want_proj = getenv("W2D_PROJ") != nullptr;
    const int want_stats = getenv("W2D_STATS") ? atoi(getenv("W2D_STATS")) : 0;
    float *dprojw = nullptr, *dproj = nullptr;
    double* dstats = nullptr;
    const size_t M = (size_t)B * H * W;
    const int parts = Cout / (16 * (fco == 9 ? 2 : fco));
    if (want_proj) {
        std::vector<float> hpw((size_t)9 * Cout);
        for (auto& v : hpw) v = nd(rng) * 0.1f;
        CK(hipMalloc(&dprojw, hpw.size() * 4)); CK(hipMemcpy(dprojw, hpw.data(), hpw.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&dproj, (size_t)parts * 9 * M * 4)); CK(hipMemset(dproj, 0, (size_t)parts * 9 * M * 4));
        p.proj_w = dprojw; p.proj_out = dproj;
    }
    const size_t nstat = (size_t)B * 64 * (want_stats ? want_stats : 1) * 2;
    if (want_stats) {
        CK(hipMalloc(&dstats, nstat * 8)); CK(hipMemset(dstats, 0, nstat * 8));
        p.gn_stats = dstats; p.gn_groups = want_stats;
    }
#ifdef EXP_STAMP
    unsigned long long* dstamp;
    const size_t nstamp = (size_t)(W / 16) * ((H + 15) / 16) * B * (Cout / (16 * (fco == 9 ? 2 : fco))) * 4 * 8;
    CK(hipMalloc(&dstamp, nstamp * 8)); CK(hipMemset(dstamp, 0, nstamp * 8));
    p.proj_out = reinterpret_cast<float*>(dstamp);
#endif
    // db: 0 single buffer, 1 double buffer, 3 persistent kernel (minw = workgroups per CU).  fco 9 (with -DEXP_W2X and KFILE w2x.hip):
    // the experimental 32x32x2 tiling
    ConvTile ct{fco, 1, 1, minw, 2, db == 3 ? 3 : (db ? 2 : 1)};
#ifdef EXP_W2X
    auto launch = [&]() { return fco == 9 ? (db == 3 ? launch_w2xp(p, minw, nullptr) : launch_w2x(p, minw, db, nullptr)) : sbgm_launch_conv_w2d(p, ct, nullptr); };
#else
    auto launch = [&]() { return sbgm_launch_conv_w2d(p, ct, nullptr); };
#endif
    if (launch()) return 1;
    CK(hipDeviceSynchronize());
    // spot check against a direct fp64 convolution (in_mode 0 and 2-without-pre only)
    std::vector<float> ho((size_t)B * H * W * Cout);
    CK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
    auto in_at = [&](int b, int y, int x, int c) -> double {
        if (y < 0 || y >= H || x < 0 || x >= W) return 0.0;
        if (in_mode != 2) return hx[(((size_t)b * H + y) * W + x) * Cin + c];
        auto src = [&](int o, int n, int& i0, int& i1, double& l) {      // PyTorch bilinear, align_corners=False
            double s = (o + 0.5) / 2.0 - 0.5; if (s < 0) s = 0; i0 = (int)s; i1 = i0 + 1 < n ? i0 + 1 : n - 1; l = s - i0; };
        int y0, y1, x0, x1; double ly, lx;
        src(y, h, y0, y1, ly); src(x, w, x0, x1, lx);
        auto g = [&](int yy, int xx) { return (double)hx[(((size_t)b * h + yy) * w + xx) * Cin + c]; };
        return (1 - ly) * ((1 - lx) * g(y0, x0) + lx * g(y0, x1)) + ly * ((1 - lx) * g(y1, x0) + lx * g(y1, x1));
    };
    double maxerr = 0, maxref = 0;
    std::uniform_int_distribution<int> ub(0, B - 1), uy(0, H - 1), ux(0, W - 1), uc(0, Cout - 1);
    for (int s = 0; s < 400; ++s) {
        int b = ub(rng), y = s < 40 ? (s & 1 ? H - 1 : 0) : uy(rng), x = s < 80 ? (s & 2 ? W - 1 : 0) : ux(rng), co = uc(rng);
        double a = hb[co];
        for (int ci = 0; ci < Cin; ++ci)
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw) a += in_at(b, y + kh - 1, x + kw - 1, ci) * hw[((size_t)co * Cin + ci) * 9 + kh * 3 + kw];
        const double got = ho[(((size_t)b * H + y) * W + x) * Cout + co];
        maxerr = fmax(maxerr, fabs(got - a)); maxref = fmax(maxref, fabs(a));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f, tot = 0.f;
    for (int round = 0; round < 3; ++round) {
        CK(hipEventRecord(e0, nullptr));
        for (int r = 0; r < reps; ++r) if (launch()) return 1;
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms / reps); tot += ms / reps;
    }
#if defined(EXP_STAMP) && defined(EXP_V2)
    {
        std::vector<unsigned long long> hs(nstamp);
        CK(hipMemcpy(hs.data(), dstamp, nstamp * 8, hipMemcpyDeviceToHost));
        double a[11] = {0}; size_t nw = 0;
        for (size_t i = 0; i + 12 <= nstamp && i < (size_t)512 * 4 * 12; i += 12) { if (!hs[i + 10]) continue; ++nw; for (int j = 0; j < 11; ++j) a[j] += (double)hs[i + j]; }
        const double st = a[10] / nw;
        const char* nm[9] = {"sweep0", "vmcnt", "barM", "dma+sweep1", "pstore", "epilogue", "vmcnt", "barE", "post-E"};
        printf("  per stage per wave (cycles, %zu waves, %.0f stages each):", nw, st);
        for (int j = 0; j < 9; ++j) printf(" %s %.0f", nm[j], a[j] / nw / st);
        printf(" | total %.0f\n", a[9] / nw / st);
    }
#elif defined(EXP_STAMP)
    {
        std::vector<unsigned long long> hs(nstamp);
        CK(hipMemcpy(hs.data(), dstamp, nstamp * 8, hipMemcpyDeviceToHost));
        double a[5] = {0, 0, 0, 0, 0};
        const size_t nw = nstamp / 8;
        for (size_t i = 0; i < nw; ++i) for (int j = 0; j < 5; ++j) a[j] += (double)hs[i * 8 + j];
        const double st = Cin / 16.0;
        printf("  per stage per wave (cycles): wait-bar0 %.0f  sweep %.0f  bar1 %.0f  store %.0f   | loop total per stage %.0f\n", a[0] / nw / st, a[1] / nw / st, a[2] / nw / st,
               a[3] / nw / st, a[4] / nw / st);
    }
#endif
    if (want_proj) {
        std::vector<float> hp((size_t)parts * 9 * M);
        CK(hipMemcpy(hp.data(), dproj, hp.size() * 4, hipMemcpyDeviceToHost));
        double a = 0, b2 = 0;
        for (size_t i = 0; i < hp.size(); ++i) { a += hp[i] * (double)((i % 251) + 1); b2 += fabs(hp[i]); }
        printf("  proj checksum %.10e  abs %.10e\n", a, b2);
    }
    if (want_stats) {
        std::vector<double> hs(nstat);
        CK(hipMemcpy(hs.data(), dstats, nstat * 8, hipMemcpyDeviceToHost));
        double a = 0, b2 = 0;
        for (size_t i = 0; i < hs.size(); ++i) { a += hs[i] * (double)((i % 251) + 1); b2 += fabs(hs[i]); }
        printf("  stats checksum %.14e  abs %.14e\n", a, b2);
    }
    {
        double a = 0;
        for (size_t i = 0; i < ho.size(); i += 7) a += ho[i] * (double)((i % 251) + 1);
        if (!want_proj) printf("  out checksum %.10e\n", a);
    }
    const double fl = 2.0 * B * H * W * (double)Cin * Cout * 9;
    printf("%d %dx%d %d->%d fco=%d minw=%d db=%d in=%d : %8.1f us (mean %8.1f)  %6.1f TF direct-equiv  %5.1f TF mfma   spot-err %.2e (ref max %.2f)\n", B, H, W, Cin, Cout, fco,
           minw, db, in_mode, best * 1e3, tot / 3 * 1e3, fl / best * 1e-9, fl * 4 / 9 / best * 1e-9, maxerr, maxref);
    return 0;
}
