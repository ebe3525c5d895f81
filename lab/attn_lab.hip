// Attention timing lab: builds vit_som_amd/csrc/attention.hip with VSOM_ATTN_STAMPS (per-workgroup real-time
// stamps at start / slices staged / phase boundary / end + the hardware id of wave 0) and prints where a
// workgroup's life goes and how many workgroups share a CU.  Standalone executable:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/attn_lab.hip -o tools/attn_lab && tools/attn_lab [B] [N] [H] [hd] [mode]
// mode = vsom_set_attention_fused(): 0 general kernels, 1 default.  With -DVSOM_ATTN_CANDIDATE -Ivit_som_amd/csrc the
// lab builds tools/attention_roll_candidate.hip instead (the LDS-DMA "rolling" kernels; mode k >= 2 = k items per
// workgroup) and checks it bit for bit against the general kernels; -DVSOM_ATTN_REPEAT=3 repeats the fused
// backward's compute three times behind one staging.
#define VSOM_ATTN_STAMPS 1
#ifdef VSOM_ATTN_CANDIDATE
#include "attention_roll_candidate.hip"
#else
#include "../vit_som_amd/csrc/attention.hip"
#endif

#include <algorithm>
#include <map>
#include <stdarg.h>
#include <vector>

namespace vsom {
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
}  // namespace vsom

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return 1;                                                               \
        }                                                                           \
    } while (0)

static void fill(std::vector<float>& v, unsigned seed, float amp) {
    unsigned s = seed * 2654435761u + 12345u;
    for (auto& x : v) {
        s = s * 1664525u + 1013904223u;
        x = amp * (((s >> 8) & 0xffff) / 32768.0f - 1.0f);
    }
}

static void report(const char* name, const std::vector<unsigned long long>& st, int grid, float kernel_us) {
    // stamps are 100 MHz ticks (10 ns)
    unsigned long long t0 = ~0ull, t3 = 0;
    for (int g = 0; g < grid; ++g) {
        t0 = std::min(t0, st[g * 16 + 0]);
        t3 = std::max(t3, st[g * 16 + 3]);
    }
    double a = 0, b = 0, c = 0, life = 0;
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> per_cu;
    for (int g = 0; g < grid; ++g) {
        const unsigned long long* s = &st[g * 16];
        a += (s[1] - s[0]) * 0.01;
        b += (s[2] - s[1]) * 0.01;
        c += (s[3] - s[2]) * 0.01;
        life += (s[3] - s[0]) * 0.01;
        const unsigned hw = (unsigned)s[4], xcc = (unsigned)s[5] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per_cu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu].push_back({s[0], s[3]});
    }
    printf("%s: kernel %.1f us (events), first start -> last end %.1f us, %d workgroups on %zu CUs\n", name, kernel_us,
           (t3 - t0) * 0.01, grid, per_cu.size());
    printf("   mean per workgroup: staging %.2f us | phase A %.2f us | phase B %.2f us | life %.2f us\n", a / grid, b / grid,
           c / grid, life / grid);
    double f6 = 0, f7 = 0;
    int n6 = 0;
    for (int g = 0; g < grid; ++g)
        if (st[g * 16 + 6]) { f6 += (st[g * 16 + 6] - st[g * 16 + 0]) * 0.01; f7 += (st[g * 16 + 7] - st[g * 16 + 0]) * 0.01; ++n6; }
    if (n6) {
        printf("   rolling: first item's slices landed at %.2f us, its score phase ended (V complete) at %.2f us\n", f6 / n6, f7 / n6);
        printf("   first item, mean time since start at:");
        const char* nm[5] = {"PV done", "token-0 partial done", "barrier C passed", "combine done", "older loads landed (wave 0)"};
        for (int k = 8; k <= 12; ++k) {
            double a2 = 0;
            for (int g = 0; g < grid; ++g) a2 += (st[g * 16 + k] - st[g * 16 + 0]) * 0.01;
            printf("  %s %.2f", nm[k - 8], a2 / grid);
        }
        printf("\n");
    }
    // mean concurrency per CU = sum of lives / span, and the start-time histogram (2 us bins)
    double conc = 0;
    for (auto& kv : per_cu) {
        double sum = 0;
        unsigned long long lo = ~0ull, hi = 0;
        for (auto& p : kv.second) {
            sum += (double)(p.second - p.first);
            lo = std::min(lo, p.first);
            hi = std::max(hi, p.second);
        }
        conc += sum / (double)(hi - lo);
    }
    printf("   mean workgroups resident per CU while it is busy: %.2f;  workgroups per CU: %.1f\n", conc / per_cu.size(),
           (double)grid / per_cu.size());
    std::vector<int> hist(64, 0);
    for (int g = 0; g < grid; ++g) {
        const int bin = (int)((st[g * 16 + 0] - t0) * 0.01 / 2.0);
        if (bin < 64) hist[bin]++;
    }
    printf("   starts per 2 us bin:");
    for (int i = 0; i < 64 && i * 2.0 < (t3 - t0) * 0.01; ++i) printf(" %d", hist[i]);
    printf("\n");
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 512, N = argc > 2 ? atoi(argv[2]) : 65, H = argc > 3 ? atoi(argv[3]) : 3,
              hd = argc > 4 ? atoi(argv[4]) : 64, mode = argc > 5 ? atoi(argv[5]) : 1;
    vsom_set_attention_fused(mode);
    const int E = H * hd;
    const size_t nq = (size_t)B * N * 3 * E, no = (size_t)B * N * E, ns = (size_t)B * H * N;
    std::vector<float> hq(nq), hdo(no);
    fill(hq, 1, 1.0f);
    fill(hdo, 2, 1.0f);
    float *qkv, *out, *lse, *dout, *dqkv, *delta;
    CK(hipMalloc(&qkv, nq * 4)); CK(hipMalloc(&out, no * 4)); CK(hipMalloc(&lse, ns * 4));
    CK(hipMalloc(&dout, no * 4)); CK(hipMalloc(&dqkv, nq * 4)); CK(hipMalloc(&delta, ns * 4));
    CK(hipMemcpy(qkv, hq.data(), nq * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dout, hdo.data(), no * 4, hipMemcpyHostToDevice));
    const int grid = B * H;
    unsigned long long* stamps;
    CK(hipMalloc(&stamps, (size_t)grid * 16 * 8));
    unsigned long long* null_stamps = nullptr;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned long long> st((size_t)grid * 16);

    // the selected mode against the general kernels (mode 0): the results must be the same bits
    {
        std::vector<float> ref_o(no), ref_l(ns), ref_g(nq), got_o(no), got_l(ns), got_g(nq);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(vsom::g_attn_stamps), &null_stamps, sizeof(void*)));
        for (int pass = 0; pass < 2; ++pass) {
            vsom_set_attention_fused(pass == 0 ? 0 : mode);
            CK(hipMemset(out, 0xff, no * 4)); CK(hipMemset(lse, 0xff, ns * 4)); CK(hipMemset(dqkv, 0xff, nq * 4));
            if (vsom_attention_fwd(qkv, out, lse, B, N, H, hd, nullptr)) return 1;
            if (vsom_attention_bwd(qkv, out, dout, lse, dqkv, delta, B, N, H, hd, nullptr)) return 1;
            CK(hipDeviceSynchronize());
            CK(hipMemcpy((pass ? got_o : ref_o).data(), out, no * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy((pass ? got_l : ref_l).data(), lse, ns * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy((pass ? got_g : ref_g).data(), dqkv, nq * 4, hipMemcpyDeviceToHost));
        }
        printf("mode %d vs mode 0: out %s, lse %s, dqkv %s\n", mode, memcmp(ref_o.data(), got_o.data(), no * 4) ? "DIFFERENT" : "same bits",
               memcmp(ref_l.data(), got_l.data(), ns * 4) ? "DIFFERENT" : "same bits",
               memcmp(ref_g.data(), got_g.data(), nq * 4) ? "DIFFERENT" : "same bits");
        int shown = 0;
        size_t nbad = 0;
        for (size_t i = 0; i < no; ++i)
            if (memcmp(&ref_o[i], &got_o[i], 4)) {
                ++nbad;
                if (shown++ < 8)
                    printf("   out[%zu] (image %zu token %zu channel %zu): %g vs %g\n", i, i / ((size_t)N * E), (i / E) % N, i % E, ref_o[i], got_o[i]);
            }
        if (nbad) printf("   %zu of %zu out elements differ\n", nbad, no);
    }

    for (int which = 0; which < 2; ++which) {
        auto run = [&]() {
            return which == 0 ? vsom_attention_fwd(qkv, out, lse, B, N, H, hd, nullptr)
                              : vsom_attention_bwd(qkv, out, dout, lse, dqkv, delta, B, N, H, hd, nullptr);
        };
        // un-instrumented timing (stamps pointer null -> the stores are skipped)
        CK(hipMemcpyToSymbol(HIP_SYMBOL(vsom::g_attn_stamps), &null_stamps, sizeof(void*)));
        for (int i = 0; i < 3; ++i)
            if (run()) return 1;
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) run();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const float plain_us = ms * 1000.f / 20;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(vsom::g_attn_stamps), &stamps, sizeof(void*)));
        CK(hipMemset(stamps, 0, (size_t)grid * 16 * 8));
        CK(hipEventRecord(e0));
        if (run()) return 1;
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(st.data(), stamps, (size_t)grid * 16 * 8, hipMemcpyDeviceToHost));
        printf("[%s] B=%d N=%d H=%d hd=%d mode=%d: %.1f us per launch without stamps\n", which == 0 ? "fwd" : "bwd", B, N, H, hd, mode, plain_us);
        // the rolling kernels run fewer, longer-lived workgroups: count the ones that wrote a stamp
        int live = 0;
        for (int g = 0; g < grid; ++g) live += st[(size_t)g * 16 + 3] != 0;
        report(which == 0 ? "fwd" : "bwd", st, live, ms * 1000.f);
    }
    return 0;
}
