// Micro-benchmark: what a cross-stream dependency (hipEventRecord + hipStreamWaitEvent) and a CU-masked stream cost per
// hop on this stack.  Explains why handing the compositing launch to another (masked) stream made the pipelined loop
// 2.3x slower.  Development aid.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/xstream_latency.hip -o /tmp/xs && /tmp/xs
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void spin(unsigned long long ticks, unsigned* sink) {   // s_memrealtime: 100 MHz
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (sink && threadIdx.x == 9999) *sink = 1;
}

static double run(const char* name, hipStream_t a, hipStream_t b, int hops, int us, int grid, bool events) {
    hipEvent_t evs[2], t0, t1;
    CHK(hipEventCreateWithFlags(&evs[0], hipEventDisableTiming));
    CHK(hipEventCreateWithFlags(&evs[1], hipEventDisableTiming));
    CHK(hipEventCreate(&t0));
    CHK(hipEventCreate(&t1));
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        CHK(hipDeviceSynchronize());
        const auto h0 = std::chrono::steady_clock::now();
        CHK(hipEventRecord(t0, a));
        for (int i = 0; i < hops; ++i) {
            spin<<<grid, 256, 0, a>>>((unsigned long long)us * 100ull, nullptr);
            if (events && b != a) {
                CHK(hipEventRecord(evs[0], a));
                CHK(hipStreamWaitEvent(b, evs[0], 0));
            }
            spin<<<grid, 256, 0, b>>>(100ull, nullptr);   // 1 us
            if (events && b != a) {
                CHK(hipEventRecord(evs[1], b));
                CHK(hipStreamWaitEvent(a, evs[1], 0));
            }
        }
        CHK(hipEventRecord(t1, a));
        CHK(hipDeviceSynchronize());
        const double host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, t0, t1));
        const double per = (host_ms * 1e3) / hops - us - 1.0;
        if (per < best) best = per;
        (void)ms;
    }
    printf("%-64s %3d hops of (%3d us kernel + 1 us kernel): overhead %.1f us per hop\n", name, hops, us, best);
    return best;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipStream_t s1, s2, m_all, m_224, m_rest;
    CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    uint32_t all[8], low[8], rest[8];
    for (int i = 0; i < 8; ++i) {
        all[i] = 0xffffffffu;
        low[i] = i < 7 ? 0xffffffffu : 0u;
        rest[i] = i < 7 ? 0u : 0xffffffffu;
    }
    CHK(hipExtStreamCreateWithCUMask(&m_all, 8, all));
    CHK(hipExtStreamCreateWithCUMask(&m_224, 8, low));
    CHK(hipExtStreamCreateWithCUMask(&m_rest, 8, rest));
    for (int us : {20, 100}) {
        for (int grid : {256, 2048}) {
            printf("-- %d us kernels, grid %d x 256\n", us, grid);
            run("same plain stream", s1, s1, 20, us, grid, false);
            run("same MASKED stream (all 256 CUs)", m_all, m_all, 20, us, grid, false);
            run("same MASKED stream (224 CUs)", m_224, m_224, 20, us, grid, false);
            run("plain -> plain, events both ways", s1, s2, 20, us, grid, true);
            run("plain -> masked(all), events both ways", s1, m_all, 20, us, grid, true);
            run("plain -> masked(224), events both ways", s1, m_224, 20, us, grid, true);
            run("masked(rest 32) -> masked(224), events both ways", m_rest, m_224, 20, us, grid, true);
        }
    }
    // concurrency: a long kernel on masked(224) next to short kernels on masked(rest): do the short ones run meanwhile?
    {
        hipEvent_t t0, t1;
        CHK(hipEventCreate(&t0));
        CHK(hipEventCreate(&t1));
        for (hipStream_t other : {s2, m_rest}) {
            CHK(hipDeviceSynchronize());
            spin<<<224 * 8, 256, 0, m_224>>>(2000ull * 100ull, nullptr);   // 2 ms on the 224-CU set
            CHK(hipEventRecord(t0, other));
            for (int i = 0; i < 20; ++i) spin<<<32, 1024, 64 * 1024, other>>>(20ull * 100ull, nullptr);   // 20 x 20 us, 1024-thread WGs with 64 KiB LDS
            CHK(hipEventRecord(t1, other));
            CHK(hipDeviceSynchronize());
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, t0, t1));
            printf("20 x 20-us kernels (32 WGs of 1024 threads, 64 KiB LDS) on %s next to a 2-ms grid on masked(224): %.3f ms\n",
                   other == s2 ? "a plain stream" : "masked(rest 32)", ms);
        }
    }
    return 0;
}
