// Micro-benchmark (round 5): questions the compositing rewrite hangs on, at the kernel's occupancy (7 waves per SIMD).
//   1. does a f32-input multi-block MFMA (v_mfma_f32_4x4x1_16b_f32: 64 lanes x 4 rows = the colour accumulation of one
//      instance for 64 pixels) run BESIDE vector FMAs of other waves, or do they share the pipe?
//   2. does a wave64 VALU instruction get cheaper when one 32-lane half of EXEC is empty?
//   3. the separate prices of v_cmp (-> vcc), v_cndmask (<- vcc), v_cmpx, v_min / v_max / v_med3, v_readfirstlane;
//   4. ds_read_b128 with one address for all lanes (the per-instance broadcast), alone and beside VALU work.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/r5_rates.hip -o tools/ubench/r5_rates && tools/ubench/r5_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 1024;
#define REP8(X) X X X X X X X X
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float s) {
    __shared__ f4 lds[256];
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float b = s * 0.5f, c = s * 0.25f;
    f4 c0 = {a0, a1, a2, a3}, c1 = c0 + 1.f, c2 = c0 + 2.f, c3 = c0 + 3.f, c4 = c0 + 4.f, c5 = c0 + 5.f, c6 = c0 + 6.f, c7 = c0 + 7.f;
    f4 r0 = c0, r1 = c0, r2 = c0, r3 = c0;
    lds[threadIdx.x] = c0;
    __syncthreads();
    const unsigned lds_addr = (unsigned)(size_t)(&lds[(threadIdx.x >> 6) * 64]);   // one address per wave
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0) {  // 8 v_fmac
            REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                              "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (MODE == 1) {  // 8 MFMA 4x4x1 (16 blocks) on 8 independent accumulators
            REP8(asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %8, %9, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %8, %9, %1\n"
                              "v_mfma_f32_4x4x1_16b_f32 %2, %8, %9, %2\n v_mfma_f32_4x4x1_16b_f32 %3, %8, %9, %3\n"
                              "v_mfma_f32_4x4x1_16b_f32 %4, %8, %9, %4\n v_mfma_f32_4x4x1_16b_f32 %5, %8, %9, %5\n"
                              "v_mfma_f32_4x4x1_16b_f32 %6, %8, %9, %6\n v_mfma_f32_4x4x1_16b_f32 %7, %8, %9, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(b), "v"(c));)
        } else if (MODE == 2) {  // 24 v_fmac + 8 MFMA, interleaved 3 : 1 (the colour FMAs moved to the matrix pipe)
            REP8(asm volatile("v_fmac_f32 %0, %12, %13\n v_fmac_f32 %1, %12, %13\n v_fmac_f32 %2, %12, %13\n v_mfma_f32_4x4x1_16b_f32 %8, %12, %13, %8\n"
                              "v_fmac_f32 %3, %12, %13\n v_fmac_f32 %4, %12, %13\n v_fmac_f32 %5, %12, %13\n v_mfma_f32_4x4x1_16b_f32 %9, %12, %13, %9\n"
                              "v_fmac_f32 %6, %12, %13\n v_fmac_f32 %7, %12, %13\n v_fmac_f32 %0, %12, %13\n v_mfma_f32_4x4x1_16b_f32 %10, %12, %13, %10\n"
                              "v_fmac_f32 %1, %12, %13\n v_fmac_f32 %2, %12, %13\n v_fmac_f32 %3, %12, %13\n v_mfma_f32_4x4x1_16b_f32 %11, %12, %13, %11\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                              : "v"(b), "v"(c));)
        } else if (MODE == 3) {  // 12 v_fmac only (the vector part of MODE 2's block x 4)
            REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n"
                              "v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n"
                              "v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n v_fmac_f32 %0, %8, %9\n"
                              "v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (MODE == 4) {  // 4 MFMA only (the matrix part of MODE 2's block)
            REP8(asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %4, %5, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %4, %5, %1\n"
                              "v_mfma_f32_4x4x1_16b_f32 %2, %4, %5, %2\n v_mfma_f32_4x4x1_16b_f32 %3, %4, %5, %3\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(b), "v"(c));)
        } else if (MODE == 5 || MODE == 6) {  // 8 v_fmac with 32 (MODE 5) / 16 (MODE 6) active lanes
            const unsigned long long m = MODE == 5 ? 0xffffffffull : 0xffffull;
            REP8(asm volatile("s_mov_b64 exec, %10\n"
                              "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                              "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                              "s_mov_b64 exec, -1\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(m));)
        } else if (MODE == 7) {  // 8 v_cmp -> vcc
            REP8(asm volatile("v_cmp_le_f32 vcc, %4, %0\n v_cmp_le_f32 vcc, %4, %1\n v_cmp_le_f32 vcc, %4, %2\n v_cmp_le_f32 vcc, %4, %3\n"
                              "v_cmp_le_f32 vcc, %5, %0\n v_cmp_le_f32 vcc, %5, %1\n v_cmp_le_f32 vcc, %5, %2\n v_cmp_le_f32 vcc, %5, %3\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");)
        } else if (MODE == 8) {  // 8 v_cndmask <- vcc (vcc fixed)
            REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                              "v_cndmask_b32 %0, %0, %5, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %5, vcc\n v_cndmask_b32 %3, %3, %5, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");)
        } else if (MODE == 9) {  // 8 v_cmpx (always true: EXEC stays full)
            REP8(asm volatile("v_cmpx_le_f32 %0, %0\n v_cmpx_le_f32 %1, %1\n v_cmpx_le_f32 %2, %2\n v_cmpx_le_f32 %3, %3\n"
                              "v_cmpx_le_f32 %0, %0\n v_cmpx_le_f32 %1, %1\n v_cmpx_le_f32 %2, %2\n v_cmpx_le_f32 %3, %3\n"
                              "s_mov_b64 exec, -1\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
        } else if (MODE == 10) {  // v_min / v_max / v_med3
            REP8(asm volatile("v_min_f32 %0, %0, %4\n v_max_f32 %1, %1, %5\n v_med3_f32 %2, %2, %4, %5\n v_min_f32 %3, %3, %4\n"
                              "v_max_f32 %0, %0, %5\n v_med3_f32 %1, %1, %4, %5\n v_min_f32 %2, %2, %5\n v_max_f32 %3, %3, %4\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if (MODE == 11) {  // 8 v_readfirstlane
            unsigned s0, s1, s2, s3;
            REP8(asm volatile("v_readfirstlane_b32 %0, %4\n v_readfirstlane_b32 %1, %5\n v_readfirstlane_b32 %2, %6\n v_readfirstlane_b32 %3, %7\n"
                              "v_readfirstlane_b32 %0, %5\n v_readfirstlane_b32 %1, %6\n v_readfirstlane_b32 %2, %7\n v_readfirstlane_b32 %3, %4\n"
                              : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
        } else if (MODE == 12) {  // 8 ds_read_b128, one address for all lanes
            REP8(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n"
                              "ds_read_b128 %0, %4 offset:64\n ds_read_b128 %1, %4 offset:80\n ds_read_b128 %2, %4 offset:96\n ds_read_b128 %3, %4 offset:112\n"
                              "s_waitcnt lgkmcnt(0)\n"
                              : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(lds_addr) : "memory");)
        } else if (MODE == 13) {  // 2 ds_read_b128 (uniform) + 1 ds_read_b32 + 24 v_fmac: the per-instance mix of the rewrite
            REP8(asm volatile("ds_read_b128 %8, %12\n ds_read_b128 %9, %12 offset:16\n ds_read_b32 %10, %12 offset:32\n"
                              "v_fmac_f32 %0, %13, %14\n v_fmac_f32 %1, %13, %14\n v_fmac_f32 %2, %13, %14\n v_fmac_f32 %3, %13, %14\n"
                              "v_fmac_f32 %4, %13, %14\n v_fmac_f32 %5, %13, %14\n v_fmac_f32 %6, %13, %14\n v_fmac_f32 %7, %13, %14\n"
                              "v_fmac_f32 %0, %13, %14\n v_fmac_f32 %1, %13, %14\n v_fmac_f32 %2, %13, %14\n v_fmac_f32 %3, %13, %14\n"
                              "v_fmac_f32 %4, %13, %14\n v_fmac_f32 %5, %13, %14\n v_fmac_f32 %6, %13, %14\n v_fmac_f32 %7, %13, %14\n"
                              "v_fmac_f32 %0, %13, %14\n v_fmac_f32 %1, %13, %14\n v_fmac_f32 %2, %13, %14\n v_fmac_f32 %3, %13, %14\n"
                              "v_fmac_f32 %4, %13, %14\n v_fmac_f32 %5, %13, %14\n v_fmac_f32 %6, %13, %14\n v_fmac_f32 %7, %13, %14\n"
                              "s_waitcnt lgkmcnt(0)\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=v"(r0), "=v"(r1), "=v"(r2.x), "=v"(r3)
                              : "v"(lds_addr), "v"(b), "v"(c) : "memory");)
        } else if (MODE == 14) {  // 8 v_exp + 8 MFMA interleaved (transcendental beside the matrix pipe)
            REP8(asm volatile("v_exp_f32 %0, %0\n v_mfma_f32_4x4x1_16b_f32 %8, %12, %13, %8\n v_exp_f32 %1, %1\n v_mfma_f32_4x4x1_16b_f32 %9, %12, %13, %9\n"
                              "v_exp_f32 %2, %2\n v_mfma_f32_4x4x1_16b_f32 %10, %12, %13, %10\n v_exp_f32 %3, %3\n v_mfma_f32_4x4x1_16b_f32 %11, %12, %13, %11\n"
                              "v_exp_f32 %4, %4\n v_mfma_f32_4x4x1_16b_f32 %8, %12, %13, %8\n v_exp_f32 %5, %5\n v_mfma_f32_4x4x1_16b_f32 %9, %12, %13, %9\n"
                              "v_exp_f32 %6, %6\n v_mfma_f32_4x4x1_16b_f32 %10, %12, %13, %10\n v_exp_f32 %7, %7\n v_mfma_f32_4x4x1_16b_f32 %11, %12, %13, %11\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                              : "v"(b), "v"(c));)
        } else if (MODE == 15) {  // new accumulate path: exp, min, fma, cmp, cndmask, sub, MFMA, mov  (7 VALU + 1 MFMA), two-instance chains
            REP8(asm volatile("v_exp_f32 %2, %0\n v_min_f32 %2, %2, %9\n v_fma_f32 %2, -%1, %2, %1\n v_cmp_gt_f32 vcc, %10, %2\n"
                              "v_cndmask_b32 %2, %2, %1, vcc\n v_sub_f32 %3, %1, %2\n v_mfma_f32_4x4x1_16b_f32 %4, %9, %3, %4\n v_mov_b32 %1, %2\n"
                              "v_exp_f32 %6, %0\n v_min_f32 %6, %6, %9\n v_fma_f32 %6, -%5, %6, %5\n v_cmp_gt_f32 vcc, %10, %6\n"
                              "v_cndmask_b32 %6, %6, %5, vcc\n v_sub_f32 %7, %5, %6\n v_mfma_f32_4x4x1_16b_f32 %8, %9, %7, %8\n v_mov_b32 %5, %6\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(c1) : "v"(b), "v"(c) : "vcc");)
        } else if (MODE == 16) {  // old accumulate path: exp, fma, cmp, cndmask, sub, 3 fmac, mov (9 VALU), two-instance chains
            REP8(asm volatile("v_exp_f32 %2, %0\n v_fma_f32 %2, -%1, %2, %1\n v_cmp_gt_f32 vcc, %10, %2\n"
                              "v_cndmask_b32 %2, %2, %1, vcc\n v_sub_f32 %3, %1, %2\n v_fmac_f32 %4, %9, %3\n v_fmac_f32 %5, %9, %3\n v_fmac_f32 %6, %9, %3\n v_mov_b32 %1, %2\n"
                              "v_exp_f32 %2, %0\n v_fma_f32 %2, -%7, %2, %7\n v_cmp_gt_f32 vcc, %10, %2\n"
                              "v_cndmask_b32 %2, %2, %7, vcc\n v_sub_f32 %3, %7, %2\n v_fmac_f32 %4, %9, %3\n v_fmac_f32 %5, %9, %3\n v_fmac_f32 %6, %9, %3\n v_mov_b32 %7, %2\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(c1) : "v"(b), "v"(c) : "vcc");)
        }
    }
    f4 cs = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + r0 + r1 + r2 + r3;
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + cs.x + cs.y + cs.z + cs.w;
}

template <int MODE>
static void run(const char* name, double inst_per_block, int wg_per_cu, float* out) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int grid = 256 * wg_per_cu;
    k<MODE><<<grid, 256>>>(out, 1.0f);
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0));
        k<MODE><<<grid, 256>>>(out, 1.0f);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    // blocks per SIMD: each workgroup = 4 waves = one per SIMD
    const double blocks_per_simd = (double)wg_per_cu * ITERS * 8.0;
    const double ns_per_block = best * 1e6 / blocks_per_simd;
    printf("%-58s w/SIMD %d: %.3f ms | %7.2f cyc@2.4GHz per block per SIMD | %5.2f cyc per instruction (%g per block)\n", name, wg_per_cu, best,
           ns_per_block * 2.4, ns_per_block * 2.4 / inst_per_block, inst_per_block);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* out;
    CHK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    for (int w : {1, 4, 7}) {
        run<0>("8 v_fmac", 8, w, out);
        run<1>("8 mfma_4x4x1_16b_f32 (8 accumulators)", 8, w, out);
        run<3>("12 v_fmac", 12, w, out);
        run<4>("4 mfma_4x4x1_16b_f32", 4, w, out);
        run<2>("12 v_fmac + 4 mfma interleaved (sum or max of the two?)", 16, w, out);
        run<5>("8 v_fmac, EXEC = low 32 lanes (+2 s_mov exec)", 8, w, out);
        run<6>("8 v_fmac, EXEC = low 16 lanes (+2 s_mov exec)", 8, w, out);
        run<7>("8 v_cmp -> vcc", 8, w, out);
        run<8>("8 v_cndmask <- vcc", 8, w, out);
        run<9>("8 v_cmpx (+1 s_mov exec)", 8, w, out);
        run<10>("8 v_min/v_max/v_med3", 8, w, out);
        run<11>("8 v_readfirstlane", 8, w, out);
        run<12>("8 ds_read_b128 uniform address + wait", 8, w, out);
        run<13>("2 ds_read_b128 + ds_read_b32 + 24 v_fmac + wait", 27, w, out);
        run<14>("8 v_exp + 8 mfma interleaved", 16, w, out);
        run<16>("old accumulate path x2 (18 VALU)", 18, w, out);
        run<15>("new accumulate path x2 (14 VALU + 2 MFMA)", 16, w, out);
    }
    return 0;
}
