// Micro-benchmark: scalar-unit issue rates next to the VALU, at the compositing kernel's occupancy.  Question it answers:
// is the compositing kernel (0.88 scalar + branch instructions per VALU instruction) bound by the CU's scalar issue rather
// than by the VALU?  Development aid (results quoted in DESIGN.md).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/salu_rates.hip -o /tmp/salu_rates && /tmp/salu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 1024;
#define REP8(X) X X X X X X X X

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float s, unsigned long long z) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float b = s * 0.5f, c = s * 0.25f;
    unsigned u0 = (unsigned)z, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    unsigned long long m0 = z, m1 = z, m2 = z, m3 = z;   // z == 0 at run time
    if (MODE == 7) a7 = -100.0f;                          // per-pixel threshold: live
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0) {  // 8 independent s_add_u32
            REP8(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                              "s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 3\n s_add_u32 %3, %3, 3\n"
                              : "+s"(u0), "+s"(u1), "+s"(u2), "+s"(u3) : : "scc");)
        } else if (MODE == 1) {  // 64-bit mask ops
            REP8(asm volatile("s_and_b64 %0, %0, exec\n s_andn2_b64 %1, exec, %1\n s_or_b64 %2, %2, %0\n s_and_b64 %3, %3, exec\n"
                              "s_andn2_b64 %0, exec, %0\n s_or_b64 %1, %1, %2\n s_and_b64 %2, %2, exec\n s_andn2_b64 %3, exec, %3\n"
                              : "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : : "scc");)
        } else if (MODE == 2) {  // VALU : SALU 1 : 1 (8 + 8)
            REP8(asm volatile("v_fmac_f32 %0, %8, %9\n s_add_u32 %10, %10, 1\n v_fmac_f32 %1, %8, %9\n s_add_u32 %11, %11, 1\n"
                              "v_fmac_f32 %2, %8, %9\n s_add_u32 %12, %12, 1\n v_fmac_f32 %3, %8, %9\n s_add_u32 %13, %13, 1\n"
                              "v_fmac_f32 %4, %8, %9\n s_add_u32 %10, %10, 3\n v_fmac_f32 %5, %8, %9\n s_add_u32 %11, %11, 3\n"
                              "v_fmac_f32 %6, %8, %9\n s_add_u32 %12, %12, 3\n v_fmac_f32 %7, %8, %9\n s_add_u32 %13, %13, 3\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c),
                                "s"(u0), "s"(u1), "s"(u2), "s"(u3) : "scc");)
        } else if (MODE == 3) {  // VALU : SALU 2 : 1 (8 + 4)
            REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n s_add_u32 %10, %10, 1\n"
                              "v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n s_add_u32 %11, %11, 1\n"
                              "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n s_add_u32 %12, %12, 3\n"
                              "v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n s_add_u32 %13, %13, 3\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c),
                                "s"(u0), "s"(u1), "s"(u2), "s"(u3) : "scc");)
        } else if (MODE == 4) {  // 8 VALU + 4 not-taken scalar branches (s_cmp + s_cbranch_scc1)
            REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n s_cmp_eq_u32 %10, 77\n s_cbranch_scc1 L0_%=\n"
                              "v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n L0_%=:\n s_cmp_eq_u32 %10, 78\n s_cbranch_scc1 L1_%=\n"
                              "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n L1_%=:\n s_cmp_eq_u32 %10, 79\n s_cbranch_scc1 L2_%=\n"
                              "v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n L2_%=:\n s_cmp_eq_u32 %10, 80\n s_cbranch_scc1 L3_%=\n L3_%=:\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c),
                                "s"(u0) : "scc");)
        } else if (MODE == 5) {  // 8 VALU + 4 TAKEN scalar branches (each jumps over nothing)
            REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n s_cmp_lg_u32 %10, 77\n s_cbranch_scc1 L0_%=\n s_nop 0\n"
                              "L0_%=:\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n s_cmp_lg_u32 %10, 78\n s_cbranch_scc1 L1_%=\n s_nop 0\n"
                              "L1_%=:\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n s_cmp_lg_u32 %10, 79\n s_cbranch_scc1 L2_%=\n s_nop 0\n"
                              "L2_%=:\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n s_cmp_lg_u32 %10, 80\n s_cbranch_scc1 L3_%=\n s_nop 0\n L3_%=:\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c),
                                "s"(u0) : "scc");)
        } else if (MODE == 6) {
            // CURRENT per-(instance, quadrant) path of k_blend_wave4e, masks in SGPRs: 2 FMA (q) + 9 VALU + 5 SALU + 2 branches
            //   a0 = q input, a1 = T, a2 = scratch, a3 = w, a4..a6 = C; m0 = dn, m1 = prem, m2 = satm
            REP8(asm volatile("s_bitcmp1_b32 %11, 0\n s_cbranch_scc0 LQ_%=\n"
                              "v_fma_f32 %2, %0, %9, %10\n v_fma_f32 %2, %2, %9, %10\n"
                              "v_cmp_le_f32 vcc, %10, %2\n s_andn2_b64 %7, vcc, %6\n s_cbranch_scc0 LQ_%=\n"
                              "v_exp_f32 %2, %2\n v_fma_f32 %2, -%1, %2, %1\n v_cmp_gt_f32 vcc, %10, %2\n s_and_b64 %8, vcc, %7\n s_andn2_b64 vcc, %7, %8\n"
                              "v_cndmask_b32 %2, %1, %2, vcc\n v_sub_f32 %3, %1, %2\n v_fmac_f32 %4, %10, %3\n v_fmac_f32 %5, %10, %3\n v_fmac_f32 %12, %10, %3\n v_mov_b32 %1, %2\n"
                              "s_or_b64 %13, %13, %8\n LQ_%=:\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+s"(m0), "+s"(m1), "+s"(m2)
                              : "v"(b), "v"(c), "s"(u1), "v"(a6), "s"(m3) : "vcc", "scc");)
        } else if (MODE == 7) {
            // CANDIDATE: no lane masks in SGPRs.  thr (a7) = per-pixel threshold (QMIN live, +inf finished); vcc only.
            //   2 FMA (q) + cmp + exp + cndmask(alpha) + fma + cmp + cndmask(T) + cndmask(thr) + sub + 3 fmac = 13 VALU, 1 SALU, 2 branches
            REP8(asm volatile("s_bitcmp1_b32 %9, 0\n s_cbranch_scc0 LQ_%=\n"
                              "v_fma_f32 %2, %0, %7, %8\n v_fma_f32 %2, %2, %7, %8\n"
                              "v_cmp_ge_f32 vcc, %2, %6\n s_cbranch_vccz LQ_%=\n"
                              "v_exp_f32 %2, %2\n v_cndmask_b32 %2, 0, %2, vcc\n v_fma_f32 %2, -%1, %2, %1\n v_cmp_gt_f32 vcc, %8, %2\n"
                              "v_cndmask_b32 %2, %2, %1, vcc\n v_cndmask_b32 %6, %6, %10, vcc\n v_sub_f32 %3, %1, %2\n v_fmac_f32 %4, %8, %3\n v_fmac_f32 %5, %8, %3\n v_fmac_f32 %11, %8, %3\n v_mov_b32 %1, %2\n"
                              "LQ_%=:\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a7)
                              : "v"(b), "v"(c), "s"(u1), "v"(a6), "v"(a6) : "vcc", "scc");)
        } else if (MODE == 8) {  // v_readfirstlane + dependent s_and (per-instance bookkeeping)
            REP8(asm volatile("v_readfirstlane_b32 %4, %0\n s_and_b32 %4, %4, %5\n v_fmac_f32 %0, %6, %7\n v_fmac_f32 %1, %6, %7\n"
                              "v_readfirstlane_b32 %4, %1\n s_and_b32 %4, %4, %5\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(u0) : "s"(u1), "v"(b), "v"(c) : "scc");)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(m0 + m1 + m2 + m3) + (float)(u0 + u1 + u2 + u3);
}

template <int MODE>
static void run(const char* name, int valu, int salu, int wg_per_cu, float* out) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int grid = 256 * wg_per_cu;
    const float sarg = (MODE == 6 || MODE == 7) ? -1.0f : 1.0f;   // q < 0, alpha < 1, nothing saturates: the full path runs every time
    k<MODE><<<grid, 256>>>(out, sarg, 0ull);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    k<MODE><<<grid, 256>>>(out, sarg, 0ull);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double blocks_per_simd = (double)wg_per_cu * ITERS * 8.0;     // asm blocks issued per SIMD (one wave of each WG per SIMD)
    const double ns_per_block = ms * 1e6 / blocks_per_simd;
    const double cyc = ns_per_block * 2.4;
    printf("%-52s w/SIMD %d: %.3f ms | %6.2f cyc per block per SIMD | VALU %2d SALU+br %2d | per CU: %.2f VALU/cyc %.2f scalar/cyc\n", name,
           wg_per_cu, ms, cyc, valu, salu, 4.0 * valu / cyc, 4.0 * salu / cyc);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* out;
    CHK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    for (int w : {1, 4, 7}) {
        run<0>("8 s_add_u32", 0, 8, w, out);
        run<1>("8 s_and/andn2/or_b64", 0, 8, w, out);
        run<2>("8 v_fmac + 8 s_add (1:1)", 8, 8, w, out);
        run<3>("8 v_fmac + 4 s_add (2:1)", 8, 4, w, out);
        run<4>("8 v_fmac + 4 (s_cmp + branch not taken)", 8, 8, w, out);
        run<5>("8 v_fmac + 4 (s_cmp + branch TAKEN)", 8, 8, w, out);
        run<6>("blend quadrant path, SGPR lane masks (now)", 11, 8, w, out);
        run<7>("blend quadrant path, VCC only (candidate)", 13, 3, w, out);
        run<8>("2 (readfirstlane + s_and) + 4 v_fmac", 6, 2, w, out);
    }
    return 0;
}
