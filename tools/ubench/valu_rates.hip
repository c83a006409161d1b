// Micro-benchmark: sustained issue cost (cycles per wave64 instruction per SIMD) of the VALU operations the compositing
// kernel is made of, at the kernel's occupancy.  Development aid (results quoted in DESIGN.md).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 2048;
#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float s) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float b = s * 0.5f, c = s * 0.25f;
    unsigned long long m0 = 0, m1 = 0;
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0) {  // v_fmac_f32 (VOP2), 8 independent chains
            REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                              "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (MODE == 1) {  // v_fma_f32 (VOP3) with a scalar operand
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c));)
        } else if (MODE == 2) {  // v_exp_f32
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                              "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 3) {  // v_cmp (VOPC -> vcc) + v_cndmask (VOP2 <- vcc): 8 instructions per block
            REP8(asm volatile("v_cmp_le_f32 vcc, %4, %0\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_le_f32 vcc, %4, %1\n v_cndmask_b32 %1, %1, %5, vcc\n"
                              "v_cmp_le_f32 vcc, %4, %2\n v_cndmask_b32 %2, %2, %5, vcc\n v_cmp_le_f32 vcc, %4, %3\n v_cndmask_b32 %3, %3, %5, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");)
        } else if (MODE == 4) {  // v_cmp to an SGPR pair (VOP3) + s_and + v_cndmask from the SGPR pair (VOP3)
            REP8(asm volatile("v_cmp_le_f32 %4, %6, %0\n s_and_b64 %4, %4, exec\n v_cndmask_b32 %0, %0, %7, %4\n"
                              "v_cmp_le_f32 %5, %6, %1\n s_and_b64 %5, %5, exec\n v_cndmask_b32 %1, %1, %7, %5\n"
                              "v_cmp_le_f32 %4, %6, %2\n s_and_b64 %4, %4, exec\n v_cndmask_b32 %2, %2, %7, %4\n"
                              "v_cmp_le_f32 %5, %6, %3\n s_and_b64 %5, %5, exec\n v_cndmask_b32 %3, %3, %7, %5\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(m0), "+s"(m1) : "v"(b), "v"(c) : "scc");)
        } else if (MODE == 5) {  // v_sub + v_mul + v_mov mix
            REP8(asm volatile("v_sub_f32 %0, %0, %4\n v_mul_f32 %1, %1, %5\n v_mov_b32 %2, %0\n v_sub_f32 %3, %3, %4\n"
                              "v_sub_f32 %0, %0, %5\n v_mul_f32 %1, %1, %4\n v_mov_b32 %2, %1\n v_sub_f32 %3, %3, %5\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if (MODE == 6) {  // the accumulate path of the compositing kernel (9 VALU + 4 SALU), two independent copies
            REP8(asm volatile("v_exp_f32 %2, %0\n v_fma_f32 %2, -%1, %2, %1\n v_cmp_gt_f32 vcc, %9, %2\n s_and_b64 %8, vcc, exec\n s_xor_b64 vcc, %8, exec\n"
                              "v_cndmask_b32 %2, %1, %2, vcc\n v_sub_f32 %3, %1, %2\n v_fmac_f32 %4, %10, %3\n v_fmac_f32 %5, %10, %3\n v_fmac_f32 %6, %10, %3\n v_mov_b32 %1, %2\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(m0) : "v"(b), "v"(c) : "vcc", "scc");)
        } else if (MODE == 7) {  // v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(const double*)&b), "v"(*(const double*)&c));)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(m0 + m1);
}

template <int MODE>
static void run(const char* name, int valu_per_block, int wg_per_cu, float* out) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int grid = 256 * wg_per_cu;
    k<MODE><<<grid, 256>>>(out, 1.0f);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    k<MODE><<<grid, 256>>>(out, 1.0f);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    // wave-instructions per SIMD: each WG = 4 waves = one per SIMD
    const double inst_per_simd = (double)wg_per_cu * ITERS * 8.0 * valu_per_block;
    const double ns_per_inst = ms * 1e6 / inst_per_simd;
    printf("%-44s waves/SIMD %d: %.3f ms, %.3f ns per wave64 VALU inst per SIMD = %.2f cycles @2.4 GHz, chip %.0f G inst/s\n", name, wg_per_cu, ms,
           ns_per_inst, ns_per_inst * 2.4, 1024.0 / ns_per_inst);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* out;
    CHK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    for (int w : {1, 2, 4, 7}) {
        run<0>("v_fmac_f32 (VOP2)", 8, w, out);
        run<1>("v_fma_f32 (VOP3, SGPR operand)", 8, w, out);
        run<2>("v_exp_f32", 8, w, out);
        run<3>("v_cmp->vcc + v_cndmask<-vcc", 8, w, out);
        run<4>("v_cmp->sgpr + s_and + v_cndmask<-sgpr", 8, w, out);
        run<5>("v_sub/v_mul/v_mov", 8, w, out);
        run<6>("accumulate path (9 VALU + 2 SALU, dependent)", 9, w, out);
        run<7>("v_pk_fma_f32 (2 fma per lane)", 8, w, out);
    }
    return 0;
}
