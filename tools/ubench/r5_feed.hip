// Micro-benchmark (round 5): what does it cost to FEED the compositing loop its per-instance constants?
// The loop needs ~10 wave-uniform dwords per instance.  Today they arrive by LDS broadcast reads (2 ds_read_b128 + 1
// ds_read_b64: 10 VGPR writes per lane per instance).  r5_rates showed 24 v_fmac + (2 ds_read_b128 + ds_read_b32) at 97
// cycles per SIMD against 61 for the v_fmac alone: this file separates latency from return-path cost and prices the
// alternative -- the constants in SGPRs via scalar loads (s_load_dwordx8 + x4 from a per-wave global ring).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/r5_feed.hip -o tools/ubench/r5_feed && tools/ubench/r5_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 8192;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i8 __attribute__((ext_vector_type(8)));
typedef int i4 __attribute__((ext_vector_type(4)));

#define FMAC24(B, C)                                                                                                 \
    asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"   \
                 "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"   \
                 "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"   \
                 "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"   \
                 "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"   \
                 "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"   \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(B), "v"(C))
// the same 24 with a scalar operand (VOP2 src0 = SGPR)
#define FMAC24S(S, C)                                                                                                \
    asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"   \
                 "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"   \
                 "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"   \
                 "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"   \
                 "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"   \
                 "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"   \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(S), "v"(C))

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, const float* ring, float s) {
    __shared__ f4 lds[4][3 * 64 + 4];
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float b = s * 0.5f, c = s * 0.25f;
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x & 63; i < 3 * 64 + 4; i += 64) lds[wave][i] = f4{a0, a1, a2, a3} * 1e-3f;
    __syncthreads();
    unsigned lds_addr = (unsigned)(size_t)(&lds[wave][0]);
    // per-wave 3-KiB ring of 64 48-byte records in global memory
    const unsigned long long ringv = (unsigned long long)(size_t)(ring) + (size_t)((blockIdx.x * 4 + wave) * 3072);
    const unsigned long long ringw = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ringv >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)ringv);
    f4 r0 = {0, 0, 0, 0}, r1 = r0, q0 = r0, q1 = r0;
    float r2 = 0.f, q2 = 0.f;
    if (MODE == 0) {
        for (int i = 0; i < ITERS; ++i) FMAC24(b, c);
    } else if (MODE == 1) {  // loads, then the work, then wait (r5_rates MODE 13)
        for (int i = 0; i < ITERS; ++i) {
            asm volatile("ds_read_b128 %0, %3\n ds_read_b128 %1, %3 offset:16\n ds_read_b32 %2, %3 offset:32" : "=v"(r0), "=v"(r1), "=v"(r2) : "v"(lds_addr) : "memory");
            FMAC24(b, c);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2));
        }
    } else if (MODE == 2) {  // software-pipelined: the loads of the NEXT block are in flight during this block's work
        asm volatile("ds_read_b128 %0, %3\n ds_read_b128 %1, %3 offset:16\n ds_read_b32 %2, %3 offset:32" : "=v"(r0), "=v"(r1), "=v"(r2) : "v"(lds_addr) : "memory");
        for (int i = 0; i < ITERS; i += 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2));
            asm volatile("ds_read_b128 %0, %3 offset:48\n ds_read_b128 %1, %3 offset:64\n ds_read_b32 %2, %3 offset:80" : "=v"(q0), "=v"(q1), "=v"(q2) : "v"(lds_addr) : "memory");
            FMAC24(r0.x, r2);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(q2));
            asm volatile("ds_read_b128 %0, %3\n ds_read_b128 %1, %3 offset:16\n ds_read_b32 %2, %3 offset:32" : "=v"(r0), "=v"(r1), "=v"(r2) : "v"(lds_addr) : "memory");
            FMAC24(q0.x, q2);
        }
    } else if (MODE == 3) {  // the VGPR writes done by the vector unit instead: 9 v_mov + 24 v_fmac
        for (int i = 0; i < ITERS; ++i) {
            asm volatile("v_mov_b32 %0, %9\n v_mov_b32 %1, %9\n v_mov_b32 %2, %9\n v_mov_b32 %3, %9\n v_mov_b32 %4, %9\n v_mov_b32 %5, %9\n v_mov_b32 %6, %9\n"
                         "v_mov_b32 %7, %9\n v_mov_b32 %8, %9"
                         : "=v"(r0.x), "=v"(r0.y), "=v"(r0.z), "=v"(r0.w), "=v"(r1.x), "=v"(r1.y), "=v"(r1.z), "=v"(r1.w), "=v"(r2) : "v"(b));
            FMAC24(r0.x, r2);
        }
    } else if (MODE == 4 || MODE == 5) {  // constants by scalar loads (x8 + x4 per block), software-pipelined, wait(0) at the top
        i8 sa, sb;
        i4 ta, tb;
        unsigned off = 0;
        asm volatile("s_load_dwordx8 %0, %2, %3\n s_load_dwordx4 %1, %2, %3 offset:32" : "=s"(sa), "=s"(ta) : "s"(ringw), "s"(off) : "memory");
        for (int i = 0; i < ITERS; i += 2) {
            if (MODE == 5 && (i & 63) == 0) {
                // what a re-staged batch costs: one wave-wide store of new records, wait for it, drop the scalar cache
                asm volatile("global_store_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)\n s_dcache_inv" : : "v"(ringv + 2048 + (threadIdx.x & 63) * 16), "v"(r0) : "memory");
            }
            off = (off + 48) % 3072;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(sa), "+s"(ta));
            asm volatile("s_load_dwordx8 %0, %2, %3\n s_load_dwordx4 %1, %2, %3 offset:32" : "=s"(sb), "=s"(tb) : "s"(ringw), "s"(off) : "memory");
            FMAC24S(__builtin_bit_cast(float, sa.x), __builtin_bit_cast(float, ta.x) * c);
            off = (off + 48) % 3072;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(sb), "+s"(tb));
            asm volatile("s_load_dwordx8 %0, %2, %3\n s_load_dwordx4 %1, %2, %3 offset:32" : "=s"(sa), "=s"(ta) : "s"(ringw), "s"(off) : "memory");
            FMAC24S(__builtin_bit_cast(float, sb.x), __builtin_bit_cast(float, tb.x) * c);
        }
    } else if (MODE == 6) {  // like 4 with a quarter of the vector work per block (6 v_fmac): is the scalar feed latency-bound?
        i8 sa, sb;
        i4 ta, tb;
        unsigned off = 0;
        asm volatile("s_load_dwordx8 %0, %2, %3\n s_load_dwordx4 %1, %2, %3 offset:32" : "=s"(sa), "=s"(ta) : "s"(ringw), "s"(off) : "memory");
        for (int i = 0; i < ITERS; i += 2) {
            off = (off + 48) % 3072;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(sa), "+s"(ta));
            asm volatile("s_load_dwordx8 %0, %2, %3\n s_load_dwordx4 %1, %2, %3 offset:32" : "=s"(sb), "=s"(tb) : "s"(ringw), "s"(off) : "memory");
            asm volatile("v_fmac_f32 %0, %6, %7\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7\n v_fmac_f32 %4, %6, %7\n v_fmac_f32 %5, %6, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "s"(sa.x), "v"(c));
            off = (off + 48) % 3072;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(sb), "+s"(tb));
            asm volatile("s_load_dwordx8 %0, %2, %3\n s_load_dwordx4 %1, %2, %3 offset:32" : "=s"(sa), "=s"(ta) : "s"(ringw), "s"(off) : "memory");
            asm volatile("v_fmac_f32 %0, %6, %7\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7\n v_fmac_f32 %4, %6, %7\n v_fmac_f32 %5, %6, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "s"(sb.x), "v"(c));
        }
    } else if (MODE == 7) {  // like 2 (LDS, pipelined) with 6 v_fmac per block
        asm volatile("ds_read_b128 %0, %3\n ds_read_b128 %1, %3 offset:16\n ds_read_b32 %2, %3 offset:32" : "=v"(r0), "=v"(r1), "=v"(r2) : "v"(lds_addr) : "memory");
        for (int i = 0; i < ITERS; i += 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2));
            asm volatile("ds_read_b128 %0, %3 offset:48\n ds_read_b128 %1, %3 offset:64\n ds_read_b32 %2, %3 offset:80" : "=v"(q0), "=v"(q1), "=v"(q2) : "v"(lds_addr) : "memory");
            asm volatile("v_fmac_f32 %0, %6, %7\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7\n v_fmac_f32 %4, %6, %7\n v_fmac_f32 %5, %6, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(r0.x), "v"(r2));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(q2));
            asm volatile("ds_read_b128 %0, %3\n ds_read_b128 %1, %3 offset:16\n ds_read_b32 %2, %3 offset:32" : "=v"(r0), "=v"(r1), "=v"(r2) : "v"(lds_addr) : "memory");
            asm volatile("v_fmac_f32 %0, %6, %7\n v_fmac_f32 %1, %6, %7\n v_fmac_f32 %2, %6, %7\n v_fmac_f32 %3, %6, %7\n v_fmac_f32 %4, %6, %7\n v_fmac_f32 %5, %6, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(q0.x), "v"(q2));
        }
    }
    f4 cs = r0 + r1 + q0 + q1;
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + cs.x + cs.y + cs.z + cs.w + r2 + q2;
}

template <int MODE>
static void run(const char* name, int wg_per_cu, float* out, float* ring) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int grid = 256 * wg_per_cu;
    k<MODE><<<grid, 256>>>(out, ring, 1.0f);
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0));
        k<MODE><<<grid, 256>>>(out, ring, 1.0f);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double blocks_per_simd = (double)wg_per_cu * ITERS;
    printf("%-78s w/SIMD %d: %.3f ms | %7.2f cyc@2.4GHz per block per SIMD\n", name, wg_per_cu, best, best * 1e6 / blocks_per_simd * 2.4);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *out, *ring;
    CHK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    CHK(hipMalloc(&ring, (size_t)256 * 8 * 4 * 3072 + 4096));
    CHK(hipMemset(ring, 0, (size_t)256 * 8 * 4 * 3072 + 4096));
    for (int w : {4, 7}) {
        run<0>("24 v_fmac", w, out, ring);
        run<1>("24 v_fmac + (2 ds_read_b128 + ds_read_b32), wait at the end of the block", w, out, ring);
        run<2>("24 v_fmac + (2 ds_read_b128 + ds_read_b32) one block ahead", w, out, ring);
        run<3>("24 v_fmac + 9 v_mov", w, out, ring);
        run<4>("24 v_fmac(SGPR operand) + (s_load_dwordx8 + x4) one block ahead", w, out, ring);
        run<5>("  ... + store / vmcnt(0) / s_dcache_inv every 64 blocks", w, out, ring);
        run<7>("6 v_fmac + (2 ds_read_b128 + ds_read_b32) one block ahead", w, out, ring);
        run<6>("6 v_fmac(SGPR operand) + (s_load_dwordx8 + x4) one block ahead", w, out, ring);
    }
    return 0;
}
