// Calibration of rocprofv3's FETCH_SIZE for GATHERS on gfx950.  The guide calibrates the counter only for
// wide streaming reads (16 B per lane, coalesced): there it reports HALF the bytes.  The TSDF sweep and the compositing stage
// gather 4- to 48-byte pieces; whether "x2" also holds for them decides what `traffic` means for those kernels.
// Every kernel below reads a KNOWN number of bytes from a 1 GiB buffer (4x the 256 MiB Infinity Cache, every address touched
// once: no reuse); run under   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/ubench/gather_fetch
// and compare FETCH_SIZE (KiB) with the byte counts the program prints (tools/ubench/gather_fetch_summary.py).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/gather_fetch.hip -o tools/ubench/gather_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// i -> a permutation of [0, n) (n a power of two): multiplication by an odd constant, then a bit swizzle
__device__ __forceinline__ unsigned scramble(unsigned i, unsigned mask) {
    unsigned x = (i * 2654435761u) & mask;
    x ^= x >> 7;
    return (x * 0x9E3779B1u) & mask;      // odd multiplier: a bijection modulo 2^k (the xorshift is one as well)
}

// stream: lane l of item i reads 16 B at 16 * i (fully coalesced)
__global__ void __launch_bounds__(256) k_stream16(const uint4* __restrict__ p, unsigned n_items, unsigned* out) {
    unsigned acc = 0;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n_items; i += gridDim.x * 256u) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// stream, 4 / 8 bytes per lane (256 / 512 contiguous bytes per wave-load): the TSDF sweep's state reads, the key lists
template <typename T>
__global__ void __launch_bounds__(256) k_stream_small(const T* __restrict__ p, unsigned n_items, unsigned* out) {
    unsigned acc = 0;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n_items; i += gridDim.x * 256u) acc += (unsigned)p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
// gather of PIECE-byte pieces at scattered, PIECE-aligned places (one piece per lane per step; every piece read once)
template <int PIECE>
__global__ void __launch_bounds__(256) k_gather(const unsigned char* __restrict__ p, unsigned n_pieces_mask, unsigned n_items, unsigned* out) {
    unsigned acc = 0;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n_items; i += gridDim.x * 256u) {
        const size_t off = (size_t)scramble(i, n_pieces_mask) * PIECE;
        if (PIECE == 4) acc += *reinterpret_cast<const unsigned*>(p + off);
        else if (PIECE == 8) { const uint2 v = *reinterpret_cast<const uint2*>(p + off); acc += v.x ^ v.y; }
        else if (PIECE == 16) { const uint4 v = *reinterpret_cast<const uint4*>(p + off); acc += v.x ^ v.y ^ v.z ^ v.w; }
        else {
            for (int k = 0; k < PIECE / 16; ++k) { const uint4 v = *reinterpret_cast<const uint4*>(p + off + 16 * k); acc += v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// "patch" gather, the TSDF sweep's shape: a wave reads 64 floats inside a compact 2-D patch of an image (8 x 8 pixels of a row-major
// image of width W), patches at scattered places
__global__ void __launch_bounds__(256) k_patch(const float* __restrict__ img, unsigned W, unsigned n_patch_mask, unsigned n_patches, unsigned* out) {
    float acc = 0.f;
    const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    for (unsigned i = wave; i < n_patches; i += gridDim.x * 4u) {
        const unsigned q = scramble(i, n_patch_mask);
        const unsigned px = (q % (W / 8u)) * 8u, py = (q / (W / 8u)) * 8u;
        acc += img[(size_t)(py + (lane >> 3)) * W + px + (lane & 7u)];
    }
    if (acc == 1.2345f) out[0] = 1;
}

int main() {
    const size_t BYTES = (size_t)1 << 30;
    unsigned char* buf;
    unsigned* out;
    CHK(hipMalloc(&buf, BYTES));
    CHK(hipMalloc(&out, 64));
    CHK(hipMemset(buf, 1, BYTES));
    CHK(hipDeviceSynchronize());
    const int grid = 256 * 16;
    // every kernel touches the whole GiB exactly once except the 4-byte gather and the patches (a quarter: one 4-byte word of
    // every 16 / the whole image)
    printf("kernel,useful_bytes,pieces,piece_bytes\n");
    k_stream16<<<grid, 256>>>((const uint4*)buf, (unsigned)(BYTES / 16), out);
    printf("k_stream16,%zu,%zu,16\n", BYTES, BYTES / 16);
    k_stream_small<unsigned><<<grid, 256>>>((const unsigned*)buf, (unsigned)(BYTES / 4), out);
    printf("k_stream_small<unsigned int>,%zu,%zu,4\n", BYTES, BYTES / 4);
    k_stream_small<unsigned long long><<<grid, 256>>>((const unsigned long long*)buf, (unsigned)(BYTES / 8), out);
    printf("k_stream_small<unsigned long long>,%zu,%zu,8\n", BYTES, BYTES / 8);
    k_gather<4><<<grid, 256>>>(buf, (unsigned)(BYTES / 4 - 1), (unsigned)(BYTES / 16), out);      // a quarter of the 4-byte words
    printf("k_gather<4>,%zu,%zu,4\n", BYTES / 4, BYTES / 16);
    k_gather<8><<<grid, 256>>>(buf, (unsigned)(BYTES / 8 - 1), (unsigned)(BYTES / 16), out);      // half of the 8-byte words
    printf("k_gather<8>,%zu,%zu,8\n", BYTES / 2, BYTES / 16);
    k_gather<16><<<grid, 256>>>(buf, (unsigned)(BYTES / 16 - 1), (unsigned)(BYTES / 16), out);
    printf("k_gather<16>,%zu,%zu,16\n", BYTES, BYTES / 16);
    k_gather<32><<<grid, 256>>>(buf, (unsigned)(BYTES / 32 - 1), (unsigned)(BYTES / 32), out);
    printf("k_gather<32>,%zu,%zu,32\n", BYTES, BYTES / 32);
    k_gather<64><<<grid, 256>>>(buf, (unsigned)(BYTES / 64 - 1), (unsigned)(BYTES / 64), out);
    printf("k_gather<64>,%zu,%zu,64\n", BYTES, BYTES / 64);
    {   // 16384 x 16384 float image = 1 GiB; 8 x 8 patches: 4 M patches, every pixel once
        const unsigned W = 16384;
        k_patch<<<grid, 256>>>((const float*)buf, W, (W / 8) * (W / 8) - 1, (W / 8) * (W / 8), out);
        printf("k_patch,%zu,%u,256\n", BYTES, (W / 8) * (W / 8));
    }
    CHK(hipDeviceSynchronize());
    return 0;
}
