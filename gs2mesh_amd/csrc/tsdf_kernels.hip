// tsdf_kernels.hip -- TSDF integration stage (compiled with -ffp-contract=off).
#include "tsdf_kernels.h"
#include "tsdf_internal.h"
#include <math.h>

void gs2m_launch_tsdf_touch(hipStream_t st, const TsdfVolume& V, const TsdfFrame& f, const float* depth,
                            const unsigned char* mask) {
    // blocks per axis the +-trunc box of a point can span
    int span = (int)floor(2.0 * f.sdf_trunc / f.unit_length) + 2;
    if (span < 2) span = 2;
    const long long n = (long long)f.nx * f.ny * span * span * span;
    GS2M_LAUNCH(k_tsdf_touch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, V, f, depth, mask, span);
}
void gs2m_launch_tsdf_integrate(hipStream_t st, int n_wg, const TsdfVolume& V, const TsdfFrame& f,
                                const float* depth, const unsigned char* color, const unsigned char* mask) {
    GS2M_LAUNCH(k_tsdf_integrate, dim3(n_wg), dim3(256), 0, st, V, f, depth, color, mask);
}
void gs2m_launch_tsdf_pack(hipStream_t st, unsigned n, const TsdfVolume& V, const int* keys, float* wsum,
                           float* weight, unsigned* rgb) {
    GS2M_LAUNCH(k_tsdf_pack, dim3(n), dim3(256), 0, st, V, keys, wsum, weight, rgb);
}
void gs2m_launch_tsdf_unpack(hipStream_t st, unsigned n, const TsdfVolume& V, const int* keys, const float* wsum,
                             const float* weight, const unsigned* rgb) {
    GS2M_LAUNCH(k_tsdf_unpack, dim3(n), dim3(256), 0, st, V, keys, wsum, weight, rgb);
}
