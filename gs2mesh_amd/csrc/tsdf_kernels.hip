// tsdf_kernels.hip -- TSDF integration stage (compiled with -ffp-contract=off).
#include "tsdf_kernels.h"
#include "tsdf_internal.h"
#include <math.h>

void gs2m_launch_tsdf_touch(hipStream_t st, const TsdfVolume& V, const TsdfFrame& f, const float* depth,
                            const unsigned char* mask) {
    const int n = f.nx * f.ny;
    GS2M_LAUNCH(k_tsdf_touch, dim3((n + 255) / 256), dim3(256), 0, st, V, f, depth, mask);
    GS2M_LAUNCH(k_tsdf_compact, dim3((V.hash_cap + 1023u) / 1024u), dim3(1024), 0, st, V, f.frame_id);
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
