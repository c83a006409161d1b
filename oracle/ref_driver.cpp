// ref_driver.cpp -- runs the REFERENCE's own rasteriser kernels on the CPU (TEST INFRASTRUCTURE ONLY).
//
// Built by oracle/build_ref.py into oracle/_ref/libref_raster.so when /root/reference is present (the build
// container); never shipped, never on the product path.  The kernel text is #included from a temporary file
// that build_ref.py cuts out of the reference sources where they lie
//   third_party/gaussian-splatting/submodules/diff-gaussian-rasterization/cuda_rasterizer/forward.cu   (device part)
//   .../rasterizer_impl.cu: checkFrustum, duplicateWithKeys, identifyTileRanges
// and compiled against oracle/ref_shim/ (CUDA-on-CPU shim).  The host sequence below restates
// CudaRasterizer::Rasterizer::forward (rasterizer_impl.cu:198-336): preprocess -> inclusive sum of tiles_touched
// -> duplicateWithKeys -> stable sort of the 64-bit keys (= cub::DeviceRadixSort::SortPairs) -> identifyTileRanges
// -> render, with the reference's own launch shapes.
#include <ucontext.h>

#include <algorithm>
#include <functional>
#include <numeric>
#include <vector>

#include "cuda_runtime.h"

namespace refshim {
thread_local ThreadCoords tc;

struct BlockRun {
    ucontext_t sched;
    std::vector<ucontext_t> ctx;
    std::vector<char> done;
    std::vector<char> stacks;
    const std::function<void()>* body = nullptr;
    int cur = 0, acc = 0, result = 0;
};
static thread_local BlockRun* cur_run = nullptr;
static const size_t kStack = 128 * 1024;

int barrier(int pred) {
    BlockRun* r = cur_run;
    if (!r) return pred ? 1 : 0;  // kernel launched without fibers (it has no barriers)
    r->acc += pred ? 1 : 0;
    swapcontext(&r->ctx[r->cur], &r->sched);
    return r->result;
}
static void fiber_main() {
    BlockRun* r = cur_run;
    (*r->body)();
    r->done[r->cur] = 1;
    swapcontext(&r->ctx[r->cur], &r->sched);
}
static void run_block_fibers(BlockRun& r, dim3 block, const std::function<void()>& body) {
    const int n = (int)(block.x * block.y * block.z);
    r.ctx.resize(n);
    r.done.assign(n, 0);
    if (r.stacks.size() < (size_t)n * kStack) r.stacks.resize((size_t)n * kStack);
    r.body = &body;
    for (int t = 0; t < n; ++t) {
        getcontext(&r.ctx[t]);
        r.ctx[t].uc_stack.ss_sp = r.stacks.data() + (size_t)t * kStack;
        r.ctx[t].uc_stack.ss_size = kStack;
        r.ctx[t].uc_link = nullptr;
        makecontext(&r.ctx[t], fiber_main, 0);
    }
    cur_run = &r;
    int alive = n;
    while (alive > 0) {
        r.acc = 0;
        for (int t = 0; t < n; ++t) {
            if (r.done[t]) continue;
            r.cur = t;
            tc.tid.x = (unsigned)t % block.x;
            tc.tid.y = ((unsigned)t / block.x) % block.y;
            tc.tid.z = (unsigned)t / (block.x * block.y);
            swapcontext(&r.sched, &r.ctx[t]);
            if (r.done[t]) --alive;
        }
        r.result = r.acc;  // every live thread has reached the same barrier
    }
    cur_run = nullptr;
}

// launch(grid, block, body): blocks in parallel over OS threads; threads of a block as fibers when the kernel
// synchronises, as a plain loop otherwise
static void launch(dim3 grid, dim3 block, bool has_barriers, const std::function<void()>& body) {
    const long long nblocks = (long long)grid.x * grid.y * grid.z;
#pragma omp parallel
    {
        BlockRun run;
#pragma omp for schedule(dynamic, 4)
        for (long long b = 0; b < nblocks; ++b) {
            tc.gdim = grid;
            tc.bdim = block;
            tc.bid.x = (unsigned)(b % grid.x);
            tc.bid.y = (unsigned)((b / grid.x) % grid.y);
            tc.bid.z = (unsigned)(b / ((long long)grid.x * grid.y));
            if (has_barriers) {
                run_block_fibers(run, block, body);
            } else {
                for (unsigned z = 0; z < block.z; ++z)
                    for (unsigned y = 0; y < block.y; ++y)
                        for (unsigned x = 0; x < block.x; ++x) {
                            tc.tid.x = x;
                            tc.tid.y = y;
                            tc.tid.z = z;
                            body();
                        }
            }
        }
    }
}
}  // namespace refshim

#include REF_KERNELS_INC

extern "C" {

// checkFrustum (rasterizer_impl.cu:54-66) == Rasterizer::markVisible
void ref_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, unsigned char* present) {
    std::vector<char> tmp(P > 0 ? P : 1);
    bool* pres = reinterpret_cast<bool*>(tmp.data());
    refshim::launch(dim3((P + 255) / 256), dim3(256), false,
                    [&] { checkFrustum(P, means3D, viewmatrix, projmatrix, pres); });
    for (int i = 0; i < P; ++i) present[i] = pres[i] ? 1 : 0;
}

// Rasterizer::forward (rasterizer_impl.cu:198-336).  Outputs: image [3,H,W], radii[P] and the per-Gaussian
// geometry buffers (means2D[P,2], depths[P], cov3D[P,6], rgb[P,3], conic_opacity[P,4], tiles_touched[P]),
// point_list[cap] (sorted Gaussian ids), ranges[tiles,2].  Returns num_rendered (or -1 if cap is too small).
long long ref_forward(int P, int D, int M, const float* background, int width, int height, const float* means3D,
                      const float* shs, const float* colors_precomp, const float* opacities, const float* scales,
                      float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                      const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
                      float* out_color, int* radii, float* means2D, float* depths, float* cov3D, float* rgb,
                      float* conic_opacity, unsigned* tiles_touched, unsigned* point_list, long long cap,
                      unsigned* ranges) {
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);
    dim3 tile_grid((width + BLOCK_X - 1) / BLOCK_X, (height + BLOCK_Y - 1) / BLOCK_Y, 1);
    dim3 block(BLOCK_X, BLOCK_Y, 1);
    std::vector<char> clamped((size_t)(P > 0 ? P : 1) * 3);
    for (int i = 0; i < P; ++i) radii[i] = 0;
    refshim::launch(dim3((P + 255) / 256), dim3(256), false, [&] {
        preprocessCUDA<NUM_CHANNELS>(P, D, M, means3D, (const glm::vec3*)scales, scale_modifier, (const glm::vec4*)rotations,
                                     opacities, shs, reinterpret_cast<bool*>(clamped.data()), cov3D_precomp, colors_precomp,
                                     viewmatrix, projmatrix, (const glm::vec3*)cam_pos, width, height, tan_fovx, tan_fovy,
                                     focal_x, focal_y, radii, (float2*)means2D, depths, cov3D, rgb, (float4*)conic_opacity,
                                     tile_grid, tiles_touched, prefiltered != 0);
    });
    std::vector<uint32_t> offsets(P > 0 ? P : 1);
    std::partial_sum(tiles_touched, tiles_touched + P, offsets.begin());  // cub::DeviceScan::InclusiveSum
    const long long num_rendered = P > 0 ? (long long)offsets[P - 1] : 0;
    const size_t ntiles = (size_t)tile_grid.x * tile_grid.y;
    memset(ranges, 0, ntiles * 2 * sizeof(unsigned));
    if (num_rendered > cap) return -1;
    std::vector<uint64_t> keys((size_t)num_rendered + 1);
    std::vector<uint32_t> vals((size_t)num_rendered + 1);
    refshim::launch(dim3((P + 255) / 256), dim3(256), false, [&] {
        duplicateWithKeys(P, (const float2*)means2D, depths, offsets.data(), keys.data(), vals.data(), radii, tile_grid);
    });
    // cub::DeviceRadixSort::SortPairs over the low 32 + bit(#tiles) key bits: a stable sort by key
    std::vector<uint32_t> order((size_t)num_rendered);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
    std::vector<uint64_t> skeys((size_t)num_rendered + 1);
    for (long long i = 0; i < num_rendered; ++i) {
        skeys[i] = keys[order[i]];
        point_list[i] = vals[order[i]];
    }
    if (num_rendered > 0)
        refshim::launch(dim3((unsigned)((num_rendered + 255) / 256)), dim3(256), false,
                        [&] { identifyTileRanges((int)num_rendered, skeys.data(), (uint2*)ranges); });
    std::vector<float> final_T((size_t)width * height);
    std::vector<uint32_t> n_contrib((size_t)width * height);
    const float* feature_ptr = colors_precomp != nullptr ? colors_precomp : rgb;
    refshim::launch(tile_grid, block, true, [&] {
        renderCUDA<NUM_CHANNELS>((const uint2*)ranges, point_list, width, height, (const float2*)means2D, feature_ptr,
                                 (const float4*)conic_opacity, final_T.data(), n_contrib.data(), background, out_color);
    });
    return num_rendered;
}

}  // extern "C"
