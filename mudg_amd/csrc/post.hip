// post.hip — per-modality post-processing of decoded frames (byte / integer work, HBM-bound):
//   frames_to_u8      clamp, (x + 1) / 2 * 255, truncate, (b c t h w) fp32 -> (b t h w c) uint8   eval_tools.py:22-27
//   depth_from_u8     mean of the three uint8 channels / 255                                      eval_tools.py:71
//   semantic_nearest  nearest of the 19 palette colours, first minimum wins                       eval_tools.py:309-347
// The arithmetic reproduces the reference's fp32 / integer operations one for one: results are bit-equal.
#include "common.h"

namespace {

__constant__ int PAL[19][3] = {{255, 120, 50}, {255, 192, 203}, {255, 255, 0}, {0, 150, 245}, {0, 255, 255}, {255, 127, 0},
                               {255, 0, 0}, {255, 240, 150}, {135, 60, 0}, {160, 32, 240}, {255, 0, 255}, {139, 137, 137},
                               {75, 0, 75}, {150, 240, 80}, {230, 230, 250}, {0, 175, 0}, {0, 255, 127}, {222, 155, 161},
                               {140, 62, 69}};

__global__ __launch_bounds__(256) void frames_to_u8_kernel(const float* __restrict__ V, uint8_t* __restrict__ O, int C, int T,
                                                            int64_t HW, int64_t total) {
    // one thread per output pixel (b, t, p): reads C planes HW*T apart, writes C consecutive bytes
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t p = i % HW, bt = i / HW;
    const int t = (int)(bt % T);
    const int64_t b = bt / T;
    for (int c = 0; c < C; ++c) {
        float x = V[((b * C + c) * T + t) * HW + p];
        x = fminf(fmaxf(x, -1.0f), 1.0f);
        const float g = __fmul_rn(__fdiv_rn(__fadd_rn(x, 1.0f), 2.0f), 255.0f);
        O[i * C + c] = (uint8_t)(int)g;                      // truncation toward zero, g in [0, 255]
    }
}

__global__ __launch_bounds__(256) void depth_from_u8_kernel(const uint8_t* __restrict__ F, float* __restrict__ D, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float s = __fadd_rn(__fadd_rn((float)F[3 * i], (float)F[3 * i + 1]), (float)F[3 * i + 2]);
    D[i] = __fdiv_rn(__fdiv_rn(s, 3.0f), 255.0f);
}

__global__ __launch_bounds__(256) void semantic_kernel(const uint8_t* __restrict__ I, uint8_t* __restrict__ Vis,
                                                        int64_t* __restrict__ L, int64_t hw) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= hw) return;
    const int r = I[i], g = I[hw + i], b = I[2 * hw + i];
    int best = 0, bd = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < 19; ++k) {
        const int dr = r - PAL[k][0], dg = g - PAL[k][1], db = b - PAL[k][2];
        const int d = dr * dr + dg * dg + db * db;
        if (d < bd) { bd = d; best = k; }                    // strict: the first minimum wins, like np.argmin
    }
    L[i] = best;
    Vis[i] = (uint8_t)PAL[best][0]; Vis[hw + i] = (uint8_t)PAL[best][1]; Vis[2 * hw + i] = (uint8_t)PAL[best][2];
}

}  // namespace

extern "C" int mudg_frames_to_u8(const float* video, uint8_t* out, int B, int C, int T, int64_t HW, void* stream) {
    MUDG_REQUIRE(video && out && B > 0 && C > 0 && T > 0 && HW > 0, "mudg_frames_to_u8: bad arguments");
    const int64_t total = (int64_t)B * T * HW;
    hipLaunchKernelGGL(frames_to_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       video, out, C, T, HW, total);
    return mudg_check_launch("mudg_frames_to_u8");
}

extern "C" int mudg_depth_from_u8(const uint8_t* frames, float* depth, int64_t pixels, void* stream) {
    MUDG_REQUIRE(frames && depth && pixels > 0, "mudg_depth_from_u8: bad arguments");
    hipLaunchKernelGGL(depth_from_u8_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       frames, depth, pixels);
    return mudg_check_launch("mudg_depth_from_u8");
}

extern "C" int mudg_semantic_nearest(const uint8_t* img, uint8_t* vis, int64_t* labels, int64_t hw, void* stream) {
    MUDG_REQUIRE(img && vis && labels && hw > 0, "mudg_semantic_nearest: bad arguments");
    hipLaunchKernelGGL(semantic_kernel, dim3((unsigned)((hw + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       img, vis, labels, hw);
    return mudg_check_launch("mudg_semantic_nearest");
}
