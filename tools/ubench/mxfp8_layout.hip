// Which 32-element K block (= which E8M0 scale) does each 16-byte chunk of a lane's A operand belong to in
// v_mfma_scale_f32_32x32x64_f8f6f4?  A = 1.0 in ONE (lane half, 16-byte chunk) of row 0, B = 1.0 everywhere,
// scale_a = 2^0 for lanes 0-31 and 2^4 for lanes 32-63, scale_b = 1.  D[0][0] = 16 -> the chunk is scaled by the low
// half's scale, 256 -> by the high half's.   hipcc --offload-arch=gfx950 -O2 mxfp8_layout.hip -o mxfp8_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void probe(int half, int chunk, int sa_lo, int sa_hi, float* out) {
    const int lane = threadIdx.x;
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b;
    for (int j = 0; j < 8; ++j) b[j] = 0x38383838;                 // e4m3 1.0
    if ((lane & 31) == 0 && (lane >> 5) == half)
        for (int j = 0; j < 4; ++j) a[chunk * 4 + j] = 0x38383838;
    f32x16 c;
    for (int j = 0; j < 16; ++j) c[j] = 0.f;
    const int sa = (lane >> 5) ? sa_hi : sa_lo, sb = 127;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    if (lane == 0) { out[0] = c[0]; out[1] = c[1]; }
    if (lane == 32) { out[2] = c[0]; }
}
int main() {
    float* d; CK(hipMalloc(&d, 64));
    const int scales[3][2] = {{127, 127}, {127, 131}, {0x7f7f7f7f, (int)0x83838383}};
    for (int s = 0; s < 3; ++s)
        for (int h = 0; h < 2; ++h) for (int c = 0; c < 2; ++c) {
            CK(hipMemset(d, 0, 64));
            probe<<<1, 64>>>(h, c, scales[s][0], scales[s][1], d);
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            float r[3]; CK(hipMemcpy(r, d, 12, hipMemcpyDeviceToHost));
            printf("scales %08x/%08x  lane half %d, 16-byte chunk %d -> D = %g %g | lane32: %g\n", scales[s][0], scales[s][1], h, c, r[0], r[1], r[2]);
        }
    return 0;
}
