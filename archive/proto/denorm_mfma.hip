#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__global__ void k(float aval, float bval, float* out) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)aval; b[i] = (_Float16)bval; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; }
}
int main() {
    float* d; hipMalloc(&d, 8);
    float vals[] = {1.0f, 1e-4f, 3e-5f, 1e-6f, 1e-7f};
    for (float v : vals) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, v, 1.0f, d);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a=%g: fp16(a)=%.9g  mfma sum over k=32: %.9g (expected %.9g)\n", v, h[1], h[0], 32.0 * h[1]);
    }
    return 0;
}
