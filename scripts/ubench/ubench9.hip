// scripts/ubench/ubench9.hip -- what does HBM take for the traffic shape of gauss_grad_march (1 B/px read, 8 B/px written) and
// of its possible replacement (1 B/px read, 4 B/px written), with no arithmetic at all?  32 frames of 3840 x 2160.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int PLANES, bool NT>
__global__ void __launch_bounds__(256) k(const unsigned *__restrict__ in, float *__restrict__ a, float *__restrict__ b, size_t quads)
{
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (size_t)gridDim.x * 256) {
        const unsigned w = in[q];  // 4 pixels
        const v4f v = {(float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u), (float)(w >> 24)};
        if (NT) {
            __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(a) + q);
            if (PLANES == 2) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(b) + q);
        } else {
            reinterpret_cast<v4f *>(a)[q] = v;
            if (PLANES == 2) reinterpret_cast<v4f *>(b)[q] = v;
        }
    }
}
template <int PLANES, bool NT>
static void run(const char *name, const unsigned *in, float *a, float *b, size_t quads, int blocks)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<PLANES, NT>), dim3(blocks), dim3(256), 0, 0, in, a, b, quads);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((k<PLANES, NT>), dim3(blocks), dim3(256), 0, 0, in, a, b, quads);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 100.0, bytes = quads * 4.0 * (1 + 4 * PLANES);
    printf("%-44s blocks %6d: %7.1f us  %6.2f TB/s\n", name, blocks, us, bytes / us * 1e-6);
}
int main()
{
    const size_t px = (size_t)32 * 3840 * 2160, quads = px / 4;
    unsigned *in; float *a, *b;
    hipMalloc(&in, px); hipMalloc(&a, px * 4); hipMalloc(&b, px * 4);
    hipMemset(in, 1, px);
    for (int blocks : {2048, 8192, 65536}) {
        run<2, true>("read 1 B/px, write 2 planes f32, nontemporal", in, a, b, quads, blocks);
        run<2, false>("read 1 B/px, write 2 planes f32, plain", in, a, b, quads, blocks);
        run<1, true>("read 1 B/px, write 1 plane f32, nontemporal", in, a, b, quads, blocks);
        run<1, false>("read 1 B/px, write 1 plane f32, plain", in, a, b, quads, blocks);
    }
    return 0;
}
