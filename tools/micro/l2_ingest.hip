// Micro-benchmark: how fast can every CU read the SAME 256 KiB image from its XCD's L2?  (the activation slice of a batch-32 step)
// variants: order of the 1 KiB pieces per wave (same everywhere / rotated per workgroup), bytes per CU, copies of the image.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

template <int NLOAD, int MODE>
__global__ __launch_bounds__(512, 2) void read_kernel(const unsigned char *__restrict__ img, size_t img_bytes, int copies, unsigned *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char *base = img + static_cast<size_t>(MODE == 2 ? (blockIdx.x % copies) : 0) * img_bytes;
    uint4_t v[NLOAD];
    const int pieces = static_cast<int>(img_bytes >> 10);   // 1 KiB pieces in the image
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        int p = wave * NLOAD + i;                            // this wave's i-th piece
        if (MODE == 1) p = (p + blockIdx.x * 7) % pieces;    // rotated: workgroups walk the image from different starting points
        v[i] = *reinterpret_cast<const uint4_t *>(base + static_cast<size_t>(p % pieces) * 1024 + lane * 16);
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) acc ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int NLOAD, int MODE> static float run(const unsigned char *img, size_t bytes, int copies, unsigned *sink, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) read_kernel<NLOAD, MODE><<<256, 512>>>(img, bytes, copies, sink);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) read_kernel<NLOAD, MODE><<<256, 512>>>(img, bytes, copies, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / reps;
}
__global__ void empty_kernel() {}

int main() {
    const size_t bytes = 256 << 10;
    unsigned char *img;
    unsigned *sink;
    hipMalloc(&img, bytes * 64);
    hipMemset(img, 1, bytes * 64);
    hipMalloc(&sink, 64);
    const int reps = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) empty_kernel<<<256, 512>>>();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("empty launch                          %6.2f us\n", ms * 1000.f / reps);
    printf("32 KiB/wave same order                %6.2f us\n", run<32, 0>(img, bytes, 1, sink, reps));
    printf("32 KiB/wave rotated per workgroup     %6.2f us\n", run<32, 1>(img, bytes, 1, sink, reps));
    printf("32 KiB/wave 8 copies (one per XCD)    %6.2f us\n", run<32, 2>(img, bytes, 8, sink, reps));
    printf("32 KiB/wave 64 copies                 %6.2f us\n", run<32, 2>(img, bytes, 64, sink, reps));
    printf("16 KiB/wave same order (128 KiB img)  %6.2f us\n", run<16, 0>(img, bytes / 2, 1, sink, reps));
    printf("8 KiB/wave same order (64 KiB img)    %6.2f us\n", run<8, 0>(img, bytes / 4, 1, sink, reps));
    printf("4 KiB/wave same order (32 KiB img)    %6.2f us\n", run<4, 0>(img, bytes / 8, 1, sink, reps));
    return 0;
}
