// Developer probe: sustained rate of v_mfma_f32_32x32x2_f32 from one wave per SIMD with N independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256, 1) void probe(float* out, int iters, float a0, float b0) {
    floatx16 acc[NACC];
#pragma unroll
    for (int f = 0; f < NACC; ++f)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[f][v] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / NACC; ++r)
#pragma unroll
            for (int f = 0; f < NACC; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[f], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < NACC; ++f)
#pragma unroll
        for (int v = 0; v < 16; ++v) s += acc[f][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(const char* name) {
    float* out;
    hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 1024;
    probe<NACC><<<blocks, 256>>>(out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NACC><<<blocks, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 64 * 4096.0;
    printf("%s: %.3f ms, %.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<4>("4 accumulators");
    run<8>("8 accumulators");
    run<16>("16 accumulators (256 acc regs)");
    return 0;
}
