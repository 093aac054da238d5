// Developer probe: how much of the MFMA rate survives when other instructions are issued between the MFMAs.
// A wave runs 8 independent chains of v_mfma_f32_32x32x2_f32 and, after every MFMA, N instructions of one kind
// (VALU fma, 16-byte LDS read, 16-byte LDS write, 16-byte global load that hits in L2), with one or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int KIND, int N, int WPS>
__global__ __launch_bounds__(256, WPS) void probe(float* out, const float* src, int iters, float a0, float b0) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 4 * 4];
    floatx16 acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[f][v] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    float x[4] = {a, b, a + 1.f, b + 1.f};
    floatx4 r[4];
    typedef float floatx2 __attribute__((ext_vector_type(2)));
    floatx2 p2[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) p2[i] = floatx2{a, b};
    unsigned long long q2[4] = {1, 2, 3, 4};
    unsigned iv = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = floatx4{a, b, a, b};
    for (int i = threadIdx.x; i < 256 * 16; i += 256) lds[i] = a;
    __syncthreads();
    const unsigned laddr = (unsigned)(size_t)(&lds[0]) + threadIdx.x * 16;
    const float* gp = src + (blockIdx.x & 63) * 1024 + threadIdx.x * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[f], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < N; ++n) {
                    if (KIND == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[n & 3]) : "v"(a), "v"(b));
                    if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(r[n & 3]) : "v"(laddr));
                    if (KIND == 3) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[n & 3]) : "v"(gp));
                    if (KIND == 4) asm volatile("ds_write_b128 %0, %1" :: "v"(laddr), "v"(r[n & 3]));
                    if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p2[n & 3]) : "v"(p2[4]), "v"(p2[5]));
                    if (KIND == 6) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p2[n & 3]) : "v"(p2[4]));
                    if (KIND == 7) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[n & 3]) : "v"(a));
                    if (KIND == 8) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q2[n & 3]) : "v"(iv), "v"(iv) : "vcc");
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    }
    float s = x[0] + x[1] + x[2] + x[3] + p2[0][0] + p2[1][1] + p2[2][0] + p2[3][1] + (float)(q2[0] + q2[1] + q2[2] + q2[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) s += r[i][0] + r[i][1] + r[i][2] + r[i][3];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int v = 0; v < 16; ++v) s += acc[f][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float* g_out;
static float* g_src;

template <int KIND, int N, int WPS>
void run(const char* name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * WPS * 4;
    probe<KIND, N, WPS><<<blocks, 256>>>(g_out, g_src, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND, N, WPS><<<blocks, 256>>>(g_out, g_src, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 32 * 4096.0;
    printf("%-14s N=%d waves/SIMD=%d: %8.3f ms  %6.1f TFLOP/s\n", name, N, WPS, ms, flops / ms / 1e9);
    fflush(stdout);
}

template <int KIND>
void sweep(const char* name) {
    run<KIND, 1, 1>(name); run<KIND, 2, 1>(name); run<KIND, 4, 1>(name); run<KIND, 8, 1>(name);
    run<KIND, 1, 2>(name); run<KIND, 2, 2>(name); run<KIND, 4, 2>(name); run<KIND, 8, 2>(name);
}

int main() {
    hipMalloc(&g_out, 4096 * 256 * 4);
    hipMalloc(&g_src, 64 * 1024 * 4 + 4096);
    hipMemset(g_src, 0, 64 * 1024 * 4 + 4096);
    run<0, 0, 1>("mfma only");
    run<0, 0, 2>("mfma only");
    sweep<1>("valu fma");
    sweep<7>("valu add");
    sweep<5>("pk_fma_f32");
    sweep<6>("pk_add_f32");
    sweep<8>("mad_u64_u32");
    return 0;
}
