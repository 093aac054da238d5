// Developer probe: how long the first loads of a convolution tile take, as a function of how the lanes of a wave are laid
// over the tile's memory and of how many CUs start such a burst at once.
// A workgroup of 512 threads requests the raw pixels of a 10 x 34 pixel tile of a (B,512,512,C) float map - the first
// 32 channels (128 bytes) of every pixel, 16 bytes per lane and load, 6 loads per thread in flight - and measures the
// shader clock from before the first request until all have arrived.
//   pattern 0: the F(4x4) kernel's order: a load covers 16 channels (2 lanes per pixel and 8 channels), so a 128-byte
//              line is touched by four different loads
//   pattern 1: 8 consecutive lanes cover the 128 bytes of a pixel: every line is requested once
//   pattern 2: pattern 0 right behind the stores of an output tile (8 x 32 pixels x 128 channels, 128 KB per workgroup),
//              as at the end of one tile and the start of the next
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/burst_probe tools/probe/burst_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int kW = 512, kH = 512, kRawW = 34, kRawH = 10, kPx = kRawW * kRawH;

template <int PATTERN>
__global__ __launch_bounds__(512, 1) void burst(const float* x, int C, int tiles_x, int tile0, long long* cycles, float* sink, float* y) {
    const int tid = threadIdx.x;
    const int tile_all = tile0 + blockIdx.x;
    const int per_image = tiles_x * (kH / 8);
    const int b = tile_all / per_image, tile = tile_all % per_image;
    const int y0 = (tile / tiles_x) * 8, x0 = (tile % tiles_x) * 32;
    const int64_t img_floats = (int64_t)kH * kW * C;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(x + b * img_floats), 0, (int)(img_floats * 4), 0x00020000);
    int off[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        int px, byte;
        if (PATTERN == 0 || PATTERN == 2) {
            // loads 0..2: channels 0..15 (sub-chunk i: 8 channels, lanes (pixel, quad)); loads 3..5: channels 16..31
            const int chunk = i / 3, j = i % 3;
            const int sub = j == 2 ? (tid >> 8) : j;
            const int f = j == 2 ? 512 + (tid & 255) : tid;
            px = f >> 1;
            byte = chunk * 64 + sub * 32 + (f & 1) * 16;
            if (f >= kPx * 2) px = -1;
        } else {
            px = (tid >> 3) + 64 * i;
            byte = (tid & 7) * 16;
        }
        const int gy = y0 - 1 + px / kRawW, gx = x0 - 1 + px % kRawW;
        const bool ok = px >= 0 && px < kPx && gy >= 0 && gy < kH && gx >= 0 && gx < kW;
        off[i] = ok ? ((gy * kW + gx) * C * 4 + byte) : (int)0x80000000;
    }
    if (PATTERN == 2) {          // the output tile of the workgroup before: 256 pixels x 512 bytes, 16 stores of 16 bytes per thread
        float* yt = y + ((int64_t)(b * kH + y0) * kW + x0) * 128;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int px = it * 16 + (tid >> 5), row = px >> 5, col = px & 31;
            *reinterpret_cast<float4*>(yt + ((int64_t)row * kW + col) * 128 + (tid & 31) * 4) = float4{1.f, 2.f, 3.f, (float)tid};
        }
    }
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    float4 r[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) r[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i], 0, 0));
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) s += r[i].x + r[i].y + r[i].z + r[i].w;
    asm volatile("" :: "v"(s));
    __syncthreads();
    const long long t2 = __builtin_readcyclecounter();
    if (tid == 0) {
        cycles[2 * blockIdx.x] = t1 - t0;
        cycles[2 * blockIdx.x + 1] = t2 - t0;
    }
    if (s == 12345.678f) sink[0] = s;
}

int main() {
    const int C = 64, B = 16, tiles_x = kW / 32;
    const size_t floats = (size_t)B * kH * kW * C;
    float* x;
    long long* d_cycles;
    float* sink;
    float* y;
    if (hipMalloc(&y, floats * 2 * 4) != hipSuccess) return 1;
    if (hipMalloc(&x, floats * 4) != hipSuccess || hipMalloc(&d_cycles, 2 * 4096 * sizeof(long long)) != hipSuccess || hipMalloc(&sink, 16) != hipSuccess) return 1;
    hipMemset(x, 0, floats * 4);
    int tile0 = 0;
    for (int pattern = 0; pattern < 3; ++pattern)
        for (int wgs : {8, 32, 64, 128, 256}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (pattern == 0) hipLaunchKernelGGL(burst<0>, dim3(wgs), dim3(512), 0, 0, x, C, tiles_x, tile0, d_cycles, sink, y);
                else if (pattern == 1) hipLaunchKernelGGL(burst<1>, dim3(wgs), dim3(512), 0, 0, x, C, tiles_x, tile0, d_cycles, sink, y);
                else hipLaunchKernelGGL(burst<2>, dim3(wgs), dim3(512), 0, 0, x, C, tiles_x, tile0, d_cycles, sink, y);
                if (hipDeviceSynchronize() != hipSuccess) return 2;
                std::vector<long long> c(2 * wgs);
                hipMemcpy(c.data(), d_cycles, 2 * wgs * sizeof(long long), hipMemcpyDeviceToHost);
                std::vector<long long> issue, all;
                for (int i = 0; i < wgs; ++i) { issue.push_back(c[2 * i]); all.push_back(c[2 * i + 1]); }
                std::sort(issue.begin(), issue.end());
                std::sort(all.begin(), all.end());
                printf("pattern %d wgs %3d rep %d (%s): issue median %6lld max %6lld | all arrived median %6lld max %6lld cycles\n", pattern, wgs, rep,
                       rep == 0 ? "cold" : "same tiles again", issue[wgs / 2], issue.back(), all[wgs / 2], all.back());
                if (rep == 1) tile0 += wgs;
            }
        }
    return 0;
}
