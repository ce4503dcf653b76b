// Development probe (stand-alone: hipcc -O3 --offload-arch=gfx950 tools/mfma_power_probe.hip -o probe; ./probe [1 = zeros]).
// MFMA-only loop, operands in registers (random bytes or zeros), 192 accumulator registers per wave, two waves per SIMD,
// whole chip.  mode 0: v_mfma_i32_32x32x32_i8, mode 1: v_mfma_i32_16x16x64_i8.  Measured on MI355X: zeros 4800 TOP/s in
// both modes; random bytes 3290 (mode 0) against 4090 TOP/s (mode 1) -- the clock the chip holds depends on the data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int MODE> __global__ __launch_bounds__(512) void k(const v4i *in, int *out, int iters) {
    const int tid = blockIdx.x * 512 + threadIdx.x;
    v4i a[6], b[2];
    for (int i = 0; i < 6; i++) a[i] = in[(tid * 8 + i) & 65535];
    for (int i = 0; i < 2; i++) b[i] = in[(tid * 8 + 6 + i) & 65535];
    int s = 0;
    if (MODE == 0) {
        v16i acc[12];
        for (int i = 0; i < 12; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int mi = 0; mi < 6; mi++)
#pragma unroll
                for (int ni = 0; ni < 2; ni++)
                    acc[mi * 2 + ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mi], b[ni], acc[mi * 2 + ni], 0, 0, 0);
            // rotate operands so that inputs change like fresh fragments would
            v4i t = a[0];
#pragma unroll
            for (int i = 0; i < 5; i++) a[i] = a[i + 1];
            a[5] = t;
        }
        for (int i = 0; i < 12; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    } else {
        v4i acc[48];
        for (int i = 0; i < 48; i++) for (int r = 0; r < 4; r++) acc[i][r] = 0;
        for (int it = 0; it < iters; it++) {
            // same MACs per iteration: 24 MFMAs of 16x16x64 = 12 of 32x32x32
#pragma unroll
            for (int q = 0; q < 24; q++)
                acc[(q * 2) % 48] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[q % 6], b[q & 1], acc[(q * 2) % 48], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 24; q++)
                acc[(q * 2 + 1) % 48] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(q + 3) % 6], b[q & 1], acc[(q * 2 + 1) % 48], 0, 0, 0);
            v4i t = a[0];
#pragma unroll
            for (int i = 0; i < 5; i++) a[i] = a[i + 1];
            a[5] = t;
        }
        for (int i = 0; i < 48; i++) for (int r = 0; r < 4; r++) s += acc[i][r];
    }
    out[tid] = s;
}
int main(int argc, char **argv) {
    const int zero = argc > 1 ? atoi(argv[1]) : 0;
    std::vector<int> h(65536 * 4);
    srand(1);
    for (auto &x : h) x = zero ? 0 : (rand() ^ (rand() << 16));
    v4i *in; int *out;
    hipMalloc(&in, h.size() * 4); hipMalloc(&out, 256 * 512 * 4 * 4);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, in, out, iters);
            else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, in, out, iters / 2);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double macs = mode == 0 ? 256.0 * 8 * iters * 12 * 32768 : 256.0 * 8 * (iters / 2) * 48 * 16384;
            printf("%s mode %d rep %d: %.2f ms, %.0f TOP/s\n", zero ? "zeros" : "random", mode, rep, ms, 2 * macs / ms / 1e9);
        }
    }
    return 0;
}
