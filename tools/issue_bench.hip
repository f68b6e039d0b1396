// single-wave FP64 issue rate: NCHAIN independent FMA chains per lane, one wave per workgroup
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int NCHAIN>
__global__ __launch_bounds__(64) void fma_chains(double *out, double a, double b, int iters)
{
    double x[NCHAIN];
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) x[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int c = 0; c < NCHAIN; c++) x[c] = fma(x[c], a, b);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < NCHAIN; c++) s += x[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NCHAIN> void run(double *d, int blocks, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(fma_chains<NCHAIN>, dim3(blocks), dim3(64), 0, 0, d, 0.999, 1e-3, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fma_chains<NCHAIN>, dim3(blocks), dim3(64), 0, 0, d, 0.999, 1e-3, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_wave = (double)iters * 8 * NCHAIN;
    const double waves_per_simd = blocks / 1024.0;
    printf("chains %d blocks %5d (%.0f waves/SIMD): %.3f ms  -> %.2f cycles per FMA per SIMD at 2.4 GHz\n", NCHAIN, blocks,
           waves_per_simd, ms, ms * 1e-3 * 2.4e9 / (inst_per_wave * (waves_per_simd < 1 ? 1 : waves_per_simd)));
}
int main()
{
    double *d; hipMalloc(&d, sizeof(double) * 64 * 8192);
    for (int blocks : {1024, 2048, 4096}) {
        run<1>(d, blocks, 2000); run<2>(d, blocks, 2000); run<4>(d, blocks, 2000); run<8>(d, blocks, 2000);
    }
    return 0;
}
