// Where do the workgroups of a one-wave-per-workgroup launch land?  Each workgroup records its XCC / SE / CU / SIMD and its
// begin and end on the GPU's wall clock; the host prints how many workgroups every CU and SIMD received and how many of them
// were resident at once.  usage: placement_probe <workgroups> <dynamic LDS bytes> [busy iterations]
// build: hipcc --offload-arch=gfx950 -O2 -o placement_probe placement_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>
#include <tuple>

__global__ __launch_bounds__(64) void probe(uint32_t *hw, uint32_t *xcc, uint64_t *t0, uint64_t *t1, int iters, double *sink)
{
    extern __shared__ double lds[];
    const uint64_t a = wall_clock64();
    double x = threadIdx.x * 1e-3, y = 1.0;
    for (int i = 0; i < iters; i++) { x = fma(x, 1.0000001, 1e-9); y = fma(y, 0.9999999, x); }
    lds[threadIdx.x] = x + y;
    __syncthreads();
    if (threadIdx.x == 0) {
        hw[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
        xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
        t0[blockIdx.x] = a; t1[blockIdx.x] = wall_clock64();
        if (lds[1] == 12345.678) *sink = lds[1];
    }
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1024, lds = argc > 2 ? atoi(argv[2]) : 17920, iters = argc > 3 ? atoi(argv[3]) : 4000;
    uint32_t *hw, *xcc; uint64_t *t0, *t1; double *sink;
    hipMalloc(&hw, 4 * n); hipMalloc(&xcc, 4 * n); hipMalloc(&t0, 8 * n); hipMalloc(&t1, 8 * n); hipMalloc(&sink, 8);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(probe, dim3(n), dim3(64), lds, 0, hw, xcc, t0, t1, iters, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    }
    std::vector<uint32_t> H(n), X(n); std::vector<uint64_t> A(n), B(n);
    hipMemcpy(H.data(), hw, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(X.data(), xcc, 4 * n, hipMemcpyDeviceToHost);
    hipMemcpy(A.data(), t0, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(B.data(), t1, 8 * n, hipMemcpyDeviceToHost);
    std::map<std::tuple<int, int, int, int>, int> per_cu; std::map<std::tuple<int, int, int, int, int>, int> per_simd;
    uint64_t first = *std::min_element(A.begin(), A.end()), last = *std::max_element(B.begin(), B.end());
    for (int i = 0; i < n; i++) {
        const int simd = (H[i] >> 4) & 3, cu = (H[i] >> 8) & 15, sh = (H[i] >> 12) & 1, se = (H[i] >> 13) & 7, x = X[i] & 15;
        per_cu[{x, se, sh, cu}]++; per_simd[{x, se, sh, cu, simd}]++;
    }
    std::map<int, int> hist_cu, hist_simd;
    for (auto &kv : per_cu) hist_cu[kv.second]++;
    for (auto &kv : per_simd) hist_simd[kv.second]++;
    // workgroups that began after the first workgroup ended = a second round
    const uint64_t first_end = *std::min_element(B.begin(), B.end());
    int late = 0; for (int i = 0; i < n; i++) late += A[i] >= first_end;
    uint64_t busy = 0; for (int i = 0; i < n; i++) busy += B[i] - A[i];
    printf("workgroups %d  lds %d B: %zu CUs, %zu SIMDs used; span %.1f us, mean workgroup %.1f us, %d began after the first one ended\n",
           n, lds, per_cu.size(), per_simd.size(), (last - first) / 100.0, busy / 100.0 / n, late);
    printf("  workgroups per CU  :"); for (auto &kv : hist_cu) printf(" %dx%d", kv.second, kv.first); printf("  (CUs x count)\n");
    printf("  workgroups per SIMD:"); for (auto &kv : hist_simd) printf(" %dx%d", kv.second, kv.first); printf("\n");
    return 0;
}
