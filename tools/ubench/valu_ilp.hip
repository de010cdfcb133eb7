// Micro-benchmark 2: FP64 FMA throughput per SIMD as a function of per-wave ILP (independent
// dependency chains) and waves per SIMD.  Answers: is a kernel with one chain per wave latency-bound?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 20000
template <int ILP>
__global__ void bench(double* out, long long* cyc, double seed)
{
    double a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = seed + threadIdx.x * 1e-3 + 0.1 * k;
    const double c = 1.0000001, d = 1e-9;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < REP; ++i) {
#pragma unroll
        for (int r = 0; r < 8 / ILP; ++r)
#pragma unroll
            for (int k = 0; k < ILP; ++k) a[k] = __builtin_fma(a[k], c, d);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int ILP>
void run()
{
    for (int w : {1, 2, 4, 8}) {
        const int threads = 64 * 4 * w;
        if (threads > 1024) {
            // 8 waves per SIMD: two 1024-thread blocks per CU (grid 512)
        }
        const int tpb = threads > 1024 ? 1024 : threads;
        const int blocks = threads > 1024 ? 512 : 256;
        double* out; long long* cyc;
        hipMalloc(&out, sizeof(double) * tpb * blocks);
        hipMalloc(&cyc, sizeof(long long) * blocks);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(bench<ILP>, dim3(blocks), dim3(tpb), 0, 0, out, cyc, 1.25);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(bench<ILP>, dim3(blocks), dim3(tpb), 0, 0, out, cyc, 1.25);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks);
        hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += v; mean /= blocks;
        // whole-kernel time (events): every wave counts, not only the oldest one
        const double ns_per = ms * 1e6 / (REP * 8.0 * w);
        printf("ILP %d, waves/SIMD %d: %.3f ns per FMA per SIMD = %.2f cycles at 2.4 GHz; wave 0 alone saw %.1f cycles between its FMAs; "
               "=> %.1f TFLOP/s FP64\n", ILP, w, ns_per, ns_per * 2.4, mean / (REP * 8.0), 1024 * 128 / ns_per * 1e-3);
        hipFree(out); hipFree(cyc);
    }
}
int main() { run<1>(); run<2>(); run<4>(); run<8>(); return 0; }
