// Micro-benchmark: issue cost of the FP64 VALU instructions the BP kernel uses, on gfx950.
// One workgroup per CU, W waves per SIMD; each wave runs N independent chains of one instruction
// in a loop; cycles from s_memtime.  Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256
template <int OP>
__global__ void bench(double* out, long long* cyc, double seed)
{
    double a0 = seed + threadIdx.x * 1e-3, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3;
    double a4 = a0 + 0.4, a5 = a0 + 0.5, a6 = a0 + 0.6, a7 = a0 + 0.7;
    const double c = 1.0000001, d = 1e-9;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < REP; ++i) {
#define APPLY(x)                                                                         \
    if (OP == 0) x = __builtin_fma(x, c, d);                                             \
    else if (OP == 1) x = __builtin_amdgcn_rcp(x);                                       \
    else if (OP == 2) x = __builtin_rint(x * c);                                         \
    else if (OP == 3) x = __builtin_fmin(x * c, 3.0);                                    \
    else if (OP == 4) x = (x * c < 2.0) ? x * c : d;                                     \
    else if (OP == 5) x = (double)(int)(x) + d;                                          \
    else if (OP == 6) x = x * c;                                                         \
    else if (OP == 7) x = __builtin_amdgcn_ldexp(x, 1) * 0.5;
        APPLY(a0) APPLY(a1) APPLY(a2) APPLY(a3) APPLY(a4) APPLY(a5) APPLY(a6) APPLY(a7)
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int insts_per_apply)
{
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 64 * 4 * waves_per_simd, blocks = 256;
        double* out; long long* cyc;
        hipMalloc(&out, sizeof(double) * threads * blocks);
        hipMalloc(&cyc, sizeof(long long) * blocks);
        hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.25);
        hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.25);
        hipDeviceSynchronize();
        std::vector<long long> h(blocks);
        hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += v; mean /= blocks;
        // s_memtime ticks at 100 MHz-derived constant rate? report raw ticks per wave-instruction per SIMD
        const double per = mean / (REP * 8.0 * insts_per_apply * waves_per_simd);
        printf("%-28s waves/SIMD %d: %.2f ticks per wave-instruction (%d inst/apply)\n", name, waves_per_simd, per, insts_per_apply);
        hipFree(out); hipFree(cyc);
    }
}

int main()
{
    run<0>("v_fma_f64", 1);
    run<6>("v_mul_f64", 1);
    run<1>("v_rcp_f64", 1);
    run<2>("v_mul+v_rndne_f64", 2);
    run<3>("v_mul+v_min_f64", 2);
    run<4>("v_mul,cmp,2cndmask(+mul)", 4);
    run<5>("cvt_i32_f64+cvt_f64_i32+add", 3);
    run<7>("v_ldexp_f64+v_mul", 2);
    return 0;
}
