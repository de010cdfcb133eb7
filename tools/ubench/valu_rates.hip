// In-run microbenchmark behind bench.py's FP64 vector-ALU roofline (measurement aid, not product).
//
// For each issue class of tools/valu_mix.py (FP64 fma / mul / add / min-max / compare / convert,
// v_rcp_f64, plain 32-bit VALU, v_readlane/v_writelane) one kernel issues a long run of independent
// instructions of that class from every wave of a chip-filling grid at the BP kernel's occupancy
// (1024-thread workgroups, one per CU: 4 waves per SIMD) and reports wave-instructions per second for
// the whole chip.  bench.py combines those rates with the kernel's static instruction mix:
//     t_min = sum_class  n_class / rate_class ;  roofline.frac = t_min / t_measured.
// Also: shader clock from a single wave of `s_nop 15` (16 cycles each).
//
//   extern "C" int ubench_valu_rate(int device, int cls, int waves_per_simd, double* wave_insts_per_s);
//   extern "C" int ubench_clock_ghz(int device, double* ghz);
//   extern "C" const char* ubench_class_name(int cls);   // NULL past the last class
#include <hip/hip_runtime.h>

#include <cstdint>

namespace {

constexpr int REP = 4000;       // loop iterations; 32 instructions each

enum { C_FMA, C_MUL, C_ADD, C_MINMAX, C_CMP, C_CVT, C_TRANS, C_B32, C_LANE, C_COUNT };
const char* const NAMES[C_COUNT] = {"fma_f64", "mul_f64", "add_f64", "minmax_f64", "cmp_f64", "cvt_f64",
                                    "trans_f64", "alu_b32", "lane_b32"};

// 8 independent chains x 4 = 32 instructions of the class per loop iteration
#define X8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define X32(S) X8(S) X8(S) X8(S) X8(S)

template <int CLS>
__global__ __launch_bounds__(1024) void rate_kernel(double* out, double seed, int rep)
{
    double a[8];
    unsigned u[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a[k] = seed + threadIdx.x * 1e-3 + 0.1 * k;
        u[k] = threadIdx.x * 2654435761u + k;
    }
    const double c = 1.0000001, d = 1e-9;
#pragma unroll 1
    for (int i = 0; i < rep; ++i) {
        if constexpr (CLS == C_FMA) {
#define S(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c), "v"(d));
            X32(S)
#undef S
        } else if constexpr (CLS == C_MUL) {
#define S(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
            X32(S)
#undef S
        } else if constexpr (CLS == C_ADD) {
#define S(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(d));
            X32(S)
#undef S
        } else if constexpr (CLS == C_MINMAX) {
#define S(k) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[k]) : "v"(d));
            X32(S)
#undef S
        } else if constexpr (CLS == C_CMP) {
#define S(k) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[k]), "v"(c) : "vcc");
            X32(S)
#undef S
        } else if constexpr (CLS == C_CVT) {
#define S(k) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[k]));
            X32(S)
#undef S
        } else if constexpr (CLS == C_TRANS) {
#define S(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));
            X32(S)
#undef S
        } else if constexpr (CLS == C_B32) {
#define S(k) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
            X32(S)
#undef S
        } else {
            unsigned s0;
#define S(k) asm volatile("v_readlane_b32 %0, %1, 3\n\tv_writelane_b32 %1, %0, 5" : "=s"(s0), "+v"(u[k]));
            X8(S) X8(S)          // 16 pairs = 32 instructions
#undef S
        }
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k] + (double)u[k];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void clock_kernel(unsigned long long* out, int rep)
{
    // one wave: rep x 64 x `s_nop 15` (16 cycles each) = rep x 1024 cycles plus the loop's three
    // scalar instructions per iteration
    const unsigned long long t0 = wall_clock64();
#pragma unroll 1
    for (int i = 0; i < rep; ++i) {
        asm volatile(
            "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\n"
            "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\n"
            "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\n"
            "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\n"
            "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\n"
            "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\n"
            "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\n"
            "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\n");
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;     // 100 MHz constant clock ticks
}

template <int CLS>
int run_rate(int waves_per_simd, int num_cu, double* out, double* rate)
{
    const int threads = 64 * 4 * (waves_per_simd > 4 ? 4 : waves_per_simd);
    const int blocks = num_cu * (waves_per_simd > 4 ? 2 : 1);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -3;
    hipLaunchKernelGGL(rate_kernel<CLS>, dim3(blocks), dim3(threads), 0, 0, out, 1.25, REP / 8);   // warm-up
    double best = 1e30;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(rate_kernel<CLS>, dim3(blocks), dim3(threads), 0, 0, out, 1.25, REP);
        hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) return -3;
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    const double waves = (double)blocks * (threads / 64);
    *rate = waves * REP * 32.0 / (best * 1e-3);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

extern "C" {

const char* ubench_class_name(int cls) { return (cls >= 0 && cls < C_COUNT) ? NAMES[cls] : nullptr; }

int ubench_valu_rate(int device, int cls, int waves_per_simd, double* wave_insts_per_s)
{
    if (!wave_insts_per_s || cls < 0 || cls >= C_COUNT || waves_per_simd < 1 || waves_per_simd > 8) return -1;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return -2;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -2;
    const int num_cu = prop.multiProcessorCount;
    double* out = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&out), sizeof(double) * 1024 * (size_t)num_cu * 2) != hipSuccess) return -3;
    int rc = -1;
    switch (cls) {
        case C_FMA: rc = run_rate<C_FMA>(waves_per_simd, num_cu, out, wave_insts_per_s); break;
        case C_MUL: rc = run_rate<C_MUL>(waves_per_simd, num_cu, out, wave_insts_per_s); break;
        case C_ADD: rc = run_rate<C_ADD>(waves_per_simd, num_cu, out, wave_insts_per_s); break;
        case C_MINMAX: rc = run_rate<C_MINMAX>(waves_per_simd, num_cu, out, wave_insts_per_s); break;
        case C_CMP: rc = run_rate<C_CMP>(waves_per_simd, num_cu, out, wave_insts_per_s); break;
        case C_CVT: rc = run_rate<C_CVT>(waves_per_simd, num_cu, out, wave_insts_per_s); break;
        case C_TRANS: rc = run_rate<C_TRANS>(waves_per_simd, num_cu, out, wave_insts_per_s); break;
        case C_B32: rc = run_rate<C_B32>(waves_per_simd, num_cu, out, wave_insts_per_s); break;
        default: rc = run_rate<C_LANE>(waves_per_simd, num_cu, out, wave_insts_per_s); break;
    }
    (void)hipFree(out);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    return rc;
}

int ubench_clock_ghz(int device, double* ghz)
{
    if (!ghz) return -1;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return -2;
    unsigned long long* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), sizeof(unsigned long long)) != hipSuccess) return -3;
    const int rep = 4000;                      // 4.1e6 cycles, about 2 ms
    hipLaunchKernelGGL(clock_kernel, dim3(1), dim3(64), 0, 0, d, rep);
    unsigned long long ticks = 0;
    const hipError_t e = hipMemcpy(&ticks, d, sizeof(ticks), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess || ticks == 0) return -3;
    *ghz = (double)rep * 1024.0 / ((double)ticks * 10.0);     // cycles / ns; one tick = 10 ns
    return 0;
}

}  // extern "C"
