// In-run microbenchmark behind bench.py's FP64 vector-ALU roofline (measurement aid, not product).
//
// For each issue class of tools/valu_mix.py (FP64 fma / mul / add / min-max / compare / convert,
// v_rcp_f64, plain 32-bit VALU, v_readlane/v_writelane) one kernel issues a long run of independent
// instructions of that class from every wave of a chip-filling grid at the BP kernel's occupancy
// (1024-thread workgroups, one per CU: 4 waves per SIMD) and reports wave-instructions per second for
// the whole chip.  bench.py combines those rates with the kernel's static instruction mix:
//     t_min = sum_class  n_class / rate_class ;  roofline.frac = t_min / t_measured.
// The mix form runs one stall-free stream with the kernel's own class proportions: its rate is the
// roofline's peak.  Also: shader clock from a single wave of `s_nop 15`.
//
//   extern "C" int ubench_valu_rate(int device, int cls, int waves_per_simd, double* wave_insts_per_s);
//   extern "C" int ubench_mix_rate(int device, const int* per_class /*[classes]*/, int waves_per_simd,
//                                  double* wave_insts_per_s);      // the same, for a MIX of classes
//   extern "C" int ubench_clock_ghz(int device, double* ghz);
//   extern "C" const char* ubench_class_name(int cls);   // NULL past the last class
#include <hip/hip_runtime.h>

#include <cstdint>

namespace {

constexpr int REP = 1000;       // loop iterations; 128 instructions each

enum { C_FMA, C_MUL, C_ADD, C_MINMAX, C_CMP, C_CVT, C_TRANS, C_B32, C_LANE, C_COUNT };
const char* const NAMES[C_COUNT] = {"fma_f64", "mul_f64", "add_f64", "minmax_f64", "cmp_f64", "cvt_f64",
                                    "trans_f64", "alu_b32", "lane_b32"};

// 8 independent chains x 16 = 128 instructions of the class per loop iteration
#define X8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define X32(S) X8(S) X8(S) X8(S) X8(S)
#define X128(S) X32(S) X32(S) X32(S) X32(S)

template <int CLS>
__global__ __launch_bounds__(1024) void rate_kernel(double* out, double seed, int rep, unsigned long long* cyc)
{
    const unsigned long long t0 = __builtin_readcyclecounter();     // s_memtime: shader-clock cycles
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
            X128(S)
#undef S
        } else if constexpr (CLS == C_MUL) {
#define S(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
            X128(S)
#undef S
        } else if constexpr (CLS == C_ADD) {
#define S(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(d));
            X128(S)
#undef S
        } else if constexpr (CLS == C_MINMAX) {
#define S(k) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[k]) : "v"(d));
            X128(S)
#undef S
        } else if constexpr (CLS == C_CMP) {
#define S(k) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[k]), "v"(c) : "vcc");
            X128(S)
#undef S
        } else if constexpr (CLS == C_CVT) {
#define S(k) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[k]));
            X128(S)
#undef S
        } else if constexpr (CLS == C_TRANS) {
#define S(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));
            X128(S)
#undef S
        } else if constexpr (CLS == C_B32) {
#define S(k) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
            X128(S)
#undef S
        } else {
            unsigned s0;
#define S(k) asm volatile("v_readlane_b32 %0, %1, 3\n\tv_writelane_b32 %1, %0, 5" : "=s"(s0), "+v"(u[k]));
            X32(S) X32(S)        // 64 pairs = 128 instructions
#undef S
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k] + (double)u[k];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    // every wave reports: the SIMD arbitrates by age, so the oldest wave of a SIMD finishes its loop
    // long before the youngest -- the issue cost is what ALL waves of the SIMD need together
    if (cyc && (threadIdx.x & 63) == 0) {
        cyc[2 * ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6))] = t0;
        cyc[2 * ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) + 1] = t1;
    }
}

// A stall-free stream with a given class mix: per outer iteration, for each class c, blocks[c] blocks
// of 8 independent instructions of that class (so the stream issues 8 * sum(blocks) instructions per
// iteration in the proportions of the kernel under study, with no LDS, no barrier and no dependency a
// wave would have to wait for).  Its rate is the issue-bound "speed of light" for that mix on this
// chip in this power state: the denominator of bench.py's roofline.
struct MixBlocks { int b[C_COUNT]; };

__global__ __launch_bounds__(1024) void mix_kernel(double* out, double seed, int rep, MixBlocks mb)
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
#define S(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c), "v"(d));
#pragma unroll 1
        for (int j = 0; j < mb.b[C_FMA]; ++j) { X8(S) }
#undef S
#define S(k) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
#pragma unroll 1
        for (int j = 0; j < mb.b[C_B32]; ++j) { X8(S) }
#undef S
#define S(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
#pragma unroll 1
        for (int j = 0; j < mb.b[C_MUL]; ++j) { X8(S) }
#undef S
#define S(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(d));
#pragma unroll 1
        for (int j = 0; j < mb.b[C_ADD]; ++j) { X8(S) }
#undef S
#define S(k) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[k]) : "v"(d));
#pragma unroll 1
        for (int j = 0; j < mb.b[C_MINMAX]; ++j) { X8(S) }
#undef S
#define S(k) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[k]), "v"(c) : "vcc");
#pragma unroll 1
        for (int j = 0; j < mb.b[C_CMP]; ++j) { X8(S) }
#undef S
#define S(k) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[k]));
#pragma unroll 1
        for (int j = 0; j < mb.b[C_CVT]; ++j) { X8(S) }
#undef S
#define S(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));
#pragma unroll 1
        for (int j = 0; j < mb.b[C_TRANS]; ++j) { X8(S) }
#undef S
        unsigned s0;
#define S(k) asm volatile("v_readlane_b32 %0, %1, 3\n\tv_writelane_b32 %1, %0, 5" : "=s"(s0), "+v"(u[k]));
#pragma unroll 1
        for (int j = 0; j < mb.b[C_LANE]; ++j) { S(0) S(1) S(2) S(3) }
#undef S
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
    const unsigned long long c0 = __builtin_readcyclecounter();
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
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = c1 - c0; }   // 100 MHz ticks; shader-clock cycles
}

template <int CLS>
int run_rate(int waves_per_simd, int num_cu, double* out, double* rate, double* cycles_per_inst)
{
    const int threads = 64 * 4 * (waves_per_simd > 4 ? 4 : waves_per_simd);
    const int blocks = num_cu * (waves_per_simd > 4 ? 2 : 1);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -3;
    unsigned long long* cyc = reinterpret_cast<unsigned long long*>(out + (size_t)1024 * num_cu * 2);
    hipLaunchKernelGGL(rate_kernel<CLS>, dim3(blocks), dim3(threads), 0, 0, out, 1.25, REP / 8, cyc);   // warm-up
    double best = 1e30;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(rate_kernel<CLS>, dim3(blocks), dim3(threads), 0, 0, out, 1.25, REP, cyc);
        hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) return -3;
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    const double waves = (double)blocks * (threads / 64);
    *rate = waves * REP * 128.0 / (best * 1e-3);
    if (cycles_per_inst) {
        // shader-clock cycles one SIMD needs per wave-instruction of this class: every SIMD of a CU
        // hosts threads / 256 waves that each issue REP * 32 instructions while the block runs
        const int nb = blocks < 128 ? blocks : 128, wpb = threads / 64;
        static unsigned long long h[128 * 16 * 2];
        if (hipMemcpy(h, cyc, sizeof(unsigned long long) * 32 * nb, hipMemcpyDeviceToHost) != hipSuccess) return -3;
        double mean = 0;
        for (int i = 0; i < nb; ++i) {
            unsigned long long lo = ~0ull, hi = 0;
            for (int w = 0; w < wpb; ++w) {
                lo = h[2 * (i * 16 + w)] < lo ? h[2 * (i * 16 + w)] : lo;
                hi = h[2 * (i * 16 + w) + 1] > hi ? h[2 * (i * 16 + w) + 1] : hi;
            }
            mean += (double)(hi - lo);
        }
        mean /= nb;
        *cycles_per_inst = mean / ((threads / 256.0) * (double)REP * 128.0);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace

extern "C" {

const char* ubench_class_name(int cls) { return (cls >= 0 && cls < C_COUNT) ? NAMES[cls] : nullptr; }

int ubench_valu_rate2(int device, int cls, int waves_per_simd, double* wave_insts_per_s, double* cycles_per_inst);
int ubench_valu_rate(int device, int cls, int waves_per_simd, double* wave_insts_per_s)
{
    return ubench_valu_rate2(device, cls, waves_per_simd, wave_insts_per_s, nullptr);
}

int ubench_valu_rate2(int device, int cls, int waves_per_simd, double* wave_insts_per_s, double* cycles_per_inst)
{
    if (!wave_insts_per_s || cls < 0 || cls >= C_COUNT || waves_per_simd < 1 || waves_per_simd > 8) return -1;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return -2;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -2;
    const int num_cu = prop.multiProcessorCount;
    double* out = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&out), sizeof(double) * 1024 * (size_t)num_cu * 2 + 16 * 16 * (size_t)num_cu * 2) != hipSuccess) return -3;
    int rc = -1;
    switch (cls) {
        case C_FMA: rc = run_rate<C_FMA>(waves_per_simd, num_cu, out, wave_insts_per_s, cycles_per_inst); break;
        case C_MUL: rc = run_rate<C_MUL>(waves_per_simd, num_cu, out, wave_insts_per_s, cycles_per_inst); break;
        case C_ADD: rc = run_rate<C_ADD>(waves_per_simd, num_cu, out, wave_insts_per_s, cycles_per_inst); break;
        case C_MINMAX: rc = run_rate<C_MINMAX>(waves_per_simd, num_cu, out, wave_insts_per_s, cycles_per_inst); break;
        case C_CMP: rc = run_rate<C_CMP>(waves_per_simd, num_cu, out, wave_insts_per_s, cycles_per_inst); break;
        case C_CVT: rc = run_rate<C_CVT>(waves_per_simd, num_cu, out, wave_insts_per_s, cycles_per_inst); break;
        case C_TRANS: rc = run_rate<C_TRANS>(waves_per_simd, num_cu, out, wave_insts_per_s, cycles_per_inst); break;
        case C_B32: rc = run_rate<C_B32>(waves_per_simd, num_cu, out, wave_insts_per_s, cycles_per_inst); break;
        default: rc = run_rate<C_LANE>(waves_per_simd, num_cu, out, wave_insts_per_s, cycles_per_inst); break;
    }
    (void)hipFree(out);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    return rc;
}

int ubench_mix_rate(int device, const int* per_class, int waves_per_simd, double* wave_insts_per_s)
{
    if (!per_class || !wave_insts_per_s || waves_per_simd < 1 || waves_per_simd > 4) return -1;
    MixBlocks mb;
    long long per_iter = 0;
    for (int c = 0; c < C_COUNT; ++c) {
        mb.b[c] = (per_class[c] + 4) / 8;                 // blocks of 8 instructions, rounded
        per_iter += 8LL * mb.b[c];
    }
    if (per_iter <= 0) return -1;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return -2;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -2;
    const int blocks = prop.multiProcessorCount, threads = 64 * 4 * waves_per_simd;
    double* out = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&out), sizeof(double) * 1024 * (size_t)blocks) != hipSuccess) return -3;
    const int rep = (int)(4000LL * 32 / per_iter) + 1;    // about as long as the single-class runs
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mix_kernel, dim3(blocks), dim3(threads), 0, 0, out, 1.25, rep / 8 + 1, mb);
    double best = 1e30;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mix_kernel, dim3(blocks), dim3(threads), 0, 0, out, 1.25, rep, mb);
        hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) { best = -1; break; }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    (void)hipFree(out);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (best <= 0 || hipGetLastError() != hipSuccess) return -3;
    *wave_insts_per_s = (double)blocks * (threads / 64) * (double)rep * (double)per_iter / (best * 1e-3);
    return 0;
}

int ubench_clock_ghz2(int device, double* ghz_from_snop, double* ghz_from_cycle_counter);
int ubench_clock_ghz(int device, double* ghz) { return ubench_clock_ghz2(device, ghz, nullptr); }

int ubench_clock_ghz2(int device, double* ghz, double* ghz_counter)
{
    // On its own non-blocking stream, results through pinned memory: the probe (one wave, ~2 ms) runs
    // BESIDE whatever the other streams of the process have queued -- called while a long queue of
    // kernels is executing it reports the clock the chip holds under that load.
    if (!ghz) return -1;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return -2;
    unsigned long long* d = nullptr;
    unsigned long long* hst = nullptr;
    hipStream_t st = nullptr;
    int rc = -3;
    if (hipMalloc(reinterpret_cast<void**>(&d), 2 * sizeof(unsigned long long)) == hipSuccess &&
        hipHostMalloc(reinterpret_cast<void**>(&hst), 2 * sizeof(unsigned long long), hipHostMallocDefault) == hipSuccess &&
        hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) {
        const int rep = 4000;                      // 64 x 64 x 4000 = 1.6e7 cycles, about 7 ms
        hst[0] = hst[1] = 0;
        hipLaunchKernelGGL(clock_kernel, dim3(1), dim3(64), 0, st, d, rep);
        if (hipMemcpyAsync(hst, d, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st) == hipSuccess &&
            hipStreamSynchronize(st) == hipSuccess && hst[0] != 0) {
            // measured on gfx950: one `s_nop 15` holds the wave for 64 clocks (16 wait states of one
            // issue quad-cycle each); 64 of them per iteration; one tick of wall_clock64 = 10 ns
            *ghz = (double)rep * 64.0 * 64.0 / ((double)hst[0] * 10.0);
            if (ghz_counter) *ghz_counter = (double)hst[1] / ((double)hst[0] * 10.0);
            rc = 0;
        }
    }
    if (st) (void)hipStreamDestroy(st);
    if (hst) (void)hipHostFree(hst);
    if (d) (void)hipFree(d);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    return rc;
}

}  // extern "C"
