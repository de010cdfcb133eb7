// Host build of qldpc_amd/csrc/qbp_math.hpp, for the CPU test suite only (ulp-error tests).
#include "../../qldpc_amd/csrc/qbp_math.hpp"
extern "C" void shim_tanh_half(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = qbp::tanh_half(x[i]); }
extern "C" void shim_atanh2(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = qbp::atanh2(x[i]); }
extern "C" void shim_div(const double* a, const double* b, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = qbp::div_nr(a[i], b[i]); }
