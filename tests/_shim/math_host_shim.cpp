// Host build of qldpc_amd/csrc/qbp_math.hpp, for the CPU test suite only (ulp-error tests).
#include "../../qldpc_amd/csrc/qbp_math.hpp"
extern "C" void shim_tanh_half(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = qbp::tanh_half(x[i]); }
extern "C" void shim_atanh2(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = qbp::atanh2(x[i]); }
extern "C" void shim_div(const double* a, const double* b, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = qbp::div_nr(a[i], b[i]); }
// numpy-exact forms: the image the kernels keep in LDS, here in host memory
alignas(16) static const qbp::NpImage g_host_image = qbp::np_make_image();
extern "C" void shim_np_tanh_half(const double* x, double* y, long n)
{
    const double* T = reinterpret_cast<const double*>(&g_host_image);
    for (long i = 0; i < n; ++i) y[i] = qbp::np_tanh_half(x[i], T);
}
extern "C" void shim_np_arctanh_x2(const double* x, double* y, long n)
{
    const double* T = reinterpret_cast<const double*>(&g_host_image);
    for (long i = 0; i < n; ++i) y[i] = qbp::np_arctanh_x2(x[i], T);
}
extern "C" void shim_np_rcp14_hi(const unsigned* v_hi, unsigned* r_hi, long n)
{
    for (long i = 0; i < n; ++i) r_hi[i] = qbp::np_rcp14_hi(v_hi[i], reinterpret_cast<const double*>(&g_host_image));
}
// the kernels' tail of the check update: 2 arctanh(clip(x * sign)) with the sign bit set at the end
extern "C" void shim_check_message(const double* x, const unsigned char* sbit, double* y, long n, int variant)
{
    const double* T = reinterpret_cast<const double*>(&g_host_image);
    for (long i = 0; i < n; ++i)
        y[i] = variant == 1 ? qbp::check_message<1>(x[i], sbit[i], T) : qbp::check_message<0>(x[i], sbit[i], T);
}
