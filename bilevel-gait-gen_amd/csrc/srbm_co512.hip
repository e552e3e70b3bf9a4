// The CO-RESIDENT kernel set of the library (see srbm_co.h): the kernels of the RTI path compiled a second time from the same headers, inside a
// namespace of their own, with
//   * 256 threads per workgroup (4 waves) and launch bounds that ask for two waves per SIMD ACROSS two workgroups (<= 256 VGPRs per lane),
//   * the packed normal matrix / factor / inverse factor of the IPM in the work record (SrbmWork::Mg, L2-resident) instead of LDS
//     (SRBM_M_GLOBAL; the tiles still live in the accumulator registers through assembly and factorisation),
//   * a dynamic-LDS request of at most 80 KB,
// so that two instances are resident on a CU at a time.  One instance alone runs slower this way (a wave carries twice the inequality rows,
// the solves read the inverse factor from L2); two of them interleaved run faster than two after one another on a CU of their own -- the
// measurement is in DESIGN.md.  Results: the same algorithm on the same data; reductions over 4 instead of 8 waves, hence equal to the standard
// set to rounding (tests/test_gpu_co.py), not bit for bit.
#define SRBM_CO 1
#define K1_THREADS 512
#define K2_THREADS 512
#define K3_THREADS 512
#define K4_THREADS 512
#define DN_THREADS 512
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>

#include "srbm_types.h"
#include "srbm_co.h"

namespace srbm_co {
#include "srbm_fused.hiph"
}  // namespace srbm_co

#define CO_CHK(x) do { if ((x) != hipSuccess) return -1; } while (0)

int srbm_co_configure(int N, size_t* lds_bytes) {
    using namespace srbm_co;
    // the fixed map of the IPM phase (no matrix in it) plus, where it fits below 80 KB, the copy of the dense state rows; kernels 1, 2, 4 of the
    // fused launch use windows of the same allocation
    size_t need = srbm_k3_lds_bytes(N);
    const size_t others = sizeof(K1Shared) > sizeof(K2Shared) ? (sizeof(K1Shared) > sizeof(K4Shared) ? sizeof(K1Shared) : sizeof(K4Shared))
                                                               : (sizeof(K2Shared) > sizeof(K4Shared) ? sizeof(K2Shared) : sizeof(K4Shared));
    if (need < others) need = others;
    if (need > K3_LDS_LAUNCH_BYTES) return -2;             // this horizon does not fit half a CU: the caller stays with the standard set
    *lds_bytes = K3_LDS_LAUNCH_BYTES;
    const int lds = (int)K3_LDS_LAUNCH_BYTES;
    CO_CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_k3_ipm), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CO_CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_k3_ipm_long), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CO_CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_rti_fused), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CO_CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_rti_fused_long), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    return 0;
}

int srbm_co_launch_step(hipStream_t stream, const SrbmParams* dp, SrbmInst* insts, SrbmWork* works, double* d_state, double* d_time, double* d_ee,
                        int batch, int N, size_t lds_bytes, double tol_step, double start_mu, hipEvent_t ev_a, hipEvent_t ev_b) {
    using namespace srbm_co;
    hipLaunchKernelGGL(srbm_k1_assemble, dim3(batch), dim3(K1_THREADS), 0, stream, dp, insts, works, d_state, d_time, d_ee);
    hipLaunchKernelGGL(srbm_k2_condense, dim3(batch), dim3(K2_THREADS), 0, stream, dp, insts, works);
    if (ev_a) CO_CHK(hipEventRecord(ev_a, stream));
    if (N <= K3_SHORT_N) hipLaunchKernelGGL(srbm_k3_ipm, dim3(batch), dim3(K3_THREADS), lds_bytes, stream, dp, insts, works, tol_step, start_mu);
    else hipLaunchKernelGGL(srbm_k3_ipm_long, dim3(batch), dim3(K3_THREADS), lds_bytes, stream, dp, insts, works, tol_step, start_mu);
    if (ev_b) CO_CHK(hipEventRecord(ev_b, stream));
    hipLaunchKernelGGL(srbm_k4_update, dim3(batch), dim3(K4_THREADS), 0, stream, dp, insts, works);
    CO_CHK(hipGetLastError());
    return 0;
}

int srbm_co_launch_fused(hipStream_t stream, const SrbmParams* dp, SrbmInst* insts, SrbmWork* works, int first_index, int steps, double* d_state,
                         double* d_time, double* d_ee, SrbmPlantArgs pl, int batch, int N, size_t lds_bytes) {
    using namespace srbm_co;
    if (N <= K3_SHORT_N)
        hipLaunchKernelGGL(srbm_rti_fused, dim3(batch), dim3(K3_THREADS), lds_bytes, stream, dp, insts, works, first_index, steps, d_state, d_time, d_ee, pl);
    else
        hipLaunchKernelGGL(srbm_rti_fused_long, dim3(batch), dim3(K3_THREADS), lds_bytes, stream, dp, insts, works, first_index, steps, d_state, d_time, d_ee, pl);
    CO_CHK(hipGetLastError());
    return 0;
}
