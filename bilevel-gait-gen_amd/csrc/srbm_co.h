// Launch interface of the CO-RESIDENT kernel set (srbm_co.hip) for srbm_capi.hip.  Same kernels, same sources as the standard set, compiled for 256
// threads per workgroup, the normal matrix of the IPM in the work record (L2) instead of LDS and <= 80 KB of LDS, so that TWO instances share a CU
// and fill each other's latency gaps -- chosen when a batch has more instances than the GPU has CUs (Config D's 512 per GPU, the 10 x batch
// candidates of a gait line search, closed-loop rollouts of large batches); the standard set stays the one for batches that fit one per CU.
#pragma once
#include <hip/hip_runtime.h>
#include "srbm_types.h"

// hipFuncSetAttribute of the set's kernels for horizon N; *lds_bytes = dynamic LDS every IPM / fused launch of the set asks for
int srbm_co_configure(int N, size_t* lds_bytes);
// one RTI step as four launches (assemble, condense, IPM, update); ev_a / ev_b (may be null) bracket the IPM kernel
int srbm_co_launch_step(hipStream_t stream, const SrbmParams* dp, SrbmInst* insts, SrbmWork* works, double* d_state, double* d_time, double* d_ee,
                        int batch, int N, size_t lds_bytes, double tol_step, double start_mu, hipEvent_t ev_a, hipEvent_t ev_b);
// K steps of the whole protocol in one launch
int srbm_co_launch_fused(hipStream_t stream, const SrbmParams* dp, SrbmInst* insts, SrbmWork* works, int first_index, int steps, double* d_state,
                         double* d_time, double* d_ee, SrbmPlantArgs pl, int batch, int N, size_t lds_bytes);
