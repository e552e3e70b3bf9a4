// Host side of the batched SRBM RTI path: the C-ABI of include/srbm_rti.h over the HIP kernels.
// Mirrors the call sequence of mpc::MPC / mpc::MPCSingleRigidBody (reference file:line cited in the header).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and prototypes only: RCCL is bound at run time (srbm_allgather_results), the library does not link it
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "srbm_k4_update.hiph"
#include "srbm_gait.hiph"
#include "srbm_plant.hiph"
#include "srbm_fused.hiph"
#include "srbm_ik.hiph"
#include "srbm_wbc.hiph"
#include "../../include/srbm_rti.h"

#ifdef SRBM_LARGE
#define SRBM_DYN_LDS(T) sizeof(T)        /* LARGE build: the working sets of kernels 1, 2, 4 exceed 64 KB of static LDS */
#else
#define SRBM_DYN_LDS(T) 0
#endif
static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return -1; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// temporaries of the debug / export entries: released on every return path
struct DevTemps {
    std::vector<void*> ptrs;
    ~DevTemps() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t alloc(T** out, size_t bytes) {
        void* p = nullptr;
        const hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) ptrs.push_back(p);
        *out = static_cast<T*>(p);
        return e;
    }
};

struct srbm_batch {
    int batch = 0, device = 0;
    SrbmParams hp{};                 // host copy of the parameters
    SrbmParams* dp = nullptr;
    SrbmInst* insts = nullptr;
    SrbmWork* works = nullptr;
    double *d_state = nullptr, *d_time = nullptr, *d_ee = nullptr;
    double *d_plant = nullptr, *d_push_time = nullptr, *d_push_impulse = nullptr;   // closed-loop harness (srbm_plant.hiph)
    bool push_set = false;
    hipStream_t stream = nullptr;
    bool owns_stream = true;
    size_t k3_lds = 0;
    int n_cu = 0;
    SrbmQueue* queues = nullptr;     // step queues of multi-step launches of a batch larger than the chip (srbm_fused.hiph), allocated at first use
    bool queued_ok = true;           // SRBM_NO_STEP_QUEUE=1 in the environment: such launches as one workgroup per instance (A/B, tests)
    bool params_dirty = true;
    // optional HIP-event timing of the dominant kernel (srbm_k3_ipm) on the launch stream
    bool timing = false;
    std::vector<hipEvent_t> ev_start, ev_stop;
    std::vector<int> ev_steps;        // RTI steps covered by each timed launch (1 for the stand-alone IPM kernel)
    size_t ev_used = 0;
    int gait_refs = 0;               // live srbm_gait handles borrowing this batch (and its stream)
    double last_tol_step = 0.0;      // tol_step of the last solve launched on this batch: > 0 means its duals may not be at the gap tolerance
    SrbmWbcParams* d_wbc = nullptr;  // whole-body QP model and gains (row f3), set by srbm_set_wbc_model
    void* d_scratch = nullptr;       // staging buffer of the small host->device entry points (grown on demand, never per call)
    size_t scratch_bytes = 0;
    void* h_stage = nullptr;         // pinned host mirror of d_scratch for the per-tick entries: ONE copy in, ONE copy out per call
    size_t stage_bytes = 0;
};
static int batch_scratch(srbm_batch* h, size_t bytes, void** out) {
    if (bytes > h->scratch_bytes) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->d_scratch) HIPCHK(hipFree(h->d_scratch));
        h->d_scratch = nullptr; h->scratch_bytes = 0;
        HIPCHK(hipMalloc(&h->d_scratch, bytes));
        h->scratch_bytes = bytes;
    }
    *out = h->d_scratch;
    return 0;
}
// device scratch + a pinned host buffer of the same size and layout
static int batch_stage(srbm_batch* h, size_t bytes, void** dev, void** host) {
    if (batch_scratch(h, bytes, dev)) return -1;
    if (bytes > h->stage_bytes) {
        if (h->h_stage) HIPCHK(hipHostFree(h->h_stage));
        h->h_stage = nullptr; h->stage_bytes = 0;
        HIPCHK(hipHostMalloc(&h->h_stage, bytes, hipHostMallocDefault));
        h->stage_bytes = bytes;
    }
    *host = h->h_stage;
    return 0;
}

__global__ void srbm_k_pack_results(const SrbmParams* __restrict__ Pp, const SrbmInst* __restrict__ insts, const SrbmWork* __restrict__ works,
                                    double* __restrict__ out, int ld) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const SrbmInst& I = insts[b];
    const int N = Pp->N, NX = 12 * (N + 1) + SRBM_NUMAX, NM = 12 * (N + 1) + 6 * SRBM_NSMAX + 16 * (N - 3) + 16;
    double* o = out + (size_t)b * ld;
    if (tid == 0) { o[0] = I.status; o[1] = I.n; o[2] = I.m; o[3] = I.cost; o[4] = I.alpha; o[5] = I.err_acc | I.err; o[6] = I.qp_iters; o[7] = I.init_time; }
    for (int i = tid; i < NX && 8 + i < ld; i += blockDim.x) o[8 + i] = i < I.n ? works[b].x[i] : 0.0;
    for (int i = tid; i < NM && 8 + NX + i < ld; i += blockDim.x) o[8 + NX + i] = i < I.m ? works[b].z[i] : 0.0;
    // contact times (Trajectory::GetContactTimes): counts of the four feet, then 8 slots per foot
    const int oc = 8 + NX + NM;
    if (tid < SRBM_NEE && oc + 4 + 8 * SRBM_NEE <= ld) {
        int n = 0;
        for (int i = 0; i < I.nk[tid]; i++)
            if (I.kind[tid][i] <= SRBM_K_TD) { if (n < 8) o[oc + 4 + 8 * tid + n] = I.knot_t[tid][i]; n++; }
        o[oc + tid] = n;
        for (int i = n; i < 8; i++) o[oc + 4 + 8 * tid + i] = 0.0;
    }
}

// Trajectory::GetForce / GetEndEffectorLocation / GetContacts of the current trajectory at time[b] (trajectory.cpp:395-410, :70-80)
__global__ void srbm_k_eval_trajectory(const SrbmParams* __restrict__ Pp, SrbmInst* __restrict__ insts, const double* __restrict__ time,
                                       double* __restrict__ force, double* __restrict__ pos, int* __restrict__ in_contact) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= Pp->batch * SRBM_NEE) return;
    const int b = w / SRBM_NEE, ee = w % SRBM_NEE;
    SrbmInst& I = insts[b];
    const FootView f{I.knot_t[ee], I.kind[ee], I.nk[ee]};
    int err = 0;
    const double t = time[b];
    double F[3], xy[2];
    srbm_force_value(f, &I.fval[ee][0][0][0], t, F, &err);
    srbm_posxy_value(f, &I.pval[ee][0][0], t, xy, &err);
    const double z = srbm_posz_value(f, t, Pp->swing_height, Pp->foot_offset, &err);
    const int lo = srbm_lower(f, SEL_POSXY, t, &err), up = srbm_upper(f, SEL_POSXY, t, &err);
    for (int c = 0; c < 3; c++) force[(size_t)w * 3 + c] = F[c];
    pos[(size_t)w * 3] = xy[0]; pos[(size_t)w * 3 + 1] = xy[1]; pos[(size_t)w * 3 + 2] = z;
    in_contact[w] = (f.kind[lo] == SRBM_K_TD && f.kind[up] == SRBM_K_LO) ? 1 : 0;      // EndEffectorSplines::IsInContact (:805-813)
    if (err) atomicOr(&I.err, err);
}

// ---------------- small device kernels of the host protocol ----------------

// default contact schedule [0,.3,.6,.9,1.2] per foot (mpc.cpp:566-608), FR and RL start in stance (trajectory.cpp:25-28),
// three force polynomials per stance (end_effector_splines.cpp:34-153)
__global__ void srbm_k_init(const SrbmParams* __restrict__ Pp, SrbmInst* __restrict__ insts) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Pp->batch) return;
    SrbmInst& I = insts[b];
    const double times[5] = {0, 0.3, 0.6, 0.9, 1.2};
    for (int ee = 0; ee < SRBM_NEE; ee++) {
        const bool start_in_contact = (ee == 1 || ee == 2);
        // pattern of 5 knot kinds, repeated until all contact times are placed
        const uint8_t pat_sw[5] = {SRBM_K_LO, SRBM_K_MID, SRBM_K_TD, SRBM_K_F, SRBM_K_F};
        const uint8_t pat_st[5] = {SRBM_K_TD, SRBM_K_F, SRBM_K_F, SRBM_K_LO, SRBM_K_MID};
        int i = 0, j = 0, k = 1, nk = 0;
        while (i < 5) {
            const uint8_t kd = start_in_contact ? pat_st[j % 5] : pat_sw[j % 5];
            double t;
            if (kd == SRBM_K_F) {
                const double d = times[i] - times[i - 1];
                const double kdv = k * d;
                const double q = kdv / 3;
                t = times[i - 1] + q;
                k++;
            } else if (kd == SRBM_K_MID) {
                const double d = times[i] - times[i - 1];
                const double hlf = d / 2;
                t = times[i - 1] + hlf;
            } else {
                t = times[i];
                i++; k = 1;
            }
            I.knot_t[ee][nk] = t; I.kind[ee][nk] = kd;
            nk++; j++;
        }
        I.nk[ee] = nk;
        for (int q = nk; q < SRBM_KMAX; q++) { I.knot_t[ee][q] = 0; I.kind[ee][q] = 0; }
        for (int c = 0; c < 3; c++) for (int q = 0; q < SRBM_KMAX; q++) { I.fval[ee][c][q][0] = 0; I.fval[ee][c][q][1] = 0; }
        for (int c = 0; c < 2; c++) for (int q = 0; q < SRBM_KMAX; q++) I.pval[ee][c][q] = 0;
    }
    for (int i = 0; i < (SRBM_NMAX + 1) * 13; i++) I.states[i] = 0;
    I.box[0] = Pp->box0[0]; I.box[1] = Pp->box0[1];
    I.init_time = 0; I.alpha = 0; I.cost = 0; I.eq_violation = 0; I.step_norm = 0; I.qp_cost = 0; I.res_primal = 0; I.res_dual = 0; I.gap = 0;
    I.status = SRBM_UNSOLVED; I.qp_iters = 0; I.n = 0; I.m = 0; I.n_eq = 0; I.n_ineq = 0; I.nfv = 0; I.npv = 0; I.n_td = 0; I.n_samples = 0;
    I.err = 0; I.run_num = 0; I.acc_iters = 0; I.acc_flops = 0;
    I.cost_sum = 0; I.merit_dd = 0; I.acc_mfma = 0; I.err_acc = 0; I.n_solves = 0; I.n_not_solved = 0; I.n_maxiter = 0;
    I.low_streak = 0; I.last_rule = 0; I.last_low = 0; I.pad_low = 0; I.low_skip = 0; I.n_low_tried = 0; I.n_low_failed = 0; I.n_step_rule = 0;
}

__global__ void srbm_k_warm_start(const SrbmParams* __restrict__ Pp, SrbmInst* __restrict__ insts, const double* __restrict__ states) {
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < (Pp->N + 1) * 13; i += blockDim.x) insts[b].states[i] = states[(size_t)b * 13 + (i % 13)];
}

// EndEffectorSplines::SetContactTimes (end_effector_splines.cpp:860-892) for every foot of every instance
__global__ void srbm_k_set_contact_times(const SrbmParams* __restrict__ Pp, SrbmInst* __restrict__ insts, const double* __restrict__ times, int ld) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= Pp->batch * SRBM_NEE) return;
    const int b = w / SRBM_NEE, ee = w % SRBM_NEE;
    SrbmInst& I = insts[b];
    srbm_apply_contact_times(I, ee, times + ((size_t)b * SRBM_NEE + ee) * ld, srbm_num_contacts(I, ee));
}

// ---------------- trajectory -> whole-body targets (srbm_ik.hiph) ----------------
__device__ __forceinline__ const SrbmLegs& srbm_legs(const SrbmParams& P) { return *reinterpret_cast<const SrbmLegs*>(&P.legs[0][0][0]); }
__global__ void srbm_k_forward_kinematics(const SrbmParams* __restrict__ Pp, const double* __restrict__ q, double* __restrict__ ee) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Pp->batch) return;
    ik_forward_kinematics(srbm_legs(*Pp), q + (size_t)b * 19, ee + (size_t)b * 12);
}
// One wave per instance (srbm_ik.hiph): every lane runs the same chain on the same data, the small vectors live in LDS, lane 0 stores.
#define SRBM_IK_THREADS 64
__global__ __launch_bounds__(SRBM_IK_THREADS) void srbm_k_inverse_kinematics(const SrbmParams* __restrict__ Pp, const double* __restrict__ state, const double* __restrict__ ee,
                                          const double* __restrict__ q_guess, double* __restrict__ q_out, int* __restrict__ iters, int* __restrict__ status) {
    const int b = blockIdx.x, lane = threadIdx.x;
    __shared__ double q[19], s13[13], e12[12];
    __shared__ int it[4];
    if (lane < 19) q[lane] = q_guess[(size_t)b * 19 + lane];
    if (lane < 13) s13[lane] = state[(size_t)b * 13 + lane];
    if (lane < 12) e12[lane] = ee[(size_t)b * 12 + lane];
    __syncthreads();
    const int failed = ik_inverse_kinematics(srbm_legs(*Pp), s13, e12, q, it);
    __syncthreads();
    if (lane < 19) q_out[(size_t)b * 19 + lane] = q[lane];
    if (lane < 4) iters[b * 4 + lane] = it[lane];
    if (lane == 0) status[b] = failed;
}
// MPCController::GetTargetsFromTraj: two IK solves per instance -- the targets at `time`, then at time + dt from the first solve's joint angles.
// TWO WAVES per instance.  Inside a solve the feet are strictly one after the other (they share the base pose), and the second solve takes from the first
// only the joint angles of a leg as the guess of that leg's foot -- final as soon as the first solve has finished THAT foot.  So wave 1 runs the second
// solve one foot behind wave 0: five foot slots instead of eight, each wave the same chain of operations on the same data as before (bitwise the same
// results), on its own SIMD of the CU.
#define SRBM_TT_THREADS (2 * SRBM_IK_THREADS)
__global__ __launch_bounds__(SRBM_TT_THREADS) void srbm_k_targets_from_traj(const SrbmParams* __restrict__ Pp, const SrbmInst* __restrict__ insts, const double* __restrict__ time_in,
                                         double* __restrict__ q_des, double* __restrict__ v_des, double* __restrict__ force_des, int* __restrict__ status) {
    const int b = blockIdx.x, lane = threadIdx.x, wv = threadIdx.x / SRBM_IK_THREADS;
    const SrbmParams& P = *Pp;
    const SrbmInst& I = insts[b];
    const int N = P.N;
    const double t0 = I.init_time, dt = P.dt;
    double time = time_in[b];
    if (time < t0) time = t0;
    const int node = (int)ceil((time - t0) / dt);
    if (node < 0 || node + 1 > N) { if (lane == 0) status[b] = 2; return; }          // GetState(node + 1) beyond the horizon: the reference's vector access throws
    __shared__ double s1[13], s2[13], ee1[12], ee2[12], F[12], q[19], q2[19];
    __shared__ int flags[2], ik_failed[2];
    auto T = [&](int k) { return t0 + dt * k; };
    const double* S = I.states;
    if (lane < 2) flags[lane] = 0;
    __syncthreads();
    if (lane < 13) {
        const int i = lane;
        if (node > 0) {
            const double a = 1 - (T(node) - time) / (T(node) - T(node - 1));
            s1[i] = (S[node * 13 + i] - S[(node - 1) * 13 + i]) * a + S[(node - 1) * 13 + i];
            const double c = 1 - (T(node + 1) - (time + dt)) / (T(node + 1) - T(node));
            s2[i] = (S[(node + 1) * 13 + i] - S[node * 13 + i]) * c + S[node * 13 + i];
        } else {
            const double a = 1 - (T(1) - time) / (T(1) - T(0));
            const double c = 1 - (T(1) - (time + dt)) / (T(1) - T(0));
            s1[i] = (S[13 + i] - S[i]) * a + S[i]; s2[i] = (S[13 + i] - S[i]) * c + S[i];
        }
    }
    if (lane >= 16 && lane < 16 + SRBM_NEE) {        // the splines of one foot per lane
        const int ee = lane - 16;
        int err = 0;
        const FootView f{I.knot_t[ee], I.kind[ee], I.nk[ee]};
        double e1[3], e2[3], ff[3];
        srbm_posxy_value(f, &I.pval[ee][0][0], time, e1, &err);
        e1[2] = srbm_posz_value(f, time, P.swing_height, P.foot_offset, &err);
        srbm_posxy_value(f, &I.pval[ee][0][0], time + dt, e2, &err);
        e2[2] = srbm_posz_value(f, time + dt, P.swing_height, P.foot_offset, &err);
        srbm_force_value(f, &I.fval[ee][0][0][0], time, ff, &err);
        for (int c = 0; c < 3; c++) { ee1[3 * ee + c] = e1[c]; ee2[3 * ee + c] = e2[c]; F[3 * ee + c] = ff[c]; }
        if (err) atomicOr(&flags[0], 1);
    }
    if (lane >= 32 && lane < 32 + 19) q[lane - 32] = q_des[(size_t)b * 19 + lane - 32];
    __syncthreads();
    int st = 0;
    if (node > 0 && time + dt < T(node)) st = 2;                              // "bad interp."
    if (flags[0]) st = 2;
    const SrbmLegs& L = srbm_legs(P);
    {
        // SingleRigidBodyModel::InverseKinematics (ik_inverse_kinematics, srbm_ik.hiph) of this wave's solve, foot by foot in step with the other wave
        const double* st13 = wv ? s2 : s1;
        const double* eed = wv ? ee2 : ee1;
        double* qo = wv ? q2 : q;
        const IkR Rdes = ik_quat_to_R(st13 + 6);
        const Ik3 pdes = {st13[0], st13[1], st13[2]};
        double p[3] = {st13[0], st13[1], st13[2]}, qt[4] = {st13[6], st13[7], st13[8], st13[9]};
        int failed = 0;
        bool any_success = false;
        #pragma unroll 1
        for (int slot = 0; slot < SRBM_NEE + 1; slot++) {
            const int ee = slot - wv;
            if (ee >= 0 && ee < SRBM_NEE) {
                const Ik3 edes = {eed[3 * ee], eed[3 * ee + 1], eed[3 * ee + 2]};
                double ang[3] = {q[7 + 3 * ee], q[8 + 3 * ee], q[9 + 3 * ee]};        // wave 0: the guess handed in; wave 1: wave 0's result of the slot before
                bool success;
                ik_solve_foot(&L.origin[ee][0][0], Rdes, pdes, edes, p, qt, ang, &success);
                qo[7 + 3 * ee] = ang[0]; qo[8 + 3 * ee] = ang[1]; qo[9 + 3 * ee] = ang[2];
                any_success = any_success || success;
                if (!any_success) failed = 1;
            }
            __syncthreads();
        }
        qo[0] = p[0]; qo[1] = p[1]; qo[2] = p[2]; qo[3] = qt[0]; qo[4] = qt[1]; qo[5] = qt[2]; qo[6] = qt[3];
        if ((lane & (SRBM_IK_THREADS - 1)) == 0) ik_failed[wv] = failed;
    }
    __syncthreads();
    if ((ik_failed[0] || ik_failed[1]) && st == 0) st = 1;
    double* v = v_des + (size_t)b * 18;
    if (lane < 3) v[lane] = s1[3 + lane] / P.mass;
    else if (lane < 6) { const int i = lane - 3; v[lane] = P.Ir_inv[3 * i] * s1[10] + P.Ir_inv[3 * i + 1] * s1[11] + P.Ir_inv[3 * i + 2] * s1[12]; }
    else if (lane < 18) { const int j = lane - 6; v[lane] = (-q[7 + j] + q2[7 + j]) / dt; }
    if (lane < 19) q_des[(size_t)b * 19 + lane] = q[lane];
    if (lane < 12) force_des[(size_t)b * 12 + lane] = F[lane];
    if (lane == 0) status[b] = st;
}

// ---------------- host helpers ----------------
static int upload_params(srbm_batch* h) {
    if (!h->params_dirty) return 0;
    h->hp.q_diag = 1;
    for (int i = 0; i < 144; i++) if (i % 13 != 0 && (h->hp.Q[i] != 0.0 || h->hp.Phi[i] != 0.0)) h->hp.q_diag = 0;
    HIPCHK(hipMemcpyAsync(h->dp, &h->hp, sizeof(SrbmParams), hipMemcpyHostToDevice, h->stream));
    h->params_dirty = false;
    return 0;
}
static void inv3(const double* m, double* r) {
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], hh = m[7], i = m[8];
    const double det = a * (e * i - f * hh) - b * (d * i - f * g) + c * (d * hh - e * g);
    r[0] = (e * i - f * hh) / det; r[1] = (c * hh - b * i) / det; r[2] = (b * f - c * e) / det;
    r[3] = (f * g - d * i) / det; r[4] = (a * i - c * g) / det; r[5] = (c * d - a * f) / det;
    r[6] = (d * hh - e * g) / det; r[7] = (b * g - a * hh) / det; r[8] = (a * e - b * d) / det;
}
// one RTI step as four launches.  exact: the solve is taken to the gap criterion whatever the batch's step rule says (the solve whose KKT
// sensitivity the gait step differentiates)
// One RTI step on inputs already in h->d_state / d_time / d_ee: four kernels.  exact: to the gap criterion whatever the batch's step rule says (the solve a gait
// gradient differentiates).
// (Measured in round 5: the same step as ONE launch of the fused kernel -- no grid-wide wait between the phases, a workgroup through with its line-search
//  candidate takes the next one -- is SLOWER: gait segment 7.06 -> 7.34 ms per step; the stand-alone IPM kernel is an entry function -- its uniform
//  loads are scalar, it spills less (scratch 396 B per lane against 988) -- and that outweighs three kernel tails.)
static int launch_step(srbm_batch* h, bool exact = false) {
    if (upload_params(h)) return -1;
    // (start_mu only in the fused K-step launches: the lower-start attempt trades a shorter mean for a longer tail, and a one-step launch ends with
    //  the slowest instance of the batch -- srbm_k3_ipm.hiph)
    //  (Tried for the 10 x batch candidates of a gait line search, where dynamic workgroup scheduling evens out the tail: a candidate's linearisation
    //  point belongs to ANOTHER contact schedule -- 17-35 % of the attempts are repeated, the gait segment goes from 8.3 to 10.4-11.8 ms per step.)
    const double tol_step = exact ? 0.0 : h->hp.tol_step, start_mu = 0.0;
    h->last_tol_step = tol_step;
    const int B = h->batch;
    hipLaunchKernelGGL(srbm_k1_assemble, dim3(B), dim3(K1_THREADS), SRBM_DYN_LDS(K1Shared), h->stream, h->dp, h->insts, h->works, h->d_state, h->d_time, h->d_ee);
    hipLaunchKernelGGL(srbm_k2_condense, dim3(B), dim3(K2_THREADS), SRBM_DYN_LDS(K2Shared), h->stream, h->dp, h->insts, h->works);
    const bool tm = h->timing && h->ev_used < h->ev_start.size();
    if (tm) HIPCHK(hipEventRecord(h->ev_start[h->ev_used], h->stream));
    if (h->hp.N <= K3_SHORT_N) hipLaunchKernelGGL(srbm_k3_ipm, dim3(B), dim3(K3_THREADS), h->k3_lds, h->stream, h->dp, h->insts, h->works, tol_step, start_mu);
    else hipLaunchKernelGGL(srbm_k3_ipm_long, dim3(B), dim3(K3_THREADS), h->k3_lds, h->stream, h->dp, h->insts, h->works, tol_step, start_mu);
    if (tm) { HIPCHK(hipEventRecord(h->ev_stop[h->ev_used], h->stream)); h->ev_steps[h->ev_used] = 1; h->ev_used++; }
    hipLaunchKernelGGL(srbm_k4_update, dim3(B), dim3(K4_THREADS), SRBM_DYN_LDS(K4Shared), h->stream, h->dp, h->insts, h->works);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" {

static int fetch_insts(srbm_batch* h, std::vector<SrbmInst>& v);
const char* srbm_last_error(void) { return g_err.c_str(); }
long srbm_bytes_per_instance(void) { return (long)(sizeof(SrbmInst) + sizeof(SrbmWork)); }
/* diagnostic builds (-DSRBM_PROFILE) only: cycles per IPM phase of one instance, 16 slots */
int srbm_debug_get_profile(srbm_batch* h, int inst, double* out16) {
    if (!h || inst < 0 || inst >= h->batch) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out16, reinterpret_cast<const char*>(h->works + inst) + offsetof(SrbmWork, prof), sizeof(double) * 16, hipMemcpyDeviceToHost));
    return 0;
}
int srbm_debug_get_profile2(srbm_batch* h, int inst, double* out96) {
    if (!h || inst < 0 || inst >= h->batch) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out96, reinterpret_cast<const char*>(h->works + inst) + offsetof(SrbmWork, prof2), sizeof(double) * 96, hipMemcpyDeviceToHost));
    return 0;
}
// unit-test hook for the dense building blocks: Cholesky of `count` packed lower-triangular n x n matrices, one workgroup each
__global__ __launch_bounds__(DN_THREADS) void srbm_k_debug_solve(int n, const double* __restrict__ Min, const double* __restrict__ rhs,
                                                               double* __restrict__ xout, double* __restrict__ Xout, int* __restrict__ ticks) {
    extern __shared__ double dbg_smem2[];
    const int np = n * (n + 1) / 2;
    double* M = dbg_smem2;
    double* panel = dbg_smem2 + (size_t)SRBM_NUMAX * (SRBM_NUMAX + 1) / 2;
    double* xv = panel + DN_PANEL_DOUBLES;
    double* tv = xv + SRBM_NUMAX;
    const double* src = Min + (size_t)blockIdx.x * np;
    for (int e = threadIdx.x; e < np; e += DN_THREADS) M[e] = src[e];
    for (int e = threadIdx.x; e < n; e += DN_THREADS) xv[e] = rhs[(size_t)blockIdx.x * n + e];
    __syncthreads();
    DnTiles T;
    int nreg = 0;
    dn_load_packed(T, M, n);
    dn_cholesky(T, M, n, panel, &nreg);
    chol_invert_diag_blocks(M, n, panel);
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    dn_trtri(M, n, panel);
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    dn_solve_inv(0, n, (int)(xv - dbg_smem2), (int)(tv - dbg_smem2), -1, 0, 0
#ifdef SRBM_M_GLOBAL
                 , M
#endif
    );
    const long long t2 = (long long)__builtin_amdgcn_s_memtime();
    for (int e = threadIdx.x; e < n; e += DN_THREADS) xout[(size_t)blockIdx.x * n + e] = xv[e];
    for (int e = threadIdx.x; e < np; e += DN_THREADS) Xout[(size_t)blockIdx.x * np + e] = M[e];
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = (int)(t1 - t0); ticks[2 * blockIdx.x + 1] = (int)(t2 - t1); }
}
__global__ __launch_bounds__(DN_THREADS) void srbm_k_debug_cholesky(int n, const double* __restrict__ Min, double* __restrict__ Lout, int* __restrict__ nreg_out) {
    extern __shared__ double dbg_smem[];
    const int np = n * (n + 1) / 2;
    double* M = dbg_smem;
    double* panel = dbg_smem + (size_t)SRBM_NUMAX * (SRBM_NUMAX + 1) / 2;
    const double* src = Min + (size_t)blockIdx.x * np;
    for (int e = threadIdx.x; e < np; e += DN_THREADS) M[e] = src[e];
    __syncthreads();
    DnTiles T;
    int nreg = 0;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    dn_load_packed(T, M, n);
    dn_cholesky(T, M, n, panel, &nreg);
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    dn_scale_factor_columns(M, n, panel);         // (the test looks at L; dn_cholesky leaves L diag(sqrt(d)))
    for (int e = threadIdx.x; e < np; e += DN_THREADS) Lout[(size_t)blockIdx.x * np + e] = M[e];
    if (threadIdx.x == 0) nreg_out[blockIdx.x] = nreg | ((int)min((long long)0x7fffff, (t1 - t0) >> 4) << 8);   // bits 8..: ticks / 16 (diagnostic)
}
// unit-test hook for the dense phase on a subset of the columns (what the IPM does with the pinned position variables): tiles loaded through
// the column map, factor + inverse of the mapped block, solve with the gather / scatter through the map; unmapped entries of x keep rhs
__global__ __launch_bounds__(DN_THREADS) void srbm_k_debug_solve_mapped(int n, int nc, const int* __restrict__ map, const double* __restrict__ Min,
                                                                      const double* __restrict__ rhs, double* __restrict__ xout, int* __restrict__ nreg_out) {
    extern __shared__ double dbg_smem3[];
    const int np = n * (n + 1) / 2;
    double* M = dbg_smem3;
    double* panel = dbg_smem3 + (size_t)SRBM_NUMAX * (SRBM_NUMAX + 1) / 2;
    double* xv = panel + DN_PANEL_DOUBLES;
    double* tv = xv + SRBM_NUMAX;
    double* xc = tv + SRBM_NUMAX;
    int* imap = reinterpret_cast<int*>(xc + SRBM_NUMAX);
    const double* src = Min + (size_t)blockIdx.x * np;
    for (int e = threadIdx.x; e < np; e += DN_THREADS) M[e] = src[e];
    for (int e = threadIdx.x; e < n; e += DN_THREADS) xv[e] = rhs[(size_t)blockIdx.x * n + e];
    for (int e = threadIdx.x; e < nc; e += DN_THREADS) imap[e] = map[e];
    __syncthreads();
    DnTiles T;
    int nreg = 0;
    dn_load_packed(T, M, nc, imap);
    dn_cholesky(T, M, nc, panel, &nreg);
    chol_invert_diag_blocks(M, nc, panel);
    dn_trtri(M, nc, panel);
    dn_solve_inv(0, nc, (int)(xv - dbg_smem3), (int)(tv - dbg_smem3), (int)(reinterpret_cast<double*>(imap) - dbg_smem3), (int)(xc - dbg_smem3), 0
#ifdef SRBM_M_GLOBAL
                 , M
#endif
    );
    for (int e = threadIdx.x; e < n; e += DN_THREADS) xout[(size_t)blockIdx.x * n + e] = xv[e];
    if (threadIdx.x == 0) nreg_out[blockIdx.x] = nreg;
}
int srbm_debug_solve_mapped(int n, int nc, const int* map, int count, const double* M_packed, const double* rhs, double* x, int* nreg) {
#ifdef SRBM_LARGE
    return fail("srbm_debug_solve_mapped: the unit-test hooks of the dense blocks exist in the standard build only");
#endif
    if (n <= 0 || n > SRBM_NUMAX || nc <= 0 || nc > n || count <= 0 || !map || !M_packed || !rhs || !x || !nreg) return fail("bad arguments");
    for (int k = 0; k < nc; k++) if (map[k] < 0 || map[k] >= n || (k > 0 && map[k] <= map[k - 1])) return fail("srbm_debug_solve_mapped: the map must be increasing and within [0, n)");
    const size_t np = (size_t)n * (n + 1) / 2, bytes = np * count * sizeof(double), vb = (size_t)n * count * sizeof(double);
    double *dM = nullptr, *dr = nullptr, *dx = nullptr; int *dmap = nullptr, *dn = nullptr;
    DevTemps tmp;
    HIPCHK(tmp.alloc(&dM, bytes)); HIPCHK(tmp.alloc(&dr, vb)); HIPCHK(tmp.alloc(&dx, vb)); HIPCHK(tmp.alloc(&dmap, sizeof(int) * nc)); HIPCHK(tmp.alloc(&dn, sizeof(int) * count));
    HIPCHK(hipMemcpy(dM, M_packed, bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dr, rhs, vb, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dmap, map, sizeof(int) * nc, hipMemcpyHostToDevice));
    const size_t lds = ((size_t)SRBM_NUMAX * (SRBM_NUMAX + 1) / 2 + DN_PANEL_DOUBLES + 3 * SRBM_NUMAX + (SRBM_NUMAX + 1) / 2) * sizeof(double);
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_k_debug_solve_mapped), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(srbm_k_debug_solve_mapped, dim3(count), dim3(DN_THREADS), lds, 0, n, nc, dmap, dM, dr, dx, dn);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(x, dx, vb, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(nreg, dn, sizeof(int) * count, hipMemcpyDeviceToHost));
    return 0;
}
/* unit-test hook: x = M^-1 rhs through Cholesky + explicit inverse of the factor; X_packed = L^-1; ticks[2*count] */
int srbm_debug_solve(int n, int count, const double* M_packed, const double* rhs, double* x, double* X_packed, int* ticks) {
#ifdef SRBM_LARGE
    return fail("srbm_debug_solve: the unit-test hooks of the dense blocks exist in the standard build only");
#endif
    if (n <= 0 || n > SRBM_NUMAX || count <= 0 || !M_packed || !rhs || !x || !X_packed || !ticks) return fail("bad arguments");
    const size_t np = (size_t)n * (n + 1) / 2, bytes = np * count * sizeof(double), vb = (size_t)n * count * sizeof(double);
    double *dM = nullptr, *dX = nullptr, *dr = nullptr, *dx = nullptr; int* dt = nullptr;
    DevTemps tmp;
    HIPCHK(tmp.alloc(&dM, bytes)); HIPCHK(tmp.alloc(&dX, bytes)); HIPCHK(tmp.alloc(&dr, vb)); HIPCHK(tmp.alloc(&dx, vb));
    HIPCHK(tmp.alloc(&dt, sizeof(int) * 2 * count));
    HIPCHK(hipMemcpy(dM, M_packed, bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dr, rhs, vb, hipMemcpyHostToDevice));
    const size_t lds = ((size_t)SRBM_NUMAX * (SRBM_NUMAX + 1) / 2 + DN_PANEL_DOUBLES + 2 * SRBM_NUMAX) * sizeof(double);
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_k_debug_solve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(srbm_k_debug_solve, dim3(count), dim3(DN_THREADS), lds, 0, n, dM, dr, dx, dX, dt);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(x, dx, vb, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(X_packed, dX, bytes, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ticks, dt, sizeof(int) * 2 * count, hipMemcpyDeviceToHost));
    return 0;
}
int srbm_debug_cholesky(int n, int count, const double* M_packed, double* L_packed, int* nreg) {
#ifdef SRBM_LARGE
    return fail("srbm_debug_cholesky: the unit-test hooks of the dense blocks exist in the standard build only");
#endif
    if (n <= 0 || n > SRBM_NUMAX || count <= 0 || !M_packed || !L_packed || !nreg) return fail("bad arguments");
    const size_t np = (size_t)n * (n + 1) / 2, bytes = np * count * sizeof(double);
    double *dM = nullptr, *dL = nullptr; int* dr = nullptr;
    DevTemps tmp;
    HIPCHK(tmp.alloc(&dM, bytes)); HIPCHK(tmp.alloc(&dL, bytes)); HIPCHK(tmp.alloc(&dr, sizeof(int) * count));
    HIPCHK(hipMemcpy(dM, M_packed, bytes, hipMemcpyHostToDevice));
    const size_t lds = ((size_t)SRBM_NUMAX * (SRBM_NUMAX + 1) / 2 + DN_PANEL_DOUBLES) * sizeof(double);
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_k_debug_cholesky), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(srbm_k_debug_cholesky, dim3(count), dim3(DN_THREADS), lds, 0, n, dM, dL, dr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(L_packed, dL, bytes, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(nreg, dr, sizeof(int) * count, hipMemcpyDeviceToHost));
    return 0;
}
int srbm_debug_get_trace(srbm_batch* h, int inst, double* out256) {
    if (!h || inst < 0 || inst >= h->batch) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out256, reinterpret_cast<const char*>(h->works + inst) + offsetof(SrbmWork, dbg), sizeof(double) * 384, hipMemcpyDeviceToHost));
    return 0;
}

// diagnostic: the spline variables of the linearisation point and of the QP minimiser of instance `inst`, with the column descriptors
// (foot, type 0 force / 1 position, coordinate, local index) and the pin / substitution mask -- scripts/dev_attempts.py
int srbm_debug_get_spline_step(srbm_batch* h, int inst, double* u_prev, double* u, int* cols4, int* fix) {
    if (!h || inst < 0 || inst >= h->batch || !u_prev || !u || !cols4 || !fix) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const char* W = reinterpret_cast<const char*>(h->works + inst);
    HIPCHK(hipMemcpy(u_prev, W + offsetof(SrbmWork, u_prev), sizeof(double) * SRBM_NUMAX, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(u, W + offsetof(SrbmWork, u), sizeof(double) * SRBM_NUMAX, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cols4, W + offsetof(SrbmWork, col_ee), sizeof(int) * SRBM_NUMAX, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cols4 + SRBM_NUMAX, W + offsetof(SrbmWork, col_type), sizeof(int) * SRBM_NUMAX, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cols4 + 2 * SRBM_NUMAX, W + offsetof(SrbmWork, col_coord), sizeof(int) * SRBM_NUMAX, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cols4 + 3 * SRBM_NUMAX, W + offsetof(SrbmWork, col_local), sizeof(int) * SRBM_NUMAX, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(fix, W + offsetof(SrbmWork, fix_mask), sizeof(int) * SRBM_NUMAX, hipMemcpyDeviceToHost));
    return 0;
}

// device buffers + kernel attributes of a batch whose host parameters (h->hp, batch, device) are set; on failure everything
// allocated so far is released by the caller through free_batch()
static void free_batch(srbm_batch* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->dp); (void)hipFree(h->insts); (void)hipFree(h->works); (void)hipFree(h->queues);
    (void)hipFree(h->d_state); (void)hipFree(h->d_time); (void)hipFree(h->d_ee);
    (void)hipFree(h->d_plant); (void)hipFree(h->d_push_time); (void)hipFree(h->d_push_impulse);
    (void)hipFree(h->d_scratch); (void)hipFree(h->d_wbc); (void)hipHostFree(h->h_stage);
    for (auto e : h->ev_start) (void)hipEventDestroy(e);
    for (auto e : h->ev_stop) (void)hipEventDestroy(e);
    if (h->owns_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}
static int alloc_batch(srbm_batch* h, hipStream_t borrowed_stream) {
    const size_t B = h->batch;
    if (borrowed_stream) { h->stream = borrowed_stream; h->owns_stream = false; }
    else { HIPCHK(hipStreamCreate(&h->stream)); h->owns_stream = true; }
    HIPCHK(hipMalloc(&h->dp, sizeof(SrbmParams)));
    HIPCHK(hipMalloc(&h->insts, sizeof(SrbmInst) * B));
    HIPCHK(hipMalloc(&h->works, sizeof(SrbmWork) * B));
    HIPCHK(hipMalloc(&h->d_state, sizeof(double) * 13 * B));
    HIPCHK(hipMalloc(&h->d_time, sizeof(double) * B));
    HIPCHK(hipMalloc(&h->d_ee, sizeof(double) * 12 * B));
    // the IPM kernel gets the whole LDS of a CU: what its fixed map leaves over holds the dense state rows (K3Smem::sig_row)
    h->k3_lds = K3_LDS_LAUNCH_BYTES;
    if (srbm_k3_lds_bytes(h->hp.N) > h->k3_lds) return fail("srbm_batch_create: LDS map exceeds 160 KB");
    h->hp.lds_doubles = (int)(h->k3_lds / sizeof(double));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_k3_ipm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->k3_lds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_k3_ipm_long), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->k3_lds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_rti_fused), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->k3_lds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_rti_fused_long), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->k3_lds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_rti_queued), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->k3_lds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_rti_queued_long), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->k3_lds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_k3_normal_matrix), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->k3_lds));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_k_gait_sensitivity), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KG_DYN_LDS_BYTES));
#ifdef SRBM_LARGE
    static_assert(sizeof(K1Shared) <= 160 * 1024 && sizeof(K2Shared) <= 160 * 1024 && sizeof(K4Shared) <= 160 * 1024, "working sets fit the LDS of a CU");
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_k1_assemble), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(K1Shared)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_k2_condense), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(K2Shared)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(srbm_k4_update), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(K4Shared)));
#endif
    HIPCHK(hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, h->device));
    { const char* e = std::getenv("SRBM_NO_STEP_QUEUE"); h->queued_ok = !(e && e[0] == '1'); }
    h->params_dirty = true;
    return 0;
}

int srbm_batch_create(srbm_batch** out, int batch, const srbm_mpc_info* info, const srbm_model* model, int device) {
    if (!out || !info || !model || batch <= 0) return fail("srbm_batch_create: bad arguments");
    if (info->num_nodes < 5 || info->num_nodes > SRBM_NMAX) return fail("srbm_batch_create: num_nodes must be in [5, " + std::to_string(SRBM_NMAX) + "]");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("srbm_batch_create: no HIP device (this library has no CPU path)");
    HIPCHK(hipSetDevice(device));
    auto* h = new srbm_batch;
    h->batch = batch; h->device = device;
    SrbmParams& p = h->hp;
    std::memset(&p, 0, sizeof(p));
    p.batch = batch; p.N = info->num_nodes; p.max_iter = 200;
    p.dt = info->integrator_dt; p.mu_fric = info->friction_coef; p.force_bound = info->force_bound;
    p.swing_height = info->swing_height; p.foot_offset = info->foot_offset; p.force_cost = info->force_cost;
    p.box0[0] = info->ee_box_size[0]; p.box0[1] = info->ee_box_size[1];
    p.mass = model->mass;
    std::memcpy(p.Ir, model->Ir, sizeof(p.Ir));
    inv3(p.Ir, p.Ir_inv);
    for (int ee = 0; ee < 4; ee++) {      // GetCOMToHip, single_rigid_body_model.cpp:289-305
        double x = model->hip_xy[2 * ee], y = model->hip_xy[2 * ee + 1];
        if (y >= 0) y += 0.1; else y -= 0.1;
        x += 0.025;
        p.hip[2 * ee] = x; p.hip[2 * ee + 1] = y;
    }
    p.merit_mu = 5000; p.td_fraction = 0.75;
    // ClarabelInterface::ConfigureForInitialRun / ConfigureForRealTime (clarabel_interface.cpp:165-175) run every solve of the
    // reference at tol_gap 1e-15, tol_feas 1e-10: the defaults here.  The gap of this QP stops improving around 1e-14..1e-15
    // in fp64 (the loop then ends on its progress test), but WHERE it stops decides how well the minimiser is determined along
    // the flat directions of the weakly convex QP: measured against the CPU restatement of the reference on identical QPs (scripts/dev_accuracy.py,
    // 1024 solves) the worst relative primal error is 1.3e-4 at 1e-13, 5e-5 at 1e-14 and 1.4e-5 at 1e-15, for 17.3 / 18.0 /
    // 18.9 IPM iterations per solve.  The parity tolerance of the path is 1e-4.  srbm_set_solver_tolerances overrides.
    p.tol_gap_abs = 1e-15; p.tol_gap_rel = 1e-15; p.tol_feas = 1e-10;
    p.tol_step = 0.0; p.start_mu = 0.0;       // the reference's criterion; srbm_set_solver_step_rule opts into the faster termination
    auto bail = [&]() { free_batch(h); return -1; };
    if (alloc_batch(h, nullptr)) return bail();
    if (hipMemsetAsync(h->works, 0, sizeof(SrbmWork) * (size_t)batch, h->stream) != hipSuccess) { fail("srbm_batch_create: memset failed"); return bail(); }
    if (upload_params(h)) return bail();
    hipLaunchKernelGGL(srbm_k_init, dim3((batch + 63) / 64), dim3(64), 0, h->stream, h->dp, h->insts);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) { fail("srbm_batch_create: initialisation kernel failed"); return bail(); }
    *out = h;
    return 0;
}

// MPC::MPC(const MPC&) (mpc.cpp:1133-1181): a deep copy with its own stream and buffers
int srbm_batch_clone(const srbm_batch* src, srbm_batch** out) {
    if (!src || !out) return fail("srbm_batch_clone: bad arguments");
    HIPCHK(hipSetDevice(src->device));
    HIPCHK(hipStreamSynchronize(src->stream));
    auto* h = new srbm_batch;
    h->batch = src->batch; h->device = src->device; h->hp = src->hp; h->push_set = src->push_set; h->last_tol_step = src->last_tol_step;
    auto bail = [&]() { free_batch(h); return -1; };
    if (alloc_batch(h, nullptr)) return bail();
    const size_t B = h->batch;
    auto cp = [&](void* d, const void* s_, size_t n) { return hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, h->stream) == hipSuccess; };
    bool ok = cp(h->insts, src->insts, sizeof(SrbmInst) * B) && cp(h->works, src->works, sizeof(SrbmWork) * B) &&
              cp(h->d_state, src->d_state, sizeof(double) * 13 * B) && cp(h->d_time, src->d_time, sizeof(double) * B) &&
              cp(h->d_ee, src->d_ee, sizeof(double) * 12 * B);
    if (ok && src->d_plant) {
        ok = hipMalloc(&h->d_plant, sizeof(double) * 13 * B) == hipSuccess && hipMalloc(&h->d_push_time, sizeof(double) * B) == hipSuccess &&
             hipMalloc(&h->d_push_impulse, sizeof(double) * 6 * B) == hipSuccess && cp(h->d_plant, src->d_plant, sizeof(double) * 13 * B) &&
             cp(h->d_push_time, src->d_push_time, sizeof(double) * B) && cp(h->d_push_impulse, src->d_push_impulse, sizeof(double) * 6 * B);
    }
    if (ok && src->d_wbc) ok = hipMalloc(&h->d_wbc, sizeof(SrbmWbcParams)) == hipSuccess && cp(h->d_wbc, src->d_wbc, sizeof(SrbmWbcParams));      // (a clone carries the complete state)
    if (!ok) { fail("srbm_batch_clone: device copy failed"); return bail(); }
    if (upload_params(h)) return bail();
    if (hipStreamSynchronize(h->stream) != hipSuccess) { fail("srbm_batch_clone: synchronisation failed"); return bail(); }
    *out = h;
    return 0;
}
int srbm_get_capacity(int* cap4) { if (!cap4) return fail("bad arguments"); cap4[0] = SRBM_NMAX; cap4[1] = SRBM_NUMAX; cap4[2] = SRBM_NSMAX; cap4[3] = SRBM_KMAX; return 0; }
int srbm_batch_size(const srbm_batch* h) { return h ? h->batch : -1; }
int srbm_num_nodes(const srbm_batch* h) { return h ? h->hp.N : -1; }

int srbm_batch_destroy(srbm_batch* h) {
    if (!h) return 0;
    if (h->gait_refs > 0) return fail("srbm_batch_destroy: srbm_gait handles still borrow this batch (destroy them first)");
    free_batch(h);
    return 0;
}

int srbm_add_quadratic_tracking_cost(srbm_batch* h, const double* state_des12, const double* Q144) {
    if (!h || !state_des12 || !Q144) return fail("bad arguments");
    std::memcpy(h->hp.Q, Q144, sizeof(double) * 144);
    for (int i = 0; i < 12; i++) {
        double a = 0;
        for (int j = 0; j < 12; j++) a += Q144[i * 12 + j] * state_des12[j];
        h->hp.w[i] = -1 * a;
    }
    h->params_dirty = true;
    return 0;
}
int srbm_set_quadratic_final_cost(srbm_batch* h, const double* Phi144) {
    if (!h || !Phi144) return fail("bad arguments");
    std::memcpy(h->hp.Phi, Phi144, sizeof(double) * 144);
    h->params_dirty = true;
    return 0;
}
int srbm_set_linear_final_cost(srbm_batch* h, const double* w12) {
    if (!h || !w12) return fail("bad arguments");
    std::memcpy(h->hp.Phi_w, w12, sizeof(double) * 12);
    h->params_dirty = true;
    return 0;
}
// MPC::AddForceCost (mpc.cpp:791-802): weight on every force spline variable
int srbm_add_force_cost(srbm_batch* h, double weight) {
    if (!h) return fail("bad arguments");
    h->hp.force_cost = weight;
    h->params_dirty = true;
    return 0;
}
int srbm_set_solver_tolerances(srbm_batch* h, double ga, double gr, double tf, int max_iter) {
    if (!h) return fail("bad arguments");
    h->hp.tol_gap_abs = ga; h->hp.tol_gap_rel = gr; h->hp.tol_feas = tf; h->hp.max_iter = max_iter;
    h->params_dirty = true;
    return 0;
}
int srbm_set_solver_step_rule(srbm_batch* h, double tol_step, double start_mu) {
    if (!h || !(tol_step >= 0.0) || !(start_mu >= 0.0)) return fail("bad arguments");
    h->hp.tol_step = tol_step; h->hp.start_mu = start_mu;
    h->params_dirty = true;
    return 0;
}
int srbm_get_solver_step_rule(const srbm_batch* h, double* tol_step, double* start_mu) {
    if (!h || !tol_step || !start_mu) return fail("bad arguments");
    *tol_step = h->hp.tol_step; *start_mu = h->hp.start_mu;
    return 0;
}
int srbm_set_state_trajectory_warm_start(srbm_batch* h, const double* states) {
    if (!h || !states) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    HIPCHK(hipMemcpyAsync(h->d_state, states, sizeof(double) * 13 * (size_t)h->batch, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(srbm_k_warm_start, dim3(h->batch), dim3(64), 0, h->stream, h->dp, h->insts, h->d_state);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

static int upload_inputs(srbm_batch* h, const double* state, const double* init_time, const double* ee) {
    const size_t B = h->batch;
    HIPCHK(hipMemcpyAsync(h->d_state, state, sizeof(double) * 13 * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_ee, ee, sizeof(double) * 12 * B, hipMemcpyHostToDevice, h->stream));
    if (init_time) HIPCHK(hipMemcpyAsync(h->d_time, init_time, sizeof(double) * B, hipMemcpyHostToDevice, h->stream));
    else HIPCHK(hipMemsetAsync(h->d_time, 0, sizeof(double) * B, h->stream));
    return 0;
}

int srbm_create_initial_run(srbm_batch* h, const double* state, const double* ee) {
    if (!h || !state || !ee) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    if (upload_inputs(h, state, nullptr, ee)) return -1;
    for (int it = 0; it < 10; it++) if (launch_step(h)) return -1;     // mpc.cpp:85-88: always 10 solves
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int srbm_get_real_time_update(srbm_batch* h, const double* state, const double* init_time, const double* ee) {
    if (!h || !state || !init_time || !ee) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    if (upload_inputs(h, state, init_time, ee)) return -1;
    if (launch_step(h)) return -1;
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int srbm_get_real_time_update_dev(srbm_batch* h, const double* state_dev, const double* time_dev, const double* ee_dev) {
    if (!h || !state_dev || !time_dev || !ee_dev) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    const size_t B = h->batch;
    HIPCHK(hipMemcpyAsync(h->d_state, state_dev, sizeof(double) * 13 * B, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_ee, ee_dev, sizeof(double) * 12 * B, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_time, time_dev, sizeof(double) * B, hipMemcpyDeviceToDevice, h->stream));
    return launch_step(h);
}
static int launch_fused(srbm_batch* h, int first_index, int steps, SrbmPlantArgs pl) {
    // (the lower-start attempt rests on the linearisation point being close to the new minimiser: true for the open-loop protocol, whose state IS
    //  node 1 of the plan; under a plant -- integration error every step, pushes -- it is repeated too often to pay: closed loop 62 k it/s with, 80 k without)
    pl.tol_step = h->hp.tol_step; pl.start_mu = pl.plant ? 0.0 : h->hp.start_mu;
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    if (steps == 0) return 0;
    h->last_tol_step = pl.tol_step;
    // one launch for all steps (double time = i*info.integrator_dt, gait_opt_playground.cpp:84, is formed on the device)
    const bool tm = h->timing && h->ev_used < h->ev_start.size();
    if (tm) HIPCHK(hipEventRecord(h->ev_start[h->ev_used], h->stream));
    // a batch larger than the chip, several steps: a resident grid takes (instance, step) items from the step queues (srbm_fused.hiph)
    const bool queued = h->queued_ok && h->batch > h->n_cu && steps > 1 && steps < 65536 && h->batch <= SRBM_NQUEUES * (SRBM_QCAP - 1);
    if (queued) {
        if (!h->queues) HIPCHK(hipMalloc(&h->queues, sizeof(SrbmQueue) * SRBM_NQUEUES));
        hipLaunchKernelGGL(srbm_k_queue_init, dim3(SRBM_NQUEUES), dim3(256), 0, h->stream, h->queues, h->batch);
        if (h->hp.N <= K3_SHORT_N)
            hipLaunchKernelGGL(srbm_rti_queued, dim3(h->n_cu), dim3(K3_THREADS), h->k3_lds, h->stream, h->dp, h->insts, h->works, first_index, steps,
                               h->d_state, h->d_time, h->d_ee, pl, h->queues, h->batch);
        else
            hipLaunchKernelGGL(srbm_rti_queued_long, dim3(h->n_cu), dim3(K3_THREADS), h->k3_lds, h->stream, h->dp, h->insts, h->works, first_index, steps,
                               h->d_state, h->d_time, h->d_ee, pl, h->queues, h->batch);
    } else if (h->hp.N <= K3_SHORT_N)
        hipLaunchKernelGGL(srbm_rti_fused, dim3(h->batch), dim3(K3_THREADS), h->k3_lds, h->stream, h->dp, h->insts, h->works, first_index, steps,
                           h->d_state, h->d_time, h->d_ee, pl);
    else
        hipLaunchKernelGGL(srbm_rti_fused_long, dim3(h->batch), dim3(K3_THREADS), h->k3_lds, h->stream, h->dp, h->insts, h->works, first_index, steps,
                           h->d_state, h->d_time, h->d_ee, pl);
    if (tm) { HIPCHK(hipEventRecord(h->ev_stop[h->ev_used], h->stream)); h->ev_steps[h->ev_used] = steps; h->ev_used++; }
    HIPCHK(hipGetLastError());
    return 0;
}
int srbm_rti_advance(srbm_batch* h, int first_index, int steps) {
    if (!h || steps < 0) return fail("bad arguments");
    return launch_fused(h, first_index, steps, SrbmPlantArgs{nullptr, nullptr, nullptr, 1, 0, 0.0, 0.0});
}

// ---- closed-loop rollout harness (srbm_plant.hiph) ----
static int plant_alloc(srbm_batch* h) {
    if (h->d_plant) return 0;
    const size_t B = h->batch;
    HIPCHK(hipMalloc(&h->d_plant, sizeof(double) * 13 * B));
    HIPCHK(hipMalloc(&h->d_push_time, sizeof(double) * B));
    HIPCHK(hipMalloc(&h->d_push_impulse, sizeof(double) * 6 * B));
    return 0;
}
int srbm_plant_set_state(srbm_batch* h, const double* state) {
    if (!h || !state) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    if (plant_alloc(h)) return -1;
    HIPCHK(hipMemcpyAsync(h->d_plant, state, sizeof(double) * 13 * h->batch, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int srbm_plant_get_state(srbm_batch* h, double* state) {
    if (!h || !state) return fail("bad arguments");
    if (!h->d_plant) return fail("the plant state has not been set (srbm_plant_set_state)");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(state, h->d_plant, sizeof(double) * 13 * h->batch, hipMemcpyDeviceToHost));
    return 0;
}
int srbm_plant_set_push(srbm_batch* h, const double* time, const double* impulse) {
    if (!h || (time == nullptr) != (impulse == nullptr)) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    if (plant_alloc(h)) return -1;
    h->push_set = time != nullptr;
    if (time) {
        HIPCHK(hipMemcpyAsync(h->d_push_time, time, sizeof(double) * h->batch, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_push_impulse, impulse, sizeof(double) * 6 * h->batch, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}
int srbm_closed_loop_advance(srbm_batch* h, int first_index, int steps, int substeps, int advance_time) {
    if (!h || steps < 0 || substeps < 1) return fail("bad arguments");
    if (!h->d_plant) return fail("the plant state has not been set (srbm_plant_set_state)");
    return launch_fused(h, first_index, steps, SrbmPlantArgs{h->d_plant, h->push_set ? h->d_push_time : nullptr, h->push_set ? h->d_push_impulse : nullptr,
                                                              substeps, advance_time ? 1 : 0});
}
// the same protocol, one kernel launch per phase and step (grid-wide synchronisation between the phases); kept for
// A/B measurements against the fused kernel
int srbm_rti_advance_unfused(srbm_batch* h, int first_index, int steps) {
    if (!h || steps < 0) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    for (int i = 0; i < steps; i++) {
        const double time = (first_index + i) * h->hp.dt;
        hipLaunchKernelGGL(srbm_k_next_inputs, dim3(h->batch), dim3(64), 0, h->stream, h->dp, h->insts, time, h->d_state, h->d_time, h->d_ee);
        if (launch_step(h)) return -1;
    }
    return 0;
}
int srbm_synchronize(srbm_batch* h) {
    if (!h) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
void* srbm_stream(srbm_batch* h) { return h ? (void*)h->stream : nullptr; }

int srbm_update_contact_times(srbm_batch* h, const double* times, int max_contacts) {
    if (!h || !times || max_contacts <= 0) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    // the reference indexes the caller's vector with its own contact count (end_effector_splines.cpp:860-892): a vector that is
    // too short is an out-of-range read there, an error here
    std::vector<SrbmInst> v;
    HIPCHK(hipStreamSynchronize(h->stream));
    v.resize(h->batch);
    HIPCHK(hipMemcpy(v.data(), h->insts, sizeof(SrbmInst) * (size_t)h->batch, hipMemcpyDeviceToHost));
    for (int b = 0; b < h->batch; b++)
        for (int ee = 0; ee < SRBM_NEE; ee++) {
            int n = 0;
            for (int i = 0; i < v[b].nk[ee]; i++) n += (v[b].kind[ee][i] <= SRBM_K_TD);
            if (n > max_contacts)
                return fail("srbm_update_contact_times: instance " + std::to_string(b) + " foot " + std::to_string(ee) + " has " + std::to_string(n) +
                            " contact times, max_contacts is " + std::to_string(max_contacts));
        }
    void* d = nullptr;
    const size_t bytes = sizeof(double) * (size_t)h->batch * SRBM_NEE * max_contacts;
    if (batch_scratch(h, bytes, &d)) return -1;
    HIPCHK(hipMemcpyAsync(d, times, bytes, hipMemcpyHostToDevice, h->stream));
    if (upload_params(h)) return -1;
    const int tot = h->batch * SRBM_NEE;
    hipLaunchKernelGGL(srbm_k_set_contact_times, dim3((tot + 63) / 64), dim3(64), 0, h->stream, h->dp, h->insts, static_cast<const double*>(d), max_contacts);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// ---------------- bilevel (gait) step: mpc::GaitOptimizer for the batch ----------------
struct srbm_gait {
    srbm_batch* h = nullptr;        // the instances being optimised
    srbm_batch* ls = nullptr;       // LS_SIZE candidates per instance, same stream
    double *xk = nullptr, *step = nullptr, *costs = nullptr, *dHdth = nullptr;   // [B][SRBM_GAIT_NV], costs [B][LS_SIZE]
    int *counts = nullptr, *imin = nullptr, *valid = nullptr, *lp_status = nullptr;   // [B][4], [B], [B], [B]
    double* pred_red = nullptr;                                                   // [B]
    int* ready = nullptr;                                                         // [B] deriv_ready of the controller
    SrbmGaitWork* gw = nullptr;                                                   // sensitivity workspace, one per instance
};

static int make_candidate_batch(const srbm_batch* h, srbm_batch** out) {
    auto* c = new srbm_batch;
    c->batch = h->batch * SRBM_LS_SIZE; c->device = h->device;
    c->hp = h->hp; c->hp.batch = c->batch;
    if (alloc_batch(c, h->stream) || hipMemsetAsync(c->works, 0, sizeof(SrbmWork) * (size_t)c->batch, c->stream) != hipSuccess) {
        free_batch(c);
        return g_err.empty() ? fail("candidate batch: allocation failed") : -1;
    }
    *out = c;
    return 0;
}
static void free_gait(srbm_gait* g) {
    if (!g) return;
    (void)hipSetDevice(g->h->device);
    (void)hipStreamSynchronize(g->h->stream);
    if (g->ls) free_batch(g->ls);                    // (borrows the batch's stream: released before the batch, enforced by gait_refs)
    (void)hipFree(g->xk); (void)hipFree(g->step); (void)hipFree(g->dHdth); (void)hipFree(g->costs);
    (void)hipFree(g->counts); (void)hipFree(g->imin); (void)hipFree(g->gw); (void)hipFree(g->valid); (void)hipFree(g->lp_status);
    (void)hipFree(g->pred_red); (void)hipFree(g->ready);
    g->h->gait_refs--;
    delete g;
}
static int alloc_gait(srbm_gait* g) {
    srbm_batch* h = g->h;
    if (make_candidate_batch(h, &g->ls)) return -1;
    const size_t B = h->batch;
    HIPCHK(hipMalloc(&g->xk, sizeof(double) * SRBM_GAIT_NV * B));
    HIPCHK(hipMalloc(&g->step, sizeof(double) * SRBM_GAIT_NV * B));
    HIPCHK(hipMalloc(&g->dHdth, sizeof(double) * SRBM_GAIT_NV * B));
    HIPCHK(hipMalloc(&g->costs, sizeof(double) * SRBM_LS_SIZE * B));
    HIPCHK(hipMalloc(&g->counts, sizeof(int) * SRBM_NEE * B));
    HIPCHK(hipMalloc(&g->imin, sizeof(int) * B));
    HIPCHK(hipMalloc(&g->valid, sizeof(int) * B));
    HIPCHK(hipMalloc(&g->lp_status, sizeof(int) * B));
    HIPCHK(hipMalloc(&g->pred_red, sizeof(double) * B));
    HIPCHK(hipMalloc(&g->ready, sizeof(int) * B));
    HIPCHK(hipMalloc(&g->gw, sizeof(SrbmGaitWork) * B));
    HIPCHK(hipMemsetAsync(g->ready, 0, sizeof(int) * B, h->stream));
    HIPCHK(hipMemsetAsync(g->lp_status, 0, sizeof(int) * B, h->stream));
    HIPCHK(hipMemsetAsync(g->pred_red, 0, sizeof(double) * B, h->stream));
    HIPCHK(hipMemsetAsync(g->valid, 0, sizeof(int) * B, h->stream));
    HIPCHK(hipMemsetAsync(g->counts, 0, sizeof(int) * SRBM_NEE * B, h->stream));
    HIPCHK(hipMemsetAsync(g->gw, 0, sizeof(SrbmGaitWork) * B, h->stream));
    HIPCHK(hipMemsetAsync(g->xk, 0, sizeof(double) * SRBM_GAIT_NV * B, h->stream));
    HIPCHK(hipMemsetAsync(g->step, 0, sizeof(double) * SRBM_GAIT_NV * B, h->stream));
    HIPCHK(hipMemsetAsync(g->dHdth, 0, sizeof(double) * SRBM_GAIT_NV * B, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

int srbm_gait_create(srbm_batch* h, srbm_gait** out) {
    if (!h || !out) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    auto* g = new srbm_gait;
    g->h = h;
    h->gait_refs++;
    if (alloc_gait(g)) { free_gait(g); return -1; }
    *out = g;
    return 0;
}
int srbm_gait_destroy(srbm_gait* g) {
    free_gait(g);
    return 0;
}
// GaitOptimizer::SetContactTimes(mpc.GetTrajectory().GetContactTimes()) (gait_optimizer.cpp:395-408)
int srbm_gait_set_contact_times_from_trajectory(srbm_gait* g) {
    if (!g) return fail("bad arguments");
    srbm_batch* h = g->h;
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    hipLaunchKernelGGL(srbm_k_gait_read_contact_times, dim3((h->batch + 63) / 64), dim3(64), 0, h->stream, h->dp, h->insts, g->xk, g->counts);
    HIPCHK(hipGetLastError());
    return 0;
}
int srbm_gait_get_contact_times(srbm_gait* g, double* xk, int* counts) {
    if (!g || !xk || !counts) return fail("bad arguments");
    HIPCHK(hipSetDevice(g->h->device));
    HIPCHK(hipStreamSynchronize(g->h->stream));
    HIPCHK(hipMemcpy(xk, g->xk, sizeof(double) * SRBM_GAIT_NV * (size_t)g->h->batch, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(counts, g->counts, sizeof(int) * SRBM_NEE * (size_t)g->h->batch, hipMemcpyDeviceToHost));
    return 0;
}
int srbm_gait_set_step(srbm_gait* g, const double* step) {
    if (!g || !step) return fail("bad arguments");
    HIPCHK(hipSetDevice(g->h->device));
    HIPCHK(hipMemcpyAsync(g->step, step, sizeof(double) * SRBM_GAIT_NV * (size_t)g->h->batch, hipMemcpyHostToDevice, g->h->stream));
    HIPCHK(hipStreamSynchronize(g->h->stream));
    return 0;
}
int srbm_gait_get_step(srbm_gait* g, double* step) {
    if (!g || !step) return fail("bad arguments");
    HIPCHK(hipSetDevice(g->h->device));
    HIPCHK(hipStreamSynchronize(g->h->stream));
    HIPCHK(hipMemcpy(step, g->step, sizeof(double) * SRBM_GAIT_NV * (size_t)g->h->batch, hipMemcpyDeviceToHost));
    return 0;
}
// MPC::ComputeDerivativeTerms (mpc.cpp:1047-1069): KKT sensitivity d = [dz; dlam; dnu] of the last QP solution
int srbm_gait_compute_sensitivity(srbm_gait* g) {
    if (!g) return fail("bad arguments");
    srbm_batch* h = g->h;
    if (h->last_tol_step > 0.0)
        return fail("srbm_gait_compute_sensitivity: the last solve of this batch ran with the step rule (tol_step > 0): its multipliers are not at the "
                    "gap tolerance the KKT sensitivity needs -- call srbm_set_solver_step_rule(h, 0, start_mu) before the solve that is differentiated "
                    "(srbm_gait_rti_advance does so by itself)");
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    hipLaunchKernelGGL(srbm_k3_normal_matrix, dim3(h->batch), dim3(K3_THREADS), h->k3_lds, h->stream, h->dp, h->insts, h->works);
    hipLaunchKernelGGL(srbm_k_gait_sensitivity, dim3(h->batch), dim3(KG_THREADS), KG_DYN_LDS_BYTES, h->stream, h->dp, h->insts, h->works, g->gw);
    HIPCHK(hipGetLastError());
    return 0;
}
// mpc_controller.cpp:518-561: SetContactTimes, ComputeDerivativeTerms / GetQPPartials, the 20 parameter partials and
// GaitOptimizer::ComputeCostFcnDerivWrtContactTimes, for every instance.  dHdth stays on the device for the LP.
int srbm_gait_compute_gradient(srbm_gait* g) {
    if (!g) return fail("bad arguments");
    srbm_batch* h = g->h;
    if (srbm_gait_set_contact_times_from_trajectory(g)) return -1;
    if (srbm_gait_compute_sensitivity(g)) return -1;
    hipLaunchKernelGGL(srbm_k_gait_gradient, dim3(h->batch), dim3(KH_THREADS), 0, h->stream, h->dp, h->insts, h->works, g->gw, g->dHdth, g->valid);
    HIPCHK(hipGetLastError());
    return 0;
}
// MPCSingleRigidBody::ComputeParamPartialsClarabel (msrb.cpp:642-792) as data: the partials of the QP of the LAST solve of instance `inst` with
// respect to contact time `idx` of foot `ee`, evaluated on the instance's current trajectory, dense and in the reference's layout
// (mpc::QPPartials, mpc/include/qp/qp_partials.h:15-35): dA [n_eq][n], dG [n_ineq][n], db [n_eq], dh [n_ineq] (zero as coded).  Debug path like
// srbm_export_qp: it runs the SAME per-item code the gradient kernel contracts (gait_param_partial_item), with a dense emitter.
int srbm_gait_get_param_partials(srbm_batch* h, int inst, int ee, int idx, double* dA, double* dG, double* db, double* dh) {
    if (!h || inst < 0 || inst >= h->batch || ee < 0 || ee >= SRBM_NEE || idx < 0 || !dA || !dG || !db || !dh) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    HIPCHK(hipStreamSynchronize(h->stream));
    SrbmInst I;
    HIPCHK(hipMemcpy(&I, h->insts + inst, sizeof(SrbmInst), hipMemcpyDeviceToHost));
    const size_t n = I.n, me = I.n_eq, mi = I.n_ineq;
    if (n == 0 || me == 0) return fail("srbm_gait_get_param_partials: no QP has been solved yet");
    double* d = nullptr;
    const size_t tot = me * n + mi * n + me + 1;
    DevTemps tmp;
    HIPCHK(tmp.alloc(&d, sizeof(double) * tot));
    HIPCHK(hipMemsetAsync(d, 0, sizeof(double) * tot, h->stream));
    int* derr = reinterpret_cast<int*>(d + me * n + mi * n + me);
    hipLaunchKernelGGL(srbm_k_gait_param_partials, dim3(1), dim3(KH_THREADS), 0, h->stream, h->dp, h->insts + inst, h->works + inst, ee, idx,
                       d, d + me * n, d + me * n + mi * n, derr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    int err = 0;
    HIPCHK(hipMemcpy(dA, d, sizeof(double) * me * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(dG, d + me * n, sizeof(double) * mi * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(db, d + me * n + mi * n, sizeof(double) * me, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&err, derr, sizeof(int), hipMemcpyDeviceToHost));
    std::memset(dh, 0, sizeof(double) * mi);
    if (err & SRBM_ERR_CAPACITY) return fail("srbm_gait_get_param_partials: contact index out of range");
    if (err) return fail("srbm_gait_get_param_partials: spline lookup failed (error bits " + std::to_string(err) + ")");
    return 0;
}
int srbm_gait_get_gradient(srbm_gait* g, double* dHdth, int* valid) {
    if (!g || !dHdth) return fail("bad arguments");
    HIPCHK(hipSetDevice(g->h->device));
    HIPCHK(hipStreamSynchronize(g->h->stream));
    HIPCHK(hipMemcpy(dHdth, g->dHdth, sizeof(double) * SRBM_GAIT_NV * (size_t)g->h->batch, hipMemcpyDeviceToHost));
    if (valid) HIPCHK(hipMemcpy(valid, g->valid, sizeof(int) * (size_t)g->h->batch, hipMemcpyDeviceToHost));
    return 0;
}
int srbm_gait_get_sensitivity(srbm_gait* g, double* d, int ld) {
    if (!g || !d || ld <= 0) return fail("bad arguments");
    srbm_batch* h = g->h;
    HIPCHK(hipSetDevice(h->device));
    void* devv = nullptr;
    const size_t bytes = sizeof(double) * (size_t)h->batch * ld;
    if (batch_scratch(h, bytes, &devv)) return -1;
    double* dev = static_cast<double*>(devv);
    HIPCHK(hipMemsetAsync(dev, 0, bytes, h->stream));
    hipLaunchKernelGGL(srbm_k_gait_pack_d, dim3(h->batch), dim3(128), 0, h->stream, h->dp, h->insts, h->works, g->gw, dev, ld);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(d, dev, bytes, hipMemcpyDeviceToHost));
    return 0;
}
// GaitOptimizer::OptimizeContactTimes (gait_optimizer.cpp:185-364): the LP over the contact-time step; time[batch]
int srbm_gait_optimize_contact_times(srbm_gait* g, const double* time) {
    if (!g || !time) return fail("bad arguments");
    srbm_batch* h = g->h;
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    HIPCHK(hipMemcpyAsync(h->d_time, time, sizeof(double) * (size_t)h->batch, hipMemcpyHostToDevice, h->stream));
    const int threads = h->batch * SRBM_NEE;
    hipLaunchKernelGGL(srbm_k_gait_lp, dim3(h->batch), dim3(LP_THREADS), 0, h->stream, h->dp, h->insts, g->xk, g->counts, g->dHdth, h->d_time,
                       g->step, g->pred_red, g->lp_status);
    HIPCHK(hipGetLastError());
    return 0;
}
int srbm_gait_get_lp_result(srbm_gait* g, int* lp_status, double* pred_red) {
    if (!g) return fail("bad arguments");
    HIPCHK(hipSetDevice(g->h->device));
    HIPCHK(hipStreamSynchronize(g->h->stream));
    if (lp_status) HIPCHK(hipMemcpy(lp_status, g->lp_status, sizeof(int) * (size_t)g->h->batch, hipMemcpyDeviceToHost));
    if (pred_red) HIPCHK(hipMemcpy(pred_red, g->pred_red, sizeof(double) * (size_t)g->h->batch, hipMemcpyDeviceToHost));
    return 0;
}
// candidates -> 10*B solves -> argmin + install; inputs already in h->d_state / d_time / d_ee
static int line_search_core(srbm_gait* g, bool use_ready_mask) {
    srbm_batch* h = g->h; srbm_batch* ls = g->ls;
    const int B = h->batch;
    if (upload_params(h)) return -1;
    ls->hp = h->hp; ls->hp.batch = ls->batch; ls->params_dirty = true;      // costs / tolerances may have changed since creation
    if (upload_params(ls)) return -1;
    const int* ready = use_ready_mask ? g->ready : nullptr;
    hipLaunchKernelGGL(srbm_k_gait_spawn_candidates, dim3(B * SRBM_LS_SIZE), dim3(128), 0, h->stream, h->dp, h->insts, ls->insts,
                       g->xk, g->step, h->d_state, h->d_time, h->d_ee, ls->d_state, ls->d_time, ls->d_ee, ready);
    HIPCHK(hipGetLastError());
    if (launch_step(ls)) return -1;
    hipLaunchKernelGGL(srbm_k_gait_select, dim3(B), dim3(128), 0, h->stream, h->dp, h->insts, ls->insts, ready, g->imin, g->costs);
    HIPCHK(hipGetLastError());
    return 0;
}
// GaitOptimizer::LineSearch (gait_optimizer.cpp:671-753)
int srbm_gait_line_search(srbm_gait* g, const double* state, const double* time, const double* ee, int* imin, double* costs) {
    if (!g || !state || !time || !ee) return fail("bad arguments");
    srbm_batch* h = g->h;
    HIPCHK(hipSetDevice(h->device));
    if (upload_inputs(h, state, time, ee)) return -1;
    if (line_search_core(g, false)) return -1;
    const int B = h->batch;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (imin) HIPCHK(hipMemcpy(imin, g->imin, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost));
    if (costs) HIPCHK(hipMemcpy(costs, g->costs, sizeof(double) * SRBM_LS_SIZE * (size_t)B, hipMemcpyDeviceToHost));
    return 0;
}
// MPCController::GaitOpt (mpc_controller.cpp:518-566): gradient + LP at `time`; sets deriv_ready per instance
static int gait_opt_core(srbm_gait* g) {        // time already in h->d_time
    srbm_batch* h = g->h;
    if (srbm_gait_compute_gradient(g)) return -1;
    const int threads = h->batch * SRBM_NEE;
    hipLaunchKernelGGL(srbm_k_gait_lp, dim3(h->batch), dim3(LP_THREADS), 0, h->stream, h->dp, h->insts, g->xk, g->counts, g->dHdth, h->d_time,
                       g->step, g->pred_red, g->lp_status);
    hipLaunchKernelGGL(srbm_k_gait_ready, dim3((h->batch + 63) / 64), dim3(64), 0, h->stream, h->dp, g->valid, g->lp_status, g->ready, -1);
    HIPCHK(hipGetLastError());
    return 0;
}
// The MPC loop of the controller with the bilevel step folded in (mpc_controller.cpp:320-346), device-resident and
// open-loop as srbm_rti_advance (state := node 1 of the previous trajectory, time = run_num * dt):
//   run_num % freq == 0 and a gradient is ready  -> LineSearch only (its 10 candidates are the RTI solves of this step)
//   (run_num + 1) % freq == 0                     -> GetRealTimeUpdate, then GaitOpt (gradient + LP for the next step)
//   otherwise                                     -> GetRealTimeUpdate
int srbm_gait_rti_advance(srbm_gait* g, int first_run_num, int steps, int gait_opt_freq) {
    if (!g || steps < 0 || gait_opt_freq <= 0) return fail("bad arguments");
    srbm_batch* h = g->h;
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    for (int i = 0; i < steps; i++) {
        const int run_num = first_run_num + i;
        const double time = run_num * h->hp.dt;
        hipLaunchKernelGGL(srbm_k_next_inputs, dim3(h->batch), dim3(64), 0, h->stream, h->dp, h->insts, time, h->d_state, h->d_time, h->d_ee);
        if (run_num % gait_opt_freq == 0 && run_num > 0) {
            // instances without a ready gradient get the plain update through the same 10-candidate batch (zero step)
            if (line_search_core(g, true)) return -1;
            hipLaunchKernelGGL(srbm_k_gait_ready, dim3((h->batch + 63) / 64), dim3(64), 0, h->stream, h->dp, g->valid, g->lp_status, g->ready, 0);
        } else if ((run_num + 1) % gait_opt_freq == 0 && run_num > 0) {
            // the solve the gradient differentiates: to the reference's gap criterion (the as-coded KKT sensitivity divides by the slacks, so it
            // needs the duals Clarabel's tolerance gives); every other solve of the protocol -- plain steps, the 10 candidates of a line
            // search, which are only compared by cost -- runs with the batch's step rule
            if (launch_step(h, true)) return -1;
            if (gait_opt_core(g)) return -1;
        } else {
            if (launch_step(h)) return -1;
            hipLaunchKernelGGL(srbm_k_gait_ready, dim3((h->batch + 63) / 64), dim3(64), 0, h->stream, h->dp, g->valid, g->lp_status, g->ready, 0);
        }
        HIPCHK(hipGetLastError());
    }
    return 0;
}
/* status / stats of the candidates of the last line search: status[batch][10], iters[batch][10] (diagnostic) */
// diagnostic / test hook: the batch of line-search candidates (LS_SIZE per instance, candidate c of instance b at index b * LS_SIZE + c), owned by
// the gait handle -- for the read-back entries (status, sizes, srbm_export_qp) only
srbm_batch* srbm_gait_debug_candidates(srbm_gait* g) { return g ? g->ls : nullptr; }
int srbm_gait_get_candidate_status(srbm_gait* g, int* status, int* err) {
    if (!g || !status || !err) return fail("bad arguments");
    std::vector<SrbmInst> v;
    if (fetch_insts(g->ls, v)) return -1;
    for (int b = 0; b < g->ls->batch; b++) { status[b] = v[b].status; err[b] = v[b].err | v[b].err_acc; }
    return 0;
}

// MPC::AdjustForCurrentContacts (mpc.cpp:1195-1203): time[batch], in_contact[batch][4]
int srbm_adjust_for_current_contacts(srbm_batch* h, const double* time, const int* in_contact) {
    if (!h || !time || !in_contact) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    void* dcv = nullptr;
    const size_t B = h->batch;
    if (batch_scratch(h, sizeof(int) * 4 * B, &dcv)) return -1;
    int* dc = static_cast<int*>(dcv);
    HIPCHK(hipMemcpyAsync(dc, in_contact, sizeof(int) * 4 * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_time, time, sizeof(double) * B, hipMemcpyHostToDevice, h->stream));
    const int tot = h->batch * SRBM_NEE;
    hipLaunchKernelGGL(srbm_k_adjust_for_current_contacts, dim3((tot + 63) / 64), dim3(64), 0, h->stream, h->dp, h->insts, h->d_time, dc);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

int srbm_enable_kernel_timing(srbm_batch* h, int max_launches) {
    if (!h || max_launches < 0) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    while ((int)h->ev_start.size() < max_launches) {
        hipEvent_t a, b;
        HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
        h->ev_start.push_back(a); h->ev_stop.push_back(b); h->ev_steps.push_back(0);
    }
    h->ev_used = 0; h->timing = max_launches > 0;
    return 0;
}
int srbm_get_kernel_timing(srbm_batch* h, double* total_ms, int* launches) {
    if (!h || !total_ms || !launches) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    double tot = 0;
    for (size_t i = 0; i < h->ev_used; i++) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, h->ev_start[i], h->ev_stop[i])); tot += ms; }
    *total_ms = tot; *launches = (int)h->ev_used;
    return 0;
}
int srbm_get_kernel_timings(srbm_batch* h, double* ms_each, int max_launches, int* launches) {
    if (!h || !ms_each || !launches || max_launches < 0) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    *launches = (int)h->ev_used;
    for (size_t i = 0; i < h->ev_used && (int)i < max_launches; i++) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, h->ev_start[i], h->ev_stop[i])); ms_each[i] = ms; }
    return 0;
}
// per instance: the running total of factorisations (diagnostic: scripts/dev_dispatch_order.py)
int srbm_debug_get_instance_iters(srbm_batch* h, double* iters /* [batch] */) {
    if (!h || !iters) return fail("bad arguments");
    std::vector<SrbmInst> v;
    if (fetch_insts(h, v)) return -1;
    for (int b = 0; b < h->batch; b++) iters[b] = v[b].acc_iters;
    return 0;
}
int srbm_get_work_counters(srbm_batch* h, double* total_ipm_iterations, double* total_algorithmic_flops) {
    if (!h) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<SrbmInst> v(h->batch);
    HIPCHK(hipMemcpy(v.data(), h->insts, sizeof(SrbmInst) * (size_t)h->batch, hipMemcpyDeviceToHost));
    double it = 0, fl = 0;
    for (auto& I : v) { it += I.acc_iters; fl += I.acc_flops; }
    if (total_ipm_iterations) *total_ipm_iterations = it;
    if (total_algorithmic_flops) *total_algorithmic_flops = fl;
    return 0;
}
int srbm_pack_results_dev(srbm_batch* h, double* out_dev, int ld) {
    if (!h || !out_dev || ld < 8) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    hipLaunchKernelGGL(srbm_k_pack_results, dim3(h->batch), dim3(64), 0, h->stream, h->dp, h->insts, h->works, out_dev, ld);
    HIPCHK(hipGetLastError());
    return 0;
}

int srbm_pack_results(srbm_batch* h, double* out, int ld) {
    if (!h || !out || ld < 8) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    void* d = nullptr;
    const size_t bytes = sizeof(double) * (size_t)h->batch * ld;
    if (batch_scratch(h, bytes, &d)) return -1;
    if (srbm_pack_results_dev(h, static_cast<double*>(d), ld)) return -1;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, d, bytes, hipMemcpyDeviceToHost));
    return 0;
}

// ---------------- multi-GPU: RCCL all-gather of the result records (include/srbm_rti.h) ----------------
// RCCL is bound at run time (types from its header, no link dependency): a process that has a copy loaded -- a C++ host linked against
// librccl.so.1, a Python process whose torch brought its own librccl.so -- gets THAT copy, so that a ncclComm_t made on the caller's side and the
// calls made here belong to the same library.
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};
static RcclApi* rccl_api_ptr() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* forced = std::getenv("SRBM_RCCL_LIB");
        const char* names[] = {"librccl.so", "librccl.so.1"};
        if (forced) api.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        for (int pass = 0; pass < 2 && !api.lib && !forced; pass++)                   // pass 0: a copy already in the process, pass 1: load one
            for (const char* n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
        if (!api.lib) { const char* e = dlerror(); api.err = std::string("RCCL not found (librccl.so / librccl.so.1; SRBM_RCCL_LIB overrides): ") + (e ? e : ""); return; }
        auto sym = [&](const char* n) { void* p = dlsym(api.lib, n); if (!p && api.err.empty()) api.err = std::string("RCCL symbol missing: ") + n; return p; };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(sym("ncclCommUserRank"));
        api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &api;
}
#define rccl_api() (*rccl_api_ptr())
static int rccl_fail(const char* what, ncclResult_t r) {
    RcclApi& A = rccl_api();
    return fail(std::string(what) + ": " + (A.GetErrorString ? A.GetErrorString(r) : "RCCL error"));
}
static_assert(sizeof(ncclUniqueId) == SRBM_RCCL_UNIQUE_ID_BYTES, "include/srbm_rti.h states the size of ncclUniqueId");

int srbm_rccl_get_unique_id(void* id_bytes) {
    if (!id_bytes) return fail("bad arguments");
    RcclApi& A = rccl_api();
    if (!A.err.empty()) return fail(A.err);
    ncclUniqueId id;
    const ncclResult_t r = A.GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    std::memcpy(id_bytes, &id, sizeof(id));
    return 0;
}
int srbm_rccl_comm_init_rank(srbm_batch* h, int world, int rank, const void* id_bytes, ncclComm_t* comm_out) {
    if (!h || !id_bytes || !comm_out || world < 1 || rank < 0 || rank >= world) return fail("bad arguments");
    RcclApi& A = rccl_api();
    if (!A.err.empty()) return fail(A.err);
    HIPCHK(hipSetDevice(h->device));
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    const ncclResult_t r = A.CommInitRank(comm_out, world, id, rank);
    if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
    return 0;
}
int srbm_rccl_comm_destroy(ncclComm_t comm) {
    if (!comm) return 0;
    RcclApi& A = rccl_api();
    if (!A.err.empty()) return fail(A.err);
    const ncclResult_t r = A.CommDestroy(comm);
    if (r != ncclSuccess) return rccl_fail("ncclCommDestroy", r);
    return 0;
}
int srbm_allgather_results(srbm_batch* h, ncclComm_t comm, double* out_dev) {
    if (!h || !comm || !out_dev) return fail("bad arguments");
    RcclApi& A = rccl_api();
    if (!A.err.empty()) return fail(A.err);
    HIPCHK(hipSetDevice(h->device));
    int rank = -1, world = 0;
    ncclResult_t r = A.CommUserRank(comm, &rank);
    if (r == ncclSuccess) r = A.CommCount(comm, &world);
    if (r != ncclSuccess) return rccl_fail("ncclCommUserRank / ncclCommCount", r);
    if (rank < 0 || rank >= world) return fail("srbm_allgather_results: communicator reports an invalid rank");
    const int ld = srbm_result_record_doubles(h->hp.N);
    const size_t count = (size_t)h->batch * ld;
    double* mine = out_dev + (size_t)rank * count;                  // in place: this rank's records go straight to their slot of the gathered array
    if (srbm_pack_results_dev(h, mine, ld)) return -1;
    r = A.AllGather(mine, out_dev, count, ncclDouble, comm, h->stream);
    if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
    return 0;
}

// ---------------- getters ----------------
static int fetch_insts(srbm_batch* h, std::vector<SrbmInst>& v) {
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    v.resize(h->batch);
    HIPCHK(hipMemcpy(v.data(), h->insts, sizeof(SrbmInst) * (size_t)h->batch, hipMemcpyDeviceToHost));
    return 0;
}
int srbm_get_sizes(srbm_batch* h, int* sizes) {
    if (!h || !sizes) return fail("bad arguments");
    std::vector<SrbmInst> v;
    if (fetch_insts(h, v)) return -1;
    for (int b = 0; b < h->batch; b++) {
        int* s = sizes + 8 * b;
        s[0] = v[b].n; s[1] = v[b].m; s[2] = v[b].n_eq; s[3] = v[b].n_ineq; s[4] = v[b].nfv; s[5] = v[b].npv; s[6] = v[b].n_td; s[7] = v[b].n_samples;
    }
    return 0;
}
int srbm_get_status(srbm_batch* h, int* status, int* err) {
    if (!h || !status) return fail("bad arguments");
    std::vector<SrbmInst> v;
    if (fetch_insts(h, v)) return -1;
    for (int b = 0; b < h->batch; b++) { status[b] = v[b].status; if (err) err[b] = v[b].err; }
    return 0;
}
// QP objective at the raw QP minimiser (the "QP Cost" column of MPC::PrintStatLineToFile, mpc.cpp:989): cost[batch]
int srbm_get_qp_cost(srbm_batch* h, double* cost) {
    if (!h || !cost) return fail("bad arguments");
    std::vector<SrbmInst> v;
    if (fetch_insts(h, v)) return -1;
    for (int b = 0; b < h->batch; b++) cost[b] = v[b].qp_cost;
    return 0;
}
int srbm_get_stats(srbm_batch* h, double* stats) {
    if (!h || !stats) return fail("bad arguments");
    std::vector<SrbmInst> v;
    if (fetch_insts(h, v)) return -1;
    for (int b = 0; b < h->batch; b++) {
        double* s = stats + 8 * b;
        s[0] = v[b].alpha; s[1] = v[b].cost; s[2] = v[b].eq_violation; s[3] = v[b].step_norm; s[4] = v[b].qp_iters;
        s[5] = v[b].res_primal; s[6] = v[b].res_dual; s[7] = v[b].gap;
    }
    return 0;
}
int srbm_get_trajectory_states(srbm_batch* h, double* states) {
    if (!h || !states) return fail("bad arguments");
    std::vector<SrbmInst> v;
    if (fetch_insts(h, v)) return -1;
    const int len = (h->hp.N + 1) * 13;
    for (int b = 0; b < h->batch; b++) std::memcpy(states + (size_t)b * len, v[b].states, sizeof(double) * len);
    return 0;
}
static int fetch_work_field(srbm_batch* h, size_t offset, size_t count, double* out, int ld) {
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if ((size_t)ld < count) return fail("leading dimension too small");
    HIPCHK(hipMemcpy2D(out, sizeof(double) * ld, reinterpret_cast<const char*>(h->works) + offset, sizeof(SrbmWork), sizeof(double) * count,
                       h->batch, hipMemcpyDeviceToHost));
    return 0;
}
int srbm_get_qp_solution(srbm_batch* h, double* x, int ld) {
    if (!h || !x) return fail("bad arguments");
    return fetch_work_field(h, offsetof(SrbmWork, x), std::min<size_t>(ld, SRBM_NXMAX), x, ld);
}
int srbm_get_raw_qp_minimiser(srbm_batch* h, double* x, int ld) {
    if (!h || !x) return fail("bad arguments");
    return fetch_work_field(h, offsetof(SrbmWork, x_qp), std::min<size_t>(ld, SRBM_NXMAX), x, ld);
}
int srbm_get_dual_solution(srbm_batch* h, double* z, double* s, int ld) {
    if (!h || !z) return fail("bad arguments");
    if (fetch_work_field(h, offsetof(SrbmWork, z), std::min<size_t>(ld, SRBM_MMAX), z, ld)) return -1;
    if (s) return fetch_work_field(h, offsetof(SrbmWork, s), std::min<size_t>(ld, SRBM_MMAX), s, ld);
    return 0;
}
int srbm_get_knots(srbm_batch* h, int inst, double* times, int* kinds, int* nk, double* fvals, double* pvals, double* box) {
    if (!h || inst < 0 || inst >= h->batch) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<SrbmInst> v(1);
    HIPCHK(hipMemcpy(v.data(), h->insts + inst, sizeof(SrbmInst), hipMemcpyDeviceToHost));
    const SrbmInst& I = v[0];
    for (int ee = 0; ee < SRBM_NEE; ee++) {
        if (nk) nk[ee] = I.nk[ee];
        for (int k = 0; k < SRBM_KMAX; k++) {
            if (times) times[ee * SRBM_KMAX + k] = I.knot_t[ee][k];
            if (kinds) kinds[ee * SRBM_KMAX + k] = I.kind[ee][k];
        }
    }
    if (fvals) std::memcpy(fvals, I.fval, sizeof(I.fval));
    if (pvals) std::memcpy(pvals, I.pval, sizeof(I.pval));
    if (box) { box[0] = I.box[0]; box[1] = I.box[1]; }
    return 0;
}

// Dense expansion of the structured QP into the reference's row/column layout.
int srbm_export_qp(srbm_batch* h, int inst, double* A, double* b, double* Pm, double* q) {
    if (!h || inst < 0 || inst >= h->batch) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<SrbmInst> vi(1);
    std::vector<SrbmWork> vw(1);
    HIPCHK(hipMemcpy(vi.data(), h->insts + inst, sizeof(SrbmInst), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(vw.data(), h->works + inst, sizeof(SrbmWork), hipMemcpyDeviceToHost));
    const SrbmInst& I = vi[0];
    const SrbmWork& W = vw[0];
    const SrbmParams& P = h->hp;
    const int N = P.N, n = I.n, m = I.m, nx = (N + 1) * 12, ns = W.n_samp;
    const double dt = P.dt;
    if (A) std::memset(A, 0, sizeof(double) * (size_t)m * n);
    if (b) std::memset(b, 0, sizeof(double) * m);
    if (Pm) std::memset(Pm, 0, sizeof(double) * (size_t)n * n);
    if (q) std::memset(q, 0, sizeof(double) * n);
    auto Aat = [&](int r, int c) -> double& { return A[(size_t)r * n + c]; };
    if (A && b) {
        // dynamics rows (msrb.cpp:218-265)
        for (int i = 0; i < 12; i++) { Aat(i, i) = -1; b[i] = -W.xbar[i]; }
        for (int k = 0; k < N; k++) {
            const int r0 = 12 * (k + 1);
            for (int i = 0; i < 12; i++) { Aat(r0 + i, 12 * k + i) += 1.0; Aat(r0 + i, 12 * (k + 1) + i) += -1.0; b[r0 + i] = -W.craw[k][i]; }
            for (int i = 0; i < 3; i++) Aat(r0 + i, 12 * k + 3 + i) += dt * (1.0 / P.mass);
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
                Aat(r0 + 6 + i, 12 * k + 9 + j) += dt * P.Ir_inv[3 * i + j];
                Aat(r0 + 9 + i, 12 * k + j) += dt * W.ALp[k][3 * i + j];
                Aat(r0 + 9 + i, 12 * k + 9 + j) += dt * W.ALL[k][3 * i + j];
            }
            for (int ee = 0; ee < SRBM_NEE; ee++) {
                const SrbmNodeRec& r = W.node[k][ee];
                const double d[3] = {r.r[0] - W.xbar[12 * k], r.r[1] - W.xbar[12 * k + 1], r.r[2] - W.xbar[12 * k + 2]};
                for (int c = 0; c < 3; c++) {
                    double e[3] = {0, 0, 0}; e[c] = 1;
                    const double dxe[3] = {d[1] * e[2] - d[2] * e[1], d[2] * e[0] - d[0] * e[2], d[0] * e[1] - d[1] * e[0]};
                    if (r.fmut) for (int p = 0; p < r.fcnt; p++) {
                        const int col = nx + W.fbase[ee][c] + r.fidx + p;
                        Aat(r0 + 3 + c, col) = dt * r.flin[p];
                        for (int qd = 0; qd < 3; qd++) Aat(r0 + 9 + qd, col) = dt * (dxe[qd] * r.flin[p]);
                    }
                    if (c < 2) {
                        const double exf[3] = {e[1] * r.f[2] - e[2] * r.f[1], e[2] * r.f[0] - e[0] * r.f[2], e[0] * r.f[1] - e[1] * r.f[0]};
                        for (int p = 0; p < r.pcnt; p++) {
                            const int col = nx + W.pbase[ee][c] + r.pidx + p;
                            for (int qd = 0; qd < 3; qd++) Aat(r0 + 9 + qd, col) = dt * (exf[qd] * r.plin[p]);
                        }
                    }
                }
            }
        }
        // force box + friction pyramid (mpc.cpp:352-414, 166-208; qp_data.cpp:256)
        int row = nx;
        for (int j = 0; j < 2; j++)
            for (int s = 0; s < ns; s++) {
                const SrbmSample& sm = W.samp[s];
                for (int p = 0; p < sm.cnt; p++) Aat(row, nx + W.fbase[sm.ee][2] + sm.idx + p) = (j == 0 ? 1.0 : -1.0) * sm.phi[p];
                b[row] = j == 0 ? P.force_bound : 0.0;
                row++;
            }
        const double mu = P.mu_fric;
        const double pyr[4][3] = {{1, 0, -mu}, {-1, 0, -mu}, {0, 1, -mu}, {0, -1, -mu}};
        for (int s = 0; s < ns; s++) {
            const SrbmSample& sm = W.samp[s];
            for (int fc = 0; fc < 4; fc++) {
                for (int c = 0; c < 3; c++) {
                    if (pyr[fc][c] == 0) continue;
                    for (int p = 0; p < sm.cnt; p++) Aat(row, nx + W.fbase[sm.ee][c] + sm.idx + p) = pyr[fc][c] * sm.phi[p];
                }
                b[row] = 0;
                row++;
            }
        }
        // foot box (msrb.cpp:381-443; qp_data.cpp:240)
        for (int i = 0; i < 2; i++)
            for (int k = 4; k <= N; k++)
                for (int ee = 0; ee < SRBM_NEE; ee++)
                    for (int c = 0; c < 2; c++) {
                        const SrbmNodeRec& r = W.node[k][ee];
                        Aat(row, 12 * k + c) = (i == 0) ? -1 : 1;
                        for (int p = 0; p < r.pcnt; p++) Aat(row, nx + W.pbase[ee][c] + r.pidx + p) = (i == 0) ? r.plin[p] : -r.plin[p];
                        const double half = W.box_used[c] / 2, hip = P.hip[2 * ee + c];
                        b[row] = (i == 0) ? (half + hip) : -(-half + hip);
                        row++;
                    }
        // TD rows then start rows
        for (int e = 0; e < W.n_td + 8; e++) {
            for (int p = 0; p < W.eq_cnt[e]; p++) Aat(row, nx + W.eq_idx[e] + p) = W.eq_coef[e][p];
            b[row] = W.eq_rhs[e];
            row++;
        }
        if (row != m) return fail("srbm_export_qp: row count mismatch");
    }
    if (Pm && q) {
        for (int k = 0; k <= N; k++) {
            const double* Qk = (k == N) ? P.Phi : P.Q;
            const double* wk = (k == N) ? P.Phi_w : P.w;
            for (int i = 0; i < 12; i++) {
                for (int j = 0; j < 12; j++) Pm[(size_t)(12 * k + i) * n + 12 * k + j] += Qk[12 * i + j];
                q[12 * k + i] = wk[i];
            }
        }
        for (int j = 0; j < W.nf; j++) Pm[(size_t)(nx + j) * n + nx + j] += P.force_cost;
        for (int i = 0; i < n; i++) Pm[(size_t)i * n + i] += 1e-3;
    }
    return 0;
}

// ---------------- mpc::Trajectory as a flat record ----------------
static void inst_to_record(const SrbmParams& P, const SrbmInst& I, srbm_trajectory* t) {
    std::memset(t, 0, sizeof(*t));
    t->num_states = P.N + 1;
    t->init_time = I.init_time; t->node_dt = P.dt; t->swing_height = P.swing_height; t->foot_offset = P.foot_offset;
    for (int k = 0; k <= P.N; k++) std::memcpy(t->states[k], &I.states[k * 13], sizeof(double) * 13);
    for (int ee = 0; ee < SRBM_NEE; ee++) {
        t->nk[ee] = I.nk[ee];
        for (int k = 0; k < SRBM_KMAX; k++) { t->knot_kind[ee][k] = I.kind[ee][k]; t->knot_time[ee][k] = I.knot_t[ee][k]; }
    }
    static_assert(SRBM_TRAJ_KMAX == SRBM_KMAX, "record and device knot capacity agree");
    std::memcpy(t->force, I.fval, sizeof(I.fval));
    std::memcpy(t->pos_xy, I.pval, sizeof(I.pval));
}
static const char* check_record(const SrbmParams& P, const srbm_trajectory& t) {
    if (t.num_states != P.N + 1) return "num_states != num_nodes + 1";
    for (int ee = 0; ee < SRBM_NEE; ee++) {
        if (t.nk[ee] < 2 || t.nk[ee] > SRBM_KMAX) return "knot count out of range";
        int contacts = 0;
        for (int k = 0; k < t.nk[ee]; k++) {
            if (t.knot_kind[ee][k] < 0 || t.knot_kind[ee][k] > SRBM_K_MID) return "unknown knot kind";
            if (k > 0 && !(t.knot_time[ee][k] >= t.knot_time[ee][k - 1])) return "knot times must be non-decreasing";
            contacts += t.knot_kind[ee][k] <= SRBM_K_TD;
        }
        if (contacts < 2) return "a foot needs at least two contact knots";
        if (t.knot_kind[ee][0] > SRBM_K_TD) return "the first knot of a foot must be a contact knot";
    }
    return nullptr;
}
int srbm_sizeof_trajectory(void) { return (int)sizeof(srbm_trajectory); }
int srbm_get_trajectory(srbm_batch* h, int first, int count, srbm_trajectory* out) {
    if (!h || !out || first < 0 || count < 0 || first + count > h->batch) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<SrbmInst> v(count);
    HIPCHK(hipMemcpy(v.data(), h->insts + first, sizeof(SrbmInst) * (size_t)count, hipMemcpyDeviceToHost));
    for (int i = 0; i < count; i++) inst_to_record(h->hp, v[i], out + i);
    return 0;
}
// MPC::SetWarmStartTrajectory (mpc.cpp:110-119)
int srbm_set_warm_start_trajectory(srbm_batch* h, int first, int count, const srbm_trajectory* trajs) {
    if (!h || !trajs || first < 0 || count < 0 || first + count > h->batch) return fail("bad arguments");
    for (int i = 0; i < count; i++)
        if (const char* why = check_record(h->hp, trajs[i])) return fail("srbm_set_warm_start_trajectory: record " + std::to_string(i) + ": " + why);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<SrbmInst> v(count);
    HIPCHK(hipMemcpy(v.data(), h->insts + first, sizeof(SrbmInst) * (size_t)count, hipMemcpyDeviceToHost));
    for (int i = 0; i < count; i++) {
        SrbmInst& I = v[i];
        const srbm_trajectory& t = trajs[i];
        for (int k = 0; k <= h->hp.N; k++) std::memcpy(&I.states[k * 13], t.states[k], sizeof(double) * 13);
        for (int ee = 0; ee < SRBM_NEE; ee++) {
            I.nk[ee] = t.nk[ee];
            for (int k = 0; k < SRBM_KMAX; k++) {
                const bool in = k < t.nk[ee];
                I.kind[ee][k] = in ? (uint8_t)t.knot_kind[ee][k] : 0; I.knot_t[ee][k] = in ? t.knot_time[ee][k] : 0.0;
            }
        }
        std::memcpy(I.fval, t.force, sizeof(I.fval));
        std::memcpy(I.pval, t.pos_xy, sizeof(I.pval));
        I.init_time = t.init_time;              // init_time_ = trajectory.GetTime(0)
        I.low_skip = 0; I.low_streak = 0;       // a trajectory from elsewhere: what the solver remembers of this instance's earlier QPs (the back-off of
                                                // the lower-start attempts, srbm_k3_ipm.hiph) no longer applies
    }
    HIPCHK(hipMemcpy(h->insts + first, v.data(), sizeof(SrbmInst) * (size_t)count, hipMemcpyHostToDevice));
    return 0;
}
// Trajectory::GetForce / GetEndEffectorLocation / EndEffectorSplines::IsInContact on a record: host arithmetic, the same
// functions the kernels use (srbm_spline.hiph is host + device)
int srbm_trajectory_eval(const srbm_trajectory* t, int ee, double time, double* force3, double* pos3, int* in_contact) {
    if (!t || ee < 0 || ee >= SRBM_NEE) return fail("bad arguments");
    if (t->nk[ee] < 2 || t->nk[ee] > SRBM_KMAX) return fail("srbm_trajectory_eval: malformed record");
    uint8_t kind[SRBM_KMAX];
    for (int k = 0; k < SRBM_KMAX; k++) kind[k] = (uint8_t)t->knot_kind[ee][k];
    const FootView f{t->knot_time[ee], kind, t->nk[ee]};
    int err = 0;
    if (force3) srbm_force_value(f, &t->force[ee][0][0][0], time, force3, &err);
    if (pos3) {
        srbm_posxy_value(f, &t->pos_xy[ee][0][0], time, pos3, &err);
        pos3[2] = srbm_posz_value(f, time, t->swing_height, t->foot_offset, &err);
    }
    if (in_contact) {
        const int lo = srbm_lower(f, SEL_POSXY, time, &err), up = srbm_upper(f, SEL_POSXY, time, &err);
        *in_contact = (kind[lo] == SRBM_K_TD && kind[up] == SRBM_K_LO) ? 1 : 0;
    }
    return err;
}
// Trajectory::SplinesAsVec (trajectory.cpp:429-452) of a record: the spline variables in the order of the QP's decision vector -- per foot and force
// coordinate (value, slope / FORCE_MULT) of every stance-interior knot (EndEffectorSplines::GetSplineAsQPVec over the mutable force nodes), then per foot
// and xy coordinate the mutable position nodes.  The same rules kernel 1 builds its linearisation point with (srbm_k1_assemble.hiph), host arithmetic.
int srbm_trajectory_splines_as_vec(const srbm_trajectory* t, double* out, int capacity, int* n_total, int* n_force) {
    if (!t || !out || !n_total) return fail("bad arguments");
    int nf = 0, np_ = 0;
    uint8_t kind[SRBM_NEE][SRBM_KMAX], ismut[SRBM_NEE][SRBM_KMAX];
    for (int ee = 0; ee < SRBM_NEE; ee++) {
        if (t->nk[ee] < 2 || t->nk[ee] > SRBM_KMAX) return fail("srbm_trajectory_splines_as_vec: malformed record");
        for (int k = 0; k < SRBM_KMAX; k++) kind[ee][k] = (uint8_t)t->knot_kind[ee][k];
        const FootView f{t->knot_time[ee], kind[ee], t->nk[ee]};
        for (int k = 0; k < t->nk[ee]; k++) nf += kind[ee][k] == SRBM_K_F ? 6 : 0;
        np_ += 2 * srbm_pos_mutable(f, ismut[ee]);
    }
    if (nf + np_ > capacity) return fail("srbm_trajectory_splines_as_vec: output buffer too small");
    int fi = 0, pi = nf;
    for (int ee = 0; ee < SRBM_NEE; ee++) {
        for (int c = 0; c < 3; c++) {
            for (int k = 0; k < t->nk[ee]; k++)
                if (kind[ee][k] == SRBM_K_F) { out[fi++] = t->force[ee][c][k][0]; out[fi++] = t->force[ee][c][k][1]; }
            if (c < 2)
                for (int k = 0; k < t->nk[ee]; k++) if (ismut[ee][k]) out[pi++] = t->pos_xy[ee][c][k];
        }
    }
    *n_total = nf + np_;
    if (n_force) *n_force = nf;
    return 0;
}
// SingleRigidBodyModel::ConvertManifoldStateToTangentState / ConvertTangentStateToManifoldState (single_rigid_body_model.cpp:188-220; the
// reference state argument is unused there: quat_ref is the identity): host arithmetic, the functions the kernels use
int srbm_convert_manifold_to_tangent(const double* state13, double* tangent12) {
    if (!state13 || !tangent12) return fail("bad arguments");
    srbm_manifold_to_tangent(state13, tangent12);
    return 0;
}
int srbm_convert_tangent_to_manifold(const double* tangent12, double* state13) {
    if (!tangent12 || !state13) return fail("bad arguments");
    srbm_tangent_to_manifold(tangent12, state13);
    return 0;
}
int srbm_eval_trajectory(srbm_batch* h, const double* time, double* force, double* pos, int* in_contact) {
    if (!h || !time) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    const size_t B = h->batch, nf = sizeof(double) * 12 * B, total = sizeof(double) * B + 2 * nf + sizeof(int) * 4 * B;
    void *dv = nullptr, *hv = nullptr;
    if (batch_stage(h, total, &dv, &hv)) return -1;
    double* d_t = static_cast<double*>(dv);
    double* d_f = d_t + B;
    double* d_p = d_f + 12 * B;
    int* d_c = reinterpret_cast<int*>(d_p + 12 * B);
    char* hb = static_cast<char*>(hv);
    memcpy(hb, time, sizeof(double) * B);
    HIPCHK(hipMemcpyAsync(d_t, hb, sizeof(double) * B, hipMemcpyHostToDevice, h->stream));
    const int tot = h->batch * SRBM_NEE;
    hipLaunchKernelGGL(srbm_k_eval_trajectory, dim3((tot + 63) / 64), dim3(64), 0, h->stream, h->dp, h->insts, d_t, d_f, d_p, d_c);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(hb + sizeof(double) * B, d_f, total - sizeof(double) * B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const char* o = hb + sizeof(double) * B;
    if (force) memcpy(force, o, nf);
    if (pos) memcpy(pos, o + nf, nf);
    if (in_contact) memcpy(in_contact, o + 2 * nf, sizeof(int) * 4 * B);
    return 0;
}
int srbm_eval_trajectory_dev(srbm_batch* h, const double* time_dev, double* force_dev, double* pos_dev, int* in_contact_dev) {
    if (!h || !time_dev || !force_dev || !pos_dev || !in_contact_dev) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    const int tot = h->batch * SRBM_NEE;
    hipLaunchKernelGGL(srbm_k_eval_trajectory, dim3((tot + 63) / 64), dim3(64), 0, h->stream, h->dp, h->insts, time_dev, force_dev, pos_dev, in_contact_dev);
    HIPCHK(hipGetLastError());
    return 0;
}
int srbm_get_ee_box_center(const srbm_batch* h, double* centers) {
    if (!h || !centers) return fail("bad arguments");
    std::memcpy(centers, h->hp.hip, sizeof(double) * 8);
    return 0;
}
int srbm_get_cost(srbm_batch* h, double* cost) {
    if (!h || !cost) return fail("bad arguments");
    std::vector<SrbmInst> v;
    if (fetch_insts(h, v)) return -1;
    for (int b = 0; b < h->batch; b++) cost[b] = v[b].cost;
    return 0;
}
// merit[b] = cost + mu |dynamics defect|_1 of the trajectory after the last solve (MPC::GetMeritValue, mpc.cpp:749-753, mu = 5000) and
// the directional derivative of the merit along the last step (GetMeritGradient, :783-788): the 'Merit' / 'Merit dd' columns
int srbm_get_merit(srbm_batch* h, double* merit, double* merit_dd) {
    if (!h || !merit) return fail("bad arguments");
    std::vector<SrbmInst> v;
    if (fetch_insts(h, v)) return -1;
    for (int b = 0; b < h->batch; b++) { merit[b] = v[b].cost + h->hp.merit_mu * v[b].eq_violation; if (merit_dd) merit_dd[b] = v[b].merit_dd; }
    return 0;
}
int srbm_get_avg_cost(srbm_batch* h, double* avg) {
    if (!h || !avg) return fail("bad arguments");
    std::vector<SrbmInst> v;
    if (fetch_insts(h, v)) return -1;
    for (int b = 0; b < h->batch; b++) avg[b] = v[b].cost_sum / v[b].run_num;      // (0/0 = NaN before the first solve, as the reference's)
    return 0;
}
int srbm_get_status_accumulated(srbm_batch* h, int* acc) {
    if (!h || !acc) return fail("bad arguments");
    std::vector<SrbmInst> v;
    if (fetch_insts(h, v)) return -1;
    for (int b = 0; b < h->batch; b++) { acc[4 * b] = v[b].err_acc | v[b].err; acc[4 * b + 1] = v[b].n_solves; acc[4 * b + 2] = v[b].n_not_solved; acc[4 * b + 3] = v[b].n_maxiter; }
    return 0;
}
__global__ void srbm_k_clear_acc(const SrbmParams* __restrict__ Pp, SrbmInst* __restrict__ insts) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Pp->batch) return;
    insts[b].err_acc = 0; insts[b].n_solves = 0; insts[b].n_not_solved = 0; insts[b].n_maxiter = 0;
    insts[b].n_low_tried = 0; insts[b].n_low_failed = 0; insts[b].n_step_rule = 0;
}
int srbm_clear_status_accumulators(srbm_batch* h) {
    if (!h) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    hipLaunchKernelGGL(srbm_k_clear_acc, dim3((h->batch + 63) / 64), dim3(64), 0, h->stream, h->dp, h->insts);
    HIPCHK(hipGetLastError());
    return 0;
}
int srbm_get_solver_counters(srbm_batch* h, long long* c4) {
    if (!h || !c4) return fail("bad arguments");
    std::vector<SrbmInst> v;
    if (fetch_insts(h, v)) return -1;
    c4[0] = c4[1] = c4[2] = c4[3] = 0;
    for (auto& I : v) { c4[0] += I.n_solves; c4[1] += I.n_step_rule; c4[2] += I.n_low_tried; c4[3] += I.n_low_failed; }
    return 0;
}
int srbm_get_solve_flags(srbm_batch* h, int* flags) {
    if (!h || !flags) return fail("bad arguments");
    std::vector<SrbmInst> v;
    if (fetch_insts(h, v)) return -1;
    for (int b = 0; b < h->batch; b++) flags[b] = (v[b].last_rule ? 1 : 0) | ((v[b].last_low & 1) ? 2 : 0) | ((v[b].last_low & 2) ? 4 : 0);
    return 0;
}
int srbm_result_record_doubles(int N) { return 8 + 12 * (N + 1) + SRBM_NUMAX + 12 * (N + 1) + 6 * SRBM_NSMAX + 16 * (N - 3) + 16 + 36; }
int srbm_get_executed_mfma(srbm_batch* h, double* total) {
    if (!h || !total) return fail("bad arguments");
    std::vector<SrbmInst> v;
    if (fetch_insts(h, v)) return -1;
    double t = 0;
    for (auto& I : v) t += I.acc_mfma;
    *total = t;
    return 0;
}

// ---------------- row f3: whole-body targets ----------------
int srbm_set_leg_kinematics(srbm_batch* h, const srbm_leg_kinematics* legs) {
    if (!h || !legs) return fail("bad arguments");
    std::memcpy(h->hp.legs, legs->origin, sizeof(h->hp.legs));
    h->hp.has_legs = 1;
    h->params_dirty = true;
    return 0;
}
static int need_legs(srbm_batch* h) { return h->hp.has_legs ? 0 : fail("the leg geometry has not been set (srbm_set_leg_kinematics)"); }
int srbm_forward_kinematics(srbm_batch* h, const double* q, double* ee) {
    if (!h || !q || !ee) return fail("bad arguments");
    if (need_legs(h)) return -1;
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    const size_t B = h->batch;
    void* dv = nullptr;
    if (batch_scratch(h, sizeof(double) * (19 + 12) * B, &dv)) return -1;
    double* dq = static_cast<double*>(dv);
    double* de = dq + 19 * B;
    HIPCHK(hipMemcpyAsync(dq, q, sizeof(double) * 19 * B, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(srbm_k_forward_kinematics, dim3((h->batch + 63) / 64), dim3(64), 0, h->stream, h->dp, dq, de);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(ee, de, sizeof(double) * 12 * B, hipMemcpyDeviceToHost));
    return 0;
}
int srbm_inverse_kinematics(srbm_batch* h, const double* state, const double* ee, const double* q_guess, double* q_out, int* iters, int* status) {
    if (!h || !state || !ee || !q_guess || !q_out) return fail("bad arguments");
    if (need_legs(h)) return -1;
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    const size_t B = h->batch;
    void* dv = nullptr;
    if (batch_scratch(h, sizeof(double) * (13 + 12 + 19 + 19) * B + sizeof(int) * 5 * B, &dv)) return -1;
    double* ds = static_cast<double*>(dv);
    double* de = ds + 13 * B;
    double* dg = de + 12 * B;
    double* dq = dg + 19 * B;
    int* di = reinterpret_cast<int*>(dq + 19 * B);
    int* dst = di + 4 * B;
    HIPCHK(hipMemcpyAsync(ds, state, sizeof(double) * 13 * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(de, ee, sizeof(double) * 12 * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(dg, q_guess, sizeof(double) * 19 * B, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(srbm_k_inverse_kinematics, dim3(h->batch), dim3(SRBM_IK_THREADS), 0, h->stream, h->dp, ds, de, dg, dq, di, dst);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(q_out, dq, sizeof(double) * 19 * B, hipMemcpyDeviceToHost));
    if (iters) HIPCHK(hipMemcpy(iters, di, sizeof(int) * 4 * B, hipMemcpyDeviceToHost));
    if (status) HIPCHK(hipMemcpy(status, dst, sizeof(int) * B, hipMemcpyDeviceToHost));
    return 0;
}
int srbm_get_targets_from_traj(srbm_batch* h, const double* time, double* q_des, double* v_des, double* force_des, int* status) {
    if (!h || !time || !q_des || !v_des || !force_des) return fail("bad arguments");
    if (need_legs(h)) return -1;
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    const size_t B = h->batch, nin = sizeof(double) * (1 + 19) * B, total = sizeof(double) * (1 + 19 + 18 + 12) * B + sizeof(int) * B;
    void *dv = nullptr, *hv = nullptr;
    if (batch_stage(h, total, &dv, &hv)) return -1;
    double* dt_ = static_cast<double*>(dv);
    double* dq = dt_ + B;
    double* dvv = dq + 19 * B;
    double* df = dvv + 18 * B;
    int* dst = reinterpret_cast<int*>(df + 12 * B);
    char* hb = static_cast<char*>(hv);
    memcpy(hb, time, sizeof(double) * B);
    memcpy(hb + sizeof(double) * B, q_des, sizeof(double) * 19 * B);
    HIPCHK(hipMemcpyAsync(dv, hb, nin, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(srbm_k_targets_from_traj, dim3(h->batch), dim3(SRBM_TT_THREADS), 0, h->stream, h->dp, h->insts, dt_, dq, dvv, df, dst);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(hb + sizeof(double) * B, dq, total - sizeof(double) * B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const char* o = hb + sizeof(double) * B;
    memcpy(q_des, o, sizeof(double) * 19 * B);
    memcpy(v_des, o + sizeof(double) * 19 * B, sizeof(double) * 18 * B);
    memcpy(force_des, o + sizeof(double) * 37 * B, sizeof(double) * 12 * B);
    if (status) memcpy(status, o + sizeof(double) * 49 * B, sizeof(int) * B);
    return 0;
}

// the same on device pointers: one launch on the batch's stream, no copy, no synchronisation (a control tick of the whole batch stays in HBM)
int srbm_get_targets_from_traj_dev(srbm_batch* h, const double* time_dev, double* q_des_dev, double* v_des_dev, double* force_des_dev, int* status_dev) {
    if (!h || !time_dev || !q_des_dev || !v_des_dev || !force_des_dev || !status_dev) return fail("bad arguments");
    if (need_legs(h)) return -1;
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    hipLaunchKernelGGL(srbm_k_targets_from_traj, dim3(h->batch), dim3(SRBM_TT_THREADS), 0, h->stream, h->dp, h->insts, time_dev, q_des_dev, v_des_dev, force_des_dev, status_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

int srbm_set_wbc_model(srbm_batch* h, const srbm_wbc_model* m) {
    if (!h || !m) return fail("bad arguments");
    HIPCHK(hipSetDevice(h->device));
    SrbmWbcParams w;
    for (int b = 0; b < 13; b++) {
        w.body[b].mass = m->body_mass[b];
        for (int i = 0; i < 3; i++) w.body[b].com[i] = m->body_com[b][i];
        for (int i = 0; i < 9; i++) w.body[b].I[i] = m->body_inertia[b][i];
    }
    for (int i = 0; i < 12; i++) { w.torque_bounds[i] = m->torque_bounds[i]; w.kp_joint[i] = m->kp_joint_gains[i]; w.kd_joint[i] = m->kd_joint_gains[i]; }
    w.kv_pos = m->base_pos_gains[0]; w.kp_pos = m->base_pos_gains[1]; w.kv_ang = m->base_ang_gains[0]; w.kp_ang = m->base_ang_gains[1];
    w.leg_weight = m->leg_tracking_weight; w.torso_weight = m->torso_tracking_weight; w.force_weight = m->force_tracking_weight;
    w.friction = m->friction_coef; w.max_grf = m->max_grf;
    if (!h->d_wbc) HIPCHK(hipMalloc(&h->d_wbc, sizeof(SrbmWbcParams)));
    HIPCHK(hipMemcpy(h->d_wbc, &w, sizeof(w), hipMemcpyHostToDevice));
    return 0;
}
int srbm_qp_control(srbm_batch* h, const double* q, const double* v, const int* contact, const double* q_des, const double* v_des,
                    const double* force_des, double* control, double* qp_sol, int* status, double* qp_dump) {
    if (!h || !q || !v || !contact || !q_des || !v_des || !force_des || !control) return fail("bad arguments");
    if (need_legs(h)) return -1;
    if (!h->d_wbc) return fail("the whole-body model has not been set (srbm_set_wbc_model)");
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    const size_t B = h->batch, DUMP = WBC_MMAX * WBC_NMAX + 2 * WBC_MMAX + 2 * WBC_NMAX;
    // layout (device = pinned host): inputs [q 19 | v 18 | q_des 19 | v_des 18 | force_des 12] doubles, contact 4 ints (padded to doubles);
    // outputs [control 36 | qp_sol 30] doubles, status 1 int (padded), then the optional dump
    const size_t n_in_d = (19 + 18 + 19 + 18 + 12) * B, n_con_d = (4 * B * sizeof(int) + sizeof(double) - 1) / sizeof(double);
    const size_t n_out_d = (36 + WBC_NMAX) * B, n_st_d = (B * sizeof(int) + sizeof(double) - 1) / sizeof(double);
    const size_t nd = n_in_d + n_con_d + n_out_d + n_st_d + (qp_dump ? DUMP * B : 0);
    void *dv = nullptr, *hv = nullptr;
    if (batch_stage(h, sizeof(double) * nd, &dv, &hv)) return -1;
    double* dq = static_cast<double*>(dv);
    double* dvl = dq + 19 * B;
    double* dqd = dvl + 18 * B;
    double* dvd = dqd + 19 * B;
    double* dfd = dvd + 18 * B;
    int* dcon = reinterpret_cast<int*>(dq + n_in_d);
    double* dctl = dq + n_in_d + n_con_d;
    double* dsol = dctl + 36 * B;
    int* dst = reinterpret_cast<int*>(dctl + n_out_d);
    double* ddump = qp_dump ? dctl + n_out_d + n_st_d : nullptr;
    double* hb = static_cast<double*>(hv);
    memcpy(hb, q, sizeof(double) * 19 * B);
    memcpy(hb + 19 * B, v, sizeof(double) * 18 * B);
    memcpy(hb + 37 * B, q_des, sizeof(double) * 19 * B);
    memcpy(hb + 56 * B, v_des, sizeof(double) * 18 * B);
    memcpy(hb + 74 * B, force_des, sizeof(double) * 12 * B);
    memcpy(hb + n_in_d, contact, sizeof(int) * 4 * B);
    HIPCHK(hipMemcpyAsync(dv, hb, sizeof(double) * (n_in_d + n_con_d), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(srbm_k_qp_control, dim3(h->batch), dim3(WBC_THREADS), 0, h->stream, h->dp, h->d_wbc, dq, dvl, dcon, dqd, dvd, dfd, dctl, dsol, dst, ddump);
    HIPCHK(hipGetLastError());
    double* ho = hb + n_in_d + n_con_d;
    HIPCHK(hipMemcpyAsync(ho, dctl, sizeof(double) * (nd - n_in_d - n_con_d), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    memcpy(control, ho, sizeof(double) * 36 * B);
    if (qp_sol) memcpy(qp_sol, ho + 36 * B, sizeof(double) * WBC_NMAX * B);
    if (status) memcpy(status, ho + n_out_d, sizeof(int) * B);
    if (qp_dump) memcpy(qp_dump, ho + n_out_d + n_st_d, sizeof(double) * DUMP * B);
    return 0;
}

int srbm_qp_control_dev(srbm_batch* h, const double* q_dev, const double* v_dev, const int* contact_dev, const double* q_des_dev, const double* v_des_dev,
                        const double* force_des_dev, double* control_dev, double* qp_sol_dev, int* status_dev) {
    if (!h || !q_dev || !v_dev || !contact_dev || !q_des_dev || !v_des_dev || !force_des_dev || !control_dev || !qp_sol_dev || !status_dev) return fail("bad arguments");
    if (need_legs(h)) return -1;
    if (!h->d_wbc) return fail("the whole-body model has not been set (srbm_set_wbc_model)");
    HIPCHK(hipSetDevice(h->device));
    if (upload_params(h)) return -1;
    hipLaunchKernelGGL(srbm_k_qp_control, dim3(h->batch), dim3(WBC_THREADS), 0, h->stream, h->dp, h->d_wbc, q_dev, v_dev, contact_dev, q_des_dev, v_des_dev,
                       force_des_dev, control_dev, qp_sol_dev, status_dev, static_cast<double*>(nullptr));
    HIPCHK(hipGetLastError());
    return 0;
}

}  // extern "C"
