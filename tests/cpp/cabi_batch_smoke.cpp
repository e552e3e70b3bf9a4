// Smoke program for the C-ABI (include/srbm_rti.h) driven from C++ for a BATCH of two instances, plain pointers and sizes only: the
// reference's call sequence -- constructor, tracking cost, warm start, CreateInitialRun, a few RTI steps (test/gait_opt_playground.cpp:
// 66-147), then one gait step (controllers/mpc_controller.cpp:518-566 + LineSearch) -- and a dump of what a caller reads back.  The
// pytest wrapper (tests/test_cpp_facade.py) generates cfg.inc from the JSON configuration, compiles this file with g++ against
// libsrbm_rti.so and compares the dump with the ctypes path.  (The reference-typed C++ layer is include/mpc_facade/mpc.h.)
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "srbm_rti.h"
#include "cfg.inc"   // kNumNodes, kDt, kMu, kForceBound, kSwing, kFootOffset, kBox[2], kForceCost, kMass, kIr[9], kHip[8], kQdiag[12], kInit[13], kTarget13[13], kTargetTangent[12]

static void check(int rc) {
    if (rc != 0) { std::fprintf(stderr, "srbm: %s\n", srbm_last_error()); std::exit(1); }
}

int main() {
    srbm_mpc_info info{};
    info.num_nodes = kNumNodes; info.integrator_dt = kDt; info.friction_coef = kMu; info.force_bound = kForceBound;
    info.swing_height = kSwing; info.foot_offset = kFootOffset; info.ee_box_size[0] = kBox[0]; info.ee_box_size[1] = kBox[1]; info.force_cost = kForceCost;
    srbm_model model{};
    model.mass = kMass;
    for (int i = 0; i < 9; i++) model.Ir[i] = kIr[i];
    for (int i = 0; i < 8; i++) model.hip_xy[i] = kHip[i];
    const int B = 2, NV = 32;
    srbm_batch* h = nullptr;
    check(srbm_batch_create(&h, B, &info, &model, 0));
    std::vector<double> Q(144, 0.0), des(kTargetTangent, kTargetTangent + 12), w(12, 0.0);
    for (int i = 0; i < 12; i++) { Q[i * 13] = kQdiag[i]; w[i] = -kQdiag[i] * des[i]; }
    check(srbm_add_quadratic_tracking_cost(h, des.data(), Q.data()));
    check(srbm_set_quadratic_final_cost(h, Q.data()));
    check(srbm_set_linear_final_cost(h, w.data()));
    check(srbm_set_solver_tolerances(h, 1e-15, 1e-15, 1e-10, 200));
    double ts = -1, mu0 = -1;
    check(srbm_get_solver_step_rule(h, &ts, &mu0));
    if (ts != 0.0 || mu0 != 0.0) { std::fprintf(stderr, "a new batch must run the reference's criterion\n"); return 1; }
    std::vector<double> state(13 * B), ee(12 * B), t(B, 0.0);
    const double ee0[12] = {0.2, 0.2, 0, 0.2, -0.2, 0, -0.2, 0.2, 0, -0.2, -0.2, 0};
    for (int b = 0; b < B; b++) { for (int i = 0; i < 13; i++) state[13 * b + i] = kInit[i]; for (int i = 0; i < 12; i++) ee[12 * b + i] = ee0[i]; }
    check(srbm_set_state_trajectory_warm_start(h, state.data()));
    check(srbm_create_initial_run(h, state.data(), ee.data()));
    check(srbm_rti_advance(h, 0, 4));
    check(srbm_synchronize(h));
    srbm_gait* g = nullptr;
    check(srbm_gait_create(h, &g));
    std::vector<double> grad(NV * B), xk(NV * B), step(NV * B);
    std::vector<int> valid(B), counts(4 * B), imin(B);
    check(srbm_gait_compute_gradient(g));
    check(srbm_gait_get_gradient(g, grad.data(), valid.data()));
    check(srbm_gait_get_contact_times(g, xk.data(), counts.data()));
    for (int b = 0; b < B; b++) t[b] = 4 * kDt;
    check(srbm_gait_optimize_contact_times(g, t.data()));
    check(srbm_gait_get_step(g, step.data()));
    std::vector<double> tr((size_t)B * (kNumNodes + 1) * 13), st1(13 * B);
    check(srbm_get_trajectory_states(h, tr.data()));
    for (int b = 0; b < B; b++) for (int i = 0; i < 13; i++) st1[13 * b + i] = tr[(size_t)b * (kNumNodes + 1) * 13 + 13 + i];
    // (foot locations at t: the line search only needs consistent inputs for the comparison)
    check(srbm_gait_line_search(g, st1.data(), t.data(), ee.data(), imin.data(), nullptr));
    std::vector<int> q(B), err(B), sz(8 * B);
    check(srbm_get_status(h, q.data(), err.data()));
    check(srbm_get_sizes(h, sz.data()));
    int cap[4];
    check(srbm_get_capacity(cap));
    const int ld = (kNumNodes + 1) * 12 + cap[1];
    std::vector<double> x((size_t)B * ld);
    check(srbm_get_qp_solution(h, x.data(), ld));
    std::printf("status %d %d err %d %d n %d m %d valid %d imin %d ncontacts %d\n", q[0], q[1], err[0], err[1], sz[0], sz[1], valid[0], imin[0],
                counts[0] + counts[1] + counts[2] + counts[3]);
    for (int i = 0; i < 20; i++) std::printf("grad %d %.17g\n", i, grad[i]);
    for (int i = 0; i < 20; i++) std::printf("step %d %.17g\n", i, step[i]);
    for (int i = 0; i < 40; i++) std::printf("x %d %.17g\n", i, x[i]);
    srbm_gait_destroy(g);
    check(srbm_batch_destroy(h));
    return 0;
}
