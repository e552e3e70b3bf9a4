// Smoke program for the C++ host side (include/srbm_rti.hpp): the reference's call sequence -- constructor, tracking cost,
// warm start, CreateInitialRun, a few GetRealTimeUpdate steps (test/gait_opt_playground.cpp:66-147), then one gait step
// (controllers/mpc_controller.cpp:518-566 + LineSearch) -- and a dump of what a caller reads back.  The pytest wrapper
// (tests/test_cpp_facade.py) generates cfg.inc from the JSON configuration, compiles this file with g++ against
// libsrbm_rti.so and compares the dump with the ctypes path.
#include <cstdio>
#include "srbm_rti.hpp"
#include "cfg.inc"   // kNumNodes, kDt, kMu, kForceBound, kSwing, kFootOffset, kBox[2], kForceCost, kMass, kIr[9], kHip[8], kQdiag[12], kInit[13], kTarget13[13], kTargetTangent[12]

int main() {
    using namespace srbm;
    MPCInfo info;
    info.num_nodes = kNumNodes; info.integrator_dt = kDt; info.friction_coef = kMu; info.force_bound = kForceBound;
    info.swing_height = kSwing; info.foot_offset = kFootOffset; info.ee_box_size = {kBox[0], kBox[1]}; info.force_cost = kForceCost;
    ModelConstants model;
    model.mass = kMass;
    for (int i = 0; i < 9; i++) model.Ir[i] = kIr[i];
    for (int i = 0; i < 8; i++) model.hip_xy[i] = kHip[i];
    const int B = 2;
    MPCSingleRigidBody mpc(info, model, B, 0);
    vector_t Q(144, 0.0), des(kTargetTangent, kTargetTangent + 12);
    for (int i = 0; i < 12; i++) Q[i * 13] = kQdiag[i];
    mpc.AddQuadraticTrackingCost(des, Q);
    mpc.SetQuadraticFinalCost(Q);
    vector_t w(12, 0.0);
    for (int i = 0; i < 12; i++) w[i] = -kQdiag[i] * des[i];
    mpc.SetLinearFinalCost(w);
    mpc.SetSolverTolerances(1e-15, 1e-15, 1e-10, 200);
    mpc.SetSolverStepRule(0.0, 0.0);      // the gradient below differentiates the last solve: gap criterion (include/srbm_rti.h)
    vector_t state(13 * B), ee(12 * B), t(B, 0.0);
    const double ee0[12] = {0.2, 0.2, 0, 0.2, -0.2, 0, -0.2, 0.2, 0, -0.2, -0.2, 0};
    for (int b = 0; b < B; b++) { for (int i = 0; i < 13; i++) state[13 * b + i] = kInit[i]; for (int i = 0; i < 12; i++) ee[12 * b + i] = ee0[i]; }
    mpc.SetStateTrajectoryWarmStart(state);
    mpc.CreateInitialRun(state, ee);
    mpc.RtiAdvance(0, 4);
    mpc.Synchronize();
    GaitOptimizer gait(mpc);
    std::vector<int> valid, counts;
    vector_t grad = gait.ComputeCostFcnDerivWrtContactTimes(&valid);
    vector_t xk = gait.GetContactTimes(&counts);
    for (int b = 0; b < B; b++) t[b] = 4 * kDt;
    vector_t step = gait.OptimizeContactTimes(t);
    const vector_t tr = mpc.GetTrajectoryStates();
    vector_t st1(13 * B);
    for (int b = 0; b < B; b++) for (int i = 0; i < 13; i++) st1[13 * b + i] = tr[(size_t)b * (kNumNodes + 1) * 13 + 13 + i];
    // foot locations at t: not part of the facade's getters; the line search only needs consistent inputs for the comparison
    std::vector<int> imin = gait.LineSearch(st1, t, ee);
    std::vector<int> err;
    const std::vector<int> q = mpc.GetSolveQuality(&err);
    const std::vector<int> sz = mpc.GetSizes();
    const vector_t x = mpc.GetQPSolution();
    std::printf("status %d %d err %d %d n %d m %d valid %d imin %d ncontacts %d\n", q[0], q[1], err[0], err[1], sz[0], sz[1], valid[0], imin[0],
                counts[0] + counts[1] + counts[2] + counts[3]);
    for (int i = 0; i < 20; i++) std::printf("grad %d %.17g\n", i, grad[i]);
    for (int i = 0; i < 20; i++) std::printf("step %d %.17g\n", i, step[i]);
    for (int i = 0; i < 40; i++) std::printf("x %d %.17g\n", i, x[i]);
    return 0;
}
