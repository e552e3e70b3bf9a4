// Behaviour driver for include/mpc_facade/mpc.h, part 2 (runs on the GPU box; builder-written): the methods the callers OUTSIDE the controller use,
// in the order those callers use them --
//   the fixed-position playground protocol (IK target state, gait step every second iteration, one RTI update per iteration, QP sizes compared
//   before / after, modified cost),
//   the finite-difference procedure for the contact-time partials (QPPartials read as matrices next to differences of GetQPData().sparse_constraint_),
//   the model's manifold <-> tangent maps and Ir on MPC::GetModelCopy().
// The reference's own text of these callers is compiled against the facade by tests/tools/extract_callsites.py in the build container; this file
// only has to CALL the same methods and print what comes back, one value per line, for tests/test_cpp_facade.py.
//
//   playground_driver <urdf> <iterations>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "mpc_facade/mpc.h"
#include "cfg.inc"

namespace {

using Vec = mpc::vector_t;
using Mat = mpc::matrix_t;
using Feet = std::vector<mpc::vector_3t>;

void dump(const char* key, const Vec& v) { for (int i = 0; i < (int)v.size(); i++) std::printf("%s %d %.17g\n", key, i, v(i)); }

double worstEntry(const Mat& analytic, const Mat& differenced) {
    if (analytic.rows() == 0 || analytic.cols() == 0) return 0.0;
    return (analytic - differenced).cwiseAbs().maxCoeff();
}

void dumpSchedule(const char* key, const std::vector<mpc::time_v>& sched) {
    int k = 0;
    for (const auto& foot : sched) for (const auto& knot : foot) std::printf("%s %d %.17g\n", key, k++, knot.GetTime());
}

// One gait step: sensitivities of the last solve, parameter partials of every contact time, gradient, LP, new schedule installed.
// Returns the modified cost before the step, or +inf when the last solve cannot be differentiated.
double gaitStep(mpc::MPCSingleRigidBody& solver, mpc::GaitOptimizer& gait, const mpc::Trajectory& plan, double cost_reduction, double t) {
    if (!solver.ComputeDerivativeTerms()) return INFINITY;
    gait.SetContactTimes(solver.GetTrajectory().GetContactTimes());
    gait.UpdateSizes(solver.GetNumDecisionVars(), solver.GetNumConstraints());
    const double before = solver.GetModifiedCost(45);
    solver.GetQPPartials(gait.GetQPPartials());
    for (int foot = 0; foot < 4; foot++) {
        const int knots = plan.GetNumContactNodes(foot);
        gait.SetNumContactTimes(foot, knots);
        for (int k = 0; k < knots; k++) solver.ComputeParamPartialsClarabel(plan, gait.GetParameterPartials(foot, k), foot, k);
    }
    gait.ModifyQPPartials(solver.GetQPSolution());
    gait.ComputeCostFcnDerivWrtContactTimes();
    gait.OptimizeContactTimes(t, cost_reduction);
    solver.UpdateContactTimes(gait.GetContactTimes());
    return before;
}

void playground(mpc::MPCSingleRigidBody& solver, mpc::GaitOptimizer& gait, const Vec& x0, Feet feet, const Vec& q_standing, double dt, int iterations) {
    const int kCostNodes = 45;
    solver.CreateInitialRun(x0, feet);
    mpc::Trajectory plan = solver.GetTrajectory();
    const auto schedule0 = plan.GetContactTimes();
    Vec q = q_standing;
    double last_cost = solver.GetModifiedCost(kCostNodes), reduction = 0, sum = 0;
    int gait_steps = 0;
    for (int it = 0; it < iterations; it++) {
        const double t = 0.0 * dt * it;                       // the fixed-position variant: every iteration at time 0
        q = solver.GetFullTargetState(t, q);
        for (int j = 0; j < 19; j++) std::printf("ik_state %d %.17g\n", 19 * it + j, q(j));
        for (size_t f = 0; f < feet.size(); f++) feet[f] = plan.GetEndEffectorLocation((int)f, t);
        if (it % 2 == 0) {
            last_cost = gaitStep(solver, gait, plan, reduction, t);
            gait_steps += std::isfinite(last_cost) ? 1 : 0;
        }
        const mpc::QPData sizes_before = solver.GetQPData();
        plan = solver.GetRealTimeUpdate(plan.GetState(0), t, feet, false);
        const mpc::QPData& sizes_after = solver.GetQPData();
        const bool same = sizes_before.num_decision_vars == sizes_after.num_decision_vars &&
                          sizes_before.GetTotalNumConstraints() == sizes_after.GetTotalNumConstraints();
        std::printf("qp_n %d %d\nqp_m %d %d\nqp_same_size %d %d\n", it, sizes_after.num_decision_vars, it, sizes_after.GetTotalNumConstraints(), it, (int)same);
        const double now = solver.GetModifiedCost(kCostNodes);
        reduction = last_cost - now;
        sum += now;
        std::printf("modified_cost %d %.17g\n", it, now);
    }
    std::printf("gait_steps 0 %d\ntotal_cost 0 %.17g\n", gait_steps, sum);
    dumpSchedule("sched_before", schedule0);
    dumpSchedule("sched_after", solver.GetTrajectory().GetContactTimes());
}

// Contact-time partials against one-sided differences of the assembled constraint matrix, block by block (dynamics rows of dA, force-box and
// friction-cone rows of dG), for every contact time with index >= 1.  Returns the worst entry; 1e9 when a perturbation changes a block size.
double partialsAgainstDifferences(mpc::MPCSingleRigidBody& base, mpc::MPCSingleRigidBody& moved, const Vec& x0, const Feet& feet) {
    base.CreateInitialRun(x0, feet);
    moved.CreateInitialRun(x0, feet);
    const double h = 1e-8;
    const mpc::Trajectory plan = base.GetTrajectory();
    const std::vector<mpc::time_v> schedule = plan.GetContactTimes();
    std::vector<std::pair<int, int>> params;
    for (int foot = 0; foot < 4; foot++) for (int k = 1; k < (int)schedule[foot].size(); k++) params.emplace_back(foot, k);
    // (the facade evaluates partials on the solver's CURRENT trajectory: take them before the update below replaces it)
    std::vector<mpc::QPPartials> analytic(params.size());
    for (size_t p = 0; p < params.size(); p++) base.ComputeParamPartialsClarabel(plan, analytic[p], params[p].first, params[p].second);
    base.GetRealTimeUpdate(x0, 0.0, feet, false);
    const mpc::QPData qp0 = base.GetQPData();
    const int n_dyn = qp0.num_dynamics_constraints, n_box = qp0.num_force_box_constraints_, n_cone = qp0.num_cone_constraints_;
    double worst = 0;
    for (size_t p = 0; p < params.size(); p++) {
        std::vector<mpc::time_v> shifted = schedule;
        mpc::SplineTimes& knot = shifted[params[p].first][params[p].second];
        knot.SetTime(knot.GetTime() + h);
        moved.SetWarmStartTrajectory(plan);
        moved.UpdateContactTimes(shifted);
        moved.GetRealTimeUpdate(x0, 0.0, feet, false);
        const mpc::QPData& qp1 = moved.GetQPData();
        if (qp1.num_force_box_constraints_ != n_box || qp1.num_cone_constraints_ != n_cone) return 1e9;
        const Mat slope = (qp1.sparse_constraint_ - qp0.sparse_constraint_) / h;
        Mat dA = Mat::Zero(qp0.num_equality_, qp0.num_decision_vars), dG = Mat::Zero(qp0.num_inequality_, qp0.num_decision_vars);
        dA += analytic[p].dA;
        dG += analytic[p].dG;
        worst = std::max(worst, worstEntry(dA.topRows(n_dyn), slope.topRows(n_dyn)));
        worst = std::max(worst, worstEntry(dG.topRows(n_box), slope.middleRows(n_dyn, n_box)));
        worst = std::max(worst, worstEntry(dG.middleRows(n_box, n_cone), slope.middleRows(n_dyn + n_box, n_cone)));
    }
    std::printf("partials_checked 0 %d\n", (int)params.size());
    return worst;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: playground_driver <urdf> <iterations>\n"); return 2; }
    const std::string urdf = argv[1];
    const int iterations = std::atoi(argv[2]);
    mpc::MPCInfo info;
    info.num_nodes = kNumNodes; info.integrator_dt = kDt; info.friction_coef = kMu; info.force_bound = kForceBound;
    info.swing_height = kSwing; info.foot_offset = kFootOffset; info.force_cost = kForceCost;
    info.ee_box_size(0) = kBox[0]; info.ee_box_size(1) = kBox[1];
    info.nom_state = Vec(19);
    Vec x0(13), x_goal(13), q_standing(19);
    for (int i = 0; i < 19; i++) info.nom_state(i) = q_standing(i) = kInitConfig[i];
    for (int i = 0; i < 13; i++) { x0(i) = kInit[i]; x_goal(i) = kTarget13[i]; }
    Mat Q = Mat::Zero(12, 12);
    for (int i = 0; i < 12; i++) Q(i, i) = kQdiag[i];
    const std::vector<Vec> warm(kNumNodes + 1, x0);

    // the URDF-taking constructor and a copy of its model (what hardware/hardware_interface.cpp keeps) ...
    mpc::MPCSingleRigidBody from_urdf(info, urdf);
    mpc::SingleRigidBodyModel model = from_urdf.GetModelCopy();
    // ... and, for the protocol, the same object from the constants of cfg.inc (bit-identical to what the Python binding of the test is given;
    // the URDF reader reproduces them to 1e-12) with the URDF's leg geometry
    srbm_model constants{};
    constants.mass = kMass;
    std::copy(kIr, kIr + 9, constants.Ir);
    std::copy(kHip, kHip + 8, constants.hip_xy);
    mpc::MPCSingleRigidBody solver(info, constants);
    solver.SetLegKinematics(mpc::LegKinematicsFromUrdf(urdf));
    solver.SetStateTrajectoryWarmStart(warm);

    const Vec goal_tangent = model.ConvertManifoldStateToTangentState(x_goal, warm.front());
    dump("des_alg", goal_tangent);
    dump("des_back", model.ConvertTangentStateToManifoldState(goal_tangent, warm.front()));
    solver.AddQuadraticTrackingCost(goal_tangent, Q);
    solver.AddForceCost(info.force_cost);
    solver.SetQuadraticFinalCost(Q);
    solver.SetLinearFinalCost(-1.0 * (Q * goal_tangent));
    const Eigen::Vector3d omega = {0.3, -0.2, 0.1};
    const Eigen::Vector3d L = model.GetIr() * omega;
    for (int i = 0; i < 3; i++) std::printf("Ir_w %d %.17g\n", i, L(i));

    const Feet feet0 = {{0.2, 0.2, 0}, {0.2, -0.2, 0}, {-0.2, 0.2, 0}, {-0.2, -0.2, 0}};
    {   // two copies of the configured solver (value semantics)
        mpc::MPCSingleRigidBody a = solver, b = solver;
        std::printf("partials_worst_fd 0 %.17g\n", partialsAgainstDifferences(a, b, x0, feet0));
    }
    mpc::GaitOptimizer gait(4, 10, 10, 10, 1, 0.05);
    playground(solver, gait, x0, feet0, q_standing, info.integrator_dt, iterations);
    dump("target_config", solver.GetTargetConfig(solver.GetTrajectory().GetTime(1)));
    dump("force_target", solver.GetForceTarget(solver.GetTrajectory().GetTime(1) + 0.01));
    return 0;
}
