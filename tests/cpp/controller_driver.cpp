// Behaviour driver for include/mpc_facade/mpc.h, part 1 (runs on the GPU box; builder-written): the controller's use of the hot path --
// cost set-up, default gait + initial run, then per MPC tick: contact adjustment, one of {line search, RTI update + gait gradient, plain RTI
// update}, foot-location check, statistics line -- with the MPC, the gait optimiser and the trajectory held BY VALUE as
// controllers/include/mpc_controller.h:82-83 holds them.  The reference's own text of MPCController::MPCUpdate / GaitOpt is compiled against the
// facade by tests/tools/extract_callsites.py in the build container; this file only has to CALL the same methods in the same order.  The robot is
// replaced by an open-loop feed (state := node 1 of the plan, feet := the plan at `time`).
//
//   controller_driver <urdf | -> <ticks> [gait_opt_freq]      ("-": model constants from cfg.inc instead of a URDF)
// Prints what the caller reads back, one value per line, for tests/test_cpp_facade.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "mpc_facade/mpc.h"
#include "cfg.inc"   // kNumNodes, kDt, kMu, kForceBound, kSwing, kFootOffset, kBox[2], kForceCost, kMass, kIr[9], kHip[8], kQdiag[12], kInit[13], kTarget13[13], kTargetTangent[12], kInitConfig[19]

namespace {
using Vec = mpc::vector_t;
using Mat = mpc::matrix_t;
using Feet = std::vector<mpc::vector_3t>;

struct ControllerLoop {
    mpc::MPCSingleRigidBody solver;        // by value
    mpc::GaitOptimizer gait;               // by value
    mpc::Trajectory plan;
    std::ofstream log;
    Vec x;
    Feet feet;
    controller::Contact contacts;
    int period, tick_count = 0, mismatches = 0, line_searches = 0;
    bool gradient_ready = false;
    double cost_before = 1e10, cost_drop = 0, running_avg = 0, line_search_cost = 0;

    ControllerLoop(const mpc::MPCInfo& info, const std::string& urdf, const srbm_model* constants, const std::vector<Vec>& warm, const Vec& goal, const Mat& Q,
                   int gait_period, const std::string& log_path)
        : solver(constants ? mpc::MPCSingleRigidBody(info, *constants) : mpc::MPCSingleRigidBody(info, urdf)), gait(4, 10, 10, 10, 1, 0.05), period(gait_period) {
        solver.SetStateTrajectoryWarmStart(warm);
        solver.AddQuadraticTrackingCost(goal, Q);
        solver.AddForceCost(info.force_cost);
        solver.SetQuadraticFinalCost(Q);
        solver.SetLinearFinalCost(-1.0 * (Q * goal));
        log.open(log_path);
    }
    void start(const Vec& x0, const Feet& feet0) {
        x = x0; feet = feet0;
        solver.SetDefaultGaitTrajectory(mpc::Gaits::Trot, 3, feet);
        solver.CreateInitialRun(x0, feet);
        solver.PrintStats();
        plan = solver.GetTrajectory();
    }
    // sensitivities of the solve just made -> gradient -> LP; true when the solve could be differentiated
    bool gaitGradient(double t) {
        const mpc::Trajectory solved = solver.GetTrajectory();
        if (!solver.ComputeDerivativeTerms()) return false;
        gait.SetContactTimes(solver.GetTrajectory().GetContactTimes());
        gait.UpdateSizes(solver.GetNumDecisionVars(), solver.GetNumConstraints());
        solver.GetQPPartials(gait.GetQPPartials());
        for (int foot = 0; foot < 4; foot++) {
            const int knots = solved.GetNumContactNodes(foot);
            gait.SetNumContactTimes(foot, knots);
            for (int k = 0; k < knots; k++) solver.ComputeParamPartialsClarabel(solved, gait.GetParameterPartials(foot, k), foot, k);
        }
        gait.ModifyQPPartials(solver.GetQPSolution());
        gait.ComputeCostFcnDerivWrtContactTimes();
        gait.OptimizeContactTimes(t, cost_drop);
        return true;
    }
    void tick(double t) {
        solver.AdjustForCurrentContacts(t, contacts);
        const bool on_period = tick_count > 0 && tick_count % period == 0, before_period = tick_count > 0 && (tick_count + 1) % period == 0;
        if (on_period && gradient_ready) {
            const auto best = gait.LineSearch(solver, t, feet, x);          // (new contact times, their cost)
            line_search_cost = best.second;
            cost_before = solver.GetCost();
            gradient_ready = false;
            line_searches++;
        } else {
            solver.GetRealTimeUpdate(x, t, feet, false);
            gradient_ready = before_period ? gaitGradient(t) : false;
        }
        const mpc::Trajectory now = solver.GetTrajectory();
        for (int foot = 0; foot < 4; foot++) {
            const mpc::vector_3t planned = now.GetEndEffectorLocation(foot, t);
            for (int c = 0; c < 2; c++) mismatches += std::abs(planned(c) - feet[foot](c)) >= 1e-4 ? 1 : 0;
        }
        cost_drop = cost_before - solver.GetCost();
        plan = now;
        tick_count++;
        solver.PrintStatLineToFile(log);
        running_avg = solver.GetAvgCost();
    }
    void feedFromPlan(double t) {
        x = plan.GetState(1);
        for (int foot = 0; foot < 4; foot++) feet[foot] = plan.GetEndEffectorLocation(foot, t);
        contacts = plan.GetDesiredContacts(t);       // the feet are where the plan says: AdjustForCurrentContacts has nothing to adjust
    }
};
}  // namespace

int main(int argc, char** argv) {
    const std::string urdf = argc > 1 ? argv[1] : "-";
    const int ticks = argc > 2 ? std::atoi(argv[2]) : 8;
    const int freq = argc > 3 ? std::atoi(argv[3]) : 5;
    mpc::MPCInfo info;
    info.num_nodes = kNumNodes; info.integrator_dt = kDt; info.friction_coef = kMu; info.force_bound = kForceBound;
    info.swing_height = kSwing; info.foot_offset = kFootOffset; info.force_cost = kForceCost;
    info.ee_box_size(0) = kBox[0]; info.ee_box_size(1) = kBox[1];
    info.nom_state = Vec(19);
    for (int i = 0; i < 19; i++) info.nom_state(i) = kInitConfig[i];
    srbm_model consts{};
    consts.mass = kMass;
    for (int i = 0; i < 9; i++) consts.Ir[i] = kIr[i];
    for (int i = 0; i < 8; i++) consts.hip_xy[i] = kHip[i];
    if (urdf != "-") {                         // constants through the facade's own URDF reader, printed for the test to compare
        const srbm_model m = mpc::ModelConstantsFromUrdf(urdf, std::vector<double>(kInitConfig, kInitConfig + 19));
        std::printf("urdf_mass 0 %.17g\n", m.mass);
        for (int i = 0; i < 9; i++) std::printf("urdf_Ir %d %.17g\n", i, m.Ir[i]);
        for (int i = 0; i < 8; i++) std::printf("urdf_hip %d %.17g\n", i, m.hip_xy[i]);
        if (ticks == 0) return 0;
    }
    Vec init(13), goal(12);
    for (int i = 0; i < 13; i++) init(i) = kInit[i];
    for (int i = 0; i < 12; i++) goal(i) = kTargetTangent[i];
    Mat Q = Mat::Zero(12, 12);
    for (int i = 0; i < 12; i++) Q(i, i) = kQdiag[i];
    const std::vector<Vec> warm(kNumNodes + 1, init);
    ControllerLoop c(info, urdf, urdf == "-" ? &consts : nullptr, warm, goal, Q, freq, "/tmp/mpc_facade_log.txt");
    const Feet ee0 = {{0.2, 0.2, 0}, {0.2, -0.2, 0}, {-0.2, 0.2, 0}, {-0.2, -0.2, 0}};
    c.start(init, ee0);
    c.contacts = c.plan.GetDesiredContacts(0.0);
    for (int i = 0; i < ticks; i++) {
        const double t = i * info.integrator_dt;
        if (i > 0) c.feedFromPlan(t);
        c.tick(t);
    }
    // value semantics: a copy made now continues exactly like the original
    mpc::MPCSingleRigidBody copy = c.solver;
    const double tn = ticks * info.integrator_dt;
    c.feedFromPlan(tn);
    copy.GetRealTimeUpdate(c.x, tn, c.feet, false);
    c.solver.GetRealTimeUpdate(c.x, tn, c.feet, false);
    const Vec xa = c.solver.GetQPSolution(), xb = copy.GetQPSolution();
    int same = xa.size() == xb.size();
    for (int i = 0; same && i < (int)xa.size(); i++) same = xa(i) == xb(i);
    std::printf("copy_equal 0 %d\n", same);
    std::printf("quality 0 %d\n", (int)c.solver.GetSolveQuality());
    std::printf("n 0 %d\nm 0 %d\n", c.solver.GetNumDecisionVars(), c.solver.GetNumConstraints());
    std::printf("run_num 0 %d\nline_searches 0 %d\nno_match 0 %d\n", c.tick_count, c.line_searches, c.mismatches);
    std::printf("cost 0 %.17g\navg_cost 0 %.17g\nls_cost 0 %.17g\n", c.solver.GetCost(), c.running_avg, c.line_search_cost);
    std::printf("mass 0 %.17g\nmanifold 0 %d\n", c.solver.GetModel()->GetMass(), c.solver.GetModel()->GetNumManifoldStates());
    for (int i = 0; i < 40; i++) std::printf("x %d %.17g\n", i, xa(i));
    const mpc::Trajectory t = c.solver.GetTrajectory();
    for (int ee = 0; ee < 4; ee++) {
        const mpc::vector_3t f = t.GetForce(ee, tn + 0.013), p = t.GetEndEffectorLocation(ee, tn + 0.013);
        for (int k = 0; k < 3; k++) std::printf("force %d %.17g\n", 3 * ee + k, f(k));
        for (int k = 0; k < 3; k++) std::printf("pos %d %.17g\n", 3 * ee + k, p(k));
    }
    int k = 0;
    for (const mpc::time_v& tv : t.GetContactTimes()) for (const mpc::SplineTimes& s : tv) std::printf("contact_time %d %.17g\n", k++, s.GetTime());
    const std::vector<Eigen::Vector2d> bc = c.solver.GetEEBoxCenter();
    for (int ee = 0; ee < 4; ee++) std::printf("box_center %d %.17g\nbox_center %d %.17g\n", 2 * ee, bc[ee](0), 2 * ee + 1, bc[ee](1));
    // statistics and trajectory dumps (mpc.cpp:818-899 PrintStats over the whole history, trajectory.cpp:146-223 PrintTrajectoryToFile)
    {
        std::ofstream stats("/tmp/mpc_facade_stats.txt");
        c.solver.PrintStats(stats);
    }
    t.PrintTrajectoryToFile("/tmp/mpc_facade_traj.txt");
    const Vec sv = t.SplinesAsVec();
    std::printf("recorded 0 %d\n", c.solver.GetNumRecordedSolves());
    for (int i = 0; i < (int)sv.size(); i++) std::printf("spline_vec %d %.17g\n", i, sv(i));
    const auto viz = c.solver.CreateVizData();
    std::printf("viz 0 %d\nviz 1 %d\n", (int)viz.size(), (int)viz.at(4).size());
    return 0;
}
