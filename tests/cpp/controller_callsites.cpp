// Signature-conformance + behaviour test of include/mpc_facade/mpc.h: the statements with which the reference's caller uses the
// hot path -- controllers/mpc_controller.cpp:57-67 (cost set-up in the constructor), :84-108 (InitSolver), :286-399 (the body of
// the MPC loop) and :518-566 (GaitOpt) -- written against `mpc::MPCSingleRigidBody mpc_; mpc::GaitOptimizer gait_opt_;
// mpc::Trajectory traj_;` held BY VALUE as controllers/include/mpc_controller.h:82-83 holds them.  What is not the MPC's business
// there (threads and mutexes, pinocchio forward kinematics, MuJoCo visualisation) is replaced by the open-loop feed of
// test/gait_opt_playground.cpp:113-126 (state := node 1 of the trajectory, foot locations := the trajectory at `time`).
//
//   controller_callsites <urdf | -> <ticks> [gait_opt_freq]      ("-": model constants from cfg.inc instead of a URDF)
// Prints what the caller reads back, one value per line, for tests/test_cpp_facade.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <tuple>

#include "mpc_facade/mpc.h"
#include "cfg.inc"   // kNumNodes, kDt, kMu, kForceBound, kSwing, kFootOffset, kBox[2], kForceCost, kMass, kIr[9], kHip[8], kQdiag[12], kInit[13], kTarget13[13], kTargetTangent[12], kInitConfig[19]

using vector_t = mpc::vector_t;
using matrix_t = mpc::matrix_t;

class MPCControllerLike {
public:
    // controllers/mpc_controller.cpp:16-82
    MPCControllerLike(const mpc::MPCInfo& info, const std::string& robot_urdf, const srbm_model* consts, const std::vector<vector_t>& warm_start_states,
                      const vector_t& des_alg, const matrix_t& Q, int gait_opt_freq, const std::string& log_file)
        : mpc_(consts ? mpc::MPCSingleRigidBody(info, *consts) : mpc::MPCSingleRigidBody(info, robot_urdf)),
          gait_opt_(4, 10, 10, 10, 1, 0.05), info_(info) {
        gait_opt_freq_ = gait_opt_freq;
        mpc_.SetStateTrajectoryWarmStart(warm_start_states);
        mpc_.AddQuadraticTrackingCost(des_alg, Q);
        mpc_.AddForceCost(info_.force_cost);
        mpc_.SetQuadraticFinalCost(1*Q);
        mpc_.SetLinearFinalCost(-1*Q*des_alg);
        log_file_.open(log_file);
    }
    // :84-118
    void InitSolver(const vector_t& mpc_state, const std::vector<mpc::vector_3t>& ee_locations) {
        state_ = mpc_state;
        ee_locations_ = ee_locations;
        mpc_.SetDefaultGaitTrajectory(mpc::Gaits::Trot, 3, ee_locations_);
        mpc_.CreateInitialRun(mpc_state, ee_locations_);
        mpc_.PrintStats();
        traj_ = mpc_.GetTrajectory();
        time_ = 0;
    }
    // one pass of the body of MPCUpdate, :313-394
    void MPCUpdateOnce(double time_act) {
        vector_t state = state_;
        double time = time_act;
        std::vector<mpc::vector_3t> ee_locations = ee_locations_;
        controller::Contact contact = contact_;

        mpc_.AdjustForCurrentContacts(time, contact);

        if (!(run_num % gait_opt_freq_) && run_num > 0 && deriv_ready) {
            std::vector<mpc::time_v> contact_times_new;
            double cost_min;
            std::tie(contact_times_new, cost_min) = gait_opt_.LineSearch(mpc_, time, ee_locations, state);
            prev_cost = mpc_.GetCost();
            deriv_ready = false;
            last_ls_cost_ = cost_min;
            n_line_searches_++;
        } else if (!((run_num + 1) % gait_opt_freq_) && run_num > 0) {
            mpc_.GetRealTimeUpdate(state, time, ee_locations, false);
            deriv_ready = GaitOpt(cost_red, time, ee_locations);
        } else {
            mpc_.GetRealTimeUpdate(state, time, ee_locations, false);
            deriv_ready = false;
        }

        for (int ee = 0; ee < 4; ee++)
            for (int i = 0; i < 2; i++)
                if (std::abs(mpc_.GetTrajectory().GetEndEffectorLocation(ee, time)(i) - ee_locations.at(ee)(i)) >= 1e-4) no_match_++;

        mpc::Trajectory traj = mpc_.GetTrajectory();
        cost_red = prev_cost - mpc_.GetCost();
        traj_ = traj;
        run_num++;
        mpc_.PrintStatLineToFile(log_file_);
        avg_cost_ = mpc_.GetAvgCost();
    }
    // :518-566
    bool GaitOpt(double cost_red, double time, const std::vector<Eigen::Vector3d>& ee_locations) {
        const mpc::Trajectory prev_traj = mpc_.GetTrajectory();
        if (mpc_.ComputeDerivativeTerms()) {
            gait_opt_.SetContactTimes(mpc_.GetTrajectory().GetContactTimes());
            gait_opt_.UpdateSizes(mpc_.GetNumDecisionVars(), mpc_.GetNumConstraints());
            double original_cost = mpc_.GetCost();
            (void)original_cost;
            mpc_.GetQPPartials(gait_opt_.GetQPPartials());
            for (int ee = 0; ee < 4; ee++) {
                gait_opt_.SetNumContactTimes(ee, prev_traj.GetNumContactNodes(ee));
                for (int idx = 0; idx < prev_traj.GetNumContactNodes(ee); idx++) {
                    mpc_.ComputeParamPartialsClarabel(prev_traj, gait_opt_.GetParameterPartials(ee, idx), ee, idx);
                }
            }
            gait_opt_.ModifyQPPartials(mpc_.GetQPSolution());
            gait_opt_.ComputeCostFcnDerivWrtContactTimes();
            gait_opt_.OptimizeContactTimes(time, cost_red);
            return true;
        } else {
            std::cerr << "Can't perform gait optimization because MPC was not solved to tolerance." << std::endl;
            return false;
        }
    }
    // the open-loop feed of test/gait_opt_playground.cpp:113-126 in place of the robot: what ComputeControlAction publishes (:142-156)
    void FeedFromTrajectory(double time) {
        state_ = traj_.GetState(1);
        for (int ee = 0; ee < 4; ee++) ee_locations_.at(ee) = traj_.GetEndEffectorLocation(ee, time);
        contact_ = traj_.GetDesiredContacts(time);        // (the feet are where the plan says: AdjustForCurrentContacts has nothing to adjust)
    }

    mpc::MPCSingleRigidBody mpc_;          // BY VALUE, controllers/include/mpc_controller.h:82
    mpc::GaitOptimizer gait_opt_;          // :83
    mpc::Trajectory traj_;
    mpc::MPCInfo info_;
    std::ofstream log_file_;
    vector_t state_;
    std::vector<mpc::vector_3t> ee_locations_;
    controller::Contact contact_;
    double time_ = 0, prev_cost = 1e10, cost_red = 0, avg_cost_ = 0, last_ls_cost_ = 0;
    int run_num = 0, gait_opt_freq_ = 5, no_match_ = 0, n_line_searches_ = 0;
    bool deriv_ready = false;
};

int main(int argc, char** argv) {
    const std::string urdf = argc > 1 ? argv[1] : "-";
    const int ticks = argc > 2 ? std::atoi(argv[2]) : 8;
    const int freq = argc > 3 ? std::atoi(argv[3]) : 5;
    mpc::MPCInfo info;
    info.num_nodes = kNumNodes; info.integrator_dt = kDt; info.friction_coef = kMu; info.force_bound = kForceBound;
    info.swing_height = kSwing; info.foot_offset = kFootOffset; info.ee_box_size(0) = kBox[0]; info.ee_box_size(1) = kBox[1];
    info.force_cost = kForceCost;
    info.nom_state = vector_t(19);
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
    vector_t init(13), des_alg(12);
    for (int i = 0; i < 13; i++) init(i) = kInit[i];
    for (int i = 0; i < 12; i++) des_alg(i) = kTargetTangent[i];
    matrix_t Q = matrix_t::Zero(12, 12);
    for (int i = 0; i < 12; i++) Q(i, i) = kQdiag[i];
    std::vector<vector_t> warm(kNumNodes + 1, init);
    MPCControllerLike c(info, urdf, urdf == "-" ? &consts : nullptr, warm, des_alg, Q, freq, "/tmp/mpc_facade_log.txt");
    std::vector<mpc::vector_3t> ee0 = {{0.2, 0.2, 0}, {0.2, -0.2, 0}, {-0.2, 0.2, 0}, {-0.2, -0.2, 0}};   // test/simulation_mpc.cpp:104-108
    c.InitSolver(init, ee0);
    c.contact_ = c.traj_.GetDesiredContacts(0.0);
    for (int i = 0; i < ticks; i++) {
        const double time = i * info.integrator_dt;
        if (i > 0) c.FeedFromTrajectory(time);
        c.MPCUpdateOnce(time);
    }
    // value semantics: a copy made now continues exactly like the original
    MPCControllerLike* c2 = nullptr;
    mpc::MPCSingleRigidBody copy = c.mpc_;
    const double tn = ticks * info.integrator_dt;
    c.FeedFromTrajectory(tn);
    copy.GetRealTimeUpdate(c.state_, tn, c.ee_locations_, false);
    c.mpc_.GetRealTimeUpdate(c.state_, tn, c.ee_locations_, false);
    (void)c2;
    const vector_t xa = c.mpc_.GetQPSolution(), xb = copy.GetQPSolution();
    int same = xa.size() == xb.size();
    for (int i = 0; same && i < (int)xa.size(); i++) same = xa(i) == xb(i);
    std::printf("copy_equal 0 %d\n", same);
    std::printf("quality 0 %d\n", (int)c.mpc_.GetSolveQuality());
    std::printf("n 0 %d\nm 0 %d\n", c.mpc_.GetNumDecisionVars(), c.mpc_.GetNumConstraints());
    std::printf("run_num 0 %d\nline_searches 0 %d\nno_match 0 %d\n", c.run_num, c.n_line_searches_, c.no_match_);
    std::printf("cost 0 %.17g\navg_cost 0 %.17g\nls_cost 0 %.17g\n", c.mpc_.GetCost(), c.avg_cost_, c.last_ls_cost_);
    std::printf("mass 0 %.17g\nmanifold 0 %d\n", c.mpc_.GetModel()->GetMass(), c.mpc_.GetModel()->GetNumManifoldStates());
    for (int i = 0; i < 40; i++) std::printf("x %d %.17g\n", i, xa(i));
    const mpc::Trajectory t = c.mpc_.GetTrajectory();
    for (int ee = 0; ee < 4; ee++) {
        const mpc::vector_3t f = t.GetForce(ee, tn + 0.013), p = t.GetEndEffectorLocation(ee, tn + 0.013);
        for (int k = 0; k < 3; k++) std::printf("force %d %.17g\n", 3 * ee + k, f(k));
        for (int k = 0; k < 3; k++) std::printf("pos %d %.17g\n", 3 * ee + k, p(k));
    }
    int k = 0;
    for (const mpc::time_v& tv : t.GetContactTimes()) for (const mpc::SplineTimes& s : tv) std::printf("contact_time %d %.17g\n", k++, s.GetTime());
    const std::vector<Eigen::Vector2d> bc = c.mpc_.GetEEBoxCenter();
    for (int ee = 0; ee < 4; ee++) std::printf("box_center %d %.17g\nbox_center %d %.17g\n", 2 * ee, bc[ee](0), 2 * ee + 1, bc[ee](1));
    // statistics and trajectory dumps (mpc.cpp:818-899 PrintStats over the whole history, trajectory.cpp:146-223 PrintTrajectoryToFile)
    {
        std::ofstream stats("/tmp/mpc_facade_stats.txt");
        c.mpc_.PrintStats(stats);
    }
    t.PrintTrajectoryToFile("/tmp/mpc_facade_traj.txt");
    const vector_t sv = t.SplinesAsVec();
    std::printf("recorded 0 %d\n", c.mpc_.GetNumRecordedSolves());
    for (int i = 0; i < (int)sv.size(); i++) std::printf("spline_vec %d %.17g\n", i, sv(i));
    const auto viz = c.mpc_.CreateVizData();
    std::printf("viz 0 %d\nviz 1 %d\n", (int)viz.size(), (int)viz.at(4).size());
    return 0;
}
