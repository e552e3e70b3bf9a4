// Behaviour driver for include/mpc_facade/controllers.h (runs on the GPU box; builder-written): the step downstream
// of the MPC as the controller performs it per 1 kHz tick -- whole-body targets from the plan (two interpolated states, two inverse-kinematics
// solves, joint velocity by differencing), desired contacts and contact forces, the whole-body QP -- with controller::QPControl,
// mpc::SingleRigidBodyModel, mpc::MPCSingleRigidBody and mpc::Trajectory held BY VALUE as controllers/include/mpc_controller.h holds them.
// The robot is replaced by "measured state = the targets with a fixed tracking error".
//
//   wbc_driver <urdf> 0            constants the facade reads from the URDF (no GPU)
//   wbc_driver <urdf> <ticks>      MPC set-up + three MPC updates, then <ticks> control ticks 1 ms apart
// Prints what the caller reads back, one value per line, for tests/test_cpp_facade.py.
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "mpc_facade/controllers.h"
#include "cfg.inc"      // as controller_driver.cpp, plus kTorqueBounds[12], kKpJoint[12], kKdJoint[12], kBasePosGains[2], kBaseAngGains[2], kLegWeight, kTorsoWeight, kForceWeight

namespace {
using Vec = mpc::vector_t;
using Mat = mpc::matrix_t;
using Feet = std::vector<mpc::vector_3t>;
using controller::Contact;

struct TickLoop {
    controller::QPControl wbc;              // by value
    mpc::MPCSingleRigidBody solver;
    mpc::Trajectory plan;
    mpc::MPCInfo info;
    mpc::SingleRigidBodyModel model;
    Vec q_target, v_target;
    Feet planned_force;
    Contact contact_used;
    int n_joints, ticks_done = 0;

    TickLoop(const std::string& urdf, const Vec& torque_bounds, const std::vector<double>& pos_gains, const std::vector<double>& ang_gains, const Vec& kp, const Vec& kd,
             const mpc::MPCInfo& info_in, const std::vector<Vec>& warm, const Vec& goal, const Mat& Q)
        : wbc(1000.0, urdf, "POINT", 18, torque_bounds, kMu, pos_gains, ang_gains, kp, kd, kLegWeight, kTorsoWeight, kForceWeight, 4, info_in.force_bound),
          solver(info_in, urdf), info(info_in), model(urdf, info_in.ee_frames, info_in.discretization_steps, info_in.integrator_dt, info_in.nom_state),
          planned_force(4), n_joints(12) {
        solver.SetStateTrajectoryWarmStart(warm);
        solver.AddQuadraticTrackingCost(goal, Q);
        solver.AddForceCost(info.force_cost);
        solver.SetQuadraticFinalCost(Q);
        solver.SetLinearFinalCost(-1.0 * (Q * goal));
    }
    void start(const Vec& q_full, const Vec& x0) {
        q_target = q_full;
        Feet feet = model.GetEndEffectorLocations(q_full);            // forward kinematics of the feet
        for (auto& p : feet) p(2) = 0;                                 // on the ground plane for the default gait
        solver.SetDefaultGaitTrajectory(mpc::Gaits::Trot, 3, feet);
        solver.CreateInitialRun(x0, feet);
        plan = solver.GetTrajectory();
    }
    void mpcUpdate(double t) {                                         // open-loop feed, as controller_driver.cpp
        Feet feet(4);
        for (int f = 0; f < 4; f++) feet[f] = plan.GetEndEffectorLocation(f, t);
        plan = solver.GetRealTimeUpdate(plan.GetState(1), t, feet, false);
    }
    // state of the plan at time t on the segment [node lo, node lo + 1]
    Vec stateOnSegment(int lo, double t) const {
        const double t_lo = plan.GetTime(lo), t_hi = plan.GetTime(lo + 1);
        return (plan.GetState(lo + 1) - plan.GetState(lo)) * (1 - (t_hi - t) / (t_hi - t_lo)) + plan.GetState(lo);
    }
    void targetsFromPlan(double t) {
        if (t < plan.GetTime(0)) t = plan.GetTime(0);
        const double dt = info.integrator_dt;
        const int node = plan.GetNode(t);
        if (node > 0 && t + dt < plan.GetTime(node)) throw std::runtime_error("bad interp.");
        const Vec x_now = stateOnSegment(node > 0 ? node - 1 : node, t), x_next = stateOnSegment(node, t + dt);
        Feet feet_now(4), feet_next(4);
        for (int f = 0; f < 4; f++) { feet_now[f] = plan.GetEndEffectorLocation(f, t); feet_next[f] = plan.GetEndEffectorLocation(f, t + dt); }
        q_target = model.InverseKinematics(x_now, feet_now, q_target, info.joint_bounds_ub, info.joint_bounds_lb);
        v_target = Vec::Zero(18);
        Vec ang_mom(3);
        for (int i = 0; i < 3; i++) { v_target(i) = x_now(3 + i) / model.GetMass(); ang_mom(i) = x_now(10 + i); }
        const Vec omega = model.GetIrInv() * ang_mom;
        for (int i = 0; i < 3; i++) v_target(3 + i) = omega(i);
        const Vec q_next = model.InverseKinematics(x_next, feet_next, q_target, info.joint_bounds_ub, info.joint_bounds_lb);
        for (int i = 0; i < n_joints; i++) v_target(6 + i) = (-q_target(7 + i) + q_next(7 + i)) / dt;
        for (int f = 0; f < 4; f++) planned_force[f] = plan.GetForce(f, t);
    }
    Vec controlAction(const Vec& q, const Vec& v, const Vec& a, const Contact& measured, double t) {
        targetsFromPlan(t);
        Contact desired = plan.GetDesiredContacts(t);
        desired.contact_frames_ = measured.contact_frames_;
        Vec stacked(desired.GetNumContacts() * 3);
        int slot = 0;
        for (int f = 0; f < (int)desired.in_contact_.size(); f++) {
            if (!desired.in_contact_[f]) continue;
            const mpc::vector_3t force = plan.GetForce(f, t);
            for (int c = 0; c < 3; c++) stacked(3 * slot + c) = force(c);
            slot++;
        }
        wbc.UpdateTargetConfig(q_target);
        wbc.UpdateTargetVel(v_target);
        wbc.UpdateForceTargets(stacked);
        wbc.UpdateDesiredContacts(desired);
        ticks_done++;
        contact_used = desired;
        return wbc.ComputeControlAction(q, v, a, desired, t);
    }
};
}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: wbc_driver <urdf> <ticks>\n"); return 2; }
    const std::string urdf = argv[1];
    const int ticks = std::atoi(argv[2]);
    if (ticks == 0) {       // what the facade's URDF reader hands to the library
        const srbm_model m = mpc::ModelConstantsFromUrdf(urdf, std::vector<double>(kInitConfig, kInitConfig + 19));
        std::printf("urdf_mass 0 %.17g\n", m.mass);
        for (int i = 0; i < 9; i++) std::printf("urdf_Ir %d %.17g\n", i, m.Ir[i]);
        for (int i = 0; i < 8; i++) std::printf("urdf_hip %d %.17g\n", i, m.hip_xy[i]);
        const srbm_leg_kinematics L = mpc::LegKinematicsFromUrdf(urdf);
        for (int i = 0; i < 48; i++) std::printf("urdf_leg %d %.17g\n", i, (&L.origin[0][0][0])[i]);
        srbm_wbc_model w{};
        mpc::WbcBodiesFromUrdf(urdf, &w);
        for (int b = 0; b < 13; b++) {
            std::printf("urdf_body_mass %d %.17g\n", b, w.body_mass[b]);
            for (int i = 0; i < 3; i++) std::printf("urdf_body_com %d %.17g\n", 3 * b + i, w.body_com[b][i]);
            for (int i = 0; i < 9; i++) std::printf("urdf_body_inertia %d %.17g\n", 9 * b + i, w.body_inertia[b][i]);
        }
        return 0;
    }
    mpc::MPCInfo info;
    info.num_nodes = kNumNodes; info.integrator_dt = kDt; info.friction_coef = kMu; info.force_bound = kForceBound;
    info.swing_height = kSwing; info.foot_offset = kFootOffset; info.ee_box_size(0) = kBox[0]; info.ee_box_size(1) = kBox[1];
    info.force_cost = kForceCost;
    info.nom_state = Vec(19);
    for (int i = 0; i < 19; i++) info.nom_state(i) = kInitConfig[i];
    Vec init(13), des_alg(12), tb(12), kp(12), kd(12);
    for (int i = 0; i < 13; i++) init(i) = kInit[i];
    for (int i = 0; i < 12; i++) { des_alg(i) = kTargetTangent[i]; tb(i) = kTorqueBounds[i]; kp(i) = kKpJoint[i]; kd(i) = kKdJoint[i]; }
    Mat Q = Mat::Zero(12, 12);
    for (int i = 0; i < 12; i++) Q(i, i) = kQdiag[i];
    std::vector<Vec> warm(kNumNodes + 1, init);
    TickLoop c(urdf, tb, {kBasePosGains[0], kBasePosGains[1]}, {kBaseAngGains[0], kBaseAngGains[1]}, kp, kd, info, warm, des_alg, Q);
    c.start(info.nom_state, init);
    for (int i = 0; i < 3; i++) c.mpcUpdate(i * info.integrator_dt);
    const double t0 = c.plan.GetTime(0);
    Contact contact(4);
    Vec ctl;
    for (int k = 0; k < ticks; k++) {
        const double time = t0 + 1e-3 * (k + 1);
        // "measured" state: the previous targets with a fixed tracking error
        Vec q = c.q_target, v = c.v_target.size() ? c.v_target : Vec::Zero(18);
        for (int i = 0; i < 12; i++) q(7 + i) += 0.01 * ((i % 3) - 1);
        for (int i = 0; i < 18; i++) v(i) = 0.9 * v(i);
        ctl = c.controlAction(q, v, Vec::Zero(18), contact, time);
        std::printf("tick_status %d %d\n", k, c.wbc.LastStatus());
    }
    for (int i = 0; i < 19; i++) std::printf("q_des %d %.17g\n", i, c.q_target(i));
    for (int i = 0; i < 18; i++) std::printf("v_des %d %.17g\n", i, c.v_target(i));
    for (int i = 0; i < (int)ctl.size(); i++) std::printf("control %d %.17g\n", i, ctl(i));
    for (int i = 0; i < 4; i++) std::printf("contact %d %d\n", i, c.contact_used.in_contact_.at(i) ? 1 : 0);
    std::printf("t0 0 %.17g\nrun_num 0 %d\n", t0, c.ticks_done);
    return 0;
}
