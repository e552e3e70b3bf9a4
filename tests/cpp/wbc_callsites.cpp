// Signature-conformance + behaviour test of include/mpc_facade/controllers.h: the statements with which the
// reference's controller uses the step downstream of the MPC -- controllers/mpc_controller.cpp:160-226 (ComputeControlAction: targets from
// the trajectory, desired contacts and forces, the whole-body QP) and :414-511 (GetTargetsFromTraj: interpolated states, two inverse-
// kinematics solves, velocity by differencing) -- written against `controller::QPControl qp_controller_; mpc::SingleRigidBodyModel model_;
// mpc::MPCSingleRigidBody mpc_; mpc::Trajectory traj_;` held BY VALUE as controllers/include/mpc_controller.h holds them.  The robot is
// replaced by "measured state = the targets with a fixed tracking error".
//
//   wbc_callsites <urdf> 0            constants the facade reads from the URDF (no GPU)
//   wbc_callsites <urdf> <ticks>      MPC set-up + three MPC updates, then <ticks> control ticks 1 ms apart
// Prints what the caller reads back, one value per line, for tests/test_cpp_facade.py.
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "mpc_facade/controllers.h"
#include "cfg.inc"      // as controller_callsites.cpp, plus kTorqueBounds[12], kKpJoint[12], kKdJoint[12], kBasePosGains[2], kBaseAngGains[2], kLegWeight, kTorsoWeight, kForceWeight

using vector_t = mpc::vector_t;
using matrix_t = mpc::matrix_t;
using controller::Contact;

class MPCControllerLike {
public:
    // controllers/mpc_controller.cpp:16-52
    MPCControllerLike(double control_rate, const std::string& robot_urdf, const std::string& foot_type, int nv, const Eigen::VectorXd& torque_bounds, double friction_coef,
                      const std::vector<double>& base_pos_gains, const std::vector<double>& base_ang_gains, const vector_t& kp_joint_gains, const vector_t& kd_joint_gains,
                      double leg_weight, double torso_weight, double force_weight, mpc::MPCInfo info, const std::vector<vector_t>& warm_start_states,
                      const vector_t& state_des, const matrix_t& Q)
        : qp_controller_(control_rate, robot_urdf, foot_type, nv, torque_bounds, friction_coef, base_pos_gains, base_ang_gains, kp_joint_gains, kd_joint_gains,
                         leg_weight, torso_weight, force_weight, 4, info.force_bound),
          mpc_(info, robot_urdf), info_(info),
          model_(robot_urdf, info.ee_frames, info.discretization_steps, info.integrator_dt, info.nom_state) {
        num_inputs_ = nv - 6;
        force_des_.resize(4);
        mpc_.SetStateTrajectoryWarmStart(warm_start_states);
        mpc_.AddQuadraticTrackingCost(state_des, Q);
        mpc_.AddForceCost(info_.force_cost);
        mpc_.SetQuadraticFinalCost(1*Q);
        mpc_.SetLinearFinalCost(-1*Q*state_des);
    }
    // :84-118
    void InitSolver(const vector_t& full_body_state, const vector_t& mpc_state) {
        q_des_ = full_body_state;
        std::vector<mpc::vector_3t> ee_locations = model_.GetEndEffectorLocations(full_body_state);     // (:95-101: forward kinematics of the feet)
        for (int ee = 0; ee < 4; ee++) ee_locations.at(ee)(2) = 0;                                        // feet on the ground plane for the default gait
        mpc_.SetDefaultGaitTrajectory(mpc::Gaits::Trot, 3, ee_locations);
        mpc_.CreateInitialRun(mpc_state, ee_locations);
        traj_ = mpc_.GetTrajectory();
    }
    void MPCUpdate(double time) {                                                                         // open-loop feed, as controller_callsites.cpp
        const vector_t state = traj_.GetState(1);
        std::vector<mpc::vector_3t> ee(4);
        for (int e = 0; e < 4; e++) ee.at(e) = traj_.GetEndEffectorLocation(e, time);
        traj_ = mpc_.GetRealTimeUpdate(state, time, ee, false);
    }
    // :120-226
    vector_t ComputeControlAction(const vector_t& q, const vector_t& v, const vector_t& a, const Contact& contact, double time) {
        GetTargetsFromTraj(traj_, time);
        Contact contact1 = traj_.GetDesiredContacts(time);
        contact1.contact_frames_ = contact.contact_frames_;

        vector_t force_des(contact1.GetNumContacts()*3);
        int j = 0;
        for (int i = 0; i < (int)contact1.in_contact_.size(); i++) {
            if (contact1.in_contact_.at(i)) {
                const mpc::vector_3t f = traj_.GetForce(i, time);
                for (int c = 0; c < 3; c++) force_des(3*j + c) = f(c);
                j++;
            }
        }

        qp_controller_.UpdateTargetConfig(q_des_);
        qp_controller_.UpdateTargetVel(v_des_);
        qp_controller_.UpdateForceTargets(force_des);
        qp_controller_.UpdateDesiredContacts(contact1);
        run_num++;
        last_contact_ = contact1;
        return qp_controller_.ComputeControlAction(q, v, a, contact1, time);
    }
    // :414-511
    void GetTargetsFromTraj(const mpc::Trajectory& traj, double time) {
        if (time < traj.GetTime(0)) {
            time = traj.GetTime(0);
        }
        const int nodes_ahead = 0;
        int node = traj.GetNode(time)+nodes_ahead;

        vector_t state_interp;
        vector_t state_interp2;
        if (node > 0) {
            state_interp = (traj.GetState(node) - traj.GetState(node - 1))
                                    * (1 - (traj.GetTime(node) - time) / (traj.GetTime(node) - traj.GetTime(node - 1)))
                                    + traj.GetState(node - 1);
            if (time + info_.integrator_dt < traj.GetTime(node)) {
                throw std::runtime_error("bad interp.");
            }
            state_interp2 = (traj.GetState(node + 1) - traj.GetState(node))
                            * (1 - (traj.GetTime(node + 1) - (time + info_.integrator_dt)) / (traj.GetTime(node + 1) - traj.GetTime(node)))
                            + traj.GetState(node);
        } else {
            state_interp = (traj.GetState(node + 1) - traj.GetState(node))
                                    * (1 - (traj.GetTime(node + 1) - time) / (traj.GetTime(node + 1) - traj.GetTime(node)))
                                    + traj.GetState(node);
            state_interp2 = (traj.GetState(node + 1) - traj.GetState(node))
                           * (1 - (traj.GetTime(node + 1) - (time + info_.integrator_dt)) / (traj.GetTime(node + 1) - traj.GetTime(node)))
                           + traj.GetState(node);
        }

        std::vector<mpc::vector_3t> traj_ee_locations(4);
        for (int ee = 0; ee < 4; ee++) {
            traj_ee_locations.at(ee) = traj.GetEndEffectorLocation(ee, time);
        }
        q_des_ = model_.InverseKinematics(state_interp,traj_ee_locations, q_des_,
                                          info_.joint_bounds_ub,
                                          info_.joint_bounds_lb);

        v_des_ = vector_t::Zero(18);
        const matrix_t IrInv = model_.GetIrInv();
        vector_t angmom(3);
        for (int i = 0; i < 3; i++) { v_des_(i) = state_interp(3 + i)/model_.GetMass(); angmom(i) = state_interp(10 + i); }
        const vector_t omega = IrInv*angmom;
        for (int i = 0; i < 3; i++) v_des_(3 + i) = omega(i);

        std::vector<mpc::vector_3t> traj_ee_locations_next(4);
        for (int ee = 0; ee < 4; ee++) {
            traj_ee_locations_next.at(ee) = traj.GetEndEffectorLocation(ee, time + info_.integrator_dt);
        }
        vector_t q_next = model_.InverseKinematics(state_interp2,traj_ee_locations_next, q_des_,
                                                   info_.joint_bounds_ub,
                                                   info_.joint_bounds_lb);
        // (node != 0: (-q_des_ + q_prev) / dt, node == 0: (q_next - q_des_) / dt -- the same numbers)
        for (int i = 0; i < num_inputs_; i++) v_des_(6 + i) = (-q_des_(7 + i) + q_next(7 + i)) / (info_.integrator_dt);

        for (int ee = 0; ee < 4; ee++) {
            force_des_.at(ee) = traj.GetForce(ee, time + nodes_ahead*info_.integrator_dt);
        }
    }

    controller::QPControl qp_controller_;   // BY VALUE, controllers/include/mpc_controller.h
    mpc::MPCSingleRigidBody mpc_;
    mpc::Trajectory traj_;
    mpc::MPCInfo info_;
    mpc::SingleRigidBodyModel model_;
    vector_t q_des_, v_des_;
    std::vector<mpc::vector_3t> force_des_;
    Contact last_contact_;
    int num_inputs_ = 12, run_num = 0;
};

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: wbc_callsites <urdf> <ticks>\n"); return 2; }
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
    info.nom_state = vector_t(19);
    for (int i = 0; i < 19; i++) info.nom_state(i) = kInitConfig[i];
    vector_t init(13), des_alg(12), tb(12), kp(12), kd(12);
    for (int i = 0; i < 13; i++) init(i) = kInit[i];
    for (int i = 0; i < 12; i++) { des_alg(i) = kTargetTangent[i]; tb(i) = kTorqueBounds[i]; kp(i) = kKpJoint[i]; kd(i) = kKdJoint[i]; }
    matrix_t Q = matrix_t::Zero(12, 12);
    for (int i = 0; i < 12; i++) Q(i, i) = kQdiag[i];
    std::vector<vector_t> warm(kNumNodes + 1, init);
    MPCControllerLike c(1000.0, urdf, "POINT", 18, tb, kMu, {kBasePosGains[0], kBasePosGains[1]}, {kBaseAngGains[0], kBaseAngGains[1]}, kp, kd,
                        kLegWeight, kTorsoWeight, kForceWeight, info, warm, des_alg, Q);
    c.InitSolver(info.nom_state, init);
    for (int i = 0; i < 3; i++) c.MPCUpdate(i * info.integrator_dt);
    const double t0 = c.traj_.GetTime(0);
    Contact contact(4);
    vector_t ctl;
    for (int k = 0; k < ticks; k++) {
        const double time = t0 + 1e-3 * (k + 1);
        // "measured" state: the previous targets with a fixed tracking error
        vector_t q = c.q_des_, v = c.v_des_.size() ? c.v_des_ : vector_t::Zero(18);
        for (int i = 0; i < 12; i++) q(7 + i) += 0.01 * ((i % 3) - 1);
        for (int i = 0; i < 18; i++) v(i) = 0.9 * v(i);
        ctl = c.ComputeControlAction(q, v, vector_t::Zero(18), contact, time);
        std::printf("tick_status %d %d\n", k, c.qp_controller_.LastStatus());
    }
    for (int i = 0; i < 19; i++) std::printf("q_des %d %.17g\n", i, c.q_des_(i));
    for (int i = 0; i < 18; i++) std::printf("v_des %d %.17g\n", i, c.v_des_(i));
    for (int i = 0; i < (int)ctl.size(); i++) std::printf("control %d %.17g\n", i, ctl(i));
    for (int i = 0; i < 4; i++) std::printf("contact %d %d\n", i, c.last_contact_.in_contact_.at(i) ? 1 : 0);
    std::printf("t0 0 %.17g\nrun_num 0 %d\n", t0, c.run_num);
    return 0;
}
