// The reference's classes downstream of the MPC over the C-ABI of include/srbm_rti.h, with the
// reference's names and signatures, for a caller that holds them by value as controllers/include/mpc_controller.h does:
//   mpc::SingleRigidBodyModel    lives in mpc_facade/mpc.h (MPCSingleRigidBody::GetModelCopy returns one by value)
//   controller::QPControl        /root/reference/controllers/include/qp_control.h:52-88 with the target setters of its base class
//                                (controllers/include/controller.h:52-54): the 1 kHz whole-body QP
// Each object owns a batch of ONE instance of the library (the kinematics / whole-body entries do not touch the MPC state of a batch);
// the geometry and the body data come from the URDF the constructors are given (mpc_facade/urdf_constants.h), as the reference takes
// them from pinocchio.  Errors the reference throws (IK did not converge) are thrown; a QP that cannot be solved returns the zero
// control action after the reference's message.
#pragma once
#include <iostream>

#include "mpc.h"


namespace controller {

class QPControl {
    using vector_t = Eigen::VectorXd;
public:
    // qp_control.cpp:20-72.  base_pos_gains / base_ang_gains = {kv, kp}
    QPControl(double control_rate, std::string robot_urdf, const std::string& /*foot_type*/, int nv, const Eigen::VectorXd& torque_bounds, double friction_coef,
              const std::vector<double>& base_pos_gains, const std::vector<double>& base_ang_gains, const vector_t& kp_joint_gains, const vector_t& kd_joint_gains,
              double leg_weight, double torso_weight, double force_weight, int num_contacts, double max_grf)
        : rate_(control_rate), num_inputs_(nv - 6), robot_urdf_(std::move(robot_urdf)), des_contact_(num_contacts) {
        if (nv != 18 || num_contacts != 4 || torque_bounds.size() != 12 || kp_joint_gains.size() != 12 || kd_joint_gains.size() != 12 ||
            base_pos_gains.size() != 2 || base_ang_gains.size() != 2)
            throw std::runtime_error("QPControl: this library is built for a floating base with four 3-joint legs (nv = 18, 4 contacts).");
        mpc::WbcBodiesFromUrdf(robot_urdf_, &model_);
        for (int i = 0; i < 12; i++) { model_.torque_bounds[i] = torque_bounds(i); model_.kp_joint_gains[i] = kp_joint_gains(i); model_.kd_joint_gains[i] = kd_joint_gains(i); }
        for (int i = 0; i < 2; i++) { model_.base_pos_gains[i] = base_pos_gains[i]; model_.base_ang_gains[i] = base_ang_gains[i]; }
        model_.leg_tracking_weight = leg_weight; model_.torso_tracking_weight = torso_weight; model_.force_tracking_weight = force_weight;
        model_.friction_coef = friction_coef; model_.max_grf = max_grf;
        legs_ = mpc::LegKinematicsFromUrdf(robot_urdf_);
        consts_ = mpc::ModelConstantsFromUrdf(robot_urdf_, std::vector<double>{0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0});
        Create();
        config_target_ = vector_t::Zero(19); config_target_(6) = 1.0;
        vel_target_ = vector_t::Zero(18); acc_target_ = vector_t::Zero(18); force_target_ = vector_t::Zero(12);
    }
    QPControl(const QPControl& o) : rate_(o.rate_), num_inputs_(o.num_inputs_), robot_urdf_(o.robot_urdf_), model_(o.model_), legs_(o.legs_), consts_(o.consts_),
                                    des_contact_(o.des_contact_), config_target_(o.config_target_), vel_target_(o.vel_target_), acc_target_(o.acc_target_),
                                    force_target_(o.force_target_) { Create(); }
    QPControl& operator=(const QPControl&) = delete;
    ~QPControl() { if (h_) srbm_batch_destroy(h_); }

    // controller.h:38-54
    double GetRate() const { return rate_; }
    int GetNumInputs() const { return num_inputs_; }
    void UpdateTargetConfig(const Eigen::VectorXd& q) { config_target_ = q; }
    void UpdateTargetVel(const Eigen::VectorXd& v) { vel_target_ = v; }
    void UpdateTargetAcc(const Eigen::VectorXd& a) { acc_target_ = a; }
    // qp_control.h:73-88
    void SetBasePosGains(double kv, double kp) { model_.base_pos_gains[0] = kv; model_.base_pos_gains[1] = kp; Upload(); }
    void SetBaseAngleGains(double kv, double kp) { model_.base_ang_gains[0] = kv; model_.base_ang_gains[1] = kp; Upload(); }
    void SetJointGains(const vector_t& kv, const vector_t& kp) {
        for (int i = 0; i < 12; i++) { model_.kd_joint_gains[i] = kv(i); model_.kp_joint_gains[i] = kp(i); }
        Upload();
    }
    void UpdateDesiredContacts(const Contact& contact) { des_contact_ = contact; }
    void UpdateForceTargets(const Eigen::VectorXd& force) { force_target_ = force; }

    // qp_control.cpp:74-135: [joint position targets (12), joint velocity targets (12), torques (12)]; zero when the QP could not be solved
    Eigen::VectorXd ComputeControlAction(const Eigen::VectorXd& q, const Eigen::VectorXd& v, const Eigen::VectorXd& /*a*/, const Contact& contact, double /*time*/) {
        if (q.size() != 19 || v.size() != 18 || contact.in_contact_.size() != 4) throw std::runtime_error("ComputeControlAction: wrong sizes.");
        des_contact_ = contact;                                   // (:85)
        double qq[19], vv[18], qd[19], vd[18], fd[12] = {0}, ctl[36], sol[30]; int con[4], status = 0;
        for (int i = 0; i < 19; i++) { qq[i] = q(i); qd[i] = config_target_(i); }
        for (int i = 0; i < 18; i++) { vv[i] = v(i); vd[i] = vel_target_(i); }
        int nc = 0;
        for (int i = 0; i < 4; i++) { con[i] = contact.in_contact_[i] ? 1 : 0; nc += con[i]; }
        if ((int)force_target_.size() < 3 * nc) throw std::runtime_error("ComputeControlAction: the force targets are shorter than 3 per foot in contact.");
        for (int i = 0; i < 3 * nc; i++) fd[i] = force_target_(i);
        if (srbm_qp_control(h_, qq, vv, con, qd, vd, fd, ctl, sol, &status, nullptr)) throw std::runtime_error(srbm_last_error());
        last_status_ = status & 255; last_iterations_ = status >> 8;
        if (last_status_ > 1) {
            std::cerr << "Could not solve WBC QP. Returning 0 control action." << std::endl;
            return vector_t::Zero(num_inputs_);               // (:88-92: the reference returns num_inputs_ zeros here)
        }
        vector_t out(3 * num_inputs_);
        for (int i = 0; i < 36; i++) out(i) = ctl[i];
        qp_sol_.assign(sol, sol + 18 + 3 * nc);
        return out;
    }
    // what the library adds for inspection: accelerations (18) then the contact forces of the last solve, its status and iteration count
    const std::vector<double>& LastQPSolution() const { return qp_sol_; }
    int LastStatus() const { return last_status_; }
    int LastIterations() const { return last_iterations_; }

private:
    void Create() {
        srbm_mpc_info ci{};
        ci.num_nodes = 10; ci.integrator_dt = 0.05; ci.friction_coef = model_.friction_coef; ci.force_bound = model_.max_grf; ci.swing_height = 0.1; ci.foot_offset = 0;
        ci.ee_box_size[0] = ci.ee_box_size[1] = 0.1; ci.force_cost = 0;
        if (srbm_batch_create(&h_, 1, &ci, &consts_, 0)) throw std::runtime_error(srbm_last_error());
        if (srbm_set_leg_kinematics(h_, &legs_) || srbm_set_wbc_model(h_, &model_)) { const std::string m = srbm_last_error(); srbm_batch_destroy(h_); h_ = nullptr; throw std::runtime_error(m); }
    }
    void Upload() { if (srbm_set_wbc_model(h_, &model_)) throw std::runtime_error(srbm_last_error()); }
    double rate_;
    int num_inputs_;
    std::string robot_urdf_;
    srbm_wbc_model model_{};
    srbm_leg_kinematics legs_{};
    srbm_model consts_{};
    Contact des_contact_;
    vector_t config_target_, vel_target_, acc_target_, force_target_;
    std::vector<double> qp_sol_;
    int last_status_ = 0, last_iterations_ = 0;
    srbm_batch* h_ = nullptr;
};

}  // namespace controller
