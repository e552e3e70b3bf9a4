// namespace mpc { MPCInfo, Trajectory, MPC / MPCSingleRigidBody, GaitOptimizer } with the reference's class names, method
// names and signatures, over the C-ABI of include/srbm_rti.h (libsrbm_rti.so, HIP / gfx950) -- the classes that
// /root/reference/controllers/mpc_controller.cpp holds BY VALUE (controllers/include/mpc_controller.h:82-83) and calls at
// :57-108 (set-up), :322-394 (MPC loop) and :523-560 (gait optimisation).  Header only, no arithmetic of the hot path here:
// every solve, the sensitivity, the gradient, the LP and the line search run on the device; what is computed on the host is
// what the reference computes on the host too -- spline evaluation of a Trajectory value (srbm_trajectory_eval) and the
// statistics table.
//
//   reference header                                   | here
//   mpc/include/mpc.h:39-62          MPCInfo           | same fields
//   mpc/include/mpc.h:70-170         MPC               | folded into MPCSingleRigidBody (the reference's only live subclass)
//   mpc/include/mpc_single_rigid_body.h:11-76          | MPCSingleRigidBody: one instance = a batch of 1 on the device
//   mpc/include/trajectory.h:20-175  Trajectory        | value type over the flat record srbm_trajectory
//   mpc/include/gait_optimizer.h:23-93 GaitOptimizer   | same call protocol; the QP partials never leave the device, so
//                                                      | QPPartials / QPPartialsDense are handles, not matrices
//   mpc/include/qp/qp_interface.h:12-22 SolveQuality   | same nine values
// Eigen: <Eigen/Core> as in the reference (mpc/include/mpc.h:8).  This container has no Eigen: the TESTS put a stand-in for the handful of types
// the interface is typed with on their include path (tests/cpp/eigen_standin/Eigen/Core); nothing of it ships with the product headers.
#pragma once
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include <Eigen/Core>

#include "../srbm_rti.h"
#include "urdf_constants.h"

namespace controller {
// controllers/include/controller.h:15-24
struct Contact {
    std::vector<bool> in_contact_;
    std::vector<int> contact_frames_;
    Contact() = default;
    explicit Contact(int num_contacts) : in_contact_(num_contacts, false), contact_frames_(num_contacts, 0) {}
    int GetNumContacts() const { int n = 0; for (const bool c : in_contact_) n += c ? 1 : 0; return n; }      // controllers/controller.cpp:13-22: the feet IN contact
};
}  // namespace controller

namespace mpc {
using vector_t = Eigen::VectorXd;
using matrix_t = Eigen::MatrixXd;
using vector_3t = Eigen::Vector3d;
using vector_2t = Eigen::Vector2d;
using matrix_33t = Eigen::Matrix3d;

enum MPCVerbosityLevel { Nothing = 0, Timing = 1, Optimization = 2, All = 3 };
enum Gaits { Trot = 0, Amble = 1, Static_Walk = 2 };
enum SolveQuality { Solved = 0, SolvedInacc = 1, MaxIter = 2, PrimalInfeasible = 3, DualInfeasible = 4, PrimalInfeasibleInacc = 5,
                    DualInfeasibleInacc = 6, Unsolved = 7, Other = 8 };
enum TimeType { LiftOff = 0, TouchDown = 1, Inter = 2 };

// mpc/include/spline/end_effector_splines.h:17-32
class SplineTimes {
public:
    SplineTimes(double time, TimeType type) : time_(time), type_(type) {}
    SplineTimes() : time_(0), type_(LiftOff) {}
    double GetTime() const { return time_; }
    TimeType GetType() const { return type_; }
    void SetTime(double time) { time_ = time; }
private:
    double time_;
    TimeType type_;
};
using time_v = std::vector<SplineTimes>;

// mpc/include/mpc.h:39-62
struct MPCInfo {
    int num_nodes = 20;
    int num_qp_iterations = 1;
    int num_contacts = 4;
    double friction_coef = 0.5;
    vector_t vel_bounds, joint_bounds_lb, joint_bounds_ub;
    std::vector<std::string> ee_frames;
    int discretization_steps = 1;
    int num_switches = 0;
    double integrator_dt = 0.05;
    double force_bound = 150;
    double swing_height = 0.075;
    double foot_offset = 0.015;
    vector_t nom_state;                 // nominal configuration: base position, base quaternion xyzw, joint angles
    vector_2t ee_box_size;
    int real_time_iters = 1;
    MPCVerbosityLevel verbose = Nothing;
    double force_cost = 0;
};

inline void check_srbm(int rc) { if (rc != 0) throw std::runtime_error(std::string("srbm: ") + srbm_last_error()); }

// mpc/include/trajectory.h:20-175: the part of the interface that callers of the MPC use on the RESULT of a solve.  A value.
class Trajectory {
public:
    Trajectory() { rec_.num_states = 0; }
    explicit Trajectory(const srbm_trajectory& rec) : rec_(rec) {}
    const srbm_trajectory& Record() const { return rec_; }
    srbm_trajectory& Record() { return rec_; }

    std::vector<vector_t> GetStates() const { std::vector<vector_t> s; for (int k = 0; k < rec_.num_states; k++) s.push_back(GetState(k)); return s; }
    vector_t GetState(int node) const {
        if (node < 0 || node >= rec_.num_states) throw std::runtime_error("Trajectory node out of range.");
        vector_t s(13);
        for (int i = 0; i < 13; i++) s(i) = rec_.states[node][i];
        return s;
    }
    void SetState(int idx, const vector_t& state) {
        if (idx < 0 || idx >= rec_.num_states || state.size() != 13) throw std::runtime_error("Trajectory::SetState: bad node or state size.");
        for (int i = 0; i < 13; i++) rec_.states[idx][i] = state(i);
    }
    double GetTime(int node) const { return rec_.init_time + rec_.node_dt * node; }                     // trajectory.cpp:412-414
    int GetNode(double time) const { return (int)std::ceil((time - rec_.init_time) / rec_.node_dt); }   // trajectory.cpp:479-481
    Eigen::Vector3d GetForce(int end_effector, double time) const {                                      // trajectory.cpp:395-402
        Eigen::Vector3d f;
        Lookup(srbm_trajectory_eval(&rec_, end_effector, time, f.data(), nullptr, nullptr), time);
        return f;
    }
    Eigen::Vector3d GetEndEffectorLocation(int end_effector, double time) const {                        // trajectory.cpp:404-410
        Eigen::Vector3d p;
        Lookup(srbm_trajectory_eval(&rec_, end_effector, time, nullptr, p.data(), nullptr), time);
        return p;
    }
    std::vector<bool> GetContacts(double time) const {
        std::vector<bool> c(4);
        for (int ee = 0; ee < 4; ee++) { int in = 0; Lookup(srbm_trajectory_eval(&rec_, ee, time, nullptr, nullptr, &in), time); c[ee] = in != 0; }
        return c;
    }
    controller::Contact GetDesiredContacts(double time) const { controller::Contact c(4); c.in_contact_ = GetContacts(time); return c; }
    int GetNumContactNodes(int ee) const { int n = 0; for (int k = 0; k < rec_.nk[ee]; k++) n += rec_.knot_kind[ee][k] <= 1; return n; }
    std::vector<time_v> GetContactTimes() const {
        std::vector<time_v> out(4);
        for (int ee = 0; ee < 4; ee++)
            for (int k = 0; k < rec_.nk[ee]; k++)
                if (rec_.knot_kind[ee][k] <= 1) out[ee].emplace_back(rec_.knot_time[ee][k], rec_.knot_kind[ee][k] == 0 ? LiftOff : TouchDown);
        return out;
    }
    double GetNextContactTime(int ee, double time) const {                                               // end_effector_splines.cpp:1033-1040
        for (int k = 0; k < rec_.nk[ee]; k++) if (rec_.knot_kind[ee][k] == 1 && rec_.knot_time[ee][k] > time) return rec_.knot_time[ee][k];
        throw std::runtime_error("No touch down after the given time.");
    }
    // Trajectory::SplinesAsVec (trajectory.cpp:429-452): the spline variables in the order of the QP's decision vector -- per foot and coordinate the
    // force variables (value, slope / FORCE_MULT of every stance-interior knot), then per foot and xy coordinate the mutable position nodes
    vector_t SplinesAsVec() const {
        std::vector<double> v(4 * (3 * 2 + 2) * 32);
        int n = 0;
        if (srbm_trajectory_splines_as_vec(&rec_, v.data(), (int)v.size(), &n, nullptr) != 0) throw std::runtime_error(std::string("srbm: ") + srbm_last_error());
        vector_t out(n);
        for (int i = 0; i < n; i++) out(i) = v[i];
        return out;
    }
    // Trajectory::PrintTrajectoryToFile (trajectory.cpp:146-223) AS CODED: the force / position / timing blocks are commented out in the reference, their
    // headings are still written; the states matrix and the spline vector go out in Eigen's default format (columns right-aligned to the widest
    // coefficient, ' ' between coefficients)
    void PrintTrajectoryToFile(const std::string& file_name) const {
        std::ofstream file;
        file.open(file_name);
        file << "states: " << std::endl;
        file << FormatDense(file, rec_.num_states, 13, [&](int r, int c) { return rec_.states[r][c]; }) << std::endl;
        file << "force spline: " << std::endl;
        file << "position spline: " << std::endl;
        file << "timings: " << std::endl;
        const vector_t sv = SplinesAsVec();
        file << "spline vec: \n" << FormatDense(file, (int)sv.size(), 1, [&](int r, int) { return sv(r); }) << std::endl;
        file.close();
    }
private:
    // Eigen's operator<< for a dense matrix with the default IOFormat: stream precision, aligned columns
    template <class Get>
    static std::string FormatDense(const std::ostream& like, int rows, int cols, Get get) {
        std::vector<std::string> cell((size_t)rows * cols);
        size_t width = 0;
        for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) {
            std::ostringstream o;
            o.copyfmt(like);
            o << get(r, c);
            cell[(size_t)r * cols + c] = o.str();
            width = std::max(width, o.str().size());
        }
        std::string out;
        for (int r = 0; r < rows; r++) {
            if (r) out += "\n";
            for (int c = 0; c < cols; c++) {
                if (c) out += " ";
                const std::string& x = cell[(size_t)r * cols + c];
                out += std::string(width - x.size(), ' ') + x;
            }
        }
        return out;
    }
    static void Lookup(int rc, double time) {
        if (rc != 0) throw std::runtime_error("Trajectory: invalid time " + std::to_string(time) + " for the spline lookup.");   // end_effector_splines.cpp:1066-1083
    }
    srbm_trajectory rec_;
};

class MPCSingleRigidBody;

// mpc/include/qp/qp_partials.h:15-57.  The reference fills these with 260x372 / 752x372 matrices on the host and contracts them in
// GaitOptimizer::ComputeCostFcnDerivWrtContactTimes; here the contraction is fused on the device (srbm_gait_compute_gradient), which is what
// GaitOptimizer uses.  The members are still DATA for a caller that reads them (test/mpc_test.cpp:181-184): ComputeParamPartialsClarabel fills
// dA, dG, db, dh from the device (srbm_gait_get_param_partials: the same entries the fused gradient contracts), GetQPPartials fills the rank-2 QP
// partials of clarabel_interface.cpp:180-260 -- unless MPCSingleRigidBody::SetPartialsAsData(false) asked for handles only (a facade extension for
// callers like MPCController::GaitOpt that never read them).  Dense matrices where the reference has Eigen::SparseMatrix (`dA += partials.dA`
// reads the same); dP, dl, du, dq of the parameter partials stay empty as they do in the reference.
struct QPPartials {
    matrix_t dA, dP, dG;
    vector_t dl, du, dq, db, dh;
    const MPCSingleRigidBody* owner = nullptr; int ee = -1, idx = -1;
    void SetZero() { dP.setZero(); dA.setZero(); dG.setZero(); dq.setZero(); dl.setZero(); du.setZero(); db.setZero(); dh.setZero(); }
};
struct QPPartialsDense {
    matrix_t dA, dP, dG;
    vector_t dl, du, dq, db, dh;
    MPCSingleRigidBody* owner = nullptr; bool modified = false;
    void SetZero() { dP.setZero(); dA.setZero(); dG.setZero(); dq.setZero(); dl.setZero(); du.setZero(); db.setZero(); dh.setZero(); modified = false; }
};

// mpc/include/qp/qp_data.h:60-135: what callers read of the QP of the last solve (test/mpc_test.cpp:125-160, test/gait_opt_playground.cpp:119-130),
// in the Clarabel form the live path uses (qp_data.cpp:195-300 with using_clarabel_): A x + s = ub_, rows in constraint order -- dynamics, force
// box, friction cone, foot box, touch-down rows, start rows; lb_ stays zero.  Dense where the reference holds Eigen::SparseMatrix.  Filled by
// MPCSingleRigidBody::GetQPData from srbm_export_qp (a read-back path: the device never forms these matrices).
struct QPData {
    matrix_t sparse_constraint_, sparse_cost_;
    vector_t lb_, ub_, cost_linear;
    int num_dynamics_constraints = 0, num_decision_vars = 0;
    int num_cone_constraints_ = 0, num_box_constraints_ = 0, num_fk_constraints_ = 0, num_force_box_constraints_ = 0, num_fk_ineq_constraints_ = 0,
        num_ee_location_constraints_ = 0, num_start_ee_constraints_ = 0, num_td_pos_constraints_ = 0, num_raibert_constraints_ = 0;
    bool using_clarabel_ = true;
    int num_equality_ = 0, num_inequality_ = 0;
    int GetTotalNumConstraints() const {            // qp_data.cpp:61-100 for the constraint list of the SRBM MPC (msrb.cpp:13-20)
        return num_dynamics_constraints + num_force_box_constraints_ + num_cone_constraints_ + num_ee_location_constraints_ + num_td_pos_constraints_ +
               num_raibert_constraints_ + num_start_ee_constraints_;
    }
};

// what MPC::GetModel() hands out (mpc/include/models/model.h): the two getters the caller uses (mpc_controller.cpp:232,239)
class Model {
public:
    explicit Model(double mass) : mass_(mass) {}
    int GetNumManifoldStates() const { return 13; }
    int GetNumTangentStates() const { return 12; }
    double GetMass() const { return mass_; }
private:
    double mass_;
};

// mpc/include/models/single_rigid_body_model.h:28-100: the entry points callers of the MPC use -- kinematics (InverseKinematics,
// GetEndEffectorLocations), the manifold <-> tangent maps (controllers/mpc_controller.cpp:60), GetIr / GetIrInv / GetMass (:258, qp_control).
// Owns a batch of ONE instance of the library (the kinematics entries do not touch the MPC state of a batch).
class SingleRigidBodyModel {
public:
    SingleRigidBodyModel(const std::string& robot_urdf, const std::vector<std::string>& frames, int discretization_steps, double dt, const vector_t& nom_state)
        : frames_(frames), discretization_steps_(discretization_steps), dt_(dt), consts_(ModelConstantsFromUrdf(robot_urdf, ToStdVec(nom_state))),
          legs_(LegKinematicsFromUrdf(robot_urdf)) { Create(); }
    SingleRigidBodyModel(const srbm_model& consts, const srbm_leg_kinematics& legs, double dt) : dt_(dt), consts_(consts), legs_(legs) { Create(); }
    SingleRigidBodyModel(const SingleRigidBodyModel& o) : frames_(o.frames_), discretization_steps_(o.discretization_steps_), dt_(o.dt_), consts_(o.consts_), legs_(o.legs_) { Create(); }
    SingleRigidBodyModel& operator=(const SingleRigidBodyModel& o) {
        if (this == &o) return *this;
        Release();
        frames_ = o.frames_; discretization_steps_ = o.discretization_steps_; dt_ = o.dt_; consts_ = o.consts_; legs_ = o.legs_;
        Create();
        return *this;
    }
    ~SingleRigidBodyModel() { Release(); }

    // single_rigid_body_model.cpp:314-425.  state: manifold SRBM state (13); state_guess: full configuration (19), its joint part is the
    // initial guess.  The joint limits are accepted and ignored, as the reference does (its clamp is commented out, :407-414).
    vector_t InverseKinematics(const vector_t& state, const std::vector<vector_3t>& end_effector_location, const vector_t& state_guess,
                               const vector_t& /*joint_limits_ub*/, const vector_t& /*joint_limits_lb*/) {
        if (state.size() != 13 || state_guess.size() != 19 || end_effector_location.size() != 4) throw std::runtime_error("InverseKinematics: wrong sizes.");
        double s[13], e[12], g[19], q[19]; int status = 0;
        for (int i = 0; i < 13; i++) s[i] = state(i);
        for (int i = 0; i < 19; i++) g[i] = state_guess(i);
        for (int ee = 0; ee < 4; ee++) for (int c = 0; c < 3; c++) e[3 * ee + c] = end_effector_location[ee](c);
        check_srbm(srbm_inverse_kinematics(h_, s, e, g, q, nullptr, &status));
        if (status) {
            std::cerr << "IK did not converge." << std::endl;
            throw std::runtime_error("IK did not converge.");
        }
        vector_t out(19);
        for (int i = 0; i < 19; i++) out(i) = q[i];
        return out;
    }
    // single_rigid_body_model.cpp:443-455
    std::vector<vector_3t> GetEndEffectorLocations(const vector_t& q) {
        if (q.size() != 19) throw std::runtime_error("GetEndEffectorLocations: wrong size.");
        double qq[19], e[12];
        for (int i = 0; i < 19; i++) qq[i] = q(i);
        check_srbm(srbm_forward_kinematics(h_, qq, e));
        std::vector<vector_3t> out(4);
        for (int ee = 0; ee < 4; ee++) for (int c = 0; c < 3; c++) out[ee](c) = e[3 * ee + c];
        return out;
    }
    double GetMass() const { return consts_.mass; }
    matrix_t GetIrInv() const {                                  // single_rigid_body_model.cpp:33-37 (Ir_inv_)
        const double* a = consts_.Ir;
        const double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
        matrix_t r = matrix_t::Zero(3, 3);
        r(0, 0) = (a[4] * a[8] - a[5] * a[7]) / det; r(0, 1) = (a[2] * a[7] - a[1] * a[8]) / det; r(0, 2) = (a[1] * a[5] - a[2] * a[4]) / det;
        r(1, 0) = (a[5] * a[6] - a[3] * a[8]) / det; r(1, 1) = (a[0] * a[8] - a[2] * a[6]) / det; r(1, 2) = (a[2] * a[3] - a[0] * a[5]) / det;
        r(2, 0) = (a[3] * a[7] - a[4] * a[6]) / det; r(2, 1) = (a[1] * a[6] - a[0] * a[7]) / det; r(2, 2) = (a[0] * a[4] - a[1] * a[3]) / det;
        return r;
    }
    matrix_33t GetIr() const {                                   // single_rigid_body_model.h:81; used at controllers/mpc_controller.cpp:258
        matrix_33t r;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = consts_.Ir[3 * i + j];
        return r;
    }
    // single_rigid_body_model.cpp:179-220 (ref_state is unused there: quat_ref is the identity); controllers/mpc_controller.cpp:60
    static vector_3t ConvertManifoldToTangentQuat(const Eigen::Vector4d& state, const Eigen::Vector4d& /*ref_state*/) {
        double s[13] = {0}, t[12];
        for (int i = 0; i < 4; i++) s[6 + i] = state(i);
        check_srbm(srbm_convert_manifold_to_tangent(s, t));
        vector_3t r;
        for (int i = 0; i < 3; i++) r(i) = t[6 + i];
        return r;
    }
    vector_t ConvertManifoldStateToTangentState(const vector_t& state, const vector_t& ref_state) const {
        if (state.size() != 13 || ref_state.size() != 13) throw std::runtime_error("ConvertManifoldStateToTangentState: 13-entry states expected.");
        vector_t t(12);
        check_srbm(srbm_convert_manifold_to_tangent(state.data(), t.data()));
        return t;
    }
    vector_t ConvertTangentStateToManifoldState(const vector_t& state, const vector_t& /*ref_state*/) const {
        if (state.size() != 12) throw std::runtime_error("ConvertTangentStateToManifoldState: a 12-entry tangent state expected.");
        vector_t s(13);
        check_srbm(srbm_convert_tangent_to_manifold(state.data(), s.data()));
        return s;
    }
    vector_3t GetCOMPosition(const vector_t& state) const { vector_3t p; for (int i = 0; i < 3; i++) p(i) = state(i); return p; }
    vector_2t GetCOMHipOffset(int ee) const { vector_2t r; r(0) = consts_.hip_xy[2 * ee]; r(1) = consts_.hip_xy[2 * ee + 1]; return r; }   // (the box centres of the MPC add the offsets of GetCOMToHip, srbm_get_ee_box_center)
    int GetNumManifoldStates() const { return 13; }
    int GetNumTangentStates() const { return 12; }
    int GetNumEndEffectors() const { return 4; }
    static constexpr int QUAT_SIZE = 4, QUAT_START = 6, ORIENTATION_START = 6, ANG_VEL_START = 9, LIN_MOM_START = 3, POS_START = 0;
    const srbm_leg_kinematics& LegKinematics() const { return legs_; }

private:
    static std::vector<double> ToStdVec(const vector_t& v) { std::vector<double> r(v.size()); for (int i = 0; i < (int)v.size(); i++) r[i] = v(i); return r; }
    void Create() {
        srbm_mpc_info ci{};
        ci.num_nodes = 10; ci.integrator_dt = dt_ > 0 ? dt_ : 0.05; ci.friction_coef = 0.5; ci.force_bound = 100; ci.swing_height = 0.1; ci.foot_offset = 0;
        ci.ee_box_size[0] = ci.ee_box_size[1] = 0.1; ci.force_cost = 0;
        check_srbm(srbm_batch_create(&h_, 1, &ci, &consts_, 0));
        if (srbm_set_leg_kinematics(h_, &legs_)) { const std::string m = srbm_last_error(); Release(); throw std::runtime_error(m); }
    }
    void Release() { if (h_) srbm_batch_destroy(h_); h_ = nullptr; }
    std::vector<std::string> frames_;
    int discretization_steps_ = 1;
    double dt_ = 0.05;
    srbm_model consts_{};
    srbm_leg_kinematics legs_{};
    srbm_batch* h_ = nullptr;
};


// mpc/include/mpc.h:70-170 + mpc/include/mpc_single_rigid_body.h:11-76
class MPCSingleRigidBody {
public:
    MPCSingleRigidBody(const MPCInfo& info, const std::string& robot_urdf)
        : MPCSingleRigidBody(info, ModelConstantsFromUrdf(robot_urdf, ToStd(info.nom_state))) { legs_ = LegKinematicsFromUrdf(robot_urdf); has_legs_ = true; }
    // the same object from constants computed elsewhere (e.g. by pinocchio in a build that has it)
    MPCSingleRigidBody(const MPCInfo& info, const srbm_model& model) : info_(info), model_consts_(model), model_(model.mass) {
        srbm_mpc_info ci{};
        ci.num_nodes = info.num_nodes; ci.integrator_dt = info.integrator_dt; ci.friction_coef = info.friction_coef;
        ci.force_bound = info.force_bound; ci.swing_height = info.swing_height; ci.foot_offset = info.foot_offset;
        ci.ee_box_size[0] = info.ee_box_size(0); ci.ee_box_size[1] = info.ee_box_size(1); ci.force_cost = info.force_cost;
        check_srbm(srbm_batch_create(&h_, 1, &ci, &model_consts_, 0));
        // the single-instance drop-in keeps ClarabelInterface's criterion (gap 1e-15) for every solve: any of them may be differentiated next
        // (ComputeDerivativeTerms).  SetSolverStepRule opts into the library's faster termination (srbm_set_solver_step_rule).
        check_srbm(srbm_set_solver_step_rule(h_, 0.0, 0.0));
    }
    void SetSolverStepRule(double tol_step, double start_mu) { check_srbm(srbm_set_solver_step_rule(h_, tol_step, start_mu)); }
    // value semantics (mpc.cpp:1133-1181, mpc_single_rigid_body.cpp:804-807)
    MPCSingleRigidBody(const MPCSingleRigidBody& other) : info_(other.info_), model_consts_(other.model_consts_), model_(other.model_),
                                                          used_log_file_(other.used_log_file_), solves_(other.solves_), last_solve_ms_(other.last_solve_ms_),
                                                          history_(other.history_), contact_sched_change_(other.contact_sched_change_),
                                                          partials_as_data_(other.partials_as_data_), legs_(other.legs_), has_legs_(other.has_legs_) {
        check_srbm(srbm_batch_clone(other.h_, &h_));
    }
    MPCSingleRigidBody& operator=(const MPCSingleRigidBody& other) {
        if (this == &other) return *this;
        Release();
        info_ = other.info_; model_consts_ = other.model_consts_; model_ = other.model_; used_log_file_ = other.used_log_file_; solves_ = other.solves_;
        last_solve_ms_ = other.last_solve_ms_; partials_as_data_ = other.partials_as_data_; legs_ = other.legs_; has_legs_ = other.has_legs_; kin_.reset();
        history_ = other.history_; contact_sched_change_ = other.contact_sched_change_;
        check_srbm(srbm_batch_clone(other.h_, &h_));
        return *this;
    }
    ~MPCSingleRigidBody() { Release(); }

    // ---- set-up (mpc_controller.cpp:57-67, 89) ----
    void SetStateTrajectoryWarmStart(const std::vector<vector_t>& states) {                              // mpc.cpp:700-706
        if ((int)states.size() < 1 || states.at(0).size() != 13) throw std::runtime_error("Warm start states are the wrong size.");
        // the reference copies node by node; every caller passes one state replicated over the horizon (test/simulation_mpc.cpp:110-113):
        // install node 0 everywhere, then the individual nodes through the trajectory record
        check_srbm(srbm_set_state_trajectory_warm_start(h_, states.at(0).data()));
        bool uniform = true;
        for (const vector_t& s : states) for (int i = 0; i < 13; i++) uniform = uniform && s(i) == states.at(0)(i);
        if (!uniform) {
            Trajectory t = GetTrajectory();
            for (int k = 0; k < (int)states.size() && k <= info_.num_nodes; k++) t.SetState(k, states[k]);
            SetWarmStartTrajectory(t);
        }
    }
    void AddQuadraticTrackingCost(const vector_t& state_des, const matrix_t& Q) {                        // mpc.cpp:533-540
        if (state_des.size() != 12 || Q.rows() != 12 || Q.cols() != 12) throw std::runtime_error("Supplied quadratic cost term is the wrong size.");
        double q[144];
        RowMajor(Q, q);
        check_srbm(srbm_add_quadratic_tracking_cost(h_, state_des.data(), q));
    }
    void AddForceCost(double weight) { check_srbm(srbm_add_force_cost(h_, weight)); }                     // mpc.cpp:791-802
    void SetQuadraticFinalCost(const matrix_t& Phi) {                                                    // mpc.cpp:137-143
        if (Phi.rows() != 12 || Phi.cols() != 12) throw std::runtime_error("Supplied quadratic cost term is the wrong size.");
        double q[144];
        RowMajor(Phi, q);
        check_srbm(srbm_set_quadratic_final_cost(h_, q));
    }
    void SetLinearFinalCost(const vector_t& w) {                                                         // mpc.cpp:145-151
        if (w.size() != 12) throw std::runtime_error("Supplied linear cost term is the wrong size.");
        check_srbm(srbm_set_linear_final_cost(h_, w.data()));
    }
    static std::vector<std::vector<double>> CreateDefaultSwitchingTimes(int, int num_ee, double) {       // mpc.cpp:566-608 (arguments ignored there too)
        return std::vector<std::vector<double>>(num_ee, {0, 0.3, 0.6, 0.9, 1.2});
    }
    void SetDefaultGaitTrajectory(Gaits gait, int, const std::vector<vector_3t>& ee_pos) {               // mpc.cpp:626-685: validation only
        if (gait != Trot) throw std::runtime_error("Only the trot gait is implemented.");
        if (ee_pos.size() != 4) throw std::runtime_error("Trot gait needs 4 end effectors.");
    }
    void SetVerbosityLevel(MPCVerbosityLevel v) { info_.verbose = v; }

    // ---- solves ----
    Trajectory CreateInitialRun(const vector_t& state, const std::vector<vector_3t>& ee_start_locations) {   // mpc.cpp:78-90
        // ten Solve calls at t = 0, each with its own row in the statistics (srbm_create_initial_run is the same ten launches without the rows)
        for (int num_iter = 0; num_iter < 10; num_iter++) Solve(state, 0, ee_start_locations);
        return GetTrajectory();
    }
    Trajectory GetRealTimeUpdate(const vector_t& state, double init_time, const std::vector<vector_3t>& ee_start_locations, bool /*high_quality*/) {   // mpc.cpp:92-108
        return Solve(state, init_time, ee_start_locations);
    }
    Trajectory Solve(const vector_t& state, double init_time, const std::vector<vector_3t>& ee_start_locations) {   // msrb.cpp:25-216
        double ee[12];
        PackEE(ee_start_locations, ee);
        CheckState(state);
        const auto t0 = std::chrono::steady_clock::now();
        check_srbm(srbm_get_real_time_update(h_, state.data(), &init_time, ee));
        last_solve_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();      // utils::Timer around Solve, msrb.cpp:27,179
        solves_ += 1;
        int st = 0, err = 0;
        check_srbm(srbm_get_status(h_, &st, &err));
        if (err) throw std::runtime_error("MPC solve raised error bits " + std::to_string(err) + " (time outside the spline range / capacity).");
        RecordStats();
        return GetTrajectory();
    }
    void SetWarmStartTrajectory(const Trajectory& trajectory) { check_srbm(srbm_set_warm_start_trajectory(h_, 0, 1, &trajectory.Record())); }   // mpc.cpp:110-119
    void UpdateContactTimes(std::vector<time_v>& contact_times) {                                        // mpc.cpp:1085-1088
        if (contact_times.size() != 4) throw std::runtime_error("Contact times for 4 end effectors expected.");
        size_t mx = 1;
        for (auto& tv : contact_times) mx = std::max(mx, tv.size());
        std::vector<double> t(4 * mx, 0.0);
        for (int ee = 0; ee < 4; ee++) for (size_t i = 0; i < contact_times[ee].size(); i++) t[ee * mx + i] = contact_times[ee][i].GetTime();
        check_srbm(srbm_update_contact_times(h_, t.data(), (int)mx));
        contact_sched_change_.push_back(solves_);                                                        // mpc.cpp:1087 (run_num_)
    }
    void AdjustForCurrentContacts(double time, const controller::Contact& contact) {                     // mpc.cpp:1195-1203
        int c[4];
        for (int i = 0; i < 4; i++) c[i] = contact.in_contact_.at(i) ? 1 : 0;
        check_srbm(srbm_adjust_for_current_contacts(h_, &time, c));
    }

    // ---- results ----
    Trajectory GetTrajectory() const { srbm_trajectory r; check_srbm(srbm_get_trajectory(h_, 0, 1, &r)); return Trajectory(r); }   // mpc.cpp:1023-1025
    double GetCost() const { double c = 0; check_srbm(srbm_get_cost(h_, &c)); return c; }
    double GetAvgCost() const { double c = 0; check_srbm(srbm_get_avg_cost(h_, &c)); return c; }           // mpc.cpp:991-998
    SolveQuality GetSolveQuality() const { int st = 0; check_srbm(srbm_get_status(h_, &st, nullptr)); return (SolveQuality)st; }
    int GetNumDecisionVars() const { return Sizes()[0]; }
    int GetNumConstraints() const { return Sizes()[1]; }
    int GetNode(double time) const { return GetTrajectory().GetNode(time); }                             // mpc.cpp:1027-1029
    vector_t GetQPSolution() const {
        const int n = GetNumDecisionVars();
        std::vector<double> x(MaxDecisionVars());
        check_srbm(srbm_get_qp_solution(h_, x.data(), (int)x.size()));
        vector_t v(n);
        for (int i = 0; i < n; i++) v(i) = x[i];
        return v;
    }
    double GetModifiedCost(int /*num_nodes*/) const { return GetCost(); }                                // msrb.cpp:798-802 (the truncation is commented out there)
    vector_t GetTargetConfig(double time) const {                                                        // mpc.cpp:708-711: states.at(node).tail(num_states_ - 5), num_states_ = 12
        const Trajectory t = GetTrajectory();
        const int node = (int)std::floor((time - t.Record().init_time) / info_.integrator_dt);
        return t.GetState(node).tail(7);
    }
    vector_t GetForceTarget(double time) const {                                                         // mpc.cpp:713-724
        const Trajectory t = GetTrajectory();
        const controller::Contact contacts = t.GetDesiredContacts(time);
        vector_t forces(contacts.GetNumContacts() * 3);
        int idx = 0;
        for (int ee = 0; ee < 4; ee++)
            if (contacts.in_contact_.at(ee)) { const Eigen::Vector3d f = t.GetForce(ee, time); for (int c = 0; c < 3; c++) forces(idx + c) = f(c); idx += 3; }
        return forces;
    }
    // msrb.cpp:477-500: inverse kinematics of the trajectory's node state and foot locations at `time`; prev_state (19) is the guess
    vector_t GetFullTargetState(double time, const vector_t& prev_state) {
        const Trajectory t = GetTrajectory();
        std::vector<vector_3t> ee_locations(4);
        for (int ee = 0; ee < 4; ee++) ee_locations.at(ee) = t.GetEndEffectorLocation(ee, time);
        return KinModel().InverseKinematics(t.GetState(t.GetNode(time)), ee_locations, prev_state, info_.joint_bounds_ub, info_.joint_bounds_lb);
    }
    SingleRigidBodyModel GetModelCopy() const { return KinModel(); }                                     // msrb.cpp:1060-1062
    // the leg geometry of a model built from constants (the URDF constructor reads it from the file)
    void SetLegKinematics(const srbm_leg_kinematics& legs) { legs_ = legs; has_legs_ = true; kin_.reset(); }
    const QPData& GetQPData() const {                                                                    // mpc.cpp:1097-1099
        const std::array<int, 8> sz = Sizes();
        const int n = sz[0], mrows = sz[1], ns = sz[7];
        QPData& d = qp_data_;
        d.num_decision_vars = n; d.num_dynamics_constraints = (info_.num_nodes + 1) * 12;
        d.num_force_box_constraints_ = 2 * ns; d.num_cone_constraints_ = 4 * ns; d.num_ee_location_constraints_ = 16 * (info_.num_nodes - 3);
        d.num_td_pos_constraints_ = sz[6]; d.num_start_ee_constraints_ = 8; d.num_raibert_constraints_ = 0;
        d.num_equality_ = sz[2]; d.num_inequality_ = sz[3];
        if (n == 0) return d;                                             // nothing solved yet
        std::vector<double> A((size_t)mrows * n), b(mrows), P((size_t)n * n), q(n);
        check_srbm(srbm_export_qp(h_, 0, A.data(), b.data(), P.data(), q.data()));
        d.sparse_constraint_ = matrix_t::Zero(mrows, n); d.sparse_cost_ = matrix_t::Zero(n, n);
        for (int r = 0; r < mrows; r++) for (int c = 0; c < n; c++) d.sparse_constraint_(r, c) = A[(size_t)r * n + c];
        for (int r = 0; r < n; r++) for (int c = 0; c < n; c++) d.sparse_cost_(r, c) = P[(size_t)r * n + c];
        d.ub_ = vector_t(mrows); d.lb_ = vector_t::Zero(mrows); d.cost_linear = vector_t(n);
        for (int r = 0; r < mrows; r++) d.ub_(r) = b[r];
        for (int c = 0; c < n; c++) d.cost_linear(c) = q[c];
        return d;
    }
    controller::Contact GetDesiredContacts(double time) const { return GetTrajectory().GetDesiredContacts(time); }
    std::vector<Eigen::Vector2d> GetEEBoxCenter() {                                                      // msrb.cpp:502-509
        double c[8];
        check_srbm(srbm_get_ee_box_center(h_, c));
        std::vector<Eigen::Vector2d> out(4);
        for (int ee = 0; ee < 4; ee++) { out[ee](0) = c[2 * ee]; out[ee](1) = c[2 * ee + 1]; }
        return out;
    }
    std::vector<std::vector<Eigen::Vector3d>> CreateVizData() {                                          // msrb.cpp:359-373
        const Trajectory t = GetTrajectory();
        std::vector<std::vector<Eigen::Vector3d>> fk(5);
        for (int ee = 0; ee < 5; ee++)
            for (int node = 0; node < info_.num_nodes + 1; node++) {
                if (ee == 4) { const vector_t s = t.GetState(node); Eigen::Vector3d p; for (int i = 0; i < 3; i++) p(i) = s(i); fk[ee].push_back(p); }
                else fk[ee].push_back(t.GetEndEffectorLocation(ee, t.GetTime(node)));
            }
        return fk;
    }
    const Model* GetModel() const { return &model_; }

    // ---- bilevel entry points (mpc.cpp:1047-1069, msrb.cpp:642-792), see GaitOptimizer below ----
    bool ComputeDerivativeTerms() {
        if (GetSolveQuality() != Solved) return false;               // mpc.cpp:1048
        check_srbm(srbm_gait_compute_sensitivity(Gait()));
        return true;
    }
    bool GetQPPartials(QPPartialsDense& partials) const {                                                // mpc.cpp:1058-1069
        if (GetSolveQuality() != Solved) return false;
        partials.owner = const_cast<MPCSingleRigidBody*>(this); partials.modified = false;
        if (partials_as_data_) FillQPPartials(partials);
        return true;
    }
    // msrb.cpp:642-792.  The partials are evaluated on the MPC's CURRENT trajectory (every caller passes mpc.GetTrajectory() or the trajectory the
    // MPC still holds: mpc_controller.cpp:535-545, test/gait_opt_playground.cpp:36-41, test/mpc_test.cpp:121-182 before the update)
    bool ComputeParamPartialsClarabel(const Trajectory&, QPPartials& partials, int ee, int idx) {
        partials.owner = this; partials.ee = ee; partials.idx = idx;
        if (partials_as_data_) {
            const std::array<int, 8> sz = Sizes();
            const int n = sz[0], me = sz[2], mi = sz[3];
            std::vector<double> dA((size_t)me * n), dG((size_t)mi * n), db(me), dh(mi);
            check_srbm(srbm_gait_get_param_partials(h_, 0, ee, idx, dA.data(), dG.data(), db.data(), dh.data()));
            partials.dA = matrix_t::Zero(me, n); partials.dG = matrix_t::Zero(mi, n); partials.db = vector_t(me); partials.dh = vector_t(mi);
            for (int r = 0; r < me; r++) { partials.db(r) = db[r]; for (int c = 0; c < n; c++) partials.dA(r, c) = dA[(size_t)r * n + c]; }
            for (int r = 0; r < mi; r++) { partials.dh(r) = dh[r]; for (int c = 0; c < n; c++) partials.dG(r, c) = dG[(size_t)r * n + c]; }
        }
        return true;
    }
    // facade extension: false = QPPartials / QPPartialsDense are handles (no read-back from the device); the fused gradient does not need them
    void SetPartialsAsData(bool as_data) { partials_as_data_ = as_data; }
    srbm_gait* Gait() { if (!gait_) check_srbm(srbm_gait_create(h_, &gait_)); return gait_; }
    srbm_batch* Handle() const { return h_; }
    const MPCInfo& Info() const { return info_; }

    // ---- statistics (mpc.cpp:804-816 RecordStats, 818-899 PrintStats, 901-989 PrintStatLineToFile) ----
    // MPC::PrintStats: the table of EVERY solve since construction, then the average compute time
    void PrintStats() { PrintStats(std::cout); }
    void PrintStats(std::ostream& os) const {
        const int col_width = 15, table_width = 10 * col_width;
        using std::setw; using std::setfill;
        os << setfill('-') << setw(table_width) << "" << std::endl;
        os << std::left << setfill(' ') << setw(table_width / 2 - 7) << "" << "MPC Statistics" << std::endl;
        os << setfill('-') << setw(table_width) << "" << std::endl;
        os << setfill(' ');
        PrintColumnNames(os);
        double avg_time = 0;
        for (int i = 0; i < (int)history_.size(); i++) {
            PrintRow(os, i);
            avg_time += history_[i].time_ms;
        }
        os << std::endl;
        avg_time = avg_time / history_.size();
        os << "Average compute time: " << avg_time << std::endl;
    }
    void PrintStatLineToFile(std::ofstream& log_file) {
        if (!used_log_file_) { PrintHeader(log_file); used_log_file_ = true; }
        if (history_.empty()) throw std::out_of_range("PrintStatLineToFile: no solve has been recorded.");     // (the reference indexes alpha_.size() - 1)
        PrintRow(log_file, (int)history_.size() - 1);
    }
    int GetNumRecordedSolves() const { return (int)history_.size(); }

private:
    static std::vector<double> ToStd(const vector_t& v) { std::vector<double> o(v.size()); for (int i = 0; i < (int)v.size(); i++) o[i] = v(i); return o; }
    static void RowMajor(const matrix_t& M, double* out) { for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) out[12 * i + j] = M(i, j); }
    static void PackEE(const std::vector<vector_3t>& ee, double* out) {
        if (ee.size() != 4) throw std::runtime_error("4 end effector locations expected.");
        for (int e = 0; e < 4; e++) for (int c = 0; c < 3; c++) out[3 * e + c] = ee[e](c);
    }
    static void CheckState(const vector_t& s) { if (s.size() != 13) throw std::runtime_error("The SRBM state has 13 entries."); }
    std::array<int, 8> Sizes() const { std::array<int, 8> s{}; check_srbm(srbm_get_sizes(h_, s.data())); return s; }
    int MaxDecisionVars() const { int cap[4]; check_srbm(srbm_get_capacity(cap)); return (info_.num_nodes + 1) * 12 + cap[1]; }     // either build of the library
    SingleRigidBodyModel& KinModel() const {
        if (!has_legs_) throw std::runtime_error("The leg kinematics are not known: construct the MPC from a URDF or call SetLegKinematics.");
        if (!kin_) kin_ = std::make_shared<SingleRigidBodyModel>(model_consts_, legs_, info_.integrator_dt);
        return *kin_;
    }
    // the rank-2 QP partials of ClarabelInterface::CalcDerivativeWrtMats / Vecs (clarabel_interface.cpp:180-260) from the device's sensitivity
    // d = [dz; dlam; dnu], the raw minimiser and the duals
    void FillQPPartials(QPPartialsDense& p) const {
        const std::array<int, 8> sz = Sizes();
        const int n = sz[0], mrows = sz[1], me = sz[2], mi = sz[3], nxs = (info_.num_nodes + 1) * 12;
        int cap[4]; check_srbm(srbm_get_capacity(cap));
        const int ldx = nxs + cap[1], ldz = nxs + 6 * cap[2] + 16 * (info_.num_nodes - 3) + 16, ldd = ldx + ldz;
        std::vector<double> x(ldx), z(ldz), s(ldz), d(ldd);
        check_srbm(srbm_get_raw_qp_minimiser(h_, x.data(), ldx));
        check_srbm(srbm_get_dual_solution(h_, z.data(), s.data(), ldz));
        check_srbm(srbm_gait_get_sensitivity(const_cast<MPCSingleRigidBody*>(this)->Gait(), d.data(), ldd));
        const double* dz = d.data(); const double* dlam = dz + n; const double* dnu = dlam + mi;
        auto nu = [&](int r) { return r < nxs ? z[r] : z[mi + r]; };          // dual vector in row order: dynamics, inequalities, touch-down + start rows
        const double* lam = z.data() + nxs;
        (void)mrows;
        p.dA = matrix_t::Zero(me, n); p.dG = matrix_t::Zero(mi, n);
        p.dq = vector_t(n); p.db = vector_t(me); p.dh = vector_t(mi);
        for (int c = 0; c < n; c++) p.dq(c) = dz[c];
        for (int r = 0; r < me; r++) { p.db(r) = -dnu[r]; for (int c = 0; c < n; c++) p.dA(r, c) = dnu[r] * x[c] + nu(r) * dz[c]; }
        for (int r = 0; r < mi; r++) { p.dh(r) = -lam[r] * dlam[r]; for (int c = 0; c < n; c++) p.dG(r, c) = lam[r] * dlam[r] * x[c] + lam[r] * dz[c]; }
    }
    void Release() {
        if (gait_) { srbm_gait_destroy(gait_); gait_ = nullptr; }
        if (h_) { srbm_batch_destroy(h_); h_ = nullptr; }
    }
    // one row per solve: what MPC::RecordStats (mpc.cpp:804-816) pushes onto its ten vectors
    struct SolveRecord { double time_ms, eq_violation, step_norm, alpha, cost_result, merit, merit_dd, cost; int solve_type; };
    void RecordStats() {
        double st[8], merit = 0, merit_dd = 0;
        check_srbm(srbm_get_stats(h_, st));
        check_srbm(srbm_get_merit(h_, &merit, &merit_dd));
        // cost_result_ and cost_ are both GetCostValue(prev_qp_sol) (mpc.cpp:809 and msrb.cpp:183-184): st[1] twice
        history_.push_back(SolveRecord{last_solve_ms_, st[2], st[3], st[0], st[1], merit, merit_dd, st[1], (int)GetSolveQuality()});
    }
    static void PrintColumnNames(std::ostream& os) {
        const int col_width = 15, table_width = 10 * col_width;
        using std::setw; using std::setfill;
        for (const char* n : {"Solve #", "Time (ms)", "Constraints", "Step Norm", "Alpha", "Cost", "Merit", "Merit dd", "Solve Type", "QP Cost"}) os << setw(col_width) << n;
        os << std::endl << setfill('-') << setw(table_width) << "" << std::endl << setfill(' ');
    }
    void PrintHeader(std::ostream& os) const {
        const int col_width = 15, table_width = 10 * col_width;
        using std::setw; using std::setfill;
        const std::time_t now = std::chrono::system_clock::to_time_t(std::chrono::system_clock::now());
        os << setfill('-') << setw(table_width) << "" << std::endl;
        os << std::left << setfill(' ') << setw(table_width / 2 - 7) << "" << "MPC Statistics" << std::endl;
        os << std::left << "MPC started at: " << std::ctime(&now);
        os << std::left << "Number of nodes: " << info_.num_nodes << std::endl;
        os << std::left << "MPC time step: " << info_.integrator_dt << std::endl;
        os << std::left << "Force bounds: " << info_.force_bound << std::endl;
        os << std::left << "End Effector box size: " << info_.ee_box_size(0) << " " << info_.ee_box_size(1) << std::endl;
        os << std::left << "Force cost: " << info_.force_cost << std::endl;
        os << std::left << "Foot offset: " << info_.foot_offset << std::endl;
        os << std::left << "Swing height: " << info_.swing_height << std::endl;
        os << setfill('-') << setw(table_width) << "" << std::endl;
        os << setfill(' ');
        PrintColumnNames(os);
    }
    void PrintRow(std::ostream& os, int i) const {
        static const char* names[] = {"Solved", "Solved Inacc", "Max Iter", "P - Infeasible", "D - Infeasible", "P - Infeasible Inacc", "D - Infeasible Inacc", "Unsolved", "Other"};
        const int col_width = 15, table_width = 10 * col_width;
        using std::setw;
        for (const int j : contact_sched_change_)
            if (i == j) std::cout << "Contact schedule changed " << setw(table_width - 25) << std::endl;       // (to std::cout in both printers, mpc.cpp:847-851, 943-947)
        const SolveRecord& r = history_.at(i);
        const int q = r.solve_type;
        // columns of mpc.cpp:979-988: i, solve_time_, equality violation, step norm, alpha, cost_result_, merit, merit dd, solve type, cost_
        os << std::left << setw(col_width) << i << setw(col_width) << r.time_ms << setw(col_width) << r.eq_violation << setw(col_width) << r.step_norm << setw(col_width) << r.alpha
           << setw(col_width) << r.cost_result << setw(col_width) << r.merit << setw(col_width) << r.merit_dd << setw(col_width) << names[q < 0 || q > 8 ? 8 : q]
           << setw(col_width) << r.cost << std::endl;
    }

    MPCInfo info_;
    srbm_model model_consts_;
    Model model_;
    srbm_batch* h_ = nullptr;
    srbm_gait* gait_ = nullptr;
    bool used_log_file_ = false;
    int solves_ = 0;
    double last_solve_ms_ = 0;
    std::vector<SolveRecord> history_;
    std::vector<int> contact_sched_change_;
    bool partials_as_data_ = true;
    srbm_leg_kinematics legs_{};
    bool has_legs_ = false;
    mutable std::shared_ptr<SingleRigidBodyModel> kin_;      // built on first use (GetFullTargetState / GetModelCopy)
    mutable QPData qp_data_;
};
using MPC = MPCSingleRigidBody;

// mpc/include/gait_optimizer.h:23-93.  The call protocol of MPCController::GaitOpt (mpc_controller.cpp:518-566) and of the line
// search (:333) is kept; the arithmetic (sensitivity, 20 parameter partials, contraction, LP, 10 candidate solves) runs on the device
// through the srbm_gait handle of the MPC the partials came from.
class GaitOptimizer {
public:
    GaitOptimizer(int num_ee, int /*num_contact_nodes*/, int num_decision_vars, int num_constraints, double /*contact_time_ub*/, double /*min_time*/)
        : num_ee_(num_ee), num_decision_vars_(num_decision_vars), num_constraints_(num_constraints), contact_times_(num_ee), num_times_(num_ee, 0) {}
    QPPartialsDense& GetQPPartials() { return qp_partials_; }
    QPPartials& GetParameterPartials(int ee, int idx) {
        if (ee < 0 || ee >= num_ee_ || idx < 0) throw std::runtime_error("Parameter partial index out of range.");
        if ((int)param_partials_.size() < num_ee_) param_partials_.resize(num_ee_);
        if ((int)param_partials_[ee].size() <= idx) param_partials_[ee].resize(idx + 1);
        return param_partials_[ee][idx];
    }
    void SetNumContactTimes(int ee, int num_times) { num_times_.at(ee) = num_times; }
    void UpdateSizes(int num_decision_vars, int num_constraints) { num_decision_vars_ = num_decision_vars; num_constraints_ = num_constraints; }
    void SetContactTimes(const std::vector<time_v>& contact_times) { contact_times_ = contact_times; }                       // gait_optimizer.cpp:395-408
    void ModifyQPPartials(const vector_t& /*xstar*/) { qp_partials_.modified = true; }                                        // dq += x*: part of the fused gradient
    void ComputeCostFcnDerivWrtContactTimes() {                                                                               // gait_optimizer.cpp:92-179
        MPCSingleRigidBody* m = Owner();
        check_srbm(srbm_gait_compute_gradient(m->Gait()));
        double g[SRBM_GAIT_NV]; int valid = 0;
        check_srbm(srbm_gait_get_gradient(m->Gait(), g, &valid));
        if (!valid) throw std::runtime_error("The cost function gradient needs a QP solved to tolerance.");
        int nv = 0;
        for (int n : num_times_) nv += n;
        dHdth_ = vector_t(nv);
        for (int i = 0; i < nv; i++) dHdth_(i) = g[i];
    }
    const vector_t& GetdHdth() const { return dHdth_; }
    void OptimizeContactTimes(double time, double /*actual_red_cost*/) {                                                      // gait_optimizer.cpp:185-364
        MPCSingleRigidBody* m = Owner();
        check_srbm(srbm_gait_optimize_contact_times(m->Gait(), &time));
        int st = 0; double pred = 0;
        check_srbm(srbm_gait_get_lp_result(m->Gait(), &st, &pred));
        if (st == 2) throw std::runtime_error("Bad gait optimization solve.");                                                // gait_optimizer.cpp:311-314
        pred_red_cost_ = pred;
        double xk[SRBM_GAIT_NV], step[SRBM_GAIT_NV]; int counts[4];
        check_srbm(srbm_gait_get_contact_times(m->Gait(), xk, counts));
        check_srbm(srbm_gait_get_step(m->Gait(), step));
        step_.assign(step, step + SRBM_GAIT_NV);
        xk_.assign(xk, xk + SRBM_GAIT_NV);
        contact_times_ = GetContactTimes(1.0);
    }
    std::vector<time_v>& GetContactTimes() { return contact_times_; }
    std::vector<time_v> GetContactTimes(double alpha) const {                                                                 // gait_optimizer.cpp:645-669
        std::vector<time_v> c = contact_times_;
        int off = 0;
        for (int ee = 0; ee < num_ee_; ee++) {
            for (size_t i = 0; i < c[ee].size(); i++) {
                double t = xk_.empty() ? c[ee][i].GetTime() : xk_[off + i] + alpha * step_[off + i];
                if (i > 0) { const double d = c[ee][i - 1].GetTime() - t; if (d <= 1e-3 && d > 0) t = c[ee][i - 1].GetTime(); }
                c[ee][i].SetTime(t);
            }
            off += (int)c[ee].size();
        }
        return c;
    }
    double GetPredictedReduction() const { return pred_red_cost_; }
    std::pair<std::vector<time_v>, double> LineSearch(MPCSingleRigidBody& mpc, double time, const std::vector<vector_3t>& ee_locations,
                                                      const vector_t& state) {                                               // gait_optimizer.cpp:671-753
        if (ee_locations.size() != 4 || state.size() != 13) throw std::runtime_error("LineSearch: 4 end effector locations and a 13-entry state expected.");
        double ee[12];
        for (int e = 0; e < 4; e++) for (int c = 0; c < 3; c++) ee[3 * e + c] = ee_locations[e](c);
        int imin = 0; double costs[SRBM_GAIT_LS_SIZE];
        check_srbm(srbm_gait_line_search(mpc.Gait(), state.data(), &time, ee, &imin, costs));
        return std::make_pair(GetContactTimes(static_cast<double>(imin) / SRBM_GAIT_LS_SIZE), costs[imin]);
    }
private:
    MPCSingleRigidBody* Owner() const {
        if (!qp_partials_.owner) throw std::runtime_error("GetQPPartials must be called (with this optimizer's partials) before the gradient.");
        return qp_partials_.owner;
    }
    int num_ee_, num_decision_vars_, num_constraints_;
    std::vector<time_v> contact_times_;
    std::vector<int> num_times_;
    QPPartialsDense qp_partials_;
    std::vector<std::vector<QPPartials>> param_partials_;      // [ee][contact time]
    vector_t dHdth_;
    std::vector<double> step_, xk_;
    double pred_red_cost_ = 0;
};

}  // namespace mpc
