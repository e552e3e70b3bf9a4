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
// Eigen: the real <Eigen/Core> when it exists, else mpc_facade/eigen_shim.h (this container has none).
#pragma once
#include <array>
#include <cmath>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define MPC_FACADE_REAL_EIGEN 1
#endif
#endif
#ifndef MPC_FACADE_REAL_EIGEN
#include "eigen_shim.h"
#endif

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
private:
    static void Lookup(int rc, double time) {
        if (rc != 0) throw std::runtime_error("Trajectory: invalid time " + std::to_string(time) + " for the spline lookup.");   // end_effector_splines.cpp:1066-1083
    }
    srbm_trajectory rec_;
};

class MPCSingleRigidBody;

// mpc/include/qp/qp_partials.h:15-57.  The reference fills these with 260x372 / 752x372 matrices on the host and contracts them in
// GaitOptimizer::ComputeCostFcnDerivWrtContactTimes; here the contraction is fused on the device (srbm_gait_compute_gradient) and
// the objects only carry which MPC they belong to.
struct QPPartials { const MPCSingleRigidBody* owner = nullptr; int ee = -1, idx = -1; void SetZero() {} };
struct QPPartialsDense { MPCSingleRigidBody* owner = nullptr; bool modified = false; void SetZero() { modified = false; } };

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

// mpc/include/mpc.h:70-170 + mpc/include/mpc_single_rigid_body.h:11-76
class MPCSingleRigidBody {
public:
    MPCSingleRigidBody(const MPCInfo& info, const std::string& robot_urdf)
        : MPCSingleRigidBody(info, ModelConstantsFromUrdf(robot_urdf, ToStd(info.nom_state))) {}
    // the same object from constants computed elsewhere (e.g. by pinocchio in a build that has it)
    MPCSingleRigidBody(const MPCInfo& info, const srbm_model& model) : info_(info), model_consts_(model), model_(model.mass) {
        srbm_mpc_info ci{};
        ci.num_nodes = info.num_nodes; ci.integrator_dt = info.integrator_dt; ci.friction_coef = info.friction_coef;
        ci.force_bound = info.force_bound; ci.swing_height = info.swing_height; ci.foot_offset = info.foot_offset;
        ci.ee_box_size[0] = info.ee_box_size(0); ci.ee_box_size[1] = info.ee_box_size(1); ci.force_cost = info.force_cost;
        check_srbm(srbm_batch_create(&h_, 1, &ci, &model_consts_, 0));
    }
    // value semantics (mpc.cpp:1133-1181, mpc_single_rigid_body.cpp:804-807)
    MPCSingleRigidBody(const MPCSingleRigidBody& other) : info_(other.info_), model_consts_(other.model_consts_), model_(other.model_),
                                                          used_log_file_(other.used_log_file_), solves_(other.solves_) {
        check_srbm(srbm_batch_clone(other.h_, &h_));
    }
    MPCSingleRigidBody& operator=(const MPCSingleRigidBody& other) {
        if (this == &other) return *this;
        Release();
        info_ = other.info_; model_consts_ = other.model_consts_; model_ = other.model_; used_log_file_ = other.used_log_file_; solves_ = other.solves_;
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
        double ee[12];
        PackEE(ee_start_locations, ee);
        CheckState(state);
        check_srbm(srbm_create_initial_run(h_, state.data(), ee));
        solves_ += 10;
        return GetTrajectory();
    }
    Trajectory GetRealTimeUpdate(const vector_t& state, double init_time, const std::vector<vector_3t>& ee_start_locations, bool /*high_quality*/) {   // mpc.cpp:92-108
        return Solve(state, init_time, ee_start_locations);
    }
    Trajectory Solve(const vector_t& state, double init_time, const std::vector<vector_3t>& ee_start_locations) {   // msrb.cpp:25-216
        double ee[12];
        PackEE(ee_start_locations, ee);
        CheckState(state);
        check_srbm(srbm_get_real_time_update(h_, state.data(), &init_time, ee));
        solves_ += 1;
        int st = 0, err = 0;
        check_srbm(srbm_get_status(h_, &st, &err));
        if (err) throw std::runtime_error("MPC solve raised error bits " + std::to_string(err) + " (time outside the spline range / capacity).");
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
        std::vector<double> x((info_.num_nodes + 1) * 12 + 160);
        check_srbm(srbm_get_qp_solution(h_, x.data(), (int)x.size()));
        vector_t v(n);
        for (int i = 0; i < n; i++) v(i) = x[i];
        return v;
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
        return true;
    }
    bool ComputeParamPartialsClarabel(const Trajectory&, QPPartials& partials, int ee, int idx) { partials.owner = this; partials.ee = ee; partials.idx = idx; return true; }
    srbm_gait* Gait() { if (!gait_) check_srbm(srbm_gait_create(h_, &gait_)); return gait_; }
    srbm_batch* Handle() const { return h_; }
    const MPCInfo& Info() const { return info_; }

    // ---- statistics (mpc.cpp:818-899, 901-989) ----
    void PrintStats() { std::ofstream null; PrintLine(std::cout, true); }
    void PrintStatLineToFile(std::ofstream& log_file) {
        if (!used_log_file_) { PrintHeader(log_file); used_log_file_ = true; }
        PrintLine(log_file, false);
    }

private:
    static std::vector<double> ToStd(const vector_t& v) { std::vector<double> o(v.size()); for (int i = 0; i < (int)v.size(); i++) o[i] = v(i); return o; }
    static void RowMajor(const matrix_t& M, double* out) { for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) out[12 * i + j] = M(i, j); }
    static void PackEE(const std::vector<vector_3t>& ee, double* out) {
        if (ee.size() != 4) throw std::runtime_error("4 end effector locations expected.");
        for (int e = 0; e < 4; e++) for (int c = 0; c < 3; c++) out[3 * e + c] = ee[e](c);
    }
    static void CheckState(const vector_t& s) { if (s.size() != 13) throw std::runtime_error("The SRBM state has 13 entries."); }
    std::array<int, 8> Sizes() const { std::array<int, 8> s{}; check_srbm(srbm_get_sizes(h_, s.data())); return s; }
    void Release() {
        if (gait_) { srbm_gait_destroy(gait_); gait_ = nullptr; }
        if (h_) { srbm_batch_destroy(h_); h_ = nullptr; }
    }
    void PrintHeader(std::ostream& os) const {
        const int col_width = 15, table_width = 10 * col_width;
        using std::setw; using std::setfill;
        os << setfill('-') << setw(table_width) << "" << std::endl;
        os << std::left << setfill(' ') << setw(table_width / 2 - 7) << "" << "MPC Statistics" << std::endl;
        os << std::left << "Number of nodes: " << info_.num_nodes << std::endl;
        os << std::left << "MPC time step: " << info_.integrator_dt << std::endl;
        os << std::left << "Force bounds: " << info_.force_bound << std::endl;
        os << std::left << "End Effector box size: " << info_.ee_box_size(0) << " " << info_.ee_box_size(1) << std::endl;
        os << std::left << "Force cost: " << info_.force_cost << std::endl;
        os << std::left << "Foot offset: " << info_.foot_offset << std::endl;
        os << std::left << "Swing height: " << info_.swing_height << std::endl;
        os << setfill('-') << setw(table_width) << "" << std::endl;
        os << setfill(' ');
        for (const char* n : {"Solve #", "Time (ms)", "Constraints", "Step Norm", "Alpha", "Cost", "Merit", "Merit dd", "Solve Type", "QP Cost"}) os << setw(col_width) << n;
        os << std::endl << setfill('-') << setw(table_width) << "" << std::endl << setfill(' ');
    }
    void PrintLine(std::ostream& os, bool with_header) const {
        if (with_header) PrintHeader(os);
        double st[8], merit = 0, merit_dd = 0, qpc = 0;
        check_srbm(srbm_get_stats(h_, st));
        check_srbm(srbm_get_merit(h_, &merit, &merit_dd));
        check_srbm(srbm_get_qp_cost(h_, &qpc));
        static const char* names[] = {"Solved", "Solved Inacc", "Max Iter", "P - Infeasible", "D - Infeasible", "P - Infeasible Inacc", "D - Infeasible Inacc", "Unsolved", "Other"};
        const int q = (int)GetSolveQuality();
        const int col_width = 15;
        using std::setw;
        os << std::left << setw(col_width) << (solves_ - 1) << setw(col_width) << 0.0 << setw(col_width) << st[2] << setw(col_width) << st[3] << setw(col_width) << st[0]
           << setw(col_width) << st[1] << setw(col_width) << merit << setw(col_width) << merit_dd << setw(col_width) << names[q < 0 || q > 8 ? 8 : q]
           << setw(col_width) << st[1] << std::endl;
    }

    MPCInfo info_;
    srbm_model model_consts_;
    Model model_;
    srbm_batch* h_ = nullptr;
    srbm_gait* gait_ = nullptr;
    bool used_log_file_ = false;
    int solves_ = 0;
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
        if ((int)param_partials_.size() <= ee * 8 + idx) param_partials_.resize(ee * 8 + idx + 1);
        return param_partials_[ee * 8 + idx];
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
    std::vector<QPPartials> param_partials_;
    vector_t dHdth_;
    std::vector<double> step_, xk_;
    double pred_red_cost_ = 0;
};

}  // namespace mpc
