// C++ host side above the C-ABI of libsrbm_rti.so: the classes a maintainer of the reference links against.
//
// mpc::MPCSingleRigidBody  (/root/reference/mpc/include/mpc_single_rigid_body.h:11-76, mpc/include/mpc.h:70-170) and
// mpc::GaitOptimizer       (/root/reference/mpc/include/gait_optimizer.h:20-170)
// with the reference's method names and argument meaning, for a BATCH of independent instances (batch = 1 is the
// reference object).  Header-only, C++17, no Eigen: vectors are std::vector<double> / std::array in the reference's
// element order; errors that the reference throws as std::runtime_error are thrown as std::runtime_error carrying
// srbm_last_error().  Nothing here computes: every method forwards to one C-ABI entry (include/srbm_rti.h).
#pragma once
#include <array>
#include <stdexcept>
#include <string>
#include <vector>

#include "srbm_rti.h"

namespace srbm {

// mpc::SolveQuality, mpc/include/qp/qp_interface.h:12-22
enum SolveQuality { Solved = 0, SolvedInacc = 1, MaxIter = 2, PrimalInfeasible = 3, DualInfeasible = 4,
                    PrimalInfeasibleInacc = 5, DualInfeasibleInacc = 6, Unsolved = 7, Other = 8 };

using vector_t = std::vector<double>;

inline void check(int rc) {
    if (rc != 0) throw std::runtime_error(std::string("srbm: ") + srbm_last_error());
}

// mpc::MPCInfo (mpc/include/mpc.h:39-62) -- the fields of the live single-rigid-body path
struct MPCInfo {
    int num_nodes = 20;
    double integrator_dt = 0.05, friction_coef = 0.5, force_bound = 150, swing_height = 0.1, foot_offset = 0.0;
    std::array<double, 2> ee_box_size{0.15, 0.15};
    double force_cost = 0.0;
};
// what the reference reads from pinocchio at construction (mpc/models/model.cpp:27, single_rigid_body_model.cpp:33-37,258-308)
struct ModelConstants {
    double mass = 0;
    std::array<double, 9> Ir{};          // row-major 3x3
    std::array<double, 8> hip_xy{};      // FL FR RL RR
};

class MPCSingleRigidBody {
public:
    // MPCSingleRigidBody::MPCSingleRigidBody(info, robot_urdf) x batch (mpc_single_rigid_body.cpp:9-23)
    MPCSingleRigidBody(const MPCInfo& info, const ModelConstants& model, int batch = 1, int device = 0)
        : info_(info), batch_(batch) {
        srbm_mpc_info ci{info.num_nodes, info.integrator_dt, info.friction_coef, info.force_bound, info.swing_height,
                         info.foot_offset, {info.ee_box_size[0], info.ee_box_size[1]}, info.force_cost};
        srbm_model cm{};
        cm.mass = model.mass;
        for (int i = 0; i < 9; i++) cm.Ir[i] = model.Ir[i];
        for (int i = 0; i < 8; i++) cm.hip_xy[i] = model.hip_xy[i];
        check(srbm_batch_create(&h_, batch, &ci, &cm, device));
    }
    ~MPCSingleRigidBody() { srbm_batch_destroy(h_); }
    MPCSingleRigidBody(const MPCSingleRigidBody&) = delete;
    MPCSingleRigidBody& operator=(const MPCSingleRigidBody&) = delete;

    int batch() const { return batch_; }
    srbm_batch* handle() const { return h_; }

    // MPC::AddQuadraticTrackingCost (mpc.cpp:533-540): state_des 12 (tangent), Q 12x12 row-major
    void AddQuadraticTrackingCost(const vector_t& state_des, const vector_t& Q) { need(state_des, 12); need(Q, 144); check(srbm_add_quadratic_tracking_cost(h_, state_des.data(), Q.data())); }
    // MPC::SetQuadraticFinalCost / SetLinearFinalCost (mpc.cpp:137-151)
    void SetQuadraticFinalCost(const vector_t& Phi) { need(Phi, 144); check(srbm_set_quadratic_final_cost(h_, Phi.data())); }
    void SetLinearFinalCost(const vector_t& w) { need(w, 12); check(srbm_set_linear_final_cost(h_, w.data())); }
    // ClarabelInterface::ConfigureForInitialRun / ConfigureForSolve tolerances (clarabel_interface.cpp:165-175)
    void SetSolverTolerances(double gap_abs, double gap_rel, double feas, int max_iter) { check(srbm_set_solver_tolerances(h_, gap_abs, gap_rel, feas, max_iter)); }
    void SetSolverStepRule(double tol_step, double start_mu) { check(srbm_set_solver_step_rule(h_, tol_step, start_mu)); }
    // MPC::SetStateTrajectoryWarmStart (mpc.cpp:700-706): states [batch][13]
    void SetStateTrajectoryWarmStart(const vector_t& states) { need(states, 13 * batch_); check(srbm_set_state_trajectory_warm_start(h_, states.data())); }
    // MPC::CreateInitialRun (mpc.cpp:78-90): state [batch][13], ee_start_locations [batch][4][3]
    void CreateInitialRun(const vector_t& state, const vector_t& ee) { need(state, 13 * batch_); need(ee, 12 * batch_); check(srbm_create_initial_run(h_, state.data(), ee.data())); }
    // MPC::GetRealTimeUpdate (mpc.cpp:92-108): one MPCSingleRigidBody::Solve per instance; init_time [batch]
    void GetRealTimeUpdate(const vector_t& state, const vector_t& init_time, const vector_t& ee) {
        need(state, 13 * batch_); need(init_time, batch_); need(ee, 12 * batch_);
        check(srbm_get_real_time_update(h_, state.data(), init_time.data(), ee.data()));
    }
    // the loop of test/gait_opt_playground.cpp:113-126, device resident
    void RtiAdvance(int first_index, int steps) { check(srbm_rti_advance(h_, first_index, steps)); }
    void Synchronize() { check(srbm_synchronize(h_)); }
    // closed-loop rollout harness: plant = CalcDynamics + RKIntegrator::CalcIntegral under the current trajectory (srbm_rti.h)
    void PlantSetState(const vector_t& state) { need(state, 13 * batch_); check(srbm_plant_set_state(h_, state.data())); }
    vector_t PlantGetState() const { vector_t s((size_t)batch_ * 13); check(srbm_plant_get_state(h_, s.data())); return s; }
    void PlantSetPush(const vector_t& time, const vector_t& impulse) { need(time, batch_); need(impulse, 6 * batch_); check(srbm_plant_set_push(h_, time.data(), impulse.data())); }
    void ClosedLoopAdvance(int first_index, int steps, int substeps, bool advance_time) { check(srbm_closed_loop_advance(h_, first_index, steps, substeps, advance_time ? 1 : 0)); }
    // MPC::UpdateContactTimes (mpc.cpp:1085-1088): times [batch][4][max_contacts]
    void UpdateContactTimes(const vector_t& times, int max_contacts) { need(times, 4 * max_contacts * batch_); check(srbm_update_contact_times(h_, times.data(), max_contacts)); }
    // MPC::AdjustForCurrentContacts (mpc.cpp:1195-1203): time [batch], in_contact [batch][4]
    void AdjustForCurrentContacts(const vector_t& time, const std::vector<int>& in_contact) {
        need(time, batch_);
        if ((int)in_contact.size() != 4 * batch_) throw std::runtime_error("srbm: in_contact must hold 4 flags per instance");
        check(srbm_adjust_for_current_contacts(h_, time.data(), in_contact.data()));
    }

    // MPC::GetQPSolution (mpc.cpp:1216-1218): [batch][ld], ld = (N+1)*12 + 160
    int SolutionStride() const { return (info_.num_nodes + 1) * 12 + 160; }
    vector_t GetQPSolution() const { vector_t x((size_t)batch_ * SolutionStride()); check(srbm_get_qp_solution(h_, x.data(), SolutionStride())); return x; }
    // MPC::GetNumDecisionVars / GetNumConstraints (mpc.cpp:1031-1040) and the other sizes: [batch][8]
    std::vector<int> GetSizes() const { std::vector<int> s((size_t)batch_ * 8); check(srbm_get_sizes(h_, s.data())); return s; }
    // MPC::GetSolveQuality (mpc.cpp:1059-1061) per instance, and the error bits (conditions on which the reference throws)
    std::vector<int> GetSolveQuality(std::vector<int>* err_bits = nullptr) const {
        std::vector<int> st(batch_), er(batch_);
        check(srbm_get_status(h_, st.data(), er.data()));
        if (err_bits) *err_bits = er;
        return st;
    }
    // alpha, MPC::GetCost, L1 dynamics defect, step norm, QP iterations, residuals, gap: [batch][8]
    vector_t GetStats() const { vector_t s((size_t)batch_ * 8); check(srbm_get_stats(h_, s.data())); return s; }
    // MPC::GetTrajectory().GetStates(): [batch][N+1][13]
    vector_t GetTrajectoryStates() const { vector_t s((size_t)batch_ * (info_.num_nodes + 1) * 13); check(srbm_get_trajectory_states(h_, s.data())); return s; }

private:
    static void need(const vector_t& v, size_t n) { if (v.size() != n) throw std::runtime_error("srbm: argument has the wrong size"); }
    MPCInfo info_;
    int batch_;
    srbm_batch* h_ = nullptr;
};

// mpc::GaitOptimizer for the instances of an MPCSingleRigidBody batch
class GaitOptimizer {
public:
    static constexpr int NV = SRBM_GAIT_NV, LS_SIZE = SRBM_GAIT_LS_SIZE;
    explicit GaitOptimizer(MPCSingleRigidBody& mpc) : mpc_(mpc) { check(srbm_gait_create(mpc.handle(), &g_)); }
    ~GaitOptimizer() { srbm_gait_destroy(g_); }
    GaitOptimizer(const GaitOptimizer&) = delete;
    GaitOptimizer& operator=(const GaitOptimizer&) = delete;

    // GaitOptimizer::SetContactTimes(mpc.GetTrajectory().GetContactTimes()) (gait_optimizer.cpp:395-408)
    void SetContactTimes() { check(srbm_gait_set_contact_times_from_trajectory(g_)); }
    vector_t GetContactTimes(std::vector<int>* counts = nullptr) const {
        vector_t x((size_t)mpc_.batch() * NV); std::vector<int> c((size_t)mpc_.batch() * 4);
        check(srbm_gait_get_contact_times(g_, x.data(), c.data()));
        if (counts) *counts = c;
        return x;
    }
    // ComputeDerivativeTerms + the parameter partials + ComputeCostFcnDerivWrtContactTimes (mpc_controller.cpp:518-561);
    // returns dHdth [batch][32]; valid[b] = 0 where the reference refuses (QP not Solved)
    vector_t ComputeCostFcnDerivWrtContactTimes(std::vector<int>* valid = nullptr) {
        check(srbm_gait_compute_gradient(g_));
        vector_t d((size_t)mpc_.batch() * NV); std::vector<int> v(mpc_.batch());
        check(srbm_gait_get_gradient(g_, d.data(), v.data()));
        if (valid) *valid = v;
        return d;
    }
    // GaitOptimizer::OptimizeContactTimes (gait_optimizer.cpp:185-364); returns the step [batch][32]
    vector_t OptimizeContactTimes(const vector_t& time) {
        if ((int)time.size() != mpc_.batch()) throw std::runtime_error("srbm: time must hold one value per instance");
        check(srbm_gait_optimize_contact_times(g_, time.data()));
        vector_t s((size_t)mpc_.batch() * NV);
        check(srbm_gait_get_step(g_, s.data()));
        return s;
    }
    // GaitOptimizer::LineSearch (gait_optimizer.cpp:671-753); returns the index of the installed candidate per instance
    std::vector<int> LineSearch(const vector_t& state, const vector_t& time, const vector_t& ee, vector_t* costs = nullptr) {
        std::vector<int> imin(mpc_.batch()); vector_t c((size_t)mpc_.batch() * LS_SIZE);
        check(srbm_gait_line_search(g_, state.data(), time.data(), ee.data(), imin.data(), c.data()));
        if (costs) *costs = c;
        return imin;
    }
    // the MPC loop of the controller with the gait step folded in (mpc_controller.cpp:320-346), device resident
    void RtiAdvance(int first_run_num, int steps, int gait_opt_freq) { check(srbm_gait_rti_advance(g_, first_run_num, steps, gait_opt_freq)); }

private:
    MPCSingleRigidBody& mpc_;
    srbm_gait* g_ = nullptr;
};

}  // namespace srbm
