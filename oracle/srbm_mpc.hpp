// ORACLE (test infrastructure, NOT product code).
// CPU restatement of mpc::MPC + mpc::MPCSingleRigidBody: one SQP real-time iteration
// (/root/reference/mpc/mpc_single_rigid_body.cpp:25-216) with the QP assembled in the reference's exact
// row/column order (/root/reference/mpc/mpc.cpp, /root/reference/mpc/qp/qp_data.cpp,
// /root/reference/utils/sparse_matrix_builder.cpp), the IPM solve (clarabel_like.hpp), the L1-merit line search
// (mpc.cpp:730-788, rk_integrator.cpp:14-30) and the trajectory write-back (msrb.cpp:275-321).
#pragma once
#include <memory>
#include "clarabel_like.hpp"
#include "srbm_traj_model.hpp"

namespace orc {

enum Constraints {   // qp_data.h:17-27
    Dynamics, JointForwardKinematics, EndEffectorLocation, ForceBox, JointBox, FrictionCone, TDPosition, Raibert,
    EndEffectorStart
};

struct MPCInfo {   // mpc.h:39-62 (fields that affect the live SRBM path)
    int num_nodes = 20;
    double integrator_dt = 0.05;
    double friction_coef = 0.5;
    double force_bound = 150;
    double swing_height = 0.075;
    double foot_offset = 0.015;
    double ee_box_size[2] = {0.15, 0.15};
    double force_cost = 0.0;
};

// utils::SparseMatrixBuilder (sparse_matrix_builder.cpp:11-38): triplets, exact zeros skipped by SetMatrix
struct TripletBuilder {
    std::vector<Triplet> t;
    void Reserve() { t.clear(); }
    void SetDiagonalMatrix(double val, int r0, int c0, int n) {
        for (int i = 0; i < n; i++) t.push_back({r0 + i, c0 + i, val});
    }
    void SetMatrix(const double* M, int rows, int cols, int ld, int r0, int c0) {
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++)
                if (M[(size_t)i * ld + j] != 0) t.push_back({r0 + i, c0 + j, M[(size_t)i * ld + j]});
    }
    void SetRow(const std::vector<double>& v, double scale, int r, int c0) {
        for (int j = 0; j < (int)v.size(); j++) {
            const double x = scale * v[j];
            if (x != 0) t.push_back({r, c0 + j, x});
        }
    }
};

struct QPData {   // qp_data.h:49-127 (Clarabel path only)
    std::vector<Constraints> constraints;
    TripletBuilder constraint_mat, cost_mat;
    std::vector<double> ub;   // Clarabel "b"
    std::vector<double> dynamics_constants, friction_cone_ub, force_box_lb, force_box_ub, ee_location_lb,
        ee_location_ub, start_ee_constants, td_pos_constants, cost_linear;
    int num_dynamics_constraints = 0, num_decision_vars = 0, num_cone_constraints = 0, num_force_box_constraints = 0,
        num_ee_location_constraints = 0, num_start_ee_constraints = 0, num_td_pos_constraints = 0;
    int num_equality = 0, num_inequality = 0;

    int GetTotalNumConstraints() const {   // qp_data.cpp:61-97
        int nc = 0;
        for (auto c : constraints) switch (c) {
            case Dynamics: nc += num_dynamics_constraints; break;
            case EndEffectorLocation: nc += num_ee_location_constraints; break;
            case EndEffectorStart: nc += num_start_ee_constraints; break;
            case ForceBox: nc += num_force_box_constraints; break;
            case FrictionCone: nc += num_cone_constraints; break;
            case TDPosition: nc += num_td_pos_constraints; break;
            default: break;
        }
        return nc;
    }
    void InitQPMats() {   // qp_data.cpp:99-167
        constraint_mat.Reserve();
        cost_mat.Reserve();
        dynamics_constants.assign(num_dynamics_constraints, 0.0);
        ee_location_lb.assign(num_ee_location_constraints / 2, 0.0);
        ee_location_ub.assign(num_ee_location_constraints / 2, 0.0);
        start_ee_constants.assign(num_start_ee_constraints, 0.0);
        force_box_lb.assign(num_force_box_constraints / 2, 0.0);
        force_box_ub.assign(num_force_box_constraints / 2, 0.0);
        friction_cone_ub.assign(num_cone_constraints, 0.0);
        td_pos_constants.assign(num_td_pos_constraints, 0.0);
        cost_linear.assign(num_decision_vars, 0.0);
        ub.assign(GetTotalNumConstraints(), 0.0);
    }
    void ConstructVectors() {   // qp_data.cpp:200-289
        int idx = 0;
        num_inequality = 0; num_equality = 0;
        auto put = [&](const std::vector<double>& v, double sgn) { for (double x : v) ub.at(idx++) = sgn * x; };
        for (auto c : constraints) switch (c) {
            case Dynamics: put(dynamics_constants, 1); num_equality += num_dynamics_constraints; break;
            case EndEffectorLocation: put(ee_location_ub, 1); put(ee_location_lb, -1);
                num_inequality += num_ee_location_constraints; break;
            case EndEffectorStart: put(start_ee_constants, 1); num_equality += num_start_ee_constraints; break;
            case ForceBox: put(force_box_ub, 1); put(force_box_lb, -1); num_inequality += num_force_box_constraints; break;
            case FrictionCone: put(friction_cone_ub, 1); num_inequality += num_cone_constraints; break;
            case TDPosition: put(td_pos_constants, 1); num_equality += num_td_pos_constraints; break;
            default: break;
        }
    }
    std::vector<Cone> Cones() const {   // clarabel_interface.cpp:29-66
        std::vector<Cone> k;
        for (auto c : constraints) switch (c) {
            case Dynamics: k.push_back({0, num_dynamics_constraints}); break;
            case EndEffectorLocation: k.push_back({1, num_ee_location_constraints}); break;
            case ForceBox: k.push_back({1, num_force_box_constraints}); break;
            case FrictionCone: k.push_back({1, num_cone_constraints}); break;
            case TDPosition: k.push_back({0, num_td_pos_constraints}); break;
            case EndEffectorStart: k.push_back({0, num_start_ee_constraints}); break;
            default: break;
        }
        return k;
    }
};

struct SolveStats { double alpha = 0, cost = 0, eq_violation = 0, step_norm = 0; SolveQuality status = Unsolved; int qp_iters = 0; };

class MPCSingleRigidBody {
public:
    static constexpr int POS_VARS = 3;
    static constexpr int FB_PER_FORCE = 10;   // mpc.h:320
    static constexpr int EE_NODE_START = 4;   // mpc_single_rigid_body.h:69
    static constexpr int NS = 12;             // num_states_ (tangent)

    // mpc.cpp:38-76 + msrb.cpp:9-23.  default contact schedule mpc.cpp:566-608
    MPCSingleRigidBody(const MPCInfo& info, const SRBModel& model)
        : info_(info), model_(model),
          prev_traj_(info.num_nodes + 1, DefaultSwitchingTimes(4), info.integrator_dt, info.swing_height, info.foot_offset) {
        data_.constraints = {Dynamics, ForceBox, FrictionCone, EndEffectorLocation, TDPosition, EndEffectorStart};
        UpdateNumInputs();
        SetFrictionPyramid();
        Phi_.assign(144, 0.0); Phi_w_.assign(12, 0.0); Q_.assign(144, 0.0); w_.assign(12, 0.0);
        ee_bounds_[0] = info_.ee_box_size[0]; ee_bounds_[1] = info_.ee_box_size[1];
        SetInitQPSizes();
        data_.InitQPMats();
        prev_qp_sol_.assign(data_.num_decision_vars, 0.0);
        // Clarabel settings as set by the reference: clarabel_interface.cpp:18-27
        solver_.settings.tol_gap_rel = 1e-8; solver_.settings.tol_gap_abs = 1e-8; solver_.settings.tol_feas = 1e-10;
    }

    static std::vector<std::vector<double>> DefaultSwitchingTimes(int num_ee) {   // mpc.cpp:566-608
        return std::vector<std::vector<double>>(num_ee, {0, 0.3, 0.6, 0.9, 1.2});
    }

    // mpc.cpp:533-540, :137-151, :700-706
    void AddQuadraticTrackingCost(const double* state_des12, const double* Q144) {
        Q_.assign(Q144, Q144 + 144);
        for (int i = 0; i < 12; i++) {
            double a = 0;
            for (int j = 0; j < 12; j++) a += Q_[i * 12 + j] * state_des12[j];
            w_[i] = -1 * a;
        }
    }
    void SetQuadraticFinalCost(const double* Phi144) { Phi_.assign(Phi144, Phi144 + 144); }
    void SetLinearFinalCost(const double* w12) { Phi_w_.assign(w12, w12 + 12); }
    void SetStateTrajectoryWarmStart(const std::vector<Vec13>& states) {
        for (int node = 0; node < info_.num_nodes + 1; node++) prev_traj_.SetState(node, states.at(node));
    }
    void SetWarmStartTrajectory(const Trajectory& t) {   // mpc.cpp:110-119
        prev_traj_ = t;
        UpdateNumInputs();
        init_time_ = t.GetTime(0);
    }
    void UpdateContactTimes(std::vector<time_v>& ct) { prev_traj_.UpdateContactTimes(ct); }   // mpc.cpp:1085-1088
    void AdjustForCurrentContacts(double time, const std::vector<bool>& in_contact) {          // mpc.cpp:1195-1203
        const std::vector<bool> tc = prev_traj_.GetDesiredContacts(time);
        for (int ee = 0; ee < 4; ee++)
            if (in_contact.at(ee) && !tc.at(ee) && std::abs(prev_traj_.GetNextContactTime(ee, time) - time) < 7e-2)
                prev_traj_.SetEEInContact(ee, time);
    }

    // mpc.cpp:78-90
    const Trajectory& CreateInitialRun(const Vec13& state, const std::vector<Vec3>& ee) {
        in_real_time_ = false;
        solver_.settings.tol_gap_rel = 1e-15; solver_.settings.tol_gap_abs = 1e-15;   // clarabel_interface.cpp:165-168
        for (int it = 0; it < 10; it++) Solve(state, 0, ee);
        return prev_traj_;
    }
    // mpc.cpp:92-108
    const Trajectory& GetRealTimeUpdate(const Vec13& state, double init_time, const std::vector<Vec3>& ee) {
        if (!in_real_time_) {
            solver_.settings.tol_gap_rel = 1e-15; solver_.settings.tol_gap_abs = 1e-15;   // clarabel_interface.cpp:170-175
            in_real_time_ = true;
        }
        return Solve(state, init_time, ee);
    }

    // msrb.cpp:25-216
    const Trajectory& Solve(const Vec13& state, double init_time, const std::vector<Vec3>& ee_start_locations) {
        init_time_ = init_time;
        prev_traj_.SetInitTime(init_time);
        prev_traj_.AddPolys(info_.integrator_dt * info_.num_nodes + init_time);
        prev_traj_.RemoveUnusedPolys(init_time);
        UpdateNumInputs();
        UpdateQPSizes();
        data_.InitQPMats();
        prev_traj_.SetState(0, state);
        prev_qp_sol_ = ConvertTrajToQPVec(prev_traj_);

        AddForceCost(info_.force_cost);
        AddHessianApproxCost();
        AddGradientCost();
        AddFinalCost();
        AddDiagonalCost();

        constraint_idx_ = 0;
        for (auto c : data_.constraints) switch (c) {
            case Dynamics: AddDynamicsConstraints(prev_traj_.GetState(0)); break;
            case ForceBox: AddForceBoxConstraints(); break;
            case FrictionCone: AddFrictionConeConstraints(); break;
            case EndEffectorLocation: AddEELocationConstraints(); break;
            case TDPosition: AddTDPositionConstraints(); break;
            case EndEffectorStart: AddEEStartConstraints(ee_start_locations); break;
            default: throw std::runtime_error("No such constraint exists.");
        }
        data_.ConstructVectors();

        // ---- solve (clarabel_interface.cpp:68-155) ----
        last_ = solver_.Solve(data_.num_decision_vars, data_.GetTotalNumConstraints(), data_.cost_mat.t, data_.cost_linear,
                              data_.constraint_mat.t, data_.ub, data_.Cones());
        solve_quality_ = last_.status;
        std::vector<double> sol = last_.x;
        if (solve_quality_ == PrimalInfeasible) sol = prev_qp_sol_;   // msrb.cpp:115-120
        if (solve_quality_ != SolvedInacc && solve_quality_ != Solved && solve_quality_ != MaxIter) IncreaseEEBox();
        else DecreaseEEBox();

        std::vector<double> p(sol.size());
        for (size_t i = 0; i < sol.size(); i++) p[i] = sol[i] - prev_qp_sol_[i];
        double alpha = 1;
        const Vec13 s0 = prev_traj_.GetState(0);
        if (sol.size() == prev_qp_sol_.size()) alpha = LineSearch(p, s0);
        prev_dual_sol_ = last_.z;
        for (size_t i = 0; i < sol.size(); i++) prev_qp_sol_[i] = (alpha * p[i]) + prev_qp_sol_[i];
        prev_traj_ = ConvertQPSolToTrajectory(prev_qp_sol_);

        stats_.alpha = alpha;
        stats_.status = solve_quality_;
        stats_.qp_iters = last_.iterations;
        stats_.cost = GetCostValue(prev_qp_sol_);
        stats_.eq_violation = l1(GetEqualityConstraintValues(prev_traj_));
        double sn = 0;
        for (double v : p) sn += v * v;
        stats_.step_norm = std::sqrt(sn);
        run_num_++;
        return prev_traj_;
    }

    // ---- accessors used by tests / the C API ----
    const QPData& GetQPData() const { return data_; }
    const Trajectory& GetTrajectory() const { return prev_traj_; }
    const std::vector<double>& GetQPSolution() const { return prev_qp_sol_; }
    const std::vector<double>& GetDualSolution() const { return prev_dual_sol_; }
    const ClarabelResult& LastQP() const { return last_; }
    const SolveStats& Stats() const { return stats_; }
    double GetCost() const { return GetCostValue(prev_qp_sol_); }
    int GetNumDecisionVars() const { return data_.num_decision_vars; }
    int GetNumConstraints() const { return data_.GetTotalNumConstraints(); }
    SolveQuality GetSolveQuality() const { return solve_quality_; }
    const SRBModel& Model() const { return model_; }
    const MPCInfo& Info() const { return info_; }
    double InitTime() const { return init_time_; }
    ClarabelLike& Solver() { return solver_; }
    int GetForceSplineStartIdx() const { return NS * (1 + info_.num_nodes); }                      // msrb.cpp:267-269
    int GetPosSplineStartIdx() const { return GetForceSplineStartIdx() + prev_traj_.GetTotalForceSplineVars(); }
    double GetTime(int node) const { return node * info_.integrator_dt + init_time_; }           // mpc.cpp:779-781
    const double* FrictionPyramid() const { return &friction_pyramid_[0][0]; }
    double TdFraction() const { return td_fraction_; }

    // msrb.cpp:343-357
    std::vector<double> ConvertTrajToQPVec(const Trajectory& traj) const {
        std::vector<double> v(traj.GetTotalVariables(), 0.0);
        for (int i = 0; i < info_.num_nodes + 1; i++) {
            const Vec12 t = SRBModel::ManifoldToTangent(traj.GetState(i));
            for (int j = 0; j < 12; j++) v[i * 12 + j] = t[j];
        }
        const std::vector<double> sp = traj.SplinesAsVec();
        for (int j = 0; j < num_inputs_; j++) v[v.size() - num_inputs_ + j] = sp[j];
        return v;
    }

    // msrb.cpp:275-321
    Trajectory ConvertQPSolToTrajectory(const std::vector<double>& qp_sol) const {
        Trajectory traj(prev_traj_);
        int force_idx = GetForceSplineStartIdx();
        int pos_idx = GetPosSplineStartIdx();
        for (int ee = 0; ee < 4; ee++) {
            for (int coord = 0; coord < POS_VARS; coord++) {
                int vars = traj.GetTotalPolyVars(Trajectory::Force, ee, coord);
                traj.UpdateForceSpline(ee, coord, &qp_sol[force_idx], vars);
                force_idx += vars;
                if (coord < 2) {
                    vars = traj.GetTotalPolyVars(Trajectory::Position, ee, coord);
                    traj.UpdatePositionSpline(ee, coord, &qp_sol[pos_idx], vars);
                    pos_idx += vars;
                }
            }
        }
        for (int node = 0; node < info_.num_nodes + 1; node++) {
            Vec13 ms = SRBModel::TangentToManifold(&qp_sol[node * NS]);
            quat_first_order_normalize(&ms[6]);
            traj.SetState(node, ms);
        }
        return traj;
    }

    // mpc.cpp:759-761
    double GetCostValue(const std::vector<double>& x) const {
        std::vector<double> Px(x.size(), 0.0);
        for (auto& t : data_.cost_mat.t) Px[t.r] += t.v * x[t.c];
        double a = 0, b = 0;
        for (size_t i = 0; i < x.size(); i++) { a += x[i] * Px[i]; b += data_.cost_linear[i] * x[i]; }
        return 0.5 * a + b;
    }
    // mpc.cpp:764-776 (+ rk_integrator.cpp:14-30: forward Euler)
    std::vector<double> GetEqualityConstraintValues(const Trajectory& traj) const {
        std::vector<double> eq((size_t)info_.num_nodes * NS, 0.0);
        for (int node = 0; node < info_.num_nodes; node++) {
            const Vec12 xn = SRBModel::ManifoldToTangent(traj.GetState(node + 1));
            const Vec12 xk = SRBModel::ManifoldToTangent(traj.GetState(node));
            const Vec12 f = model_.CalcDynamics(xk.data(), traj, GetTime(node));
            for (int j = 0; j < NS; j++) eq[node * NS + j] = xn[j] - (xk[j] + info_.integrator_dt * f[j]);
        }
        return eq;
    }
    double GetMeritValue(const std::vector<double>& x, double mu) const {   // mpc.cpp:749-753
        const Trajectory t = ConvertQPSolToTrajectory(x);
        return mu * l1(GetEqualityConstraintValues(t)) + GetCostValue(x);
    }
    double GetMeritGradient(const std::vector<double>& x, const std::vector<double>& p, double mu) const {   // :783-788
        const Trajectory t = ConvertQPSolToTrajectory(x);
        std::vector<double> g(x.size(), 0.0);
        for (auto& tr : data_.cost_mat.t) g[tr.r] += tr.v * x[tr.c];
        double d = 0;
        for (size_t i = 0; i < x.size(); i++) d += (g[i] + data_.cost_linear[i]) * p[i];
        return d - mu * l1(GetEqualityConstraintValues(t));
    }
    // mpc.cpp:730-747
    double LineSearch(const std::vector<double>& direction, const Vec13& /*init_state*/) {
        double alpha = 1;
        const double merit = GetMeritValue(prev_qp_sol_, mu_);
        std::vector<double> tmp(direction.size());
        for (size_t i = 0; i < tmp.size(); i++) tmp[i] = alpha * direction[i] + prev_qp_sol_[i];
        double merit_step = GetMeritValue(tmp, mu_);
        const double merit_directional = GetMeritGradient(prev_qp_sol_, direction, mu_);
        int i = 0;
        while ((merit - merit_step) < -0.00001 * alpha * merit_directional && i < 10) {
            alpha *= 0.5;
            for (size_t k = 0; k < tmp.size(); k++) tmp[k] = (alpha * direction[k]) + prev_qp_sol_[k];
            merit_step = GetMeritValue(tmp, mu_);
            i++;
        }
        return alpha;
    }

    // mpc.cpp:1101-1127
    int GetNumForceBoxConstraints() const { return 2 * FB_PER_FORCE * CountStancePhases(); }
    int GetNumFricConeConstraints() const { return 4 * FB_PER_FORCE * CountStancePhases(); }
    // mpc.cpp:1205-1214
    int GetNumTDConstraints() const {
        int n = 0;
        for (int ee = 0; ee < 4; ee++)
            if (prev_traj_.GetNextContactTime(ee, init_time_) - init_time_ < td_fraction_ * prev_traj_.GetCurrentSwingTime(ee))
                n += 2;
        return n;
    }

protected:
    static double l1(const std::vector<double>& v) { double s = 0; for (double x : v) s += std::abs(x); return s; }
    int CountStancePhases() const {
        int n = 0;
        const std::vector<time_v> ct = prev_traj_.GetContactTimes();
        for (int ee = 0; ee < 4; ee++)
            for (int i = 0; i < (int)ct[ee].size() - 1; i++)
                if (ct[ee][i].type == TouchDown) n++;
        return n;
    }
    int UpdateNumInputs() {   // mpc.cpp:1076-1083
        num_inputs_ = prev_traj_.GetTotalPosSplineVars() + prev_traj_.GetTotalForceSplineVars();
        return num_inputs_;
    }
    void SetFrictionPyramid() {   // mpc.cpp:153-163
        const double mu = info_.friction_coef;
        const double fp[4][3] = {{1, 0, -mu}, {-1, -0.0, -mu}, {0, 1, -mu}, {-0.0, -1, -mu}};
        for (int i = 0; i < 4; i++) for (int j = 0; j < 3; j++) friction_pyramid_[i][j] = fp[i][j];
    }
    void SetInitQPSizes() {   // msrb.cpp:323-341
        data_.num_decision_vars = (info_.num_nodes + 1) * NS + num_inputs_;
        data_.num_dynamics_constraints = (info_.num_nodes + 1) * NS;
        data_.num_cone_constraints = GetNumFricConeConstraints();
        data_.num_force_box_constraints = GetNumForceBoxConstraints();
        data_.num_ee_location_constraints = 2 * (info_.num_nodes - (EE_NODE_START - 1)) * 2 * 4;
        data_.num_td_pos_constraints = GetNumTDConstraints();
        data_.num_start_ee_constraints = 2 * 4;
    }
    void UpdateQPSizes() {   // mpc.cpp:610-624
        UpdateNumInputs();
        data_.num_decision_vars = (info_.num_nodes + 1) * NS + num_inputs_;
        data_.num_force_box_constraints = GetNumForceBoxConstraints();
        data_.num_cone_constraints = GetNumFricConeConstraints();
        data_.num_td_pos_constraints = GetNumTDConstraints();
    }

    // ---- costs: mpc.cpp:791-802, :542-564, :1090-1095 ----
    void AddForceCost(double weight) {
        const int nf = prev_traj_.GetTotalForceSplineVars();
        Q_forces_diag_.assign(nf, weight);
    }
    void AddHessianApproxCost() {
        for (int node = 0; node < info_.num_nodes; node++) data_.cost_mat.SetMatrix(Q_.data(), 12, 12, 12, node * NS, node * NS);
        const int fs = GetForceSplineStartIdx();
        for (int i = 0; i < (int)Q_forces_diag_.size(); i++)
            if (Q_forces_diag_[i] != 0) data_.cost_mat.t.push_back({fs + i, fs + i, Q_forces_diag_[i]});
    }
    void AddGradientCost() {
        for (int node = 0; node < info_.num_nodes; node++)
            for (int j = 0; j < NS; j++) data_.cost_linear[node * NS + j] = w_[j];
    }
    void AddFinalCost() {
        data_.cost_mat.SetMatrix(Phi_.data(), 12, 12, 12, info_.num_nodes * NS, info_.num_nodes * NS);
        for (int j = 0; j < NS; j++) data_.cost_linear[info_.num_nodes * NS + j] = Phi_w_[j];
    }
    void AddDiagonalCost() { data_.cost_mat.SetDiagonalMatrix(1e-3, 0, 0, data_.num_decision_vars); }

    // ---- msrb.cpp:218-265 ----
    void AddDynamicsConstraints(const Vec13& state) {
        data_.constraint_mat.SetDiagonalMatrix(-1, 0, 0, NS);
        const Vec12 t0 = SRBModel::ManifoldToTangent(state);
        for (int j = 0; j < NS; j++) data_.dynamics_constants[j] = -t0[j];
        const double dt = info_.integrator_dt;
        std::vector<double> A, B;
        Vec12 C;
        for (int node = 0; node < info_.num_nodes; node++) {
            const double time = GetTime(node);
            model_.GetLinearDynamics(prev_traj_.GetState(node), prev_traj_, time, A, B, C);
            for (int r = 0; r < 12; r++)
                for (int c = 0; c < 12; c++) A[r * 12 + c] = (r == c ? 1.0 : 0.0) + dt * A[r * 12 + c];
            for (auto& v : B) v = dt * v;
            for (auto& v : C) v = dt * v;
            const int r0 = constraint_idx_ + (node + 1) * NS;
            data_.constraint_mat.SetMatrix(A.data(), 12, 12, 12, r0, node * NS);
            data_.constraint_mat.SetDiagonalMatrix(-1, r0, (node + 1) * NS, NS);
            data_.constraint_mat.SetMatrix(B.data(), 12, num_inputs_, num_inputs_, r0, GetForceSplineStartIdx());
            for (int j = 0; j < NS; j++) data_.dynamics_constants[r0 + j] = -C[j];
        }
        constraint_idx_ += (info_.num_nodes + 1) * NS;
    }

    // ---- mpc.cpp:352-414 ----
    void AddForceBoxConstraints() {
        const int force_idx = GetForceSplineStartIdx();
        int row_idx = 0;
        const int coord = 2;
        const std::vector<time_v> ct = prev_traj_.GetContactTimes();
        for (int j = 0; j < 2; j++)
            for (int ee = 0; ee < 4; ee++)
                for (int ti = 0; ti < (int)ct[ee].size() - 1; ti++)
                    if (ct[ee][ti].type == TouchDown)
                        for (int i = 0; i < FB_PER_FORCE; i++) {
                            const double lower_time = ct[ee][ti].time, upper_time = ct[ee][ti + 1].time;
                            const double time = (static_cast<double>(i) / static_cast<double>(FB_PER_FORCE)) *
                                                    (upper_time - lower_time) + lower_time;
                            if (!prev_traj_.IsForceMutable(ee, time)) throw std::runtime_error("Force is not mutable here.");
                            auto [vi, va] = prev_traj_.GetForceSplineIndex(ee, time, coord);
                            (void)va;
                            const std::vector<double> lin = prev_traj_.GetSplineLin(Trajectory::Force, ee, coord, time);
                            data_.constraint_mat.SetRow(lin, j == 0 ? 1.0 : -1.0, constraint_idx_ + row_idx, force_idx + vi);
                            if (j == 0) {
                                data_.force_box_lb[row_idx] = -0.0;
                                data_.force_box_ub[row_idx] = info_.force_bound;
                            }
                            row_idx++;
                        }
        assert(row_idx == data_.num_force_box_constraints);
        constraint_idx_ += row_idx;
    }
    // ---- mpc.cpp:166-208 ----
    void AddFrictionConeConstraints() {
        const int force_idx = GetForceSplineStartIdx();
        int row_idx = 0;
        const std::vector<time_v> ct = prev_traj_.GetContactTimes();
        for (int ee = 0; ee < 4; ee++)
            for (int ti = 0; ti < (int)ct[ee].size() - 1; ti++)
                if (ct[ee][ti].type == TouchDown)
                    for (int i = 0; i < FB_PER_FORCE; i++) {
                        for (int coord = 0; coord < POS_VARS; coord++) {
                            const double lower_time = ct[ee][ti].time, upper_time = ct[ee][ti + 1].time;
                            const double time = (static_cast<double>(i) / static_cast<double>(FB_PER_FORCE)) *
                                                    (upper_time - lower_time) + lower_time;
                            auto [vi, va] = prev_traj_.GetForceSplineIndex(ee, time, coord);
                            (void)va;
                            const std::vector<double> lin = prev_traj_.GetSplineLin(Trajectory::Force, ee, coord, time);
                            for (int fc = 0; fc < 4; fc++) {
                                data_.constraint_mat.SetRow(lin, friction_pyramid_[fc][coord], constraint_idx_ + row_idx + fc,
                                                            force_idx + vi);
                                data_.friction_cone_ub[row_idx + fc] = 0;
                            }
                        }
                        row_idx += 4;
                    }
        assert(row_idx == data_.num_cone_constraints);
        constraint_idx_ += row_idx;
    }
    // ---- msrb.cpp:381-443 (dense scratch matrix, "=" assignment semantics kept) ----
    void AddEELocationConstraints() {
        const int pos_start_idx = GetPosSplineStartIdx();
        const double bounds[2] = {info_.ee_box_size[0] / 2, info_.ee_box_size[1] / 2};
        const int rows = data_.num_ee_location_constraints, cols = data_.num_decision_vars;
        std::vector<double> A((size_t)rows * cols, 0.0);
        int idx = 0;
        for (int i = 0; i < 2; i++)
            for (int node = EE_NODE_START; node < info_.num_nodes + 1; node++)
                for (int ee = 0; ee < 4; ee++) {
                    if (i == 0) {
                        const Vec3 hip = model_.GetCOMToHip(ee);
                        for (int c = 0; c < 2; c++) {
                            data_.ee_location_ub[idx + c] = bounds[c] + hip[c];
                            data_.ee_location_lb[idx + c] = -bounds[c] + hip[c];
                        }
                    }
                    for (int coord = 0; coord < 2; coord++) {
                        A[(size_t)idx * cols + node * NS + coord] = (i == 0) ? -1 : 1;
                        auto [vi, va] = prev_traj_.GetPositionSplineIndex(ee, GetTime(node), coord);
                        const std::vector<double> lin = prev_traj_.GetSplineLin(Trajectory::Position, ee, coord, GetTime(node));
                        for (int p = 0; p < va; p++) A[(size_t)idx * cols + pos_start_idx + vi + p] = (i == 0) ? lin.at(p) : -lin.at(p);
                        idx++;
                    }
                }
        data_.constraint_mat.SetMatrix(A.data(), rows, cols, cols, constraint_idx_, 0);
        assert(idx == data_.num_ee_location_constraints);
        constraint_idx_ += idx;
    }
    // ---- msrb.cpp:849-887 ----
    void AddTDPositionConstraints() {
        const int start_pos_idx = GetPosSplineStartIdx();
        int row_idx = 0;
        for (int ee = 0; ee < 4; ee++) {
            if (prev_traj_.GetNextContactTime(ee, init_time_) - init_time_ < td_fraction_ * prev_traj_.GetCurrentSwingTime(ee)) {
                const double td_time = prev_traj_.GetNextContactTime(ee, init_time_);
                const Vec3 loc = prev_traj_.GetEndEffectorLocation(ee, td_time);
                data_.td_pos_constants[row_idx] = loc[0];
                data_.td_pos_constants[row_idx + 1] = loc[1];
                for (int coord = 0; coord < 2; coord++) {
                    auto [vi, va] = prev_traj_.GetPositionSplineIndex(ee, td_time, coord);
                    (void)va;
                    const std::vector<double> lin = prev_traj_.GetSplineLin(Trajectory::Position, ee, coord, td_time);
                    data_.constraint_mat.SetRow(lin, 1.0, constraint_idx_ + row_idx, start_pos_idx + vi);
                    row_idx++;
                }
            }
        }
        assert(row_idx == data_.num_td_pos_constraints);
        constraint_idx_ += row_idx;
    }
    // ---- msrb.cpp:445-475 ----
    void AddEEStartConstraints(const std::vector<Vec3>& ee_start) {
        int idx = 0;
        const int pv = prev_traj_.GetTotalPosSplineVars();
        std::vector<double> M((size_t)data_.num_start_ee_constraints * pv, 0.0);
        for (int ee = 0; ee < 4; ee++) {
            data_.start_ee_constants[idx] = ee_start.at(ee)[0];
            data_.start_ee_constants[idx + 1] = ee_start.at(ee)[1];
            for (int coord = 0; coord < 2; coord++) {
                auto [vi, va] = prev_traj_.GetPositionSplineIndex(ee, GetTime(0), coord);
                const std::vector<double> lin = prev_traj_.GetSplineLin(Trajectory::Position, ee, coord, GetTime(0));
                for (int p = 0; p < va; p++) M[(size_t)idx * pv + vi + p] = lin.at(p);
                idx++;
            }
        }
        data_.constraint_mat.SetMatrix(M.data(), data_.num_start_ee_constraints, pv, pv, constraint_idx_, GetPosSplineStartIdx());
        constraint_idx_ += idx;
    }
    void IncreaseEEBox() { info_.ee_box_size[0] += 0.05; info_.ee_box_size[1] += 0.05; }   // msrb.cpp:929-937
    void DecreaseEEBox() {
        info_.ee_box_size[0] = std::max(info_.ee_box_size[0] - 0.05, ee_bounds_[0]);
        info_.ee_box_size[1] = std::max(info_.ee_box_size[1] - 0.05, ee_bounds_[1]);
    }

    MPCInfo info_;
    SRBModel model_;
    Trajectory prev_traj_;
    QPData data_;
    int num_inputs_ = 0;
    double friction_pyramid_[4][3];
    std::vector<double> Q_, w_, Phi_, Phi_w_, Q_forces_diag_;
    std::vector<double> prev_qp_sol_, prev_dual_sol_;
    double init_time_ = 0;
    double mu_ = 5000;          // mpc.cpp:65
    double td_fraction_ = 0.75; // mpc.cpp:73
    int run_num_ = 0, constraint_idx_ = 0;
    bool in_real_time_ = false;
    double ee_bounds_[2];
    ClarabelLike solver_;
    ClarabelResult last_;
    SolveQuality solve_quality_ = Unsolved;
    SolveStats stats_;
};

}  // namespace orc
