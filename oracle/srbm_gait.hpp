// ORACLE (test infrastructure, NOT product code).
// CPU restatement of the bilevel sensitivity step:
//   * KKT sensitivity of the QP solution (as coded)   /root/reference/mpc/qp/clarabel_interface.cpp:182-260,262-612
//   * partials of the QP data w.r.t. one contact time /root/reference/mpc/mpc_single_rigid_body.cpp:642-792,889-927
//                                                     /root/reference/mpc/mpc.cpp:240-325,416-531
//   * dH/dtheta, the contact-time LP and the 10-way line search   /root/reference/mpc/gait_optimizer.cpp
// The LP is handed to OSQP in the reference (third-party, absent, unpinned: mpc/CMakeLists.txt:68-70); here it is
// solved by the same IPM restatement as the MPC QP (unique optimum => same answer; ties => parity unpinned).
#pragma once
#include "srbm_mpc.hpp"

namespace orc {

struct QPPartials {   // qp_partials.h:15-35 (sparse: triplets, duplicates summed)
    std::vector<Triplet> dA, dG;
    std::vector<double> dq, db, dh;
};

// Dense LU with partial pivoting, solves M x = rhs in place (M row-major n x n, destroyed)
inline void dense_lu_solve(int n, std::vector<double>& M, std::vector<double>& rhs) {
    std::vector<int> piv(n);
    for (int k = 0; k < n; k++) {
        int p = k;
        double best = std::abs(M[(size_t)k * n + k]);
        for (int i = k + 1; i < n; i++)
            if (std::abs(M[(size_t)i * n + k]) > best) { best = std::abs(M[(size_t)i * n + k]); p = i; }
        if (best == 0) throw std::runtime_error("Could not factor the differential matrix.");
        if (p != k) {
            for (int j = 0; j < n; j++) std::swap(M[(size_t)k * n + j], M[(size_t)p * n + j]);
            std::swap(rhs[k], rhs[p]);
        }
        const double inv = 1.0 / M[(size_t)k * n + k];
        for (int i = k + 1; i < n; i++) {
            const double f = M[(size_t)i * n + k] * inv;
            if (f == 0) continue;
            double* ri = &M[(size_t)i * n];
            const double* rk = &M[(size_t)k * n];
            for (int j = k + 1; j < n; j++) ri[j] -= f * rk[j];
            rhs[i] -= f * rhs[k];
        }
    }
    for (int i = n - 1; i >= 0; i--) {
        double a = rhs[i];
        for (int j = i + 1; j < n; j++) a -= M[(size_t)i * n + j] * rhs[j];
        rhs[i] = a / M[(size_t)i * n + i];
    }
}

// Solves the reference's differential system AS CODED (clarabel_interface.cpp:262-602):
//   [ P   G' diag(lam)   A' ] [dz  ]     [dl/dx]
//   [ G   diag(s)        0  ] [dlam] = - [  0  ]        (note: +diag(s), not diag(Gz-h))
//   [ A   0              0  ] [dnu ]     [  0  ]
// The diagonal (2,2) block is eliminated exactly (dlam = -(G dz)/s) and the remaining (n+n_eq) system is solved by
// dense LU with partial pivoting (the reference uses Eigen::SparseLU on the full matrix: same linear system).
// G, A dense row-major.  Output d = [dz; dlam; dnu].
inline std::vector<double> solve_differential(int n, int n_ineq, int n_eq, const std::vector<double>& Pd,
                                              const std::vector<double>& G, const std::vector<double>& A,
                                              const std::vector<double>& lam, const std::vector<double>& s,
                                              const std::vector<double>& dldx) {
    const int N = n + n_eq;
    std::vector<double> M((size_t)N * N, 0.0), rhs(N, 0.0);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) M[(size_t)i * N + j] = Pd[(size_t)i * n + j];
    for (int r = 0; r < n_ineq; r++) {
        const double w = lam[r] / s[r];
        const double* g = &G[(size_t)r * n];
        for (int i = 0; i < n; i++) {
            if (g[i] == 0) continue;
            const double gi = g[i] * w;
            for (int j = 0; j < n; j++) M[(size_t)i * N + j] -= gi * g[j];
        }
    }
    for (int r = 0; r < n_eq; r++)
        for (int j = 0; j < n; j++) {
            M[(size_t)(n + r) * N + j] = A[(size_t)r * n + j];
            M[(size_t)j * N + n + r] = A[(size_t)r * n + j];
        }
    for (int i = 0; i < n; i++) rhs[i] = -dldx[i];
    dense_lu_solve(N, M, rhs);
    std::vector<double> d(n + n_ineq + n_eq, 0.0);
    for (int i = 0; i < n; i++) d[i] = rhs[i];
    for (int r = 0; r < n_ineq; r++) {
        double gz = 0;
        for (int j = 0; j < n; j++) gz += G[(size_t)r * n + j] * rhs[j];
        d[n + r] = -gz / s[r];
    }
    for (int r = 0; r < n_eq; r++) d[n + n_ineq + r] = rhs[n + r];
    return d;
}

// generic-QP form used by the 3-variable fixture test (rows ordered [eq; ineq])
inline void DenseSensitivity(int n, int n_eq, int n_ineq, const double* Pdense, const double* Adense, const double* q,
                             const double* x, const double* z, const double* s, double* dA, double* dG, double* dq,
                             double* db, double* dh) {
    std::vector<double> P(Pdense, Pdense + (size_t)n * n), A(Adense, Adense + (size_t)n_eq * n),
        G(Adense + (size_t)n_eq * n, Adense + (size_t)(n_eq + n_ineq) * n);
    std::vector<double> lam(z + n_eq, z + n_eq + n_ineq), sl(s + n_eq, s + n_eq + n_ineq), nu(z, z + n_eq);
    std::vector<double> dldx(n);
    for (int i = 0; i < n; i++) {
        double a = q[i];
        for (int j = 0; j < n; j++) a += P[(size_t)i * n + j] * x[j];
        dldx[i] = a;
    }
    const std::vector<double> d = solve_differential(n, n_ineq, n_eq, P, G, A, lam, sl, dldx);
    const double* dz = &d[0];
    const double* dlam = &d[n];
    const double* dnu = &d[n + n_ineq];
    for (int i = 0; i < n; i++) dq[i] = dz[i];
    for (int r = 0; r < n_ineq; r++) dh[r] = lam[r] * -1 * dlam[r];
    for (int r = 0; r < n_eq; r++) db[r] = -dnu[r];
    for (int r = 0; r < n_eq; r++)
        for (int j = 0; j < n; j++) dA[(size_t)r * n + j] = dnu[r] * x[j] + nu[r] * dz[j];
    for (int r = 0; r < n_ineq; r++)
        for (int j = 0; j < n; j++) dG[(size_t)r * n + j] = lam[r] * dlam[r] * x[j] + lam[r] * dz[j];
}

class GaitOptimizer {
public:
    static constexpr int LS_SIZE = 10;   // gait_optimizer.h:164
    explicit GaitOptimizer(int num_ee) : num_ee_(num_ee) {}

    // mpc_controller.cpp:518-561 up to (and including) ComputeCostFcnDerivWrtContactTimes
    bool ComputeGradient(MPCSingleRigidBody& mpc) {
        if (mpc.GetSolveQuality() != Solved) return false;   // mpc.cpp:1048,1059
        const QPData& data = mpc.GetQPData();
        const int n = data.num_decision_vars, m = data.GetTotalNumConstraints();
        const int n_eq = data.num_equality, n_ineq = data.num_inequality;
        // dense P, rows split in the reference's KKT order (clarabel_interface.cpp:296-470)
        std::vector<double> Pd((size_t)n * n, 0.0), Afull((size_t)m * n, 0.0);
        for (auto& t : data.cost_mat.t) Pd[(size_t)t.r * n + t.c] += t.v;
        for (auto& t : data.constraint_mat.t) Afull[(size_t)t.r * n + t.c] += t.v;
        const std::vector<double>& dual = mpc.LastQP().z;
        const std::vector<double>& slack = mpc.LastQP().s;
        std::vector<double> G((size_t)n_ineq * n), A((size_t)n_eq * n), lam(n_ineq), sl(n_ineq), nu(n_eq);
        int gen = 0, ei = 0, ii = 0;
        auto take = [&](int cnt, bool eq) {
            for (int r = 0; r < cnt; r++) {
                const double* src = &Afull[(size_t)(gen + r) * n];
                if (eq) { std::copy(src, src + n, &A[(size_t)(ei + r) * n]); nu[ei + r] = dual[gen + r]; }
                else { std::copy(src, src + n, &G[(size_t)(ii + r) * n]); lam[ii + r] = dual[gen + r]; sl[ii + r] = slack[gen + r]; }
            }
            gen += cnt;
            if (eq) ei += cnt; else ii += cnt;
        };
        for (auto c : data.constraints) switch (c) {
            case Dynamics: take(data.num_dynamics_constraints, true); break;
            case ForceBox: take(data.num_force_box_constraints, false); break;
            case FrictionCone: take(data.num_cone_constraints, false); break;
            case EndEffectorLocation: take(data.num_ee_location_constraints, false); break;
            case TDPosition: take(data.num_td_pos_constraints, true); break;
            case EndEffectorStart: take(data.num_start_ee_constraints, true); break;
            default: break;
        }
        // dx = P x* + q with x* = prev_qp_sol (mpc.cpp:1049, clarabel_interface.cpp:604-612)
        const std::vector<double>& xs = mpc.GetQPSolution();
        std::vector<double> dldx(n);
        for (int i = 0; i < n; i++) {
            double a = data.cost_linear[i];
            for (int j = 0; j < n; j++) a += Pd[(size_t)i * n + j] * xs[j];
            dldx[i] = a;
        }
        d_ = solve_differential(n, n_ineq, n_eq, Pd, G, A, lam, sl, dldx);
        const double* dz = &d_[0];
        const double* dlam = &d_[n];
        const double* dnu = &d_[n + n_ineq];
        const std::vector<double>& primal = mpc.LastQP().x;   // clarabel_interface.cpp:151 (raw QP minimiser)

        // QP partials (clarabel_interface.cpp:182-260) + ModifyQPPartials (gait_optimizer.cpp:536-539)
        std::vector<double> dq(n), db(n_eq), dh(n_ineq);
        for (int i = 0; i < n; i++) dq[i] = dz[i] + xs[i];
        for (int r = 0; r < n_ineq; r++) dh[r] = lam[r] * -1 * dlam[r];
        for (int r = 0; r < n_eq; r++) db[r] = -dnu[r];

        // parameter partials per contact time and the contraction (gait_optimizer.cpp:92-179)
        const Trajectory traj = mpc.GetTrajectory();
        contact_times_ = traj.GetContactTimes();
        int total = 0;
        for (auto& tv : contact_times_) total += (int)tv.size();
        dHdth_.assign(total, 0.0);
        int off = 0;
        for (int ee = 0; ee < num_ee_; ee++) {
            for (int idx = 0; idx < (int)contact_times_[ee].size(); idx++) {
                QPPartials pp;
                ComputeParamPartials(mpc, traj, pp, ee, idx);
                double acc = 0;
                // duplicates in the triplet lists are summed by setFromTriplets before the cwiseProduct; the
                // contraction is linear in the parameter partial so summing term-by-term is identical
                for (auto& t : pp.dG)
                    acc += (lam[t.r] * dlam[t.r] * primal[t.c] + lam[t.r] * dz[t.c]) * t.v;
                for (auto& t : pp.dA) acc += (dnu[t.r] * primal[t.c] + nu[t.r] * dz[t.c]) * t.v;
                for (int i = 0; i < n; i++) acc += dq[i] * pp.dq[i];
                for (int r = 0; r < n_eq; r++) acc += db[r] * pp.db[r];
                for (int r = 0; r < n_ineq; r++) acc += dh[r] * pp.dh[r];
                dHdth_[off + idx] = acc;
            }
            off += (int)contact_times_[ee].size();
        }
        // SetContactTimes (gait_optimizer.cpp:395-408)
        xkp1_.clear();
        for (auto& tv : contact_times_) for (auto& t : tv) xkp1_.push_back(t.time);
        return true;
    }

    // mpc_single_rigid_body.cpp:642-792
    void ComputeParamPartials(const MPCSingleRigidBody& mpc, const Trajectory& traj, QPPartials& partials, int ee,
                              int contact_time_idx) const {
        const QPData& data = mpc.GetQPData();
        const MPCInfo& info = mpc.Info();
        const int NS = 12, num_ee = 4;
        const double dt = info.integrator_dt;
        const int n = data.num_decision_vars;
        partials.dq.assign(n, 0.0);
        partials.db.assign(data.num_equality, 0.0);
        partials.dh.assign(data.num_inequality, 0.0);
        partials.dA.clear(); partials.dG.clear();
        TripletBuilder Ab, Gb;
        const int fs = mpc.GetForceSplineStartIdx(), ps = mpc.GetPosSplineStartIdx();
        const int num_inputs = traj.GetTotalForceSplineVars() + traj.GetTotalPosSplineVars();
        int eq_idx = 0, ineq_idx = 0;
        for (auto c : data.constraints) {
            if (c == Dynamics) {
                std::vector<double> dA, dB;
                Vec12 dC;
                for (int node = 0; node < info.num_nodes; node++) {
                    mpc.Model().ComputeLinearizationPartialWrtContactTimes(dA, dB, dC, traj.GetState(node), traj,
                                                                          mpc.GetTime(node), ee, contact_time_idx);
                    for (auto& v : dA) v = dt * v;
                    for (auto& v : dB) v = dt * v;
                    for (auto& v : dC) v = dt * v;
                    Ab.SetMatrix(dA.data(), 12, 12, 12, eq_idx + (node + 1) * NS, node * NS);
                    Ab.SetMatrix(dB.data(), 12, num_inputs, num_inputs, eq_idx + (node + 1) * NS, fs);
                    for (int j = 0; j < NS; j++) partials.db[eq_idx + (node + 1) * NS + j] = -dC[j];
                }
                eq_idx += data.num_dynamics_constraints;
            } else if (c == EndEffectorLocation) {
                int idx = 2 * ee;
                for (int node = MPCSingleRigidBody::EE_NODE_START; node < info.num_nodes + 1; node++) {
                    const double time = mpc.GetTime(node);
                    for (int coord = 0; coord < 2; coord++) {
                        auto [vi, va] = traj.GetPositionSplineIndex(ee, time, coord);
                        (void)va;
                        const std::vector<double> pcp = traj.GetPositionCoefPartialsWrtContactTime(ee, coord, time, contact_time_idx);
                        Gb.SetRow(pcp, 1.0, ineq_idx + idx, ps + vi);
                        Gb.SetRow(pcp, -1.0, ineq_idx + idx + data.num_ee_location_constraints / 2, ps + vi);
                        idx++;
                    }
                    idx += 2 * (num_ee - 1);
                }
                // (as coded: ineq_idx is NOT advanced here, msrb.cpp:705-734)
            } else if (c == EndEffectorStart) {
                const int pv = traj.GetTotalPosSplineVars();
                std::vector<double> M((size_t)data.num_start_ee_constraints * pv, 0.0);
                int idx = 0;   // as coded: rows 0,1 whatever the foot (msrb.cpp:740)
                for (int coord = 0; coord < 2; coord++) {
                    auto [vi, va] = mpc.GetTrajectory().GetPositionSplineIndex(ee, mpc.GetTime(0), coord);
                    const std::vector<double> pcp = traj.GetPositionCoefPartialsWrtContactTime(ee, coord, mpc.GetTime(0), contact_time_idx);
                    for (int p = 0; p < va; p++) M[(size_t)idx * pv + vi + p] = pcp.at(p);
                    idx++;
                }
                Ab.SetMatrix(M.data(), data.num_start_ee_constraints, pv, pv, eq_idx, ps);
                eq_idx += data.num_start_ee_constraints;
                ineq_idx += data.num_ee_location_constraints;
            } else if (c == ForceBox) {
                AddForceBoxConstraintPartials(mpc, Gb, contact_time_idx, ineq_idx, ee);
                ineq_idx += data.num_force_box_constraints;
            } else if (c == FrictionCone) {
                AddFrictionConeConstraintPartials(mpc, Gb, contact_time_idx, ineq_idx, ee);
                ineq_idx += data.num_cone_constraints;
            } else if (c == TDPosition) {
                AddTDPositionConstraintPartial(mpc, Ab, partials.db, contact_time_idx, eq_idx, ee);
                eq_idx += data.num_td_pos_constraints;
            }
        }
        partials.dA = Ab.t;
        partials.dG = Gb.t;
    }

    // gait_optimizer.cpp:185-364 (P = 0 LP; trust region Delta = 1; BFGS and trust-region adaptation are commented out
    // in the reference)
    void OptimizeContactTimes(double time) {
        const int nv = (int)xkp1_.size();
        const int nc = nv + nv + 3 * num_ee_;
        lb_.assign(nc, 0.0); ub_.assign(nc, 0.0);
        A_.clear();
        int next_row = CreatePolytopeConstraint(0, time);
        next_row = CreateStartConstraint(next_row);
        next_row = CreateTrustRegionConstraint(next_row);
        CreateNextNodeConstraints(next_row, time);
        // hand the LP  min dHdth's  s.t. lb <= A s <= ub  to the IPM: equality rows -> zero cone, others -> two
        // one-sided nonnegative rows
        std::vector<Triplet> At;
        std::vector<double> b;
        std::vector<Cone> cones;
        std::vector<std::vector<std::pair<int, double>>> rows(nc);
        for (auto& t : A_) rows[t.r].push_back({t.c, t.v});
        int r = 0;
        std::vector<int> eqrows, inrows;
        for (int i = 0; i < nc; i++) (lb_[i] == ub_[i] ? eqrows : inrows).push_back(i);
        for (int i : eqrows) { for (auto& e : rows[i]) At.push_back({r, e.first, e.second}); b.push_back(ub_[i]); r++; }
        cones.push_back({0, (int)eqrows.size()});
        for (int i : inrows) { for (auto& e : rows[i]) At.push_back({r, e.first, e.second}); b.push_back(ub_[i]); r++; }
        for (int i : inrows) { for (auto& e : rows[i]) At.push_back({r, e.first, -e.second}); b.push_back(-lb_[i]); r++; }
        cones.push_back({1, 2 * (int)inrows.size()});
        ClarabelLike lp;
        lp.settings.tol_gap_abs = 1e-10; lp.settings.tol_gap_rel = 1e-10; lp.settings.tol_feas = 1e-10;
        ClarabelResult res = lp.Solve(nv, r, {}, dHdth_, At, b, cones);
        if (res.status != Solved && res.status != SolvedInacc && res.status != MaxIter)
            throw std::runtime_error("Bad gait optimization solve");
        lp_status_ = res.status;
        step_ = res.x;
        xk_ = xkp1_;
        xkp1_.resize(nv);
        for (int i = 0; i < nv; i++) xkp1_[i] = xk_[i] + step_[i];
        contact_times_ = ConvertQPVecToContactTimes(xkp1_);
        pred_red_cost_ = 0;
        for (int i = 0; i < nv; i++) pred_red_cost_ -= dHdth_[i] * step_[i];
    }

    // gait_optimizer.cpp:645-669
    std::vector<time_v> GetContactTimes(double alpha) const {
        std::vector<double> v(xk_.size());
        for (size_t i = 0; i < v.size(); i++) v[i] = xk_[i] + alpha * step_[i];
        return ConvertQPVecToContactTimes(v);
    }

    // gait_optimizer.cpp:671-753.  Returns argmin index (0 if every candidate was primal infeasible).
    // (quality_out: the solve quality of every candidate, for the tests; may be nullptr)
    int LineSearch(MPCSingleRigidBody& mpc, double time, const std::vector<Vec3>& ee, const Vec13& state, double* costs_out, int* quality_out = nullptr) {
        double costs[LS_SIZE];
        SolveQuality quality[LS_SIZE];
        std::vector<Trajectory> trajs(LS_SIZE, mpc.GetTrajectory());
        for (int i = 0; i < LS_SIZE; i++) {
            MPCSingleRigidBody mpc_ls = mpc;
            std::vector<time_v> ct = GetContactTimes(static_cast<double>(i) / LS_SIZE);
            mpc_ls.UpdateContactTimes(ct);
            mpc_ls.GetRealTimeUpdate(state, time, ee);
            costs[i] = mpc_ls.GetCost() / mpc_ls.GetNumDecisionVars();
            trajs[i] = mpc_ls.GetTrajectory();
            quality[i] = mpc_ls.GetSolveQuality();
            if (costs_out) costs_out[i] = costs[i];
            if (quality_out) quality_out[i] = (int)quality[i];
        }
        int imin = -1;
        double cmin = 1e10;
        for (int i = 0; i < LS_SIZE; i++)
            if (costs[i] < cmin && quality[i] != PrimalInfeasible) { cmin = costs[i]; imin = i; }
        if (imin == -1) imin = 0;
        mpc.SetWarmStartTrajectory(trajs[imin]);
        return imin;
    }

    const std::vector<double>& dHdth() const { return dHdth_; }
    const std::vector<double>& d() const { return d_; }
    const std::vector<double>& step() const { return step_; }
    const std::vector<time_v>& ContactTimes() const { return contact_times_; }
    double PredictedReduction() const { return pred_red_cost_; }

private:
    int NumTimeNodes(int ee) const {
        int n = 0;
        for (int i = 0; i < ee; i++) n += (int)contact_times_[i].size();
        return n;
    }
    // gait_optimizer.cpp:410-464
    int CreatePolytopeConstraint(int start_row, double time) {
        constexpr double MIN_TIME = 0.2;
        int end_row = start_row;
        for (int ee = 0; ee < num_ee_; ee++) {
            const auto& ct = contact_times_[ee];
            const int nodes = (int)ct.size();
            const int base = start_row + NumTimeNodes(ee);
            int next_node = -1;
            for (int j = 1; j < nodes; j++)
                if (ct[j].time >= time) { next_node = j; break; }
            if (ct.at(next_node).type == TouchDown) {
                ub_[base + next_node - 1] = ct[next_node].time - ct[next_node - 1].time;
                lb_[base + next_node - 1] = -3;
            }
            for (int i = 1; i < nodes; i++) {
                A_.push_back({base + i - 1, NumTimeNodes(ee) + i - 1, 1.0});
                A_.push_back({base + i - 1, NumTimeNodes(ee) + i, -1.0});
                if (i != next_node || ct[next_node].type != TouchDown) {
                    ub_[base + i - 1] = ct[i].time - ct[i - 1].time - MIN_TIME;
                    lb_[base + i - 1] = -2;
                }
            }
            A_.push_back({base + nodes - 1, NumTimeNodes(ee) + nodes - 1, 1.0});
            lb_[base + nodes - 1] = 0;
            ub_[base + nodes - 1] = 1;
            end_row += nodes;
        }
        return end_row;
    }
    int CreateStartConstraint(int start_row) {   // :491-499
        for (int ee = 0; ee < num_ee_; ee++) {
            lb_[start_row + ee] = 0; ub_[start_row + ee] = 0;
            A_.push_back({start_row + ee, NumTimeNodes(ee), 1.0});
        }
        return start_row + num_ee_;
    }
    int CreateTrustRegionConstraint(int start_row) {   // :501-509
        const int nv = NumTimeNodes(num_ee_);
        for (int i = 0; i < nv; i++) {
            lb_[start_row + i] = -Delta_; ub_[start_row + i] = Delta_;
            A_.push_back({start_row + i, i, 1.0});
        }
        return start_row + nv;
    }
    int CreateNextNodeConstraints(int start_row, double time) {   // :511-534
        int idx = 0;
        for (int ee = 0; ee < num_ee_; ee++) {
            const auto& ct = contact_times_[ee];
            int next_node = -1;
            for (int i = 1; i < (int)ct.size(); i++)
                if (ct[i].time >= time) { next_node = i; break; }
            if (ct.at(next_node).type == TouchDown) {
                A_.push_back({start_row + idx, NumTimeNodes(ee) + next_node - 1, 1.0});
                A_.push_back({start_row + idx + 1, NumTimeNodes(ee) + next_node, 1.0});
                idx += 2;   // bounds stay 0 (lb = ub = 0)
            }
        }
        return start_row + idx;
    }
    // :651-669
    std::vector<time_v> ConvertQPVecToContactTimes(const std::vector<double>& vec) const {
        std::vector<time_v> contacts = contact_times_;
        for (int ee = 0; ee < num_ee_; ee++)
            for (int idx = 0; idx < (int)contacts[ee].size(); idx++) {
                contacts[ee][idx].time = vec[NumTimeNodes(ee) + idx];
                if (idx > 0) {
                    const double diff = contacts[ee][idx - 1].time - contacts[ee][idx].time;
                    if (diff <= 1e-3 && diff > 0) contacts[ee][idx] = contacts[ee][idx - 1];
                }
            }
        return contacts;
    }

    // mpc.cpp:416-531
    static void AddForceBoxConstraintPartials(const MPCSingleRigidBody& mpc, TripletBuilder& builder, int contact_idx,
                                              int start_idx, int ee) {
        const int FB = MPCSingleRigidBody::FB_PER_FORCE;
        const Trajectory& traj = mpc.GetTrajectory();
        const int force_idx = mpc.GetForceSplineStartIdx();
        const std::vector<time_v> ct = traj.GetContactTimes();
        int row_idx = 0;
        for (int i = 0; i < ee; i++)
            for (int c = 0; c < (int)ct[i].size() - 1; c++)
                if (ct[i][c].type == TouchDown) row_idx += FB;
        for (int c = 0; c < contact_idx; c++)
            if (ct[ee][c].type == TouchDown) row_idx += FB;
        if (ct[ee][contact_idx].type == LiftOff && contact_idx > 0) row_idx -= FB;
        const int coord = 2;
        for (int j = 0; j < 2; j++) {
            const bool td = ct[ee][contact_idx].type == TouchDown && contact_idx < (int)ct[ee].size() - 1;
            const bool lo = ct[ee][contact_idx].type == LiftOff && contact_idx > 0;
            if (td || lo) {
                const double lower_time = td ? ct[ee][contact_idx].time : ct[ee][contact_idx - 1].time;
                const double upper_time = td ? ct[ee][contact_idx + 1].time : ct[ee][contact_idx].time;
                for (int i = 0; i < FB; i++) {
                    const double frac = static_cast<double>(i) / static_cast<double>(FB);
                    const double time = frac * (upper_time - lower_time) + lower_time;
                    const double dtimedth = td ? (-frac + 1.0) : frac;
                    auto [vi, va] = traj.GetForceSplineIndex(ee, time, coord);
                    (void)va;
                    const std::vector<double> vp = traj.GetForceCoefPartialsWrtContactTime(ee, coord, time, contact_idx, dtimedth);
                    builder.SetRow(vp, j == 0 ? 1.0 : -1.0, start_idx + row_idx, force_idx + vi);
                    row_idx++;
                }
            }
            row_idx += mpc.GetQPData().num_force_box_constraints / 2 - FB;
        }
    }
    // mpc.cpp:240-325
    static void AddFrictionConeConstraintPartials(const MPCSingleRigidBody& mpc, TripletBuilder& builder, int contact_idx,
                                                  int start_idx, int ee) {
        const int FB = MPCSingleRigidBody::FB_PER_FORCE;
        const Trajectory& traj = mpc.GetTrajectory();
        const int force_idx = mpc.GetForceSplineStartIdx();
        const double* fp = mpc.FrictionPyramid();
        const std::vector<time_v> ct = traj.GetContactTimes();
        int row_idx = 0;
        for (int i = 0; i < ee; i++)
            for (int c = 0; c < (int)ct[i].size() - 1; c++)
                if (ct[i][c].type == TouchDown) row_idx += 4 * FB;
        for (int c = 0; c < contact_idx; c++)
            if (ct[ee][c].type == TouchDown) row_idx += 4 * FB;
        if (ct[ee][contact_idx].type == LiftOff && contact_idx > 0) row_idx -= 4 * FB;
        const bool td = ct[ee][contact_idx].type == TouchDown && contact_idx < (int)ct[ee].size() - 1;
        const bool lo = !td && contact_idx > 0 && ct[ee][contact_idx].type == LiftOff;
        if (!(td || lo)) return;
        const double lower_time = td ? ct[ee][contact_idx].time : ct[ee][contact_idx - 1].time;
        const double upper_time = td ? ct[ee][contact_idx + 1].time : ct[ee][contact_idx].time;
        for (int i = 0; i < FB; i++) {
            for (int coord = 0; coord < 3; coord++) {
                const double frac = static_cast<double>(i) / static_cast<double>(FB);
                const double time = frac * (upper_time - lower_time) + lower_time;
                const double dtimedth = td ? (-frac + 1.0) : frac;
                auto [vi, va] = traj.GetForceSplineIndex(ee, time, coord);
                (void)va;
                const std::vector<double> vp = traj.GetForceCoefPartialsWrtContactTime(ee, coord, time, contact_idx, dtimedth);
                for (int fc = 0; fc < 4; fc++) builder.SetRow(vp, fp[fc * 3 + coord], start_idx + row_idx + fc, force_idx + vi);
            }
            row_idx += 4;
        }
    }
    // msrb.cpp:889-927 (note the /2 where the constraint itself uses td_fraction_ = 0.75: as coded)
    static void AddTDPositionConstraintPartial(const MPCSingleRigidBody& mpc, TripletBuilder& builder, std::vector<double>& b,
                                               int contact_idx, int eq_idx, int ee) {
        const Trajectory& traj = mpc.GetTrajectory();
        const double t0 = mpc.InitTime();
        const int start_pos_idx = mpc.GetPosSplineStartIdx();
        int row_idx = 0;
        for (int i = 0; i < ee; i++)
            if (traj.GetNextContactTime(i, t0) - t0 < mpc.TdFraction() * traj.GetCurrentSwingTime(i)) row_idx += 2;
        if (traj.GetNextContactTime(ee, t0) - t0 < traj.GetCurrentSwingTime(ee) / 2) {
            const double td_time = traj.GetNextContactTime(ee, t0);
            const Vec3 pp = traj.GetPositionPartialWrtContactTime(ee, td_time, contact_idx);
            b.at(eq_idx + row_idx) = pp[0];
            b.at(eq_idx + row_idx + 1) = pp[1];
            for (int coord = 0; coord < 2; coord++) {
                auto [vi, va] = traj.GetPositionSplineIndex(ee, td_time, coord);
                (void)va;
                const std::vector<double> vl = traj.GetPositionCoefPartialsWrtContactTime(ee, coord, td_time, contact_idx);
                builder.SetRow(vl, 1.0, eq_idx + row_idx, start_pos_idx + vi);
                row_idx++;
            }
        }
    }

    int num_ee_;
    std::vector<time_v> contact_times_;
    std::vector<double> dHdth_, d_, step_, xk_, xkp1_, lb_, ub_;
    std::vector<Triplet> A_;
    double Delta_ = 1;   // gait_optimizer.cpp:43
    double pred_red_cost_ = 0;
    SolveQuality lp_status_ = Unsolved;
};

}  // namespace orc
