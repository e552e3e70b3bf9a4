// ORACLE (test infrastructure, NOT product code).  C entry points over the CPU restatement so that tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive it through ctypes.
#include <chrono>
#include <cstring>
#include "srbm_gait.hpp"

using namespace orc;

extern "C" {

struct orc_config {
    int num_nodes;
    double dt, friction_coef, force_bound, swing_height, foot_offset, box_x, box_y, force_cost;
    double mass;
    double Ir[9];
    double hip_xy[8];
    double Q_diag[12];
    double des_state[13];
};

struct OrcMPC {
    std::unique_ptr<MPCSingleRigidBody> mpc;
    std::unique_ptr<GaitOptimizer> gait;
    std::string err;
};

static std::vector<Vec3> ee_from(const double* ee12) {
    std::vector<Vec3> v(4);
    for (int e = 0; e < 4; e++) v[e] = {ee12[3 * e], ee12[3 * e + 1], ee12[3 * e + 2]};
    return v;
}
static Vec13 st_from(const double* s) { Vec13 v; std::memcpy(v.data(), s, 13 * sizeof(double)); return v; }

void* orc_mpc_create(const orc_config* c) {
    MPCInfo info;
    info.num_nodes = c->num_nodes; info.integrator_dt = c->dt; info.friction_coef = c->friction_coef;
    info.force_bound = c->force_bound; info.swing_height = c->swing_height; info.foot_offset = c->foot_offset;
    info.ee_box_size[0] = c->box_x; info.ee_box_size[1] = c->box_y; info.force_cost = c->force_cost;
    SRBModel model;
    model.mass = c->mass;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) model.Ir.m[i][j] = c->Ir[3 * i + j];
    model.Ir_inv = model.Ir.inverse();
    for (int e = 0; e < 4; e++) { model.hip_xy[e][0] = c->hip_xy[2 * e]; model.hip_xy[e][1] = c->hip_xy[2 * e + 1]; }
    auto* h = new OrcMPC;
    h->mpc = std::make_unique<MPCSingleRigidBody>(info, model);
    // cost set-up as the caller does it: /root/reference/controllers/mpc_controller.cpp:57-67
    double Q[144] = {0};
    for (int i = 0; i < 12; i++) Q[13 * i] = c->Q_diag[i];
    const Vec12 des = SRBModel::ManifoldToTangent(st_from(c->des_state));
    h->mpc->AddQuadraticTrackingCost(des.data(), Q);
    h->mpc->SetQuadraticFinalCost(Q);
    double w[12];
    for (int i = 0; i < 12; i++) w[i] = -1 * Q[13 * i] * des[i];
    h->mpc->SetLinearFinalCost(w);
    h->gait = std::make_unique<GaitOptimizer>(4);
    return h;
}
void orc_mpc_destroy(void* p) { delete static_cast<OrcMPC*>(p); }
void* orc_mpc_clone(void* p) {
    auto* src = static_cast<OrcMPC*>(p);
    auto* h = new OrcMPC;
    h->mpc = std::make_unique<MPCSingleRigidBody>(*src->mpc);
    h->gait = std::make_unique<GaitOptimizer>(4);
    return h;
}
const char* orc_mpc_error(void* p) { return static_cast<OrcMPC*>(p)->err.c_str(); }

void orc_mpc_set_warmstart(void* p, const double* state13) {
    auto* h = static_cast<OrcMPC*>(p);
    std::vector<Vec13> st(h->mpc->Info().num_nodes + 1, st_from(state13));
    h->mpc->SetStateTrajectoryWarmStart(st);
}
// returns SolveQuality of the last solve, or -1 on exception (message via orc_mpc_error)
int orc_mpc_initial_run(void* p, const double* state13, const double* ee12) {
    auto* h = static_cast<OrcMPC*>(p);
    try { h->mpc->CreateInitialRun(st_from(state13), ee_from(ee12)); }
    catch (const std::exception& e) { h->err = e.what(); return -1; }
    return (int)h->mpc->GetSolveQuality();
}
int orc_mpc_solve(void* p, const double* state13, double t, const double* ee12) {   // raw Solve (no tolerance switch)
    auto* h = static_cast<OrcMPC*>(p);
    try { h->mpc->Solve(st_from(state13), t, ee_from(ee12)); }
    catch (const std::exception& e) { h->err = e.what(); return -1; }
    return (int)h->mpc->GetSolveQuality();
}
int orc_mpc_rti(void* p, const double* state13, double t, const double* ee12) {
    auto* h = static_cast<OrcMPC*>(p);
    try { h->mpc->GetRealTimeUpdate(st_from(state13), t, ee_from(ee12)); }
    catch (const std::exception& e) { h->err = e.what(); return -1; }
    return (int)h->mpc->GetSolveQuality();
}
void orc_mpc_set_max_iter(void* p, int it) { static_cast<OrcMPC*>(p)->mpc->Solver().settings.max_iter = it; }

// sizes[0..7] = n, m, n_eq, n_ineq, n_force_vars, n_pos_vars, num_force_box, num_cone, [8]=num_ee_loc, [9]=num_td, [10]=num_start
void orc_mpc_sizes(void* p, int* sizes) {
    auto* h = static_cast<OrcMPC*>(p);
    const QPData& d = h->mpc->GetQPData();
    sizes[0] = d.num_decision_vars; sizes[1] = d.GetTotalNumConstraints(); sizes[2] = d.num_equality; sizes[3] = d.num_inequality;
    sizes[4] = h->mpc->GetTrajectory().GetTotalForceSplineVars(); sizes[5] = h->mpc->GetTrajectory().GetTotalPosSplineVars();
    sizes[6] = d.num_force_box_constraints; sizes[7] = d.num_cone_constraints; sizes[8] = d.num_ee_location_constraints;
    sizes[9] = d.num_td_pos_constraints; sizes[10] = d.num_start_ee_constraints;
}
void orc_mpc_get_x(void* p, double* x) {
    auto& v = static_cast<OrcMPC*>(p)->mpc->GetQPSolution();
    std::memcpy(x, v.data(), v.size() * sizeof(double));
}
void orc_mpc_get_qp_x(void* p, double* x) {   // raw QP minimiser (before the line search)
    auto& v = static_cast<OrcMPC*>(p)->mpc->LastQP().x;
    std::memcpy(x, v.data(), v.size() * sizeof(double));
}
void orc_mpc_get_z(void* p, double* z) {
    auto& v = static_cast<OrcMPC*>(p)->mpc->GetDualSolution();
    std::memcpy(z, v.data(), v.size() * sizeof(double));
}
void orc_mpc_get_s(void* p, double* s) {
    auto& v = static_cast<OrcMPC*>(p)->mpc->LastQP().s;
    std::memcpy(s, v.data(), v.size() * sizeof(double));
}
void orc_mpc_get_states(void* p, double* states) {   // (N+1) x 13
    auto* h = static_cast<OrcMPC*>(p);
    const Trajectory& t = h->mpc->GetTrajectory();
    for (int k = 0; k < t.NumNodesPlus1(); k++) std::memcpy(states + 13 * k, t.GetState(k).data(), 13 * sizeof(double));
}
// stats[0..5] = alpha, cost, eq_violation(L1), step_norm, qp_iters, status ; [6..9] = res_primal,res_dual,gap_abs,gap_rel ; [10],[11]= ee box
void orc_mpc_get_stats(void* p, double* s) {
    auto* h = static_cast<OrcMPC*>(p);
    const SolveStats& st = h->mpc->Stats();
    s[0] = st.alpha; s[1] = st.cost; s[2] = st.eq_violation; s[3] = st.step_norm; s[4] = st.qp_iters; s[5] = (double)st.status;
    const ClarabelResult& r = h->mpc->LastQP();
    s[6] = r.res_primal; s[7] = r.res_dual; s[8] = r.gap_abs; s[9] = r.gap_rel;
    s[10] = h->mpc->Info().ee_box_size[0]; s[11] = h->mpc->Info().ee_box_size[1];
}
// dense export of the assembled QP of the LAST solve: A (m x n row-major, duplicates summed), b (m), P (n x n), q (n)
void orc_mpc_get_qp_dense(void* p, double* A, double* b, double* P, double* q) {
    auto* h = static_cast<OrcMPC*>(p);
    const QPData& d = h->mpc->GetQPData();
    const int n = d.num_decision_vars, m = d.GetTotalNumConstraints();
    if (A) { std::memset(A, 0, sizeof(double) * (size_t)m * n); for (auto& t : d.constraint_mat.t) A[(size_t)t.r * n + t.c] += t.v; }
    if (P) { std::memset(P, 0, sizeof(double) * (size_t)n * n); for (auto& t : d.cost_mat.t) P[(size_t)t.r * n + t.c] += t.v; }
    if (b) std::memcpy(b, d.ub.data(), m * sizeof(double));
    if (q) std::memcpy(q, d.cost_linear.data(), n * sizeof(double));
}
int orc_mpc_constraint_nnz(void* p) { return (int)static_cast<OrcMPC*>(p)->mpc->GetQPData().constraint_mat.t.size(); }

// knot tables of one foot: returns number of knots; arrays sized >= 64.
// ftype/ptype: [coord*64 + k]; fvals/pvals: [(coord*64 + k)*2 + {0,1}]
int orc_mpc_get_knots(void* p, int ee, double* times, int* ttypes, int* ftype, double* fvals, int* ptype, double* pvals) {
    auto* h = static_cast<OrcMPC*>(p);
    const EndEffectorSplines& s = h->mpc->GetTrajectory().EE(ee);
    const int K = s.GetNumNodes();
    for (int k = 0; k < K; k++) {
        times[k] = s.RawTimes()[k].time; ttypes[k] = (int)s.RawTimes()[k].type;
        for (int c = 0; c < 3; c++) {
            const auto& f = s.RawNodes(orc::Force, c)[k];
            const auto& q = s.RawNodes(orc::Position, c)[k];
            ftype[c * 64 + k] = (int)f.type; fvals[(c * 64 + k) * 2] = f.v0; fvals[(c * 64 + k) * 2 + 1] = f.v1;
            ptype[c * 64 + k] = (int)q.type; pvals[(c * 64 + k) * 2] = q.v0; pvals[(c * 64 + k) * 2 + 1] = q.v1;
        }
    }
    return K;
}
double orc_mpc_init_time(void* p) { return static_cast<OrcMPC*>(p)->mpc->InitTime(); }
// contact times of one foot (max 32): returns count
int orc_mpc_get_contact_times(void* p, int ee, double* times, int* types) {
    auto* h = static_cast<OrcMPC*>(p);
    const time_v ct = h->mpc->GetTrajectory().EE(ee).GetContactTimes();
    for (size_t i = 0; i < ct.size(); i++) { times[i] = ct[i].time; types[i] = (int)ct[i].type; }
    return (int)ct.size();
}
int orc_mpc_set_contact_times(void* p, int ee_count, const int* counts, const double* times) {
    auto* h = static_cast<OrcMPC*>(p);
    try {
        std::vector<time_v> ct = h->mpc->GetTrajectory().GetContactTimes();
        int o = 0;
        for (int ee = 0; ee < ee_count; ee++)
            for (int i = 0; i < counts[ee]; i++) ct.at(ee).at(i).time = times[o++];
        h->mpc->UpdateContactTimes(ct);
    } catch (const std::exception& e) { h->err = e.what(); return -1; }
    return 0;
}
// MPC::AdjustForCurrentContacts (mpc.cpp:1195-1203)
int orc_mpc_adjust_for_current_contacts(void* p, double time, const int* in_contact4) {
    auto* h = static_cast<OrcMPC*>(p);
    try {
        std::vector<bool> c(4);
        for (int i = 0; i < 4; i++) c[i] = in_contact4[i] != 0;
        h->mpc->AdjustForCurrentContacts(time, c);
    } catch (const std::exception& e) { h->err = e.what(); return -1; }
    return 0;
}
double orc_mpc_ee_value(void* p, int ee, int is_position, int coord, double t) {
    auto* h = static_cast<OrcMPC*>(p);
    return h->mpc->GetTrajectory().EE(ee).ValueAt(is_position ? orc::Position : orc::Force, coord, t);
}

// plant of the closed-loop harness: RKIntegrator::CalcIntegral (rk_integrator.cpp:14-30) under the current trajectory
int orc_mpc_plant_integrate(void* p, const double* state13, double time, double dt, int num_steps, int advance_time, double* out13) {
    auto* h = static_cast<OrcMPC*>(p);
    try {
        orc::Vec13 ic;
        for (int i = 0; i < 13; i++) ic[i] = state13[i];
        const orc::Vec13 r = h->mpc->Model().CalcIntegral(ic, h->mpc->GetTrajectory(), time, dt, num_steps, advance_time);
        for (int i = 0; i < 13; i++) out13[i] = r[i];
    } catch (const std::exception& e) { h->err = e.what(); return -1; }
    return 0;
}

// ---- bilevel sensitivity step (gait_optimizer.cpp / clarabel_interface.cpp:262-612) ----
// dHdth out: concatenated per foot (foot-major); returns number of entries, -1 if the last solve was not "Solved"
int orc_gait_gradient(void* p, double* dHdth) {
    auto* h = static_cast<OrcMPC*>(p);
    try {
        if (!h->gait->ComputeGradient(*h->mpc)) return -1;
        const auto& g = h->gait->dHdth();
        std::memcpy(dHdth, g.data(), g.size() * sizeof(double));
        return (int)g.size();
    } catch (const std::exception& e) { h->err = e.what(); return -2; }
}
// KKT sensitivity vector d = -K^-1 [dl/dx; 0; 0] (n + n_ineq + n_eq)
int orc_gait_get_d(void* p, double* d) {
    auto* h = static_cast<OrcMPC*>(p);
    const auto& v = h->gait->d();
    std::memcpy(d, v.data(), v.size() * sizeof(double));
    return (int)v.size();
}
// dense parameter partials of one contact time: dA (n_eq x n), dG (n_ineq x n), db (n_eq), dh (n_ineq)
// traj_src: handle whose CURRENT trajectory plays the role of the `traj` argument of ComputeParamPartialsClarabel
// (msrb.cpp:642); pass NULL to use p's own trajectory (what mpc_controller.cpp:523-546 does)
int orc_gait_param_partials(void* p, void* traj_src, int ee, int idx, double* dA, double* dG, double* db, double* dh) {
    auto* h = static_cast<OrcMPC*>(p);
    try {
        QPPartials pr;
        const Trajectory& tr = traj_src ? static_cast<OrcMPC*>(traj_src)->mpc->GetTrajectory() : h->mpc->GetTrajectory();
        h->gait->ComputeParamPartials(*h->mpc, tr, pr, ee, idx);
        const QPData& d = h->mpc->GetQPData();
        const int n = d.num_decision_vars;
        std::memset(dA, 0, sizeof(double) * (size_t)d.num_equality * n);
        std::memset(dG, 0, sizeof(double) * (size_t)d.num_inequality * n);
        for (auto& t : pr.dA) dA[(size_t)t.r * n + t.c] += t.v;
        for (auto& t : pr.dG) dG[(size_t)t.r * n + t.c] += t.v;
        std::memcpy(db, pr.db.data(), pr.db.size() * sizeof(double));
        std::memcpy(dh, pr.dh.data(), pr.dh.size() * sizeof(double));
    } catch (const std::exception& e) { h->err = e.what(); return -1; }
    return 0;
}
// gait LP: returns 0 on success; step (num contact times), new contact times written back into the optimizer
int orc_gait_optimize(void* p, double time, double* step, double* new_times) {
    auto* h = static_cast<OrcMPC*>(p);
    try {
        h->gait->OptimizeContactTimes(time);
        const auto& s = h->gait->step();
        std::memcpy(step, s.data(), s.size() * sizeof(double));
        int o = 0;
        for (auto& tv : h->gait->ContactTimes()) for (auto& t : tv) new_times[o++] = t.time;
    } catch (const std::exception& e) { h->err = e.what(); return -1; }
    return 0;
}
// 10-candidate line search (gait_optimizer.cpp:671-753): returns argmin index, installs the winning trajectory
int orc_gait_line_search(void* p, const double* state13, double t, const double* ee12, double* costs10) {
    auto* h = static_cast<OrcMPC*>(p);
    try { return h->gait->LineSearch(*h->mpc, t, ee_from(ee12), st_from(state13), costs10); }
    catch (const std::exception& e) { h->err = e.what(); return -1; }
}
// the same with the solve quality of every candidate (mpc::SolveQuality values)
int orc_gait_line_search_q(void* p, const double* state13, double t, const double* ee12, double* costs10, int* quality10) {
    auto* h = static_cast<OrcMPC*>(p);
    try { return h->gait->LineSearch(*h->mpc, t, ee_from(ee12), st_from(state13), costs10, quality10); }
    catch (const std::exception& e) { h->err = e.what(); return -1; }
}

// ---- generic QP entry (used to pin the solver restatement on the reference's 3-variable fixture) ----
int orc_qp_solve(int n, int m, int nnzP, const int* Pr, const int* Pc, const double* Pv, const double* q, int nnzA,
                 const int* Ar, const int* Ac, const double* Av, const double* b, int ncones, const int* cone_nonneg,
                 const int* cone_dim, double tol_gap, double tol_feas, double* x, double* z, double* s, int* iters) {
    ClarabelLike solver;
    solver.settings.tol_gap_abs = tol_gap; solver.settings.tol_gap_rel = tol_gap; solver.settings.tol_feas = tol_feas;
    std::vector<Triplet> P, A;
    for (int i = 0; i < nnzP; i++) P.push_back({Pr[i], Pc[i], Pv[i]});
    for (int i = 0; i < nnzA; i++) A.push_back({Ar[i], Ac[i], Av[i]});
    std::vector<Cone> cones;
    for (int i = 0; i < ncones; i++) cones.push_back({cone_nonneg[i], cone_dim[i]});
    ClarabelResult r = solver.Solve(n, m, P, std::vector<double>(q, q + n), A, std::vector<double>(b, b + m), cones);
    std::memcpy(x, r.x.data(), n * sizeof(double));
    std::memcpy(z, r.z.data(), m * sizeof(double));
    std::memcpy(s, r.s.data(), m * sizeof(double));
    if (iters) *iters = r.iterations;
    return (int)r.status;
}
// KKT sensitivity on a generic QP with rows ordered [eq ; ineq(nonneg)] as the fixture has them
// (clarabel_interface.cpp:262-602 + :182-260).  dA: n_eq x n, dG: n_ineq x n.
void orc_qp_sensitivity(int n, int n_eq, int n_ineq, const double* Pdense, const double* Adense /* (n_eq+n_ineq) x n */,
                        const double* q, const double* x, const double* z, const double* s, double* dA, double* dG,
                        double* dq, double* db, double* dh) {
    DenseSensitivity(n, n_eq, n_ineq, Pdense, Adense, q, x, z, s, dA, dG, dq, db, dh);
}

// ---- spline object API (mirrors /root/reference/test/splines_tests.cpp usage) ----
void* orc_spline_create(int num_contacts, const double* times, int start_in_contact, int num_force_polys) {
    return new EndEffectorSplines(num_contacts, std::vector<double>(times, times + num_contacts), start_in_contact != 0,
                                  num_force_polys);
}
void* orc_spline_clone(void* s) { return new EndEffectorSplines(*static_cast<EndEffectorSplines*>(s)); }
void orc_spline_destroy(void* s) { delete static_cast<EndEffectorSplines*>(s); }
static SplineType ST(int t) { return t == 0 ? orc::Force : orc::Position; }
double orc_spline_value_at(void* s, int type, int coord, double t) {
    return static_cast<EndEffectorSplines*>(s)->ValueAt(ST(type), coord, t);
}
int orc_spline_lin(void* s, int type, int coord, double t, double* out) {
    try {
        auto v = static_cast<EndEffectorSplines*>(s)->GetPolyVarsLin(ST(type), coord, t);
        for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
        return (int)v.size();
    } catch (const std::exception&) { return -1; }
}
int orc_spline_vars_idx(void* s, int type, int coord, double t, int* idx) {
    try {
        auto pr = static_cast<EndEffectorSplines*>(s)->GetVarsIdx(ST(type), coord, t);
        *idx = pr.first;
        return pr.second;
    } catch (const std::exception&) { return -1; }
}
int orc_spline_is_force_mutable(void* s, double t) { return static_cast<EndEffectorSplines*>(s)->IsForceMutable(t) ? 1 : 0; }
void orc_spline_add_poly(void* s, double dt) { static_cast<EndEffectorSplines*>(s)->AddPoly(dt); }
int orc_spline_remove_poly(void* s, double t) {
    try { static_cast<EndEffectorSplines*>(s)->RemovePoly(t); } catch (const std::exception&) { return -1; }
    return 0;
}
int orc_spline_set_vars(void* s, int type, int coord, int node, double a, double b) {
    try { static_cast<EndEffectorSplines*>(s)->SetVars(ST(type), coord, node, a, b); } catch (const std::exception&) { return -1; }
    return 0;
}
int orc_spline_mutable_nodes(void* s, int type, int coord, int* out) {
    auto v = static_cast<EndEffectorSplines*>(s)->GetMutableNodes(ST(type), coord);
    for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
    return (int)v.size();
}
int orc_spline_times(void* s, double* out, int* types) {
    auto& v = static_cast<EndEffectorSplines*>(s)->RawTimes();
    for (size_t i = 0; i < v.size(); i++) { out[i] = v[i].time; if (types) types[i] = (int)v[i].type; }
    return (int)v.size();
}
int orc_spline_node_type(void* s, int type, int coord, int node) { return (int)static_cast<EndEffectorSplines*>(s)->GetNodeType(ST(type), coord, node); }
int orc_spline_qp_vec(void* s, int type, int coord, double* out) {
    auto v = static_cast<EndEffectorSplines*>(s)->GetSplineAsQPVec(ST(type), coord);
    for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
    return (int)v.size();
}
double orc_spline_end_time(void* s) { return static_cast<EndEffectorSplines*>(s)->GetEndTime(); }
double orc_spline_start_time(void* s) { return static_cast<EndEffectorSplines*>(s)->GetStartTime(); }
int orc_spline_contact_times(void* s, double* out) {
    auto v = static_cast<EndEffectorSplines*>(s)->GetContactTimes();
    for (size_t i = 0; i < v.size(); i++) out[i] = v[i].time;
    return (int)v.size();
}
int orc_spline_set_contact_times(void* s, int n, const double* t) {
    auto* sp = static_cast<EndEffectorSplines*>(s);
    time_v ct = sp->GetContactTimes();
    if ((int)ct.size() != n) return -1;
    for (int i = 0; i < n; i++) ct[i].time = t[i];
    try { sp->SetContactTimes(ct); } catch (const std::exception&) { return -2; }
    return 0;
}
double orc_spline_partial_wrt_time(void* s, int type, int coord, double t, int time_idx) {
    return static_cast<EndEffectorSplines*>(s)->ComputePartialWrtTime(ST(type), coord, t, time_idx);
}
int orc_spline_coef_partial_wrt_time(void* s, int type, int coord, double t, int time_idx, double dtwdth, double* out) {
    try {
        auto v = static_cast<EndEffectorSplines*>(s)->ComputeCoefPartialWrtTime(ST(type), coord, t, time_idx, dtwdth);
        for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
        return (int)v.size();
    } catch (const std::exception&) { return -1; }
}

// so(3) helpers for tests
void orc_quat_exp3(const double* v, double* q) { quat_exp3({v[0], v[1], v[2]}, q); }
void orc_quat_log3(const double* q, double* v) { Vec3 r = quat_log3(q); v[0] = r[0]; v[1] = r[1]; v[2] = r[2]; }

}  // extern "C"
