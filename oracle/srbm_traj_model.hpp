// ORACLE (test infrastructure, NOT product code).
// CPU restatement of mpc::Trajectory (/root/reference/mpc/trajectory.cpp) and
// mpc::SingleRigidBodyModel (/root/reference/mpc/models/single_rigid_body_model.cpp,
// constants from /root/reference/mpc/models/model.cpp:14-37).  pinocchio's log3/exp3/
// firstOrderNormalize (third-party, absent) are restated from their closed forms.
#pragma once
#include <algorithm>
#include <cstring>
#include "srbm_splines.hpp"

namespace orc {

using Vec3 = std::array<double, 3>;
using Vec13 = std::array<double, 13>;
using Vec12 = std::array<double, 12>;

inline Vec3 cross(const Vec3& a, const Vec3& b) {
    return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
}
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline Vec3 operator*(const Vec3& a, double s) { return {a[0] * s, a[1] * s, a[2] * s}; }
inline Vec3 operator-(const Vec3& a) { return {-a[0], -a[1], -a[2]}; }
inline Vec3 unit3(int i) { Vec3 e{0, 0, 0}; e[i] = 1; return e; }

struct Mat3 {
    double m[3][3];
    Vec3 col(int j) const { return {m[0][j], m[1][j], m[2][j]}; }
    Vec3 mul(const Vec3& v) const {
        return {m[0][0] * v[0] + m[0][1] * v[1] + m[0][2] * v[2], m[1][0] * v[0] + m[1][1] * v[1] + m[1][2] * v[2],
                m[2][0] * v[0] + m[2][1] * v[1] + m[2][2] * v[2]};
    }
    Mat3 inverse() const {
        const double a = m[0][0], b = m[0][1], c = m[0][2], d = m[1][0], e = m[1][1], f = m[1][2], g = m[2][0],
                     h = m[2][1], i = m[2][2];
        const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
        Mat3 r;
        r.m[0][0] = (e * i - f * h) / det; r.m[0][1] = (c * h - b * i) / det; r.m[0][2] = (b * f - c * e) / det;
        r.m[1][0] = (f * g - d * i) / det; r.m[1][1] = (a * i - c * g) / det; r.m[1][2] = (c * d - a * f) / det;
        r.m[2][0] = (d * h - e * g) / det; r.m[2][1] = (b * g - a * h) / det; r.m[2][2] = (a * e - b * d) / det;
        return r;
    }
};

// ---- so(3) maps (pinocchio::quaternion::{log3,exp3,firstOrderNormalize}; quaternion stored xyzw) ----
inline Vec3 quat_log3(const double q[4]) {
    // log of a unit quaternion: theta * axis, theta in (-pi, pi]
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    const double n = std::sqrt(n2);
    const double w = q[3];
    if (n < 1e-8) {
        // theta/sin(theta/2) ~ 2/w * (1 - n^2/(3 w^2))
        const double s = (2.0 / w) * (1.0 - n2 / (3.0 * w * w));
        return {s * q[0], s * q[1], s * q[2]};
    }
    double theta = (w >= 0) ? 2.0 * std::atan2(n, w) : -2.0 * std::atan2(n, -w);
    const double s = theta / n;
    return {s * q[0], s * q[1], s * q[2]};
}
inline void quat_exp3(const Vec3& v, double q[4]) {
    const double t2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const double t = std::sqrt(t2);
    if (t > 1.220703125e-4) {  // eps^(1/4)
        const double s = std::sin(t / 2) / t;
        q[0] = s * v[0]; q[1] = s * v[1]; q[2] = s * v[2]; q[3] = std::cos(t / 2);
    } else {
        const double s = 0.5 - t2 / 48;
        q[0] = s * v[0]; q[1] = s * v[1]; q[2] = s * v[2]; q[3] = 1.0 - t2 / 8;
    }
}
inline void quat_first_order_normalize(double q[4]) {
    const double N2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const double alpha = (3.0 - N2) / 2.0;
    for (int i = 0; i < 4; i++) q[i] *= alpha;
}

// ---- Trajectory (trajectory.cpp) ----
class Trajectory {
public:
    enum SplineTypes { Position, Force };
    static constexpr int POS_VARS = 3;

    // trajectory.cpp:11-48  (FR and RL, i.e. ee 1 and 2, start in contact; 3 force polys)
    Trajectory(int len, const std::vector<std::vector<double>>& switching_times, double node_dt, double swing_height,
               double foot_offset)
        : swing_height_(swing_height), foot_offset_(foot_offset), node_dt_(node_dt) {
        states_.assign(len, Vec13{});
        for (int i = 0; i < (int)switching_times.size(); i++) {
            const bool in_contact = (i == 1 || i == 2);
            init_time_ = 0;
            ee_splines_.emplace_back((int)switching_times[i].size(), switching_times[i], in_contact, 3);
        }
        UpdateSplineVarsCount();
        SetSwingPosZ();
    }

    int NumNodesPlus1() const { return (int)states_.size(); }
    void SetState(int idx, const Vec13& s) { states_.at(idx) = s; }
    const Vec13& GetState(int node) const { return states_.at(node); }
    int GetTotalPosSplineVars() const { return pos_spline_vars_; }
    int GetTotalForceSplineVars() const { return force_spline_vars_; }
    int NumEE() const { return (int)ee_splines_.size(); }
    const EndEffectorSplines& EE(int ee) const { return ee_splines_.at(ee); }
    EndEffectorSplines& EE(int ee) { return ee_splines_.at(ee); }

    // trajectory.cpp:85-111
    void UpdateForceSpline(int ee, int coord, const double* vars, int nvars) {
        int idx = 0;
        for (int node : ee_splines_.at(ee).GetMutableNodes(orc::Force, coord)) {
            ee_splines_.at(ee).SetVars(orc::Force, coord, node, vars[idx], vars[idx + 1]);
            idx += 2;
        }
        assert(idx == nvars); (void)nvars;
    }
    void UpdatePositionSpline(int ee, int coord, const double* vars, int nvars) {
        int idx = 0;
        for (int node : ee_splines_.at(ee).GetMutableNodes(orc::Position, coord)) {
            ee_splines_.at(ee).SetVars(orc::Position, coord, node, vars[idx], 0);
            idx++;
        }
        assert(idx == nvars); (void)nvars;
    }

    // trajectory.cpp:113-133 (quirk kept: the inner loop adds the `coord` count `coord` times)
    std::pair<int, int> GetPositionSplineIndex(int end_effector, double time, int coord) const {
        if (coord == 2) throw std::runtime_error("The chosen spline is not mutable and thus does not provide a index.");
        int before = 0;
        for (int ee = 0; ee < end_effector; ee++) before += 2 * ee_splines_.at(ee).GetTotalPolyVars(orc::Position, coord);
        int into = 0;
        for (int j = 0; j < coord; j++) into += ee_splines_.at(end_effector).GetTotalPolyVars(orc::Position, coord);
        auto [vi, va] = ee_splines_.at(end_effector).GetVarsIdx(orc::Position, coord, time);
        return {before + into + vi, va};
    }
    // trajectory.cpp:363-378
    std::pair<int, int> GetForceSplineIndex(int end_effector, double time, int coord) const {
        int before = 0;
        for (int ee = 0; ee < end_effector; ee++) before += 3 * ee_splines_.at(ee).GetTotalPolyVars(orc::Force, coord);
        int into = 0;
        for (int j = 0; j < coord; j++) into += ee_splines_.at(end_effector).GetTotalPolyVars(orc::Force, coord);
        auto [vi, va] = ee_splines_.at(end_effector).GetVarsIdx(orc::Force, coord, time);
        return {before + into + vi, va};
    }

    // trajectory.cpp:225-246
    void AddPolys(double final_time) {
        for (auto& ee_spline : ee_splines_) {
            while (ee_spline.GetEndTime() < final_time) {
                const time_v ct = ee_spline.GetContactTimes();
                const double last_diff = ct.at(ct.size() - 1).time - ct.at(ct.size() - 2).time;
                ee_spline.AddPoly(std::max(last_diff, 0.2));
            }
        }
        SetSwingPosZ();
        UpdateSplineVarsCount();
    }
    void RemoveUnusedPolys(double init_time) {
        for (auto& ee_spline : ee_splines_) ee_spline.RemovePoly(init_time);
        UpdateSplineVarsCount();
    }
    void SetInitTime(double t) { init_time_ = t; }

    // trajectory.cpp:286-301
    std::vector<bool> GetContacts(double time) const {
        std::vector<bool> c(ee_splines_.size());
        for (int ee = 0; ee < (int)ee_splines_.size(); ee++)
            c[ee] = ee_splines_[ee].GetVarsIdx(orc::Position, 0, time).second == 1;
        return c;
    }
    int GetNumContactNodes(int ee) const { return ee_splines_.at(ee).GetNumContacts(); }
    std::vector<time_v> GetContactTimes() const {
        std::vector<time_v> ct(ee_splines_.size());
        for (int ee = 0; ee < (int)ee_splines_.size(); ee++) ct[ee] = ee_splines_[ee].GetContactTimes();
        return ct;
    }

    // trajectory.cpp:348-361
    std::vector<double> GetSplineLin(SplineTypes st, int ee, int coord, double time) const {
        if (st == Force) return ee_splines_.at(ee).GetPolyVarsLin(orc::Force, coord, time);
        if (coord == 2) throw std::runtime_error("You cannot request spline linearizations for position z axis.");
        return ee_splines_.at(ee).GetPolyVarsLin(orc::Position, coord, time);
    }
    int GetTotalPolyVars(SplineTypes st, int ee, int coord) const {
        return ee_splines_.at(ee).GetTotalPolyVars(st == Force ? orc::Force : orc::Position, coord);
    }
    Vec3 GetForce(int ee, double time) const {
        Vec3 f;
        for (int c = 0; c < 3; c++) f[c] = ee_splines_.at(ee).ValueAt(orc::Force, c, time);
        return f;
    }
    Vec3 GetEndEffectorLocation(int ee, double time) const {
        Vec3 p;
        for (int c = 0; c < 3; c++) p[c] = ee_splines_.at(ee).ValueAt(orc::Position, c, time);
        return p;
    }
    double GetTime(int node) const { return init_time_ + node_dt_ * node; }                 // :417-419
    bool IsForceMutable(int ee, double time) const { return ee_splines_.at(ee).IsForceMutable(time); }
    int GetTotalVariables() const { return force_spline_vars_ + pos_spline_vars_ + (int)states_.size() * 12; }

    // trajectory.cpp:429-452
    std::vector<double> SplinesAsVec() const {
        std::vector<double> v(force_spline_vars_ + pos_spline_vars_, 0.0);
        int fi = 0, pi = force_spline_vars_;
        for (int ee = 0; ee < (int)ee_splines_.size(); ee++) {
            for (int coord = 0; coord < POS_VARS; coord++) {
                for (double x : ee_splines_[ee].GetSplineAsQPVec(orc::Force, coord)) v.at(fi++) = x;
                if (coord < 2)
                    for (double x : ee_splines_[ee].GetSplineAsQPVec(orc::Position, coord)) v.at(pi++) = x;
            }
        }
        assert(fi == force_spline_vars_);
        assert(pi == force_spline_vars_ + pos_spline_vars_);
        return v;
    }
    int GetNode(double time) const { return (int)std::ceil((time - init_time_) / node_dt_); }   // :479-481
    std::vector<bool> GetDesiredContacts(double time) const {                                   // :483-497
        std::vector<bool> c(ee_splines_.size());
        for (int ee = 0; ee < (int)ee_splines_.size(); ee++) c[ee] = ee_splines_[ee].IsInContact(time);
        return c;
    }
    // :499-536
    Vec3 GetForcePartialWrtContactTime(int ee, double time, int contact_idx) const {
        Vec3 r{0, 0, 0};
        for (int c = 0; c < 3; c++) r[c] = ee_splines_.at(ee).ComputePartialWrtTime(orc::Force, c, time, contact_idx);
        return r;
    }
    Vec3 GetPositionPartialWrtContactTime(int ee, double time, int contact_idx) const {
        Vec3 r{0, 0, 0};
        for (int c = 0; c < 2; c++) r[c] = ee_splines_.at(ee).ComputePartialWrtTime(orc::Position, c, time, contact_idx);
        return r;
    }
    std::vector<double> GetForceCoefPartialsWrtContactTime(int ee, int coord, double time, int contact_idx,
                                                           double dtwdth = 0) const {
        return ee_splines_.at(ee).ComputeCoefPartialWrtTime(orc::Force, coord, time, contact_idx, dtwdth);
    }
    std::vector<double> GetPositionCoefPartialsWrtContactTime(int ee, int coord, double time, int contact_idx) const {
        return ee_splines_.at(ee).ComputeCoefPartialWrtTime(orc::Position, coord, time, contact_idx);
    }
    void UpdateContactTimes(std::vector<time_v>& ct) {
        for (int ee = 0; ee < (int)ee_splines_.size(); ee++) ee_splines_[ee].SetContactTimes(ct.at(ee));
    }
    double GetNextContactTime(int ee, double time) const { return ee_splines_.at(ee).GetNextTouchDownTime(time); }
    void SetEEInContact(int ee, double time) { ee_splines_.at(ee).SetToTouchdown(time); }
    double GetCurrentSwingTime(int ee) const { return ee_splines_.at(ee).GetSwingTime(init_time_); }
    double InitTime() const { return init_time_; }

private:
    void UpdateSplineVarsCount() {                                        // :252-266
        pos_spline_vars_ = 0;
        force_spline_vars_ = 0;
        for (const auto& s : ee_splines_) {
            pos_spline_vars_ += 2 * s.GetTotalPolyVars(orc::Position, 0);
            force_spline_vars_ += 3 * s.GetTotalPolyVars(orc::Force, 0);
        }
    }
    void SetSwingPosZ() {                                                 // :303-317
        for (auto& s : ee_splines_) {
            for (int node : s.GetMutableNodes(orc::Position, 2)) {
                if (s.GetNodeType(orc::Position, 2, node) == FullDeriv) s.SetVars(orc::Position, 2, node, swing_height_, 0);
                else s.SetVars(orc::Position, 2, node, foot_offset_, 0);
            }
        }
    }

    std::vector<Vec13> states_;
    std::vector<EndEffectorSplines> ee_splines_;
    int pos_spline_vars_ = 0, force_spline_vars_ = 0;
    double swing_height_, foot_offset_;
    double init_time_ = 0, node_dt_;
};

// ---- SRBM model (single_rigid_body_model.cpp) ----
struct SRBModel {
    double mass = 0;
    Mat3 Ir{}, Ir_inv{};
    double hip_xy[4][2] = {};     // trunk-frame hip joint origins (x,y) per ee, FL FR RL RR
    Vec3 gravity{0, 0, -9.81};    // model.cpp:16
    int num_ee = 4;

    // single_rigid_body_model.cpp:258-308
    Vec3 GetCOMToHip(int ee) const {
        Vec3 t{hip_xy[ee][0], hip_xy[ee][1], 0.0};
        if (t[1] >= 0) t[1] += 0.1; else t[1] -= 0.1;
        t[0] += 0.025;
        return t;
    }

    // :171-177 + :179-192 (the reference state is ignored: identity is used, :174)
    static Vec12 ManifoldToTangent(const Vec13& s) {
        Vec12 t;
        for (int i = 0; i < 6; i++) t[i] = s[i];
        const Vec3 l = quat_log3(&s[6]);
        t[6] = l[0]; t[7] = l[1]; t[8] = l[2];
        t[9] = s[10]; t[10] = s[11]; t[11] = s[12];
        return t;
    }
    // :194-214
    static Vec13 TangentToManifold(const double* t) {
        Vec13 s;
        for (int i = 0; i < 6; i++) s[i] = t[i];
        quat_exp3({t[6], t[7], t[8]}, &s[6]);
        s[10] = t[9]; s[11] = t[10]; s[12] = t[11];
        return s;
    }

    // :222-256
    Vec12 CalcDynamics(const double* tan_state, const Trajectory& traj, double time) const {
        const Vec13 sm = TangentToManifold(tan_state);
        const Vec3 omega{tan_state[9], tan_state[10], tan_state[11]};
        Vec12 xd;
        for (int i = 0; i < 3; i++) xd[i] = tan_state[3 + i] / mass;
        for (int i = 0; i < 3; i++) xd[3 + i] = mass * gravity[i];
        const Vec3 io = Ir_inv.mul(omega);
        for (int i = 0; i < 3; i++) xd[6 + i] = io[i];
        const Vec3 am = -cross(omega, Ir.mul(omega));
        for (int i = 0; i < 3; i++) xd[9 + i] = am[i];
        const Vec3 com{sm[0], sm[1], sm[2]};
        for (int i = 0; i < num_ee; i++) {
            const Vec3 force = traj.GetForce(i, time);
            for (int c = 0; c < 3; c++) xd[3 + c] += force[c];
            const Vec3 tq = cross(traj.GetEndEffectorLocation(i, time) - com, force);
            for (int c = 0; c < 3; c++) xd[9 + c] += tq[c];
        }
        return xd;
    }

    // RKIntegrator::CalcIntegral, /root/reference/mpc/rk_integrator.cpp:14-30: num_steps explicit Euler steps of size dt on the
    // tangent state, every one of them evaluated at the SAME time init_time (as coded; the midpoint variant is commented
    // out there).  advance_time != 0 is the harness's variant with the time moving with the sub-steps.
    Vec13 CalcIntegral(const Vec13& ic, const Trajectory& traj, double init_time, double dt, int num_steps, int advance_time = 0) const {
        Vec12 ts = ManifoldToTangent(ic);
        for (int i = 0; i < num_steps; i++) {
            const Vec12 f = CalcDynamics(ts.data(), traj, advance_time ? init_time + i * dt : init_time);
            for (int c = 0; c < 12; c++) ts[c] = ts[c] + dt * f[c];
        }
        return TangentToManifold(ts.data());
    }

    // :55-169.  A is 12x12 row-major, B is 12 x num_inputs row-major, C is 12.
    void GetLinearDynamics(const Vec13& state, const Trajectory& traj, double time, std::vector<double>& A,
                           std::vector<double>& B, Vec12& C) const {
        const Vec3 omega{state[10], state[11], state[12]};
        A.assign(144, 0.0);
        for (int i = 0; i < 3; i++) A[(0 + i) * 12 + 3 + i] = 1.0 / mass;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) A[(6 + i) * 12 + 9 + j] = Ir_inv.m[i][j];
        const Vec3 Iw = Ir.mul(omega);
        for (int i = 0; i < 3; i++) {
            const Vec3 c1 = -cross(unit3(i), Iw) - cross(omega, Ir.col(i));
            for (int r = 0; r < 3; r++) A[(9 + r) * 12 + 9 + i] = c1[r];
            for (int ee = 0; ee < num_ee; ee++) {
                const Vec3 c2 = -cross(unit3(i), traj.GetForce(ee, time));
                for (int r = 0; r < 3; r++) A[(9 + r) * 12 + 0 + i] += c2[r];
            }
        }
        const int num_inputs = traj.GetTotalForceSplineVars() + traj.GetTotalPosSplineVars();
        const int pos_spline_start = traj.GetTotalForceSplineVars();
        B.assign((size_t)12 * num_inputs, 0.0);
        const Vec3 com{state[0], state[1], state[2]};
        for (int ee = 0; ee < num_ee; ee++) {
            const Vec3 ee_pos_wrt_com = traj.GetEndEffectorLocation(ee, time) - com;
            const Vec3 force = traj.GetForce(ee, time);
            for (int coord = 0; coord < 3; coord++) {
                if (traj.IsForceMutable(ee, time)) {
                    const std::vector<double> vars_lin = traj.GetSplineLin(Trajectory::Force, ee, coord, time);
                    auto [vars_idx, vars_affecting] = traj.GetForceSplineIndex(ee, time, coord);
                    for (int p = 0; p < vars_affecting; p++) B[(3 + coord) * num_inputs + vars_idx + p] = vars_lin.at(p);
                    const Vec3 ce = cross(ee_pos_wrt_com, unit3(coord));
                    for (int p = 0; p < (int)vars_lin.size(); p++)
                        for (int r = 0; r < 3; r++) B[(9 + r) * num_inputs + vars_idx + p] = ce[r] * vars_lin[p];
                }
                if (coord != 2) {
                    const std::vector<double> vars_lin = traj.GetSplineLin(Trajectory::Position, ee, coord, time);
                    auto [vars_idx, vars_affecting] = traj.GetPositionSplineIndex(ee, time, coord);
                    (void)vars_affecting;
                    const Vec3 cf = cross(unit3(coord), force);
                    for (int p = 0; p < (int)vars_lin.size(); p++)
                        for (int r = 0; r < 3; r++)
                            B[(9 + r) * num_inputs + pos_spline_start + vars_idx + p] = cf[r] * vars_lin[p];
                }
            }
        }
        const Vec12 ts = ManifoldToTangent(state);
        const std::vector<double> u = traj.SplinesAsVec();
        const Vec12 f = CalcDynamics(ts.data(), traj, time);
        for (int r = 0; r < 12; r++) {
            double acc = 0;
            for (int c = 0; c < 12; c++) acc -= A[r * 12 + c] * ts[c];
            double accb = 0;
            for (int c = 0; c < num_inputs; c++) accb += B[(size_t)r * num_inputs + c] * u[c];
            C[r] = acc - accb + f[r];
        }
    }

    // :458-555.  dA 12x12, dB 12 x num_inputs, dC 12 (all row-major)
    void ComputeLinearizationPartialWrtContactTimes(std::vector<double>& dA, std::vector<double>& dB, Vec12& dC,
                                                    const Vec13& state, const Trajectory& traj, double time,
                                                    int end_effector, int contact_time_idx) const {
        const int num_inputs = traj.GetTotalPosSplineVars() + traj.GetTotalForceSplineVars();
        dA.assign(144, 0.0);
        dB.assign((size_t)12 * num_inputs, 0.0);
        const Vec3 force_partial = traj.GetForcePartialWrtContactTime(end_effector, time, contact_time_idx);
        const Vec3 position_partial = traj.GetPositionPartialWrtContactTime(end_effector, time, contact_time_idx);
        for (int coord = 0; coord < 3; coord++) {
            const Vec3 c = -cross(unit3(coord), force_partial);
            for (int r = 0; r < 3; r++) dA[(9 + r) * 12 + coord] += c[r];
        }
        const Vec3 com{state[0], state[1], state[2]};
        const Vec3 ee_pos_wrt_com = traj.GetEndEffectorLocation(end_effector, time) - com;
        const Vec3 force = traj.GetForce(end_effector, time);
        const int pos_spline_start = traj.GetTotalForceSplineVars();
        for (int coord = 0; coord < 3; coord++) {
            if (traj.IsForceMutable(end_effector, time)) {
                const std::vector<double> fcp =
                    traj.GetForceCoefPartialsWrtContactTime(end_effector, coord, time, contact_time_idx);
                const std::vector<double> vars_lin = traj.GetSplineLin(Trajectory::Force, end_effector, coord, time);
                auto [vars_idx, vars_affecting] = traj.GetForceSplineIndex(end_effector, time, coord);
                for (int p = 0; p < vars_affecting; p++) dB[(3 + coord) * num_inputs + vars_idx + p] = fcp.at(p);
                const Vec3 ce = cross(ee_pos_wrt_com, unit3(coord));
                const Vec3 pe = cross(position_partial, unit3(coord));
                for (int p = 0; p < (int)fcp.size(); p++)
                    for (int r = 0; r < 3; r++)
                        dB[(9 + r) * num_inputs + vars_idx + p] = ce[r] * fcp[p] + pe[r] * vars_lin.at(p);
            }
            if (coord != 2) {
                const std::vector<double> pcp =
                    traj.GetPositionCoefPartialsWrtContactTime(end_effector, coord, time, contact_time_idx);
                const std::vector<double> vars_lin = traj.GetSplineLin(Trajectory::Position, end_effector, coord, time);
                auto [vars_idx, vars_affecting] = traj.GetPositionSplineIndex(end_effector, time, coord);
                (void)vars_affecting;
                const Vec3 cf = cross(unit3(coord), force);
                const Vec3 cfp = cross(unit3(coord), force_partial);
                for (int p = 0; p < (int)pcp.size(); p++)
                    for (int r = 0; r < 3; r++)
                        dB[(9 + r) * num_inputs + pos_spline_start + vars_idx + p] = cf[r] * pcp[p] + cfp[r] * vars_lin.at(p);
            }
        }
        const Vec12 ts = ManifoldToTangent(state);
        const std::vector<double> u = traj.SplinesAsVec();
        for (int r = 0; r < 12; r++) {
            double acc = 0;
            for (int c = 0; c < 12; c++) acc -= dA[r * 12 + c] * ts[c];
            double accb = 0;
            for (int c = 0; c < num_inputs; c++) accb += dB[(size_t)r * num_inputs + c] * u[c];
            dC[r] = acc - accb;
        }
        for (int c = 0; c < 3; c++) dC[3 + c] += force_partial[c];
        const Vec3 t = cross(ee_pos_wrt_com, force_partial) + cross(position_partial, force);
        for (int c = 0; c < 3; c++) dC[9 + c] += t[c];
    }
};

}  // namespace orc
