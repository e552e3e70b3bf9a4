// ORACLE (test infrastructure, NOT product code).
// Restatement of the interior-point method the reference calls through Clarabel.cpp
// (/root/reference/mpc/qp/clarabel_interface.cpp:18-27 settings, :29-70 cone list, :72-155 solve + status map).
// Clarabel (Rust core + Clarabel.cpp binding) is a third-party dependency that is NOT in /root/reference and is
// not version-pinned there (mpc/CMakeLists.txt:79-83 points at the author's home directory); the algorithm
// below restates the published method (P. Goulart, Y. Chen, "Clarabel: An interior-point solver for conic
// programs with quadratic objectives", 2024): Ruiz equilibration, homogeneous self-dual embedding for QPs,
// Mehrotra predictor-corrector with NT scaling (diagonal for the nonnegative cone), statically + dynamically
// regularised LDL^T of the quasi-definite KKT matrix with iterative refinement, and Clarabel's termination tests.
// Only the zero cone and the nonnegative cone are needed by the reference's call sites.
//
//   minimise 1/2 x'Px + q'x   s.t.  A x + s = b,  s in K = Zero x Nonneg x ...
#pragma once
#include <cmath>
#include <limits>
#include <string>
#include <vector>
#include "sparse_ldl.hpp"

namespace orc {

// status codes = mpc::SolveQuality (/root/reference/mpc/include/qp/qp_interface.h:12-22)
enum SolveQuality {
    Solved = 0, SolvedInacc = 1, MaxIter = 2, PrimalInfeasible = 3, DualInfeasible = 4,
    PrimalInfeasibleInacc = 5, DualInfeasibleInacc = 6, Unsolved = 7, Other = 8
};

struct Triplet { int r, c; double v; };
struct Cone { int is_nonneg; int dim; };   // is_nonneg = 0: zero cone

struct ClarabelSettings {
    int max_iter = 200;
    double max_step_fraction = 0.99;
    double tol_gap_abs = 1e-8, tol_gap_rel = 1e-8, tol_feas = 1e-8;
    double tol_infeas_abs = 1e-8, tol_infeas_rel = 1e-8, tol_ktratio = 1e-6;
    double reduced_tol_gap_abs = 5e-5, reduced_tol_gap_rel = 5e-5, reduced_tol_feas = 1e-4;
    double reduced_tol_infeas_abs = 5e-5, reduced_tol_infeas_rel = 5e-5, reduced_tol_ktratio = 1e-4;
    bool equilibrate_enable = true;
    int equilibrate_max_iter = 10;
    double equilibrate_min_scaling = 1e-4, equilibrate_max_scaling = 1e4;
    double min_terminate_step_length = 1e-4;
    double static_reg_constant = 1e-8, static_reg_proportional = 4.930380657631324e-32;  // eps^2
    double dynamic_reg_eps = 1e-13, dynamic_reg_delta = 2e-7;
    int refine_max_iter = 10;
    double refine_reltol = 1e-13, refine_abstol = 1e-12, refine_stop_ratio = 5.0;
};

struct ClarabelResult {
    std::vector<double> x, z, s;
    SolveQuality status = Unsolved;
    int iterations = 0;
    double obj_val = 0, res_primal = 0, res_dual = 0, gap_abs = 0, gap_rel = 0;
};

class ClarabelLike {
public:
    ClarabelSettings settings;

    // P: symmetric, given as triplets of the FULL matrix or the upper triangle (only r <= c entries are used;
    // duplicates are summed).  A: m x n triplets (duplicates summed).
    ClarabelResult Solve(int n, int m, const std::vector<Triplet>& Ptr, const std::vector<double>& q_in,
                         const std::vector<Triplet>& Atr, const std::vector<double>& b_in,
                         const std::vector<Cone>& cones) {
        n_ = n; m_ = m;
        // ---- compress inputs (sum duplicates) ----
        compress(Ptr, n, true, P_);
        compress(Atr, n, false, A_);
        q_ = q_in; b_ = b_in;
        is_nn_.assign(m, 0);
        int off = 0, degree = 0;
        for (auto& c : cones) {
            for (int i = 0; i < c.dim; i++) is_nn_[off + i] = c.is_nonneg;
            off += c.dim;
            if (c.is_nonneg) degree += c.dim;
        }
        if (off != m) throw std::runtime_error("cone dimensions do not match constraint rows");
        degree_ = degree;

        equilibrate();
        setup_kkt();

        ClarabelResult res;
        x_.assign(n, 0); z_.assign(m, 0); s_.assign(m, 0);
        initialise();

        std::vector<double> rx(n), rz(m), Px(n);
        std::vector<double> x2(n), z2(m), x1(n), z1(m);
        std::vector<double> dx(n), dz(m), ds(m), dx_a(n), dz_a(m), ds_a(m);
        std::vector<double> Wdiag(m);   // H = s/z on nonneg rows, 0 on zero rows
        std::vector<double> ds_const(m);
        SolveQuality status = Unsolved;
        double prev_res_primal = 1e300, prev_res_dual = 1e300, prev_gap_abs = 1e300, prev_gap_rel = 1e300;
        int iter = 0;
        for (;; iter++) {
            // ---- residuals (scaled problem) ----
            sym_mul(P_, x_, Px);
            double xPx = dot(x_, Px);
            for (int i = 0; i < n; i++) rx[i] = -Px[i] - q_[i] * tau_;
            std::vector<double> ATz(n, 0.0);
            for (auto& t : A_) ATz[t.c] += t.v * z_[t.r];
            for (int i = 0; i < n; i++) rx[i] -= ATz[i];
            std::vector<double> Axs(m, 0.0);
            for (auto& t : A_) Axs[t.r] += t.v * x_[t.c];
            for (int i = 0; i < m; i++) Axs[i] += s_[i];
            for (int i = 0; i < m; i++) rz[i] = Axs[i] - b_[i] * tau_;
            const double dot_qx = dot(q_, x_), dot_bz = dot(b_, z_);
            const double rtau = dot_qx + dot_bz + kappa_ + xPx / tau_;
            double sz = 0;
            for (int i = 0; i < m; i++) if (is_nn_[i]) sz += s_[i] * z_[i];
            const double mu = (sz + tau_ * kappa_) / (degree_ + 1);

            // ---- convergence info on the unscaled problem ----
            const double tinv = 1.0 / tau_;
            const double xPx_t2 = xPx * tinv * tinv / 2;
            const double cost_primal = (dot_qx * tinv + xPx_t2) / c_;
            const double cost_dual = (-dot_bz * tinv - xPx_t2) / c_;
            const double normx = norm_inf_scaled(x_, D_) * tinv;
            const double normz = norm_inf_scaled(z_, E_) * tinv / c_;
            const double norms = norm_inf_scaled_inv(s_, E_) * tinv;
            const double res_primal = norm_inf_scaled_inv(rz, E_) * tinv / std::max(1.0, normb_ + normx + norms);
            const double res_dual = norm_inf_scaled_inv(rx, D_) * tinv / c_ / std::max(1.0, normq_ + normx + normz);
            const double gap_abs = std::abs(cost_primal - cost_dual);
            const double gap_rel = gap_abs / std::max(1.0, std::min(std::abs(cost_primal), std::abs(cost_dual)));
            const double ktratio = kappa_ / tau_;
            res.res_primal = res_primal; res.res_dual = res_dual; res.gap_abs = gap_abs; res.gap_rel = gap_rel;
            res.obj_val = cost_primal;

            // infeasibility certificates (unscaled)
            const double res_primal_inf = norm_inf_scaled_inv(ATz, D_) / c_;     // ||A'z||
            std::vector<double> Pxv = Px;
            const double res_dual_inf_P = norm_inf_scaled_inv(Pxv, D_) / c_;    // ||Px||
            const double res_dual_inf_A = norm_inf_scaled_inv(Axs, E_);         // ||Ax+s||
            const double u_dot_bz = dot_bz / c_, u_dot_qx = dot_qx / c_;

            auto check = [&](double tga, double tgr, double tf, double tia, double tir, double tkt, SolveQuality ok,
                             SolveQuality pinf, SolveQuality dinf) -> SolveQuality {
                if (ktratio <= 1.0 && (gap_abs < tga || gap_rel < tgr) && res_primal < tf && res_dual < tf) return ok;
                if (ktratio > 1000.0 / tkt) {
                    if (u_dot_bz < -tia && res_primal_inf < -tir * std::max(1.0, normx + normz) * u_dot_bz) return pinf;
                    if (u_dot_qx < -tia && res_dual_inf_P < -tir * std::max(1.0, normx) * u_dot_qx &&
                        res_dual_inf_A < -tir * std::max(1.0, normx + norms) * u_dot_qx)
                        return dinf;
                }
                return Unsolved;
            };
            const ClarabelSettings& S = settings;
            status = check(S.tol_gap_abs, S.tol_gap_rel, S.tol_feas, S.tol_infeas_abs, S.tol_infeas_rel, S.tol_ktratio,
                           Solved, PrimalInfeasible, DualInfeasible);
            bool stop = (status != Unsolved);
            bool limited = false;
            if (!stop && iter > 0 && (res_dual > prev_res_dual || res_primal > prev_res_primal)) {
                // insufficient progress (Clarabel: residuals grow while ktratio is tiny and the gap already met)
                if (ktratio < 100 * std::numeric_limits<double>::epsilon() &&
                    (prev_gap_abs < S.tol_gap_abs || prev_gap_rel < S.tol_gap_rel)) {
                    limited = true; stop = true;
                }
            }
            if (!stop && iter >= S.max_iter) { status = MaxIter; limited = true; stop = true; }
            if (stop) {
                if (limited) {
                    SolveQuality red = check(S.reduced_tol_gap_abs, S.reduced_tol_gap_rel, S.reduced_tol_feas,
                                             S.reduced_tol_infeas_abs, S.reduced_tol_infeas_rel, S.reduced_tol_ktratio,
                                             SolvedInacc, PrimalInfeasibleInacc, DualInfeasibleInacc);
                    if (red != Unsolved) status = red;
                    else if (status != MaxIter) status = Other;   // InsufficientProgress / NumericalError -> "Other"
                }
                break;
            }
            prev_res_primal = res_primal; prev_res_dual = res_dual; prev_gap_abs = gap_abs; prev_gap_rel = gap_rel;

            // ---- scaling + KKT factorisation ----
            for (int i = 0; i < m; i++) Wdiag[i] = is_nn_[i] ? s_[i] / z_[i] : 0.0;
            if (!factor_kkt(Wdiag)) { status = Other; limited = true; stop = true; }
            // constant solve: K [x2; z2] = [-q; b]
            {
                std::vector<double> rhs(n + m);
                for (int i = 0; i < n; i++) rhs[i] = -q_[i];
                for (int i = 0; i < m; i++) rhs[n + i] = b_[i];
                kkt_solve(rhs, Wdiag);
                for (int i = 0; i < n; i++) x2[i] = rhs[i];
                for (int i = 0; i < m; i++) z2[i] = rhs[n + i];
            }
            std::vector<double> xi(n), Pxi(n);
            for (int i = 0; i < n; i++) xi[i] = x_[i] * tinv;
            sym_mul(P_, xi, Pxi);
            const double xiPxi = dot(xi, Pxi);
            double tau_den = kappa_ / tau_ + xiPxi - dot(b_, z2);
            for (int i = 0; i < n; i++) tau_den -= (q_[i] + 2 * Pxi[i]) * x2[i];

            auto solve_step = [&](const std::vector<double>& d_x, const std::vector<double>& d_z, double d_tau,
                                  double d_kappa, std::vector<double>& ox, std::vector<double>& oz,
                                  std::vector<double>& os, double& otau, double& okappa) {
                std::vector<double> rhs(n + m);
                for (int i = 0; i < n; i++) rhs[i] = d_x[i];
                for (int i = 0; i < m; i++) rhs[n + i] = ds_const[i] - d_z[i];
                kkt_solve(rhs, Wdiag);
                for (int i = 0; i < n; i++) x1[i] = rhs[i];
                for (int i = 0; i < m; i++) z1[i] = rhs[n + i];
                double tau_num = d_tau - d_kappa / tau_ + dot(b_, z1);
                for (int i = 0; i < n; i++) tau_num += (q_[i] + 2 * Pxi[i]) * x1[i];
                otau = tau_num / tau_den;
                for (int i = 0; i < n; i++) ox[i] = x1[i] + otau * x2[i];
                for (int i = 0; i < m; i++) oz[i] = z1[i] + otau * z2[i];
                for (int i = 0; i < m; i++) os[i] = is_nn_[i] ? -(ds_const[i] + Wdiag[i] * oz[i]) : 0.0;
                okappa = -(d_kappa + kappa_ * otau) / tau_;
            };

            // ---- affine (predictor) step ----
            for (int i = 0; i < m; i++) ds_const[i] = is_nn_[i] ? s_[i] : 0.0;   // (s o z)/z
            double dtau_a, dkap_a;
            solve_step(rx, rz, rtau, kappa_ * tau_, dx_a, dz_a, ds_a, dtau_a, dkap_a);
            const double alpha_a = step_length(dz_a, ds_a, dtau_a, dkap_a, 1.0);
            const double sigma = std::pow(1.0 - alpha_a, 3);

            // ---- combined (corrector) step ----
            std::vector<double> cx(n), cz(m);
            for (int i = 0; i < n; i++) cx[i] = (1 - sigma) * rx[i];
            for (int i = 0; i < m; i++) cz[i] = (1 - sigma) * rz[i];
            for (int i = 0; i < m; i++)
                ds_const[i] = is_nn_[i] ? (s_[i] * z_[i] + ds_a[i] * dz_a[i] - sigma * mu) / z_[i] : 0.0;
            double dtau, dkap;
            solve_step(cx, cz, (1 - sigma) * rtau, kappa_ * tau_ + dkap_a * dtau_a - sigma * mu, dx, dz, ds, dtau, dkap);
            double alpha = step_length(dz, ds, dtau, dkap, 1.0) * S.max_step_fraction;
            bool finite_step = std::isfinite(alpha) && std::isfinite(dtau) && std::isfinite(dkap);
            for (int i = 0; i < n && finite_step; i++) finite_step = std::isfinite(dx[i]);
            for (int i = 0; i < m && finite_step; i++) finite_step = std::isfinite(dz[i]) && std::isfinite(ds[i]);
            if (!finite_step) alpha = 0;   // Clarabel: NumericalError -> keep the iterate, post-process with reduced tolerances
            if (alpha < S.min_terminate_step_length) {
                // Clarabel: step too small -> InsufficientProgress -> post-processed with the reduced tolerances
                status = check(S.reduced_tol_gap_abs, S.reduced_tol_gap_rel, S.reduced_tol_feas, S.reduced_tol_infeas_abs,
                               S.reduced_tol_infeas_rel, S.reduced_tol_ktratio, SolvedInacc, PrimalInfeasibleInacc,
                               DualInfeasibleInacc);
                if (status == Unsolved) status = Other;
                break;
            }
            for (int i = 0; i < n; i++) x_[i] += alpha * dx[i];
            for (int i = 0; i < m; i++) { z_[i] += alpha * dz[i]; s_[i] += alpha * ds[i]; }
            tau_ += alpha * dtau;
            kappa_ += alpha * dkap;
        }
        res.iterations = iter;
        res.status = status;
        // ---- unscale: x = D xhat / tau, z = E zhat/(c tau), s = shat/(E tau) ----
        double scale = 1.0 / tau_;
        if (status == PrimalInfeasible || status == DualInfeasible || status == PrimalInfeasibleInacc ||
            status == DualInfeasibleInacc)
            scale = 1.0;   // certificates are returned unnormalised
        res.x.resize(n); res.z.resize(m); res.s.resize(m);
        for (int i = 0; i < n; i++) res.x[i] = D_[i] * x_[i] * scale;
        for (int i = 0; i < m; i++) { res.z[i] = E_[i] * z_[i] * scale / c_; res.s[i] = s_[i] / E_[i] * scale; }
        return res;
    }

private:
    // ---------- helpers ----------
    static double dot(const std::vector<double>& a, const std::vector<double>& b) {
        double s = 0;
        for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i];
        return s;
    }
    static double norm_inf_scaled(const std::vector<double>& v, const std::vector<double>& d) {
        double r = 0;
        for (size_t i = 0; i < v.size(); i++) r = std::max(r, std::abs(v[i] * d[i]));
        return r;
    }
    static double norm_inf_scaled_inv(const std::vector<double>& v, const std::vector<double>& d) {
        double r = 0;
        for (size_t i = 0; i < v.size(); i++) r = std::max(r, std::abs(v[i] / d[i]));
        return r;
    }
    static void compress(const std::vector<Triplet>& in, int ncols, bool upper_only, std::vector<Triplet>& out) {
        std::vector<Triplet> t;
        t.reserve(in.size());
        for (auto& e : in) {
            if (upper_only && e.r > e.c) continue;
            t.push_back(e);
        }
        std::sort(t.begin(), t.end(), [&](const Triplet& a, const Triplet& b) {
            return (long long)a.c * 1000000 + a.r < (long long)b.c * 1000000 + b.r;
        });
        out.clear();
        for (auto& e : t) {
            if (!out.empty() && out.back().r == e.r && out.back().c == e.c) out.back().v += e.v;
            else out.push_back(e);
        }
        (void)ncols;
    }
    // y = P x with P stored as upper triangle
    static void sym_mul(const std::vector<Triplet>& P, const std::vector<double>& x, std::vector<double>& y) {
        std::fill(y.begin(), y.end(), 0.0);
        for (auto& t : P) {
            y[t.r] += t.v * x[t.c];
            if (t.r != t.c) y[t.c] += t.v * x[t.r];
        }
    }

    void equilibrate() {
        const int n = n_, m = m_;
        D_.assign(n, 1.0); E_.assign(m, 1.0); c_ = 1.0;
        // norms of the ORIGINAL data for the convergence tests
        normq_ = 0; for (double v : q_) normq_ = std::max(normq_, std::abs(v));
        normb_ = 0; for (double v : b_) normb_ = std::max(normb_, std::abs(v));
        if (!settings.equilibrate_enable) return;
        const double lo = settings.equilibrate_min_scaling, hi = settings.equilibrate_max_scaling;
        std::vector<double> dw(n), ew(m);
        for (int it = 0; it < settings.equilibrate_max_iter; it++) {
            std::fill(dw.begin(), dw.end(), 0.0);
            std::fill(ew.begin(), ew.end(), 0.0);
            for (auto& t : P_) {
                dw[t.c] = std::max(dw[t.c], std::abs(t.v));
                dw[t.r] = std::max(dw[t.r], std::abs(t.v));
            }
            for (auto& t : A_) {
                dw[t.c] = std::max(dw[t.c], std::abs(t.v));
                ew[t.r] = std::max(ew[t.r], std::abs(t.v));
            }
            auto lim = [&](double v) {
                if (v == 0) v = 1.0;   // zero rows/columns are left alone
                v = std::min(std::max(v, lo), hi);
                return 1.0 / std::sqrt(v);
            };
            for (int i = 0; i < n; i++) dw[i] = lim(dw[i]);
            for (int i = 0; i < m; i++) ew[i] = lim(ew[i]);
            for (auto& t : P_) t.v *= dw[t.r] * dw[t.c];
            for (auto& t : A_) t.v *= ew[t.r] * dw[t.c];
            for (int i = 0; i < n; i++) { q_[i] *= dw[i]; D_[i] *= dw[i]; }
            for (int i = 0; i < m; i++) { b_[i] *= ew[i]; E_[i] *= ew[i]; }
            // cost scaling
            std::vector<double> colP(n, 0.0);
            for (auto& t : P_) {
                colP[t.c] = std::max(colP[t.c], std::abs(t.v));
                colP[t.r] = std::max(colP[t.r], std::abs(t.v));
            }
            double meanP = 0;
            for (double v : colP) meanP += v;
            meanP /= std::max(1, n);
            double qinf = 0;
            for (double v : q_) qinf = std::max(qinf, std::abs(v));
            double sc = std::max(meanP, qinf);
            if (sc == 0) sc = 1.0;
            sc = std::min(std::max(sc, lo), hi);
            const double ctmp = 1.0 / sc;
            for (auto& t : P_) t.v *= ctmp;
            for (auto& v : q_) v *= ctmp;
            c_ *= ctmp;
        }
    }

    // KKT = [P + eps I, A'; A, -(H + eps I)]; pattern fixed, H diagonal updated each iteration
    void setup_kkt() {
        const int n = n_, m = m_;
        kkt_.clear();
        diag_pos_.assign(n + m, -1);
        std::vector<char> has_diag(n, 0);
        for (auto& t : P_) {
            if (t.r == t.c) { diag_pos_[t.r] = (int)kkt_.size(); has_diag[t.r] = 1; }
            kkt_.push_back({t.c, t.r, t.v});   // upper (r<=c) -> lower entry (c, r)
        }
        for (int i = 0; i < n; i++)
            if (!has_diag[i]) { diag_pos_[i] = (int)kkt_.size(); kkt_.push_back({i, i, 0.0}); }
        for (auto& t : A_) kkt_.push_back({n + t.r, t.c, t.v});
        for (int i = 0; i < m; i++) { diag_pos_[n + i] = (int)kkt_.size(); kkt_.push_back({n + i, n + i, 0.0}); }
        Pdiag_.assign(n, 0.0);
        for (auto& t : P_) if (t.r == t.c) Pdiag_[t.r] = t.v;
        sign_.assign(n + m, 1);
        for (int i = 0; i < m; i++) sign_[n + i] = -1;
        ldl_.Analyze(n + m, kkt_);
    }
    bool factor_kkt(const std::vector<double>& H) {
        const int n = n_, m = m_;
        double maxdiag = 0;
        for (int i = 0; i < n; i++) maxdiag = std::max(maxdiag, std::abs(Pdiag_[i]));
        for (int i = 0; i < m; i++) maxdiag = std::max(maxdiag, std::abs(H[i]));
        eps_ = settings.static_reg_constant + settings.static_reg_proportional * maxdiag;
        for (int i = 0; i < n; i++) kkt_[diag_pos_[i]].v = Pdiag_[i] + eps_;
        for (int i = 0; i < m; i++) kkt_[diag_pos_[n + i]].v = -H[i] - eps_;
        ldl_.Factor(kkt_, sign_, settings.dynamic_reg_eps, settings.dynamic_reg_delta);
        return true;
    }
    // y = K x with the UNregularised K (for iterative refinement)
    void kkt_mul(const std::vector<double>& x, const std::vector<double>& H, std::vector<double>& y) const {
        const int n = n_, m = m_;
        std::fill(y.begin(), y.end(), 0.0);
        for (auto& t : P_) {
            y[t.r] += t.v * x[t.c];
            if (t.r != t.c) y[t.c] += t.v * x[t.r];
        }
        for (auto& t : A_) {
            y[n + t.r] += t.v * x[t.c];
            y[t.c] += t.v * x[n + t.r];
        }
        for (int i = 0; i < m; i++) y[n + i] -= H[i] * x[n + i];
    }
    void kkt_solve(std::vector<double>& rhs, const std::vector<double>& H) {
        const int N = n_ + m_;
        const std::vector<double> b = rhs;
        std::vector<double> x = rhs, r(N), dx(N);
        ldl_.Solve(x);
        double normb = 0;
        for (double v : b) normb = std::max(normb, std::abs(v));
        auto resid = [&](const std::vector<double>& xx) {
            kkt_mul(xx, H, r);
            double e = 0;
            for (int i = 0; i < N; i++) { r[i] = b[i] - r[i]; e = std::max(e, std::abs(r[i])); }
            return e;
        };
        double norme = resid(x);
        for (int it = 0; it < settings.refine_max_iter; it++) {
            if (norme <= settings.refine_abstol + settings.refine_reltol * normb) break;
            dx = r;
            ldl_.Solve(dx);
            std::vector<double> xn(N);
            for (int i = 0; i < N; i++) xn[i] = x[i] + dx[i];
            std::vector<double> rsave = r;
            const double newe = resid(xn);
            if (newe >= norme / settings.refine_stop_ratio) {
                if (newe < norme) { x = xn; norme = newe; } else { r = rsave; }
                break;
            }
            x = xn; norme = newe;
        }
        rhs = x;
    }

    void initialise() {
        const int n = n_, m = m_;
        // QP start: solve [P A'; A -I][x; z] = [-q; b], s = -z, then shift both into the cone
        std::vector<double> H(m, 1.0);
        factor_kkt(H);
        std::vector<double> rhs(n + m);
        for (int i = 0; i < n; i++) rhs[i] = -q_[i];
        for (int i = 0; i < m; i++) rhs[n + i] = b_[i];
        kkt_solve(rhs, H);
        for (int i = 0; i < n; i++) x_[i] = rhs[i];
        for (int i = 0; i < m; i++) { z_[i] = rhs[n + i]; s_[i] = -rhs[n + i]; }
        auto shift = [&](std::vector<double>& v, bool primal) {
            double minm = std::numeric_limits<double>::infinity(), posm = 0;
            for (int i = 0; i < m; i++) {
                if (!is_nn_[i]) continue;
                minm = std::min(minm, v[i]);
                posm += std::max(v[i], 0.0);
            }
            const double target = std::max(1.0, 0.1 * posm / std::max(1, degree_));
            double sh = 0;
            if (degree_ > 0) {
                if (minm <= 0) sh = -minm + target;
                else if (minm < target) sh = target - minm;
            }
            for (int i = 0; i < m; i++) {
                if (is_nn_[i]) v[i] += sh;
                else if (primal) v[i] = 0.0;   // zero cone: s = 0, z free
            }
        };
        shift(s_, true);
        shift(z_, false);
        tau_ = 1.0; kappa_ = 1.0;
    }

    double step_length(const std::vector<double>& dz, const std::vector<double>& ds, double dtau, double dkap,
                       double amax) const {
        double a = amax;
        if (dtau < 0) a = std::min(a, -tau_ / dtau);
        if (dkap < 0) a = std::min(a, -kappa_ / dkap);
        for (int i = 0; i < m_; i++) {
            if (!is_nn_[i]) continue;
            if (dz[i] < 0) a = std::min(a, -z_[i] / dz[i]);
            if (ds[i] < 0) a = std::min(a, -s_[i] / ds[i]);
        }
        return a;
    }

    int n_ = 0, m_ = 0, degree_ = 0;
    std::vector<Triplet> P_, A_;
    std::vector<double> q_, b_, D_, E_, Pdiag_;
    std::vector<char> is_nn_;
    double c_ = 1, normq_ = 0, normb_ = 0, eps_ = 0;
    std::vector<double> x_, z_, s_;
    double tau_ = 1, kappa_ = 1;
    std::vector<SymTriplet> kkt_;
    std::vector<int> diag_pos_, sign_;
    SparseLDL ldl_;
};

}  // namespace orc
