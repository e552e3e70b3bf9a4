// Shared host/device data layout of the batched SRBM real-time-iteration path (MI355X / gfx950).
//
// One "instance" = one mpc::MPCSingleRigidBody of the reference (/root/reference/mpc/include/mpc_single_rigid_body.h:11-76):
//   SrbmInst  -- the persistent part (what the reference keeps in prev_traj_ + solver status): node states, the four
//                per-foot knot tables (times, knot kinds, force/position node values), the current foot-box size.
//   SrbmWork  -- per-solve scratch that lives in HBM/L2 between the kernels of one RTI step (structured QP:
//                per-node linearisation records, condensed Hessian, dense state rows of the foot-box constraints,
//                inequality sample tables, primal/dual iterates).
// Everything is fp64 (the reference computes in Eigen::VectorXd) + int32 indices.
#pragma once
#include <stdint.h>

#define SRBM_NEE 4
/* Where the packed normal matrix of the IPM lives: LDS (standard build: one workgroup of 512 threads per CU) or the work record in global
   memory / L2 (SRBM_M_GLOBAL: the LARGE capacities, whose matrix does not fit the LDS) */
#if defined(SRBM_LARGE)
#define SRBM_M_GLOBAL 1
#endif
#define SRBM_KMAX 32          /* knots per foot inside the horizon window */
#ifdef SRBM_LARGE
/* The LARGE build of the same sources (libsrbm_rti_large.so): the reference's own limit of 101 trajectory nodes
   (mpc/include/trajectory.h:165-166), up to 240 spline variables and 20 stance phases in the window -- N = 40 at dt = 0.05 (2 s
   horizon: 232 variables, 160 force samples) and the short-phase schedules the gait LP may produce (MIN_TIME 0.2,
   mpc/gait_optimizer.cpp:412).  The packed normal matrix (231 KB at n_u = 240) no longer fits the 160 KB of LDS beside the iterates: it lives in the work record (L2 / Infinity
   Cache), the tiles still in the accumulator registers during the factorisation.  Same algorithm, same results, slower. */
#define SRBM_NMAX 100
#define SRBM_NUMAX 240
#define SRBM_NSMAX 200
#else
#define SRBM_NMAX 50          /* horizon nodes (reference configs use 10 / 20 / 50) */
#define SRBM_NUMAX 160        /* spline (input) variables */
#define SRBM_NSMAX 120        /* force samples: FB_PER_FORCE(10) x stance phases in the window */
#endif
#define SRBM_NEEROW (2 * (SRBM_NMAX - 3))          /* dense state rows: (node 4..N) x (x,y) */
#define SRBM_MIMAX (6 * SRBM_NSMAX + 16 * (SRBM_NMAX - 3))
#define SRBM_NXMAX ((SRBM_NMAX + 1) * 12 + SRBM_NUMAX)
#define SRBM_MMAX ((SRBM_NMAX + 1) * 12 + SRBM_MIMAX + 16)
#define SRBM_HPACK (SRBM_NUMAX * (SRBM_NUMAX + 1) / 2)

/* knot kinds (end_effector_splines.cpp:45-100 pattern): force / position-xy / position-z node types follow from the kind */
enum { SRBM_K_LO = 0, SRBM_K_TD = 1, SRBM_K_F = 2, SRBM_K_MID = 3 };

/* mpc::SolveQuality, /root/reference/mpc/include/qp/qp_interface.h:12-22 */
enum { SRBM_SOLVED = 0, SRBM_SOLVED_INACC = 1, SRBM_MAX_ITER = 2, SRBM_PRIMAL_INFEASIBLE = 3, SRBM_DUAL_INFEASIBLE = 4,
       SRBM_PRIMAL_INFEASIBLE_INACC = 5, SRBM_DUAL_INFEASIBLE_INACC = 6, SRBM_UNSOLVED = 7, SRBM_OTHER = 8 };

/* error bits (SrbmInst.err): conditions on which the reference throws */
enum { SRBM_ERR_TIME_SMALL = 1, SRBM_ERR_TIME_LARGE = 2, SRBM_ERR_INVALID_TIME = 4, SRBM_ERR_FORCE_NOT_MUTABLE = 8,
       SRBM_ERR_CAPACITY = 16, SRBM_ERR_REMOVE_POLY = 32, SRBM_ERR_TD_INDEX = 64, SRBM_ERR_CHOLESKY = 128,
       SRBM_ERR_STRUCTURE = 256 /* internal invariant: a dense state row has a non-zero outside the force variables of its coordinate (srbm_k2_condense.hiph) */,
       SRBM_ERR_QUEUE = 512 /* a bounded wait of the step queue of a batch larger than the chip ran out (srbm_fused.hiph: srbm_rti_queued) */ };

typedef struct SrbmParams {
    int batch, N;
    int max_iter, lds_doubles;      /* IPM iteration limit; doubles of dynamic LDS the IPM kernel is launched with */
    int q_diag, pad0;               /* Q and Phi are diagonal (every configuration of the reference): the condensing kernel then skips the 12x12 products */
    double dt, mu_fric, force_bound, swing_height, foot_offset, force_cost;
    double box0[2];                 /* configured ee_box_size (ee_bounds_, msrb.cpp:22) */
    double mass, Ir[9], Ir_inv[9];
    double hip[8];                  /* GetCOMToHip(ee).xy, single_rigid_body_model.cpp:258-308 */
    double Q[144], w[12], Phi[144], Phi_w[12];
    double merit_mu, td_fraction;   /* mpc.cpp:65, :73 */
    double tol_gap_abs, tol_gap_rel, tol_feas;
    /* (host-side record of srbm_set_solver_step_rule; the kernels receive the two values as launch arguments)  tol_step > 0 ends a solve as soon as the affine (predictor) Newton step -- the distance to the KKT point of
       the QP as the factor at hand sees it -- is below tol_step * max(1, |u|_inf): the iterate then takes that step and the solve is over;
       start_mu > 0 (fused open-loop launches only): every solve is first attempted from the LINEARISATION POINT with slacks h - G u and centred
       multipliers lambda = start_mu / s.
       Both are 0 in a new batch: the reference's gap criterion (include/srbm_rti.h, srbm_set_solver_step_rule) */
    double tol_step, start_mu;
    double legs[SRBM_NEE][4][3];    /* leg geometry for the IK of row f3 (srbm_ik.hiph): joint origins hip / thigh / calf / foot */
    int has_legs, pad2;
} SrbmParams;

typedef struct SrbmInst {
    double states[(SRBM_NMAX + 1) * 13];          /* manifold states [p, lin-mom, quat xyzw, ang-mom] */
    double knot_t[SRBM_NEE][SRBM_KMAX];
    double fval[SRBM_NEE][3][SRBM_KMAX][2];       /* force node (value, slope/FORCE_MULT) */
    double pval[SRBM_NEE][2][SRBM_KMAX];          /* position xy node values */
    double box[2];
    double init_time;
    double alpha, cost, eq_violation, step_norm, qp_cost, res_primal, res_dual, gap;
    int nk[SRBM_NEE];
    uint8_t kind[SRBM_NEE][SRBM_KMAX];
    int status, qp_iters, n, m, n_eq, n_ineq, nfv, npv, n_td, n_samples, err, run_num;
    double acc_iters, acc_flops;                  /* running totals: IPM iterations, algorithmic flops (SURVEY.md 8d formula) */
    /* sticky accumulators over ALL solves since srbm_clear_status_accumulators: `err` / `status` describe the last solve only
       (kernel 1 restarts them), these keep every error bit raised and count the solves by outcome; cost_sum / run_num is
       MPC::GetAvgCost (mpc.cpp:991-998: mean of the cost_ entries RecordStats pushes, one per solve) */
    double cost_sum;
    double merit_dd;                              /* directional derivative of the L1 merit along the step of the last solve (mpc.cpp:783-788; the 'Merit dd' column of the statistics log) */
    double acc_mfma;                              /* v_mfma_f64_16x16x4_f64 instructions EXECUTED (per wave) by the condensing and IPM phases: the executed-flop side of the roofline */
    int err_acc, n_solves, n_not_solved, n_maxiter;       /* n_not_solved: status not in {Solved, SolvedInacc}; n_maxiter: of those, MaxIter */
    int low_streak, last_rule;                            /* last_rule: 1 if the LAST solve ended through the step rule (its duals are then not at the reference's gap tolerance: the
                                                             gait gradient is marked invalid); low_streak: consecutive failed attempts: no back-off after the first failure of a streak, K3_LOW_BACKOFF << (streak - 1) solves (at most 48) from the second on */
    int last_low, pad_low;                                /* last solve: bit 0 began with a lower-start attempt, bit 1 the attempt was repeated from the standard start */
    int low_skip, n_low_tried, n_low_failed, n_step_rule; /* lower-start attempts (srbm_k3_ipm.hiph): solves left that skip the attempt, attempts, attempts repeated
                                                             from the standard start; solves ended by the step rule */
} SrbmInst;

/* per (node, foot) linearisation record */
typedef struct SrbmNodeRec {
    double f[3];        /* force at the node time */
    double r[3];        /* foot location at the node time */
    double flin[4];     /* Hermite basis coefficients of the force variables (same for x,y,z) */
    double plin[2];     /* basis coefficients of the position variables (same for x,y) */
    int fidx, fcnt, fmut, pidx, pcnt, pad;   /* indices local to the (foot, coord) variable block */
} SrbmNodeRec;

typedef struct SrbmSample {   /* one force sample time (10 per stance phase): rows of ForceBox + FrictionCone */
    double phi[4];
    int idx, cnt, ee, pad;
} SrbmSample;

typedef struct SrbmWork {
    /* ---- assembly (kernel 1) ---- */
    SrbmNodeRec node[SRBM_NMAX + 1][SRBM_NEE];
    double ALp[SRBM_NMAX][9], ALL[SRBM_NMAX][9];  /* continuous-time blocks dL/dp, dL/dL of A_k */
    double cbar[SRBM_NMAX][12];                   /* dt*C_k + columns of fixed / substituted variables */
    double craw[SRBM_NMAX][12];                   /* dt*C_k */
    double xbar[(SRBM_NMAX + 1) * 12];            /* tangent states of the linearisation point */
    double u_prev[SRBM_NUMAX];                    /* spline variables of the linearisation point */
    SrbmSample samp[SRBM_NSMAX];
    double eq_rhs[16];                            /* TD rows then start rows (reference order) */
    double eq_coef[16][2];
    int eq_idx[16], eq_cnt[16];
    double fix_val[SRBM_NUMAX];                   /* value of pinned variables */
    double sub_kappa[SRBM_NUMAX], sub_rho[SRBM_NUMAX];
    int fix_mask[SRBM_NUMAX];                     /* 1: pinned to fix_val, 2: u_j = kappa + rho*u_{sub_src}, 0: free */
    int sub_src[SRBM_NUMAX];                      /* for a free column: index of a variable substituted INTO it, or -1 */
    int sub_tgt[SRBM_NUMAX];                      /* for a substituted variable: the free variable it depends on */
    int col_ee[SRBM_NUMAX], col_type[SRBM_NUMAX], col_coord[SRBM_NUMAX], col_local[SRBM_NUMAX];
    int fbase[SRBM_NEE][3], pbase[SRBM_NEE][2], nfv_ee[SRBM_NEE], npv_ee[SRBM_NEE];
    int nu, nf, np, n_samp, n_td, n_eq_u, td_ee_mask, pad1;
    double box_used[2];                           /* foot-box size in effect for this solve */
    /* ---- condensing (kernel 2) ---- */
    double H[SRBM_HPACK];                         /* reduced Hessian, packed lower, row-major */
    double g[SRBM_NUMAX];
    double Sig[SRBM_NEEROW][SRBM_NUMAX];          /* rows c of S_k, k=4..N, c in {x,y}: row index 2*(k-4)+c; COMPACT: entry q = force variable q of coordinate c over the four feet
                                                     (the only non-zeros of the row; srbm_k3_ipm.hiph, K3Smem) */
    double sig0[SRBM_NEEROW];                     /* affine part s_k[c] */
    /* ---- IPM (kernel 3) ---- */
    double u[SRBM_NUMAX];                         /* QP minimiser, spline variables (full vector incl. pinned) */
    double lam[SRBM_MIMAX], slack[SRBM_MIMAX];
    /* ---- recovery (kernel 4) ---- */
    double x_qp[SRBM_NXMAX];                      /* raw QP minimiser in the reference's decision-vector layout */
    double x[SRBM_NXMAX];                         /* prev_qp_sol after the line search */
    double z[SRBM_MMAX];                          /* dual vector in the reference's row order */
    double s[SRBM_MMAX];
#ifdef SRBM_M_GLOBAL
    double Mg[SRBM_HPACK];                        /* the normal matrix / its factor / the inverse of the factor of the IPM (in LDS otherwise); followed by Ms: the one-block-ahead
                                                     operand prefetch of dn_trtri_column reads up to 15 doubles past the packed matrix (values masked), which must stay inside the record */
#endif
    double Ms[SRBM_HPACK];                        /* gait step: H + G' diag(lambda/s) G of the last solution (srbm_k3_normal_matrix) */
    double w0[SRBM_MIMAX];                        /* IPM: unit weight of the row/cost-scaled problem, e_r^2 / c (kernel 3 scratch) */
    double hrow_g[SRBM_MIMAX];                    /* IPM: right-hand sides of the inequality rows (large build: read from here instead of held in registers) */
    double prof2[96];                             /* diagnostic builds only: fine-grained stamps (K3_FINE) */
    double prof[16];                              /* diagnostic builds only (-DSRBM_PROFILE): cycles per IPM phase */
    double dbg[4 * 64];
    double dbg2[4 * 32];                          /* diagnostic builds only: worst refinement row (index, s, lambda, e2) */                           /* diagnostic builds only: per-iteration (mu, alpha_aff, alpha, gap_rel) */
} SrbmWork;

/* launch arguments of the fused RTI kernel (srbm_fused.hiph) besides the batch: the closed-loop mode (plant != nullptr, srbm_plant.hiph) and the two
   settings of srbm_set_solver_step_rule */
typedef struct SrbmPlantArgs {
    double* plant; const double* push_time; const double* push_impulse;
    int substeps, advance_time;
    double tol_step, start_mu;
} SrbmPlantArgs;
