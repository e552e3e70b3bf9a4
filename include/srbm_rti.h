/* C-ABI of the MI355X-native batched SRBM real-time-iteration path (libsrbm_rti.so).
 *
 * Every entry point replaces, for a BATCH of independent MPC instances, one public method of the reference's
 * mpc::MPC / mpc::MPCSingleRigidBody (cited per function, paths relative to /root/reference).  Instance b of the batch
 * is exactly one reference object: same constructor arguments (srbm_mpc_info + model constants), same default contact
 * schedule, same state.  Plain pointers and sizes only; all array arguments are HOST pointers unless the function
 * name ends in _dev.  Return value: 0 on success, negative on error (srbm_last_error() gives the text).
 * The library fails loudly: there is no CPU fallback of any kind.
 */
#ifndef SRBM_RTI_H
#define SRBM_RTI_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct srbm_batch srbm_batch;

/* mpc::MPCInfo, mpc/include/mpc.h:39-62 (fields that affect the live SRBM path) */
typedef struct srbm_mpc_info {
    int num_nodes;
    double integrator_dt, friction_coef, force_bound, swing_height, foot_offset;
    double ee_box_size[2];
    double force_cost;
} srbm_mpc_info;

/* constants the reference reads from pinocchio at construction: mpc/models/model.cpp:27 (mass),
 * mpc/models/single_rigid_body_model.cpp:33-37 (Ir), :258-308 (hip joint origins, trunk frame, FL FR RL RR) */
typedef struct srbm_model {
    double mass;
    double Ir[9];
    double hip_xy[8];
} srbm_model;

/* MPCSingleRigidBody::MPCSingleRigidBody x batch  (mpc/mpc_single_rigid_body.cpp:9-23, mpc/mpc.cpp:38-76) */
int srbm_batch_create(srbm_batch** out, int batch, const srbm_mpc_info* info, const srbm_model* model, int device);
int srbm_batch_destroy(srbm_batch* h);
const char* srbm_last_error(void);

/* value semantics of the reference object: MPC::MPC(const MPC&) / operator= (mpc/mpc.cpp:1133-1181, mpc_single_rigid_body.cpp:
 * 804-807; the gait line search copies one MPC per thread, mpc/gait_optimizer.cpp:696).  The clone owns its own stream and
 * device buffers and carries the complete state of `src` (parameters, costs, tolerances, trajectories, last QP, plant). */
int srbm_batch_clone(const srbm_batch* src, srbm_batch** out);
int srbm_batch_size(const srbm_batch* h);
/* capacities of this build: cap4 = {max horizon nodes N, max spline variables n_u, max force samples, max knots per foot}.
 * libsrbm_rti.so: {50, 160, 120, 32}, the configurations the reference ships (its normal matrix lives in LDS);
 * libsrbm_rti_large.so (same sources, -DSRBM_LARGE): {100, 240, 200, 32}, the reference's own limit of 101 trajectory nodes
 * (mpc/include/trajectory.h:165-166); same C-ABI, same results, slower (normal matrix in L2). */
int srbm_get_capacity(int* cap4);
int srbm_num_nodes(const srbm_batch* h);

/* MPC::AddQuadraticTrackingCost (mpc/mpc.cpp:533-540): Q 12x12 row-major, state_des in tangent coordinates (12) */
int srbm_add_quadratic_tracking_cost(srbm_batch* h, const double* state_des12, const double* Q144);
/* MPC::SetQuadraticFinalCost / SetLinearFinalCost (mpc/mpc.cpp:137-151) */
int srbm_set_quadratic_final_cost(srbm_batch* h, const double* Phi144);
int srbm_set_linear_final_cost(srbm_batch* h, const double* w12);
/* MPC::AddForceCost (mpc/mpc.cpp:791-802): weight of every force spline variable (replaces srbm_mpc_info.force_cost) */
int srbm_add_force_cost(srbm_batch* h, double weight);
/* MPC::SetStateTrajectoryWarmStart (mpc/mpc.cpp:700-706): states[batch][13], replicated over the horizon */
int srbm_set_state_trajectory_warm_start(srbm_batch* h, const double* states);
/* ClarabelInterface tolerances (mpc/qp/clarabel_interface.cpp:18-27,165-175) for the on-device IPM.
 * Defaults: the reference's own -- gap 1e-15, feasibility 1e-10 -- and 200 iterations (Clarabel's max_iter) */
int srbm_set_solver_tolerances(srbm_batch* h, double tol_gap_abs, double tol_gap_rel, double tol_feas, int max_iter);
/* Two OPT-IN settings of the on-device solver that have no counterpart in ClarabelInterface (mpc/qp/clarabel_interface.cpp:72-155 runs Clarabel
 * to its 1e-15 gap).  A new batch has both at 0: every solve then ends by exactly the gap criterion of srbm_set_solver_tolerances, i.e. the
 * reference's (clarabel_interface.cpp:165-175), its multipliers are at that tolerance and any solve may be differentiated afterwards.
 *   tol_step > 0: a solve ALSO ends Solved as soon as the affine Newton step -- which measures the distance of the iterate to the minimiser of
 *     the QP -- is below tol_step * max(1, |u|_inf); the iterate then takes that step (what is left is <= 0.2 tol_step).  Applies to EVERY
 *     launch path (srbm_get_real_time_update[_dev], srbm_rti_advance[_unfused], srbm_closed_loop_advance, the line-search candidates and the
 *     plain RTI steps of srbm_gait_rti_advance).  The primal point is then accurate to tol_step, the multipliers only to that order: a solve
 *     that ended through the rule is flagged (srbm_get_solve_flags bit 0), residuals / gap / QP cost reported for it are re-evaluated at the
 *     returned iterate, and srbm_gait_compute_sensitivity / _gradient REFUSE it (error return, not a silent valid = 0) -- srbm_gait_rti_advance
 *     runs the one solve it differentiates at the gap criterion by itself, a caller that drives that protocol passes tol_step = 0 first.
 *   start_mu > 0: every solve is first ATTEMPTED from the linearisation point (the shifted solution of the previous RTI step) with slacks
 *     h - G u and perfectly centred multipliers lambda = start_mu / s -- five to six decades further down the central path than Clarabel's
 *     starting point -- with an iteration budget, and repeated from the standard point unless the attempt ends Solved (through the step rule
 *     or, with tol_step = 0, through the gap criterion itself: (0, start_mu) keeps the reference's TERMINATION criterion for every solve and only
 *     changes where the iteration starts).  Honoured by the device-resident open-loop launch srbm_rti_advance ONLY (there a repeated attempt of
 *     one instance is averaged over its K steps and the state is node 1 of the plan); IGNORED by srbm_get_real_time_update[_dev],
 *     srbm_rti_advance_unfused, srbm_create_initial_run (one-step launches wait for the slowest instance every time), by
 *     srbm_closed_loop_advance (integration error and pushes make the attempts fail) and by srbm_gait_rti_advance (its candidates belong to
 *     other contact schedules).
 * SRBM_FAST_TOL_STEP / SRBM_FAST_START_MU are the values bench.py opts into for its headline line (1e-4 relative primal accuracy is the bar of
 * the path; parity of that mode: tests/test_gpu_resync.py); the same line carries the run at (0, 0). */
#define SRBM_FAST_TOL_STEP 1e-5
#define SRBM_FAST_START_MU 0.1
int srbm_set_solver_step_rule(srbm_batch* h, double tol_step, double start_mu);
int srbm_get_solver_step_rule(const srbm_batch* h, double* tol_step, double* start_mu);
/* flags[batch] of the LAST solve: bit 0 = ended through the step rule (duals not at the gap tolerance), bit 1 = began with a lower-start attempt,
 * bit 2 = that attempt was repeated from the standard starting point */
int srbm_get_solve_flags(srbm_batch* h, int* flags);

/* MPC::CreateInitialRun (mpc/mpc.cpp:78-90): 10 solves at t = 0.   state[batch][13], ee[batch][4][3] */
int srbm_create_initial_run(srbm_batch* h, const double* state, const double* ee_start_locations);
/* MPC::GetRealTimeUpdate (mpc/mpc.cpp:92-108) == one MPCSingleRigidBody::Solve (mpc_single_rigid_body.cpp:25-216).
 * init_time[batch] */
int srbm_get_real_time_update(srbm_batch* h, const double* state, const double* init_time, const double* ee_start_locations);
/* same, inputs already resident in HBM (device pointers), asynchronous on the handle's stream */
int srbm_get_real_time_update_dev(srbm_batch* h, const double* state_dev, const double* init_time_dev, const double* ee_dev);
/* Device-resident open-loop protocol of test/gait_opt_playground.cpp:113-126: `steps` RTI iterations with
 * state := node 1 of the previous trajectory, foot locations := previous trajectory at t, t_i = (first_index+i)*dt.
 * No host round trip between iterations.  Asynchronous; srbm_synchronize() to wait.
 * A batch of at most one instance per CU runs as one workgroup per instance for all steps; a LARGER batch with steps > 1 runs on a resident grid
 * that takes (instance, step) items from per-XCD queues (csrc/srbm_fused.hiph: the launch no longer ends with the instance whose `steps` solves
 * happen to be the longest) -- bitwise the same results (tests/test_gpu_queue.py); SRBM_NO_STEP_QUEUE=1 in the environment at srbm_create time
 * selects the first form for every batch.  The same holds for srbm_closed_loop_advance. */
int srbm_rti_advance(srbm_batch* h, int first_index, int steps);
/* same protocol with one kernel launch per phase and step (A/B measurements against the fused kernel) */
int srbm_rti_advance_unfused(srbm_batch* h, int first_index, int steps);

/* ---- closed-loop rollout harness ----
 * What the reference closes over MuJoCo (test/simulation_mpc.cpp:186-215), device resident for a batch: the plant is
 * the single-rigid-body model itself, SingleRigidBodyModel::CalcDynamics (mpc/models/single_rigid_body_model.cpp:
 * 222-256) integrated by RKIntegrator::CalcIntegral (mpc/rk_integrator.cpp:14-30: explicit Euler on the tangent state)
 * under the forces and foot locations of the current trajectory, plus one push (momentum impulse) per instance.
 * Iteration i, t = (first_index+i)*dt (the current trajectory must start at t, e.g. first_index = 0 after
 * srbm_create_initial_run at time 0):
 *     x <- CalcIntegral(x, trajectory, t, substeps steps of dt/substeps)     advance_time = 0: every sub-step at time t
 *                                                                            (as coded), 1: time moves with the sub-steps
 *     if t < push_time <= t+dt:  lin-mom += impulse[0..2], ang-mom += impulse[3..5]
 *     MPC::GetRealTimeUpdate(x, t+dt, foot locations of the trajectory at t+dt)
 * srbm_plant_set_state: x[batch][13] (manifold state, as srbm_create_initial_run); srbm_plant_set_push: time[batch],
 * impulse[batch][6], both NULL to clear.  srbm_closed_loop_advance is asynchronous like srbm_rti_advance. */
int srbm_plant_set_state(srbm_batch* h, const double* state);
int srbm_plant_get_state(srbm_batch* h, double* state);
int srbm_plant_set_push(srbm_batch* h, const double* time, const double* impulse);
int srbm_closed_loop_advance(srbm_batch* h, int first_index, int steps, int substeps, int advance_time);
int srbm_synchronize(srbm_batch* h);
void* srbm_stream(srbm_batch* h);            /* hipStream_t the kernels are launched on */

/* ---- mpc::Trajectory as a flat record (mpc/include/trajectory.h:20-175): what MPC::GetTrajectory returns by value and
 * MPC::SetWarmStartTrajectory takes.  Knot tables as srbm_get_knots: kind 0 lift-off, 1 touch-down, 2 stance-interior
 * (force node with value + slope/FORCE_MULT), 3 mid-swing; force[ee][coord][knot] = {value, slope/100} (used on kind-2
 * knots), pos_xy[ee][coord][knot] (used on kind-0/1 knots); z follows from swing_height / foot_offset
 * (trajectory.cpp:303-317). */
#define SRBM_TRAJ_KMAX 32
#define SRBM_TRAJ_NODES_MAX 101          /* trajectory.h:165-166 */
typedef struct srbm_trajectory {
    int num_states;                       /* num_nodes + 1 */
    int nk[4];
    int knot_kind[4][SRBM_TRAJ_KMAX];
    double init_time, node_dt, swing_height, foot_offset;
    double states[SRBM_TRAJ_NODES_MAX][13];
    double knot_time[4][SRBM_TRAJ_KMAX];
    double force[4][3][SRBM_TRAJ_KMAX][2];
    double pos_xy[4][2][SRBM_TRAJ_KMAX];
} srbm_trajectory;
/* sizeof(srbm_trajectory) as the library was compiled: bindings check their own struct layout against it */
int srbm_sizeof_trajectory(void);
/* MPC::GetTrajectory (mpc/mpc.cpp:1023-1025) for instances [first, first + count): out[count] */
int srbm_get_trajectory(srbm_batch* h, int first, int count, srbm_trajectory* out);
/* MPC::SetWarmStartTrajectory (mpc/mpc.cpp:110-119) for instances [first, first + count): prev_traj_ = trajectory,
 * init_time_ = trajectory.GetTime(0).  The record is validated (knot counts, kinds, ordering); -1 on a malformed one.  The solver's memory of the instance (the back-off
 * of the lower-start attempts, srbm_set_solver_step_rule) is reset: two instances given the same trajectory solve the same next QP bit for bit. */
int srbm_set_warm_start_trajectory(srbm_batch* h, int first, int count, const srbm_trajectory* trajs);
/* Trajectory::GetForce / GetEndEffectorLocation / GetContacts at a time (mpc/trajectory.cpp:395-410, :70-80): pure host
 * arithmetic on a record (no GPU needed) -- what controllers/mpc_controller.cpp:171-186,352,415-509 evaluates at 1 kHz.
 * force[3], pos[3]; returns 0, or the error bits of the lookup (time outside the knot range: the reference throws). */
int srbm_trajectory_eval(const srbm_trajectory* traj, int ee, double time, double* force3, double* pos3, int* in_contact);
/* Trajectory::SplinesAsVec (mpc/trajectory.cpp:429-452) of a record: force spline variables (per foot, per coordinate: value and slope / FORCE_MULT of
 * every stance-interior knot) then position variables (per foot, x then y: the mutable nodes) -- the order of the spline part of the QP's decision
 * vector.  out[capacity]; *n_total = entries written, *n_force (may be NULL) = how many of them are force variables.  Host arithmetic. */
int srbm_trajectory_splines_as_vec(const srbm_trajectory* t, double* out, int capacity, int* n_total, int* n_force);
/* SingleRigidBodyModel::ConvertManifoldStateToTangentState / ConvertTangentStateToManifoldState (mpc/models/single_rigid_body_model.cpp:188-220;
 * used by the caller at controllers/mpc_controller.cpp:60): [p, lin-mom, quat xyzw, ang-mom] (13) <-> [p, lin-mom, log3(quat), ang-mom] (12).
 * Host arithmetic, the same functions the kernels use; the reference's ref_state argument is unused there (its quat_ref is the identity). */
int srbm_convert_manifold_to_tangent(const double* state13, double* tangent12);
int srbm_convert_tangent_to_manifold(const double* tangent12, double* state13);
/* the same for the CURRENT trajectory of every instance on the device: time[batch] -> force[batch][4][3], pos[batch][4][3],
 * in_contact[batch][4] (any output may be NULL) */
int srbm_eval_trajectory(srbm_batch* h, const double* time, double* force, double* pos, int* in_contact);
/* ... the same on DEVICE pointers (one launch on the batch's stream, no copy, no synchronisation): the contact flags the whole-body QP of the
 * same tick takes (srbm_qp_control_dev) without a round trip through the host; all four pointers required */
int srbm_eval_trajectory_dev(srbm_batch* h, const double* time_dev, double* force_dev, double* pos_dev, int* in_contact_dev);
/* MPCSingleRigidBody::GetEEBoxCenter (mpc/mpc_single_rigid_body.cpp:502-509): centers[4][2] = GetCOMToHip(ee).xy */
int srbm_get_ee_box_center(const srbm_batch* h, double* centers);
/* MPC::GetCost (cost of prev_qp_sol, cost[batch]) and MPC::GetAvgCost (mpc/mpc.cpp:991-998: mean over all solves so far) */
int srbm_get_cost(srbm_batch* h, double* cost);
int srbm_get_avg_cost(srbm_batch* h, double* avg_cost);
/* MPC::GetMeritValue / GetMeritGradient of the last solve (mpc/mpc.cpp:749-753, :783-788): the 'Merit' and 'Merit dd' columns of
 * MPC::PrintStatLineToFile; merit_dd may be NULL */
int srbm_get_merit(srbm_batch* h, double* merit, double* merit_dd);

/* MPC::UpdateContactTimes (mpc/mpc.cpp:1085-1088): times[batch][4][max_contacts], counts must match the current
 * number of contact knots of every foot */
int srbm_update_contact_times(srbm_batch* h, const double* times, int max_contacts);

/* MPC::AdjustForCurrentContacts (mpc/mpc.cpp:1195-1203) -> EndEffectorSplines::SetToTouchdown
 * (mpc/spline/end_effector_splines.cpp:1042-1060): time[batch], in_contact[batch][4] (0/1 per foot) */
int srbm_adjust_for_current_contacts(srbm_batch* h, const double* time, const int* in_contact);

/* ---- bilevel (gait) step: mpc::GaitOptimizer (mpc/include/gait_optimizer.h:20-170) for every instance of a batch ----
 * Contact-time vectors are laid out as in the reference's QP vector (gait_optimizer.cpp:395-408): foot after foot,
 * counts[batch][4] contact times per foot, row stride SRBM_GAIT_NV = 32 doubles per instance. */
#define SRBM_GAIT_NV 32
#define SRBM_GAIT_LS_SIZE 10      /* gait_optimizer.h:164 */
typedef struct srbm_gait srbm_gait;
/* GaitOptimizer::GaitOptimizer + UpdateSizes (gait_optimizer.cpp:15-63): allocates the 10 line-search candidates per instance */
int srbm_gait_create(srbm_batch* h, srbm_gait** out);
int srbm_gait_destroy(srbm_gait* g);
/* GaitOptimizer::SetContactTimes(mpc.GetTrajectory().GetContactTimes()) (gait_optimizer.cpp:395-408, mpc_controller.cpp:528) */
int srbm_gait_set_contact_times_from_trajectory(srbm_gait* g);
int srbm_gait_get_contact_times(srbm_gait* g, double* xk, int* counts);
/* MPC::ComputeDerivativeTerms -> ClarabelInterface::Computedx (mpc/mpc.cpp:1047-1069, mpc/qp/clarabel_interface.cpp:262-612):
 * KKT sensitivity of the last QP solution, d[batch][ld] = [dz (n); dlam (n_ineq, constraint order); dnu (n_eq)] */
int srbm_gait_compute_sensitivity(srbm_gait* g);
int srbm_gait_get_sensitivity(srbm_gait* g, double* d, int ld);
/* The gradient of the optimal cost w.r.t. the contact times (mpc_controller.cpp:518-561): SetContactTimes,
 * ComputeDerivativeTerms, MPCSingleRigidBody::ComputeParamPartialsClarabel for every contact time (mpc_single_rigid_body.cpp:
 * 642-792) and GaitOptimizer::ComputeCostFcnDerivWrtContactTimes (gait_optimizer.cpp:92-179).  dHdth[batch][32];
 * valid[batch] = 0 where the reference refuses (last QP not Solved, mpc.cpp:1048) -- may be NULL */
int srbm_gait_compute_gradient(srbm_gait* g);
int srbm_gait_get_gradient(srbm_gait* g, double* dHdth, int* valid);
/* MPCSingleRigidBody::ComputeParamPartialsClarabel (mpc/mpc_single_rigid_body.cpp:642-792; read as matrices at test/mpc_test.cpp:181-184 and
 * mpc/gait_optimizer.cpp:92-179): partials of the QP of the last solve of instance `inst` w.r.t. contact time `idx` of foot `ee`, on the instance's
 * current trajectory, dense, in the layout of mpc::QPPartials (mpc/include/qp/qp_partials.h:15-35): dA [n_eq][n], dG [n_ineq][n], db [n_eq],
 * dh [n_ineq] (zero as coded); sizes from srbm_get_sizes.  Debug path like srbm_export_qp (the gradient kernel contracts the same entries on the fly). */
int srbm_gait_get_param_partials(srbm_batch* h, int inst, int ee, int idx, double* dA, double* dG, double* db, double* dh);
/* GaitOptimizer::OptimizeContactTimes (gait_optimizer.cpp:185-364): the LP  min dHdth' s  over the polytope of
 * gait_optimizer.cpp:410-534 at time[batch]; the step stays on the device for the line search.
 * lp_status[batch]: 0 solved, 1 iteration limit, 2 numerical failure (the reference throws "Bad gait optimization
 * solve"); pred_red[batch] = -dHdth' s (gait_optimizer.cpp:352-356) */
int srbm_gait_optimize_contact_times(srbm_gait* g, const double* time);
int srbm_gait_get_lp_result(srbm_gait* g, int* lp_status, double* pred_red);
/* step of the outer problem (result of OptimizeContactTimes, or supplied by the caller): step[batch][32] */
int srbm_gait_set_step(srbm_gait* g, const double* step);
int srbm_gait_get_step(srbm_gait* g, double* step);
/* GaitOptimizer::LineSearch (gait_optimizer.cpp:671-753): 10 schedules x_k + (i/10) step per instance, each a full
 * MPC::GetRealTimeUpdate(state, time, ee) on a copy of the instance (inputs as srbm_get_real_time_update); the cheapest
 * candidate (cost / n, primal-infeasible ones excluded, index 0 if all are) is installed with
 * MPC::SetWarmStartTrajectory.  imin[batch], costs[batch][10] may be NULL. */
int srbm_gait_line_search(srbm_gait* g, const double* state, const double* init_time, const double* ee, int* imin, double* costs);
/* The MPC loop of the controller with the gait step folded in (controllers/mpc_controller.cpp:320-346), device
 * resident and open loop like srbm_rti_advance (test/gait_opt_playground.cpp:113-126): for run_num = first_run_num ...
 *   run_num % gait_opt_freq == 0, run_num > 0 : GaitOptimizer::LineSearch where a gradient is ready (plain update elsewhere)
 *   (run_num + 1) % gait_opt_freq == 0        : MPC::GetRealTimeUpdate, then MPCController::GaitOpt (gradient + LP)
 *   otherwise                                 : MPC::GetRealTimeUpdate
 * Asynchronous on the batch's stream; srbm_synchronize(h) to wait. */
int srbm_gait_rti_advance(srbm_gait* g, int first_run_num, int steps, int gait_opt_freq);
/* solver status / error bits of the candidates of the last line search: status[batch*10], err[batch*10] */
int srbm_gait_get_candidate_status(srbm_gait* g, int* status, int* err);
/* test hook: the candidate batch of the last line search (borrowed: never destroy it), candidate c of instance b at b * 10 + c; for the
 * read-back entries (srbm_get_status, srbm_get_sizes, srbm_export_qp) */
srbm_batch* srbm_gait_debug_candidates(srbm_gait* g);

/* ---- trajectory -> whole-body targets: the step right downstream of the MPC in the reference's
 * controller.  q = [base position (3), base quaternion xyzw (4), 12 joint angles in pinocchio's model order FL FR RL RR x
 * (hip, thigh, calf)]; v = [base linear, base angular velocity (base frame), 12 joint rates]. ---- */
/* leg geometry (what the reference reads from the URDF through pinocchio): per leg FL FR RL RR the origins of the hip joint in
 * the trunk, the thigh joint in the hip, the calf joint in the thigh and the foot in the calf; joint axes x, y, y, no rotation in
 * the origins (A1: models/a1_description/urdf/a1.urdf:363-466) */
typedef struct srbm_leg_kinematics { double origin[4][4][3]; } srbm_leg_kinematics;
int srbm_set_leg_kinematics(srbm_batch* h, const srbm_leg_kinematics* legs);
/* SingleRigidBodyModel::GetEndEffectorLocations (mpc/models/single_rigid_body_model.cpp:443-455): q[batch][19] -> ee[batch][4][3] */
int srbm_forward_kinematics(srbm_batch* h, const double* q, double* ee);
/* SingleRigidBodyModel::InverseKinematics (mpc/models/single_rigid_body_model.cpp:314-425): damped least squares on (foot position,
 * base pose), one foot after the other, eps 5e-6, step 0.1, <= 1000 iterations per foot.  state[batch][13] (base pose from it),
 * ee[batch][4][3] desired foot positions, q_guess[batch][19] (joint part used) -> q_out[batch][19]; iters[batch][4] and
 * status[batch] (0 converged, 1 not converged: the reference throws "IK did not converge.") may be NULL */
int srbm_inverse_kinematics(srbm_batch* h, const double* state, const double* ee, const double* q_guess, double* q_out, int* iters, int* status);
/* MPCController::GetTargetsFromTraj (controllers/mpc_controller.cpp:414-511) on the CURRENT trajectory of every instance at time[batch]:
 * interpolated state -> IK at `time` and `time + dt` -> q_des (in: the previous q_des as the IK guess, out), v_des[batch][18],
 * force_des[batch][4][3] = Trajectory::GetForce(ee, time).  status[batch]: 0 ok, 1 IK not converged, 2 time outside the trajectory
 * ("bad interp." / spline range: the reference throws) */
int srbm_get_targets_from_traj(srbm_batch* h, const double* time, double* q_des, double* v_des, double* force_des, int* status);
/* ... on device pointers (same shapes): ONE launch on the batch's stream, no copy, no synchronisation */
int srbm_get_targets_from_traj_dev(srbm_batch* h, const double* time_dev, double* q_des_dev, double* v_des_dev, double* force_des_dev, int* status_dev);

/* QPControl (controllers/qp_control.cpp): the 1 kHz whole-body inverse-dynamics QP, for every instance of the batch.
 * Model: the rigid bodies pinocchio builds from the URDF -- trunk, then FL FR RL RR x (hip, thigh, calf), links on fixed joints
 * merged into their parent -- each with mass, centre of mass and rotational inertia about it (row-major 3x3) in the frame of
 * its joint; gains and weights as the QPControl constructor takes them (qp_control.cpp:20-52: base_pos_gains = {kv, kp}, ...) */
typedef struct srbm_wbc_model {
    double body_mass[13], body_com[13][3], body_inertia[13][9];
    double torque_bounds[12], kp_joint_gains[12], kd_joint_gains[12];
    double base_pos_gains[2], base_ang_gains[2];
    double leg_tracking_weight, torso_tracking_weight, force_tracking_weight, friction_coef, max_grf;
} srbm_wbc_model;
int srbm_set_wbc_model(srbm_batch* h, const srbm_wbc_model* model);
/* QPControl::ComputeControlAction (qp_control.cpp:74-135) with the targets of UpdateTargetConfig / Vel / ForceTargets:
 * q[batch][19], v[batch][18] measured; contact[batch][4] (0/1: in contact, measured and desired); q_des[batch][19], v_des[batch][18],
 * force_des[batch][12] = 3 per foot IN CONTACT, stacked in foot order (the reference's force_target_).
 * control[batch][36] = joint position targets (12), joint velocity targets (12), torques (12) -- all zero where the QP failed, as the
 * reference returns; qp_sol[batch][30] = accelerations (18) then contact forces; status[batch] = SolveQuality | iterations << 8.
 * qp_dump (may be NULL): the assembled QP in the reference's layout per instance: A[50][30], lb[50], ub[50], diag P[30], w[30].
 * The solver relies on the rows being the controller's (every torque row and every contact-force row two-sided): with a torque bound or max_grf of
 * exactly zero such a row is an equality, and the instance is reported with status 8 (SRBM_OTHER) and a zero control action. */
int srbm_qp_control(srbm_batch* h, const double* q, const double* v, const int* contact, const double* q_des, const double* v_des,
                    const double* force_des, double* control, double* qp_sol, int* status, double* qp_dump);
/* ... on device pointers: control_dev[batch][36], qp_sol_dev[batch][30], status_dev[batch]; one launch, no copy, no synchronisation */
int srbm_qp_control_dev(srbm_batch* h, const double* q_dev, const double* v_dev, const int* contact_dev, const double* q_des_dev, const double* v_des_dev,
                        const double* force_des_dev, double* control_dev, double* qp_sol_dev, int* status_dev);

/* ---- results (all copied to host) ---- */
/* sizes[batch][8] = n, m, n_eq, n_ineq, n_force_vars, n_pos_vars, n_td_rows, n_force_samples */
int srbm_get_sizes(srbm_batch* h, int* sizes);
/* status[batch] = mpc::SolveQuality (mpc/include/qp/qp_interface.h:12-22); err[batch] = error bits (0 = none).
 * Both describe the LAST solve only.  Error bits (conditions on which the reference throws, plus two of this library): 1 time before the first knot,
 * 2 time beyond the last knot, 4 invalid time, 8 force node not mutable, 16 beyond the capacity of the build (status Other), 32 RemovePoly on an empty
 * spline, 64 touch-down index, 128 a pivot of the normal matrix was regularised, 256 internal invariant violated (a dense state row with a non-zero
 * outside the force variables of its coordinate: never observed; the compact storage of those rows rests on it), 512 a bounded wait of the step
 * queue ran out (multi-step launches of a batch larger than the chip hand out (instance, step) items to a resident grid; never observed). */
int srbm_get_status(srbm_batch* h, int* status, int* err);
/* Sticky accumulators over every solve since creation / the last clear (multi-step launches overwrite status and err each
 * step): acc[batch][4] = {all error bits raised, solves, solves not in {Solved, SolvedInacc}, of those MaxIter} */
int srbm_get_status_accumulated(srbm_batch* h, int* acc);
int srbm_clear_status_accumulators(srbm_batch* h);
/* counters over the same span: c[4] = {solves, solves ended by the step rule, lower-start attempts, attempts repeated from the standard start} */
int srbm_get_solver_counters(srbm_batch* h, long long* c4);
/* stats[batch][8] = alpha, cost (GetCost), L1 dynamics defect, step norm, qp iterations, res_primal, res_dual, gap_rel */
int srbm_get_stats(srbm_batch* h, double* stats);
/* objective of the QP at its raw minimiser, cost[batch] (the "QP Cost" column of MPC::PrintStatLineToFile, mpc/mpc.cpp:974-989) */
int srbm_get_qp_cost(srbm_batch* h, double* cost);
/* MPC::GetQPSolution (prev_qp_sol after the line search), x[batch][ld]; ld >= n_max */
int srbm_get_qp_solution(srbm_batch* h, double* x, int ld);
int srbm_get_raw_qp_minimiser(srbm_batch* h, double* x, int ld);
/* ClarabelInterface::GetDualSolution in the reference's row order; z[batch][ld] */
int srbm_get_dual_solution(srbm_batch* h, double* z, double* s, int ld);
/* Trajectory::GetStates, states[batch][N+1][13] */
int srbm_get_trajectory_states(srbm_batch* h, double* states);
/* knot tables of one instance: times[4][32], kinds[4][32] (0 LO, 1 TD, 2 stance-interior, 3 mid-swing), nk[4],
 * fvals[4][3][32][2], pvals[4][2][32], box[2] */
int srbm_get_knots(srbm_batch* h, int inst, double* times, int* kinds, int* nk, double* fvals, double* pvals, double* box);
/* Dense expansion of the structured QP of the LAST solve of one instance into the reference's layout
 * (rows/cols as SURVEY.md Appendix A): A[m][n], b[m], P[n][n], q[n].  Debug / parity-test aid. */
int srbm_export_qp(srbm_batch* h, int inst, double* A, double* b, double* P, double* q);
/* result records for collection across GPUs (one RCCL all-gather in bench.py): out_dev[batch][ld] on
 * the handle's stream.  Layout of one record (doubles), NX = 12 (N+1) + n_u_max, NM = 12 (N+1) + 6 * samples_max + 16 (N-3) + 16 with the capacities of the loaded build
 * (srbm_get_capacity: 160 / 120 in the standard build, 240 / 200 in the LARGE one):
 *   [0..8)  status, n, m, cost, alpha, err (sticky bits since the last clear), qp_iters, init_time
 *   [8 .. 8+NX)           x[0..n)   primal (prev_qp_sol), zero padded
 *   [8+NX .. 8+NX+NM)     z[0..m)   dual vector in the reference's row order, zero padded
 *   [8+NX+NM .. +36)      contact-time counts of the 4 feet, then 4 x 8 contact times
 * srbm_result_record_doubles(N) = 8 + NX + NM + 36.  A shorter ld truncates the record (ld >= 8). */
int srbm_result_record_doubles(int num_nodes);
int srbm_pack_results_dev(srbm_batch* h, double* out_dev, int ld);
/* the same records into a HOST buffer out[batch][ld] (synchronous) */
int srbm_pack_results(srbm_batch* h, double* out, int ld);

/* ---- multi-GPU: collecting the solved trajectories of all shards (north_star: "partitioned across the 8 GPUs of one node with an RCCL all-gather
 * over xGMI only to collect solved trajectories"; the reference's only batch is the 10-thread line search, mpc/gait_optimizer.cpp:688-721; the host
 * that would call this is the MPC thread of controllers/mpc_controller.cpp:286-399, one per GPU) ----
 * One process per GPU, each owning a batch of the SAME size (contiguous shards of the global batch).  srbm_allgather_results packs this rank's
 * result records (srbm_pack_results_dev layout, ld = srbm_result_record_doubles(N)) straight into its slot of out_dev and runs ONE in-place
 * ncclAllGather (RCCL) on the batch's stream: out_dev[world * batch][ld] on every rank, rank r's records at rows [r * batch, (r + 1) * batch).
 * Asynchronous like every *_dev entry: srbm_synchronize (or work queued behind it on srbm_get_stream) before out_dev is read.
 * `comm` is a plain ncclComm_t of <rccl/rccl.h> -- the caller's own (ncclCommInitRank on its side), or one made by the helpers below for hosts
 * that do not link RCCL themselves (ctypes, bench.py).  The library does not link RCCL: it binds the copy already loaded in the process
 * (librccl.so / librccl.so.1), else loads librccl.so.1, at the first of these calls; SRBM_RCCL_LIB overrides the name.  No RCCL, no call: they fail loudly. */
struct ncclComm;
typedef struct ncclComm* ncclComm_t;               /* identical to the typedef in <rccl/rccl.h> */
#define SRBM_RCCL_UNIQUE_ID_BYTES 128              /* sizeof(ncclUniqueId) */
int srbm_allgather_results(srbm_batch* h, ncclComm_t comm, double* out_dev);
/* ncclGetUniqueId on one rank (bytes travel to the others by whatever the host uses for its rendezvous), ncclCommInitRank on the batch's device, ncclCommDestroy */
int srbm_rccl_get_unique_id(void* id_bytes);
int srbm_rccl_comm_init_rank(srbm_batch* h, int world, int rank, const void* id_bytes, ncclComm_t* comm_out);
int srbm_rccl_comm_destroy(ncclComm_t comm);
/* measurement aids: HIP-event timing of the dominant kernel (srbm_k3_ipm) on the launch stream, and running totals
 * of executed IPM iterations / algorithmic flops (SURVEY.md section 8d formula) summed over the batch */
int srbm_enable_kernel_timing(srbm_batch* h, int max_launches);
int srbm_get_kernel_timing(srbm_batch* h, double* total_ms, int* launches);
/* ... and launch by launch: ms_each[min(launches, max_launches)] in launch order (bench.py prices the MEDIAN region with its own launch) */
int srbm_get_kernel_timings(srbm_batch* h, double* ms_each, int max_launches, int* launches);
int srbm_get_work_counters(srbm_batch* h, double* total_ipm_iterations, double* total_algorithmic_flops);
/* matrix-core instructions (v_mfma_f64_16x16x4_f64, 2048 flop each, counted per wave) EXECUTED by the condensing and IPM
 * phases, summed over the batch: the executed-flop side of the roofline (the algorithmic figure counts a dense SYRK that the
 * structured assembly never performs) */
int srbm_get_executed_mfma(srbm_batch* h, double* total_mfma_instructions);
/* bytes of HBM held per instance (persistent record + per-solve workspace) */
long srbm_bytes_per_instance(void);

#ifdef __cplusplus
}
#endif
#endif
