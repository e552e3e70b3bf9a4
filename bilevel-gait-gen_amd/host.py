"""Python binding of the C-ABI in include/srbm_rti.h (libsrbm_rti.so, HIP/gfx950).

`BatchMPC` mirrors the public surface of the reference's mpc::MPCSingleRigidBody for a batch of instances
(/root/reference/mpc/include/mpc.h:70-170, mpc_single_rigid_body.h:11-76): same method names (snake_case), same
argument meaning.  There is NO CPU fallback: if the HIP library or a GPU is missing this module raises.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SRBM_RTI_LIB', os.path.join(HERE, 'libsrbm_rti.so'))   # override: A/B builds of scripts/dev_ab.py
# the LARGE-capacity build of the same sources (N <= 100, n_u <= 240, csrc/srbm_types.h): same C-ABI, slower (normal matrix in L2)
LIB_PATH_LARGE = os.environ.get('SRBM_RTI_LIB_LARGE', os.path.join(HERE, 'libsrbm_rti_large.so'))
CONFIG_DIR = os.path.join(HERE, 'configs')

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

SOLVE_QUALITY = ['Solved', 'SolvedInacc', 'MaxIter', 'PrimalInfeasible', 'DualInfeasible', 'PrimalInfeasibleInacc',
                 'DualInfeasibleInacc', 'Unsolved', 'Other']


class MPCInfo(C.Structure):          # srbm_mpc_info
    _fields_ = [('num_nodes', C.c_int), ('integrator_dt', C.c_double), ('friction_coef', C.c_double),
                ('force_bound', C.c_double), ('swing_height', C.c_double), ('foot_offset', C.c_double),
                ('ee_box_size', C.c_double * 2), ('force_cost', C.c_double)]


class WbcModel(C.Structure):         # srbm_wbc_model
    _fields_ = [('body_mass', C.c_double * 13), ('body_com', (C.c_double * 3) * 13), ('body_inertia', (C.c_double * 9) * 13),
                ('torque_bounds', C.c_double * 12), ('kp_joint_gains', C.c_double * 12), ('kd_joint_gains', C.c_double * 12),
                ('base_pos_gains', C.c_double * 2), ('base_ang_gains', C.c_double * 2),
                ('leg_tracking_weight', C.c_double), ('torso_tracking_weight', C.c_double), ('force_tracking_weight', C.c_double),
                ('friction_coef', C.c_double), ('max_grf', C.c_double)]


class Model(C.Structure):            # srbm_model
    _fields_ = [('mass', C.c_double), ('Ir', C.c_double * 9), ('hip_xy', C.c_double * 8)]


KMAX, NODES_MAX = 32, 101
# srbm_set_solver_step_rule: a new batch runs every solve to the reference's gap criterion (0, 0); these are the values bench.py opts into for
# its headline line (include/srbm_rti.h: SRBM_FAST_TOL_STEP, SRBM_FAST_START_MU)
FAST_TOL_STEP, FAST_START_MU = 1e-5, 0.1


class Trajectory(C.Structure):       # srbm_trajectory: mpc::Trajectory as a flat record (include/srbm_rti.h)
    _fields_ = [('num_states', C.c_int), ('nk', C.c_int * 4), ('knot_kind', (C.c_int * KMAX) * 4),
                ('init_time', C.c_double), ('node_dt', C.c_double), ('swing_height', C.c_double), ('foot_offset', C.c_double),
                ('states', (C.c_double * 13) * NODES_MAX), ('knot_time', (C.c_double * KMAX) * 4),
                ('force', (((C.c_double * 2) * KMAX) * 3) * 4), ('pos_xy', ((C.c_double * KMAX) * 2) * 4)]

    # ---- the part of mpc::Trajectory's interface the caller uses (controllers/mpc_controller.cpp:171-186,415-509) ----
    def get_time(self, node):                      # Trajectory::GetTime (trajectory.cpp:412-414)
        return self.init_time + self.node_dt * node

    def get_node(self, time):                      # Trajectory::GetNode (trajectory.cpp:479-481)
        return int(np.ceil((time - self.init_time) / self.node_dt))

    def get_states(self):
        return np.ctypeslib.as_array(self.states)[:self.num_states].copy()

    def _eval(self, ee, time):
        f = (C.c_double * 3)(); p = (C.c_double * 3)(); c = C.c_int(0)
        rc = lib().srbm_trajectory_eval(C.byref(self), int(ee), C.c_double(time), f, p, C.byref(c))
        if rc != 0:
            raise RuntimeError('trajectory lookup failed at t=%g (error bits %d)' % (time, rc))   # the reference throws std::runtime_error
        return np.array(f[:]), np.array(p[:]), bool(c.value)

    def get_force(self, ee, time):                 # Trajectory::GetForce (trajectory.cpp:395-402)
        return self._eval(ee, time)[0]

    def get_end_effector_location(self, ee, time):     # Trajectory::GetEndEffectorLocation (trajectory.cpp:404-410)
        return self._eval(ee, time)[1]

    def get_contacts(self, time):                  # Trajectory::GetContacts
        return [self._eval(ee, time)[2] for ee in range(4)]

    def get_contact_times(self):                   # Trajectory::GetContactTimes: per foot the times of the LO / TD knots
        out = []
        for ee in range(4):
            out.append([self.knot_time[ee][k] for k in range(self.nk[ee]) if self.knot_kind[ee][k] <= 1])
        return out


def build(force=False):
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU).  Serialised by a file lock: the ranks of a multi-GPU run that
    find a library missing (a fresh box) must not run `make` on the same tree at the same time."""
    if force or not (os.path.exists(LIB_PATH) and os.path.exists(LIB_PATH_LARGE)):
        import fcntl
        with open(os.path.join(HERE, 'csrc', '.build.lock'), 'w') as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            try:
                subprocess.check_call(['make', '-s', '-j2', '-C', os.path.join(HERE, 'csrc')])
            finally:
                fcntl.flock(lk, fcntl.LOCK_UN)
    return LIB_PATH


_libs = {}


def lib(large=False):
    path = LIB_PATH_LARGE if large else LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError('%s is not built (run __graft_entry__.build()); there is no CPU fallback' % os.path.basename(path))
        L = C.CDLL(path)
        L.srbm_last_error.restype = C.c_char_p
        L.srbm_stream.restype = C.c_void_p
        L.srbm_bytes_per_instance.restype = C.c_long
        cap = (C.c_int * 4)()
        L.srbm_get_capacity(cap)
        L.capacity = dict(N=cap[0], nu=cap[1], samples=cap[2], knots=cap[3])
        _libs[path] = L
    return _libs[path]


def load_config(name='a1_configuration', **overrides):
    cfg = json.load(open(os.path.join(CONFIG_DIR, name + '.json')))
    cfg.update(overrides)
    return cfg


def load_yaml_config(yaml_path, model_constants):
    """The reference's own configuration files (apps/*.yaml, keys as parsed in /root/reference/test/simulation_mpc.cpp:55-89
    and controllers/mpc_controller.cpp:27-67) -> the cfg dict of BatchMPC.  model_constants: dict with 'mass', 'Ir' (3x3)
    and 'hip_xy' (FL FR RL RR) -- what the reference asks pinocchio for at construction."""
    import yaml
    y = yaml.safe_load(open(yaml_path))
    keys = ['num_nodes', 'integrator_dt', 'friction_coef', 'force_bound', 'swing_height', 'foot_offset', 'ee_box_size', 'force_cost',
            'Q_srbd_diag', 'srb_init']
    missing = [k for k in keys if k not in y]
    if missing:
        raise KeyError('%s: missing keys %s' % (yaml_path, missing))
    cfg = {k: y[k] for k in keys}
    if 'srb_target' in y:
        cfg['srb_target'] = y['srb_target']
    else:       # configurations that predate srb_target give x_des / y_des (apps/a1_gait_opt_config.yaml:126-129)
        tgt = list(y['srb_init']); tgt[0] = y.get('x_des', tgt[0]); tgt[1] = y.get('y_des', tgt[1]); cfg['srb_target'] = tgt
    cfg['gait_opt_freq'] = y.get('gait_opt_freq', 5)
    cfg['mass'] = float(model_constants['mass'])
    cfg['Ir'] = np.asarray(model_constants['Ir'], float).reshape(3, 3).tolist()
    cfg['hip_xy'] = np.asarray(model_constants['hip_xy'], float).reshape(4, 2).tolist()
    cfg['source'] = os.path.basename(yaml_path)
    return cfg


SOLVE_TYPE_NAMES = {0: 'Solved', 1: 'Solved Inacc', 2: 'Max Iter', 3: 'P - Infeasible', 4: 'D - Infeasible', 5: 'P - Infeasible Inacc',
                    6: 'D - Infeasible Inacc', 7: 'Unsolved'}      # MPC::PrintStatLineToFile, mpc.cpp:944-972


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def quat_log3(q):
    """log of a unit quaternion (xyzw) -- closed form of pinocchio::quaternion::log3"""
    v = np.asarray(q[:3], float)
    n = np.linalg.norm(v)
    w = q[3]
    if n < 1e-8:
        return (2.0 / w) * (1.0 - n * n / (3.0 * w * w)) * v
    th = 2.0 * np.arctan2(n, w) if w >= 0 else -2.0 * np.arctan2(n, -w)
    return th / n * v


def manifold_to_tangent(s13):
    """SingleRigidBodyModel::ConvertManifoldStateToTangentState through the library's own host function (srbm_convert_manifold_to_tangent: no GPU
    needed), so that a Python caller and a C++ caller of the facade hand the MPC the same bits"""
    a = np.ascontiguousarray(s13, dtype=np.float64)
    t = np.zeros(12)
    if lib().srbm_convert_manifold_to_tangent(_d(a), _d(t)) != 0:
        raise RuntimeError('srbm: ' + lib().srbm_last_error().decode())
    return t


class BatchMPC:
    """batch x mpc::MPCSingleRigidBody on one MI355X."""

    def __init__(self, cfg, batch, device=0, large=None):
        self.N = int(cfg['num_nodes'])
        # horizons beyond the standard build's 50 nodes (or an explicit request) go to the LARGE-capacity build
        self.large = bool(cfg.get('large', self.N > lib().capacity['N'])) if large is None else bool(large)
        self.L = lib(self.large)
        self.NUMAX, self.NSMAX = self.L.capacity['nu'], self.L.capacity['samples']
        self.cfg = cfg
        self.batch = int(batch)
        info = MPCInfo()
        info.num_nodes = self.N
        info.integrator_dt = cfg['integrator_dt']
        info.friction_coef = cfg['friction_coef']
        info.force_bound = cfg['force_bound']
        info.swing_height = cfg['swing_height']
        info.foot_offset = cfg['foot_offset']
        info.ee_box_size[:] = [float(v) for v in cfg['ee_box_size']]
        info.force_cost = cfg['force_cost']
        model = Model()
        model.mass = cfg['mass']
        model.Ir[:] = list(np.asarray(cfg['Ir'], float).reshape(-1))
        model.hip_xy[:] = list(np.asarray(cfg['hip_xy'], float).reshape(-1))
        self.h = C.c_void_p()
        self._chk(self.L.srbm_batch_create(C.byref(self.h), self.batch, C.byref(info), C.byref(model), int(device)))
        # cost set-up exactly as the caller does it: /root/reference/controllers/mpc_controller.cpp:57-67
        Q = np.diag(np.asarray(cfg['Q_srbd_diag'], float))
        des = manifold_to_tangent(cfg['srb_target'])
        self.add_quadratic_tracking_cost(des, Q)
        self.set_quadratic_final_cost(Q)
        self.set_linear_final_cost(-1 * Q @ des)
        if 'leg_origins' in cfg:            # leg geometry for the whole-body targets (row f3)
            lo = np.ascontiguousarray(cfg['leg_origins'], dtype=np.float64).reshape(4, 4, 3)
            self._chk(self.L.srbm_set_leg_kinematics(self.h, _d(lo)))
        if 'body_model' in cfg and 'torque_bounds' in cfg:        # whole-body QP of the low-level controller (row f3)
            w = WbcModel()
            for b, body in enumerate(cfg['body_model']):
                w.body_mass[b] = body['mass']
                w.body_com[b][:] = body['com']
                w.body_inertia[b][:] = list(np.asarray(body['inertia'], float).reshape(-1))
            w.torque_bounds[:] = [float(x) for x in cfg['torque_bounds']]
            w.kp_joint_gains[:] = [float(x) for x in cfg['kp_joint_gains']]; w.kd_joint_gains[:] = [float(x) for x in cfg['kd_joint_gains']]
            w.base_pos_gains[:] = [float(x) for x in cfg['base_pos_gains']]; w.base_ang_gains[:] = [float(x) for x in cfg['base_ang_gains']]
            w.leg_tracking_weight = cfg['leg_tracking_weight']; w.torso_tracking_weight = cfg['torso_tracking_weight']
            w.force_tracking_weight = cfg['force_tracking_weight']; w.friction_coef = cfg['friction_coef']; w.max_grf = cfg['force_bound']
            self._chk(self.L.srbm_set_wbc_model(self.h, C.byref(w)))

    def close(self):
        """srbm_batch_destroy.  A batch that gait handles still borrow is NOT released by the library (return code -1): the handle is kept, so that
        it is released once the optimiser is gone, instead of leaking the device batch silently"""
        if self.h:
            for g in list(getattr(self, '_gait_handles', [])):        # optimisers created on this batch go first
                g.close()
            if self.L.srbm_batch_destroy(self.h) == 0:
                self.h = C.c_void_p()
            else:
                import warnings
                warnings.warn('srbm_batch_destroy refused: ' + self.L.srbm_last_error().decode())

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError('srbm: ' + self.L.srbm_last_error().decode())

    def clone(self):
        """MPC copy constructor (mpc.cpp:1133-1181): a deep copy with its own stream and device buffers"""
        c = object.__new__(BatchMPC)
        c.L, c.cfg, c.batch, c.N, c.large, c.NUMAX, c.NSMAX = self.L, self.cfg, self.batch, self.N, self.large, self.NUMAX, self.NSMAX
        c.h = C.c_void_p()
        self._chk(self.L.srbm_batch_clone(self.h, C.byref(c.h)))
        return c

    # ---- mpc::Trajectory in and out ----
    def get_trajectory(self, first=0, count=None):
        """MPC::GetTrajectory for instances [first, first + count): a ctypes array of Trajectory records"""
        count = self.batch - first if count is None else count
        arr = (Trajectory * count)()
        self._chk(self.L.srbm_get_trajectory(self.h, int(first), int(count), arr))
        return arr

    def set_warm_start_trajectory(self, trajs, first=0):
        """MPC::SetWarmStartTrajectory (mpc.cpp:110-119); trajs: ctypes array (or list) of Trajectory records"""
        if not isinstance(trajs, C.Array):
            trajs = (Trajectory * len(trajs))(*trajs)
        self._chk(self.L.srbm_set_warm_start_trajectory(self.h, int(first), len(trajs), trajs))

    def eval_trajectory(self, time):
        """Trajectory::GetForce / GetEndEffectorLocation / GetContacts of every instance's current trajectory at time[batch]"""
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(time, dtype=np.float64), (self.batch,)))
        f = np.zeros((self.batch, 4, 3)); p = np.zeros((self.batch, 4, 3)); c = np.zeros((self.batch, 4), np.int32)
        self._chk(self.L.srbm_eval_trajectory(self.h, _d(t), _d(f), _d(p), _i(c)))
        return f, p, c

    def ee_box_center(self):
        a = np.zeros((4, 2))
        self._chk(self.L.srbm_get_ee_box_center(self.h, _d(a)))
        return a

    def cost(self):
        a = np.zeros(self.batch)
        self._chk(self.L.srbm_get_cost(self.h, _d(a)))
        return a

    def merit(self):
        m = np.zeros(self.batch); d = np.zeros(self.batch)
        self._chk(self.L.srbm_get_merit(self.h, _d(m), _d(d)))
        return m, d

    def add_force_cost(self, weight):
        self._chk(self.L.srbm_add_force_cost(self.h, C.c_double(weight)))

    def avg_cost(self):
        a = np.zeros(self.batch)
        self._chk(self.L.srbm_get_avg_cost(self.h, _d(a)))
        return a

    def status_accumulated(self):
        """sticky accumulators over all solves since the last clear: [batch][4] = error bits, solves, not-solved, of those MaxIter"""
        a = np.zeros((self.batch, 4), np.int32)
        self._chk(self.L.srbm_get_status_accumulated(self.h, _i(a)))
        return a

    def clear_status_accumulators(self):
        self._chk(self.L.srbm_clear_status_accumulators(self.h))

    def executed_mfma(self):
        v = C.c_double(0)
        self._chk(self.L.srbm_get_executed_mfma(self.h, C.byref(v)))
        return v.value

    def result_record_doubles(self):
        return int(self.L.srbm_result_record_doubles(self.N))

    # ---- set-up (mpc.h:92-110) ----
    def add_quadratic_tracking_cost(self, state_des12, Q):
        a = np.ascontiguousarray(state_des12, dtype=np.float64); q = np.ascontiguousarray(Q, dtype=np.float64)
        self._chk(self.L.srbm_add_quadratic_tracking_cost(self.h, _d(a), _d(q)))

    def set_quadratic_final_cost(self, Phi):
        q = np.ascontiguousarray(Phi, dtype=np.float64)
        self._chk(self.L.srbm_set_quadratic_final_cost(self.h, _d(q)))

    def set_linear_final_cost(self, w):
        a = np.ascontiguousarray(w, dtype=np.float64)
        self._chk(self.L.srbm_set_linear_final_cost(self.h, _d(a)))

    def set_solver_step_rule(self, tol_step, start_mu=0.0):
        self._chk(self.L.srbm_set_solver_step_rule(self.h, C.c_double(tol_step), C.c_double(start_mu)))

    def enable_fast_termination(self, start_mu=FAST_START_MU):
        """opt into the step rule (and, for srbm_rti_advance, the lower-start attempt) at the values the bench line is taken with"""
        self.set_solver_step_rule(FAST_TOL_STEP, start_mu)

    def enable_lower_start(self, start_mu=FAST_START_MU):
        """the lower starting point alone: every solve still ends by the reference's gap criterion (tol_step 0), srbm_rti_advance first attempts it
        from the linearisation point"""
        self.set_solver_step_rule(0.0, start_mu)

    def solve_flags(self):
        """per instance, of the LAST solve: bit 0 ended through the step rule, bit 1 began with a lower-start attempt, bit 2 the attempt was repeated"""
        f = np.zeros(self.batch, np.int32)
        self._chk(self.L.srbm_get_solve_flags(self.h, _i(f)))
        return f

    def solver_step_rule(self):
        a = C.c_double(0); b = C.c_double(0)
        self._chk(self.L.srbm_get_solver_step_rule(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def solver_counters(self):
        c = (C.c_longlong * 4)()
        self._chk(self.L.srbm_get_solver_counters(self.h, c))
        return dict(solves=c[0], step_rule=c[1], low_tried=c[2], low_failed=c[3])

    def set_state_trajectory_warm_start(self, states):
        a = self._bcast(states, 13)
        self._chk(self.L.srbm_set_state_trajectory_warm_start(self.h, _d(a)))

    def set_solver_tolerances(self, gap_abs, gap_rel, feas, max_iter=200):
        self._chk(self.L.srbm_set_solver_tolerances(self.h, C.c_double(gap_abs), C.c_double(gap_rel), C.c_double(feas), int(max_iter)))

    def _bcast(self, a, width):
        a = np.asarray(a, dtype=np.float64)
        if a.size == width:
            a = np.tile(a.reshape(1, width), (self.batch, 1))
        return np.ascontiguousarray(a.reshape(self.batch, width))

    # ---- solves ----
    def create_initial_run(self, state, ee):
        s = self._bcast(state, 13); e = self._bcast(ee, 12)
        self._chk(self.L.srbm_create_initial_run(self.h, _d(s), _d(e)))

    def get_real_time_update(self, state, init_time, ee):
        s = self._bcast(state, 13); e = self._bcast(ee, 12)
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(init_time, dtype=np.float64), (self.batch,)))
        self._chk(self.L.srbm_get_real_time_update(self.h, _d(s), _d(t), _d(e)))

    def get_real_time_update_dev(self, state_ptr, time_ptr, ee_ptr):
        self._chk(self.L.srbm_get_real_time_update_dev(self.h, C.c_void_p(state_ptr), C.c_void_p(time_ptr), C.c_void_p(ee_ptr)))

    def rti_advance(self, first_index, steps):
        self._chk(self.L.srbm_rti_advance(self.h, int(first_index), int(steps)))

    def rti_advance_unfused(self, first_index, steps):
        self._chk(self.L.srbm_rti_advance_unfused(self.h, int(first_index), int(steps)))

    # ---- closed-loop rollout harness (include/srbm_rti.h: srbm_plant_*, srbm_closed_loop_advance) ----
    def plant_set_state(self, state):
        self._chk(self.L.srbm_plant_set_state(self.h, _d(self._bcast(state, 13))))

    def plant_state(self):
        out = np.zeros((self.batch, 13))
        self._chk(self.L.srbm_plant_get_state(self.h, _d(out)))
        return out

    def plant_set_push(self, time=None, impulse=None):
        """one push per instance: lin-mom += impulse[:3], ang-mom += impulse[3:] when the plant passes `time`; None clears"""
        if time is None:
            self._chk(self.L.srbm_plant_set_push(self.h, None, None))
        else:
            self._chk(self.L.srbm_plant_set_push(self.h, _d(self._bcast(time, 1)), _d(self._bcast(impulse, 6))))

    def closed_loop_advance(self, first_index, steps, substeps=1, advance_time=False):
        self._chk(self.L.srbm_closed_loop_advance(self.h, int(first_index), int(steps), int(substeps), int(bool(advance_time))))

    def synchronize(self):
        self._chk(self.L.srbm_synchronize(self.h))

    def stream(self):
        return self.L.srbm_stream(self.h)

    def update_contact_times(self, times):
        a = np.ascontiguousarray(times, dtype=np.float64)
        assert a.ndim == 3 and a.shape[0] == self.batch and a.shape[1] == 4
        self._chk(self.L.srbm_update_contact_times(self.h, _d(a), a.shape[2]))

    def adjust_for_current_contacts(self, time, in_contact):
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(time, dtype=np.float64), (self.batch,)))
        c = np.ascontiguousarray(np.broadcast_to(np.asarray(in_contact, dtype=np.int32), (self.batch, 4)))
        self._chk(self.L.srbm_adjust_for_current_contacts(self.h, _d(t), _i(c)))

    # ---- trajectory -> whole-body targets ----
    def forward_kinematics(self, q):
        """SingleRigidBodyModel::GetEndEffectorLocations: q[batch][19] -> [batch][4][3]"""
        qq = self._bcast(q, 19); ee = np.zeros((self.batch, 4, 3))
        self._chk(self.L.srbm_forward_kinematics(self.h, _d(qq), _d(ee)))
        return ee

    def inverse_kinematics(self, state, ee, q_guess):
        """SingleRigidBodyModel::InverseKinematics for every instance -> (q [batch][19], iterations [batch][4], status [batch])"""
        s = self._bcast(state, 13); e = self._bcast(ee, 12); g = self._bcast(q_guess, 19)
        q = np.zeros((self.batch, 19)); it = np.zeros((self.batch, 4), np.int32); st = np.zeros(self.batch, np.int32)
        self._chk(self.L.srbm_inverse_kinematics(self.h, _d(s), _d(e), _d(g), _d(q), _i(it), _i(st)))
        return q, it, st

    def get_targets_from_traj_dev(self, time_ptr, q_des_ptr, v_des_ptr, force_des_ptr, status_ptr):
        """the same on device pointers (ints, e.g. torch tensors' data_ptr()): one launch on the batch's stream, no copy, no synchronisation"""
        self._chk(self.L.srbm_get_targets_from_traj_dev(self.h, C.c_void_p(time_ptr), C.c_void_p(q_des_ptr), C.c_void_p(v_des_ptr), C.c_void_p(force_des_ptr), C.c_void_p(status_ptr)))

    def eval_trajectory_dev(self, time_ptr, force_ptr, pos_ptr, in_contact_ptr):
        """Trajectory::GetForce / GetEndEffectorLocation / contact flags on device pointers (one launch, no copy)"""
        self._chk(self.L.srbm_eval_trajectory_dev(self.h, C.c_void_p(time_ptr), C.c_void_p(force_ptr), C.c_void_p(pos_ptr), C.c_void_p(in_contact_ptr)))

    def qp_control_dev(self, q, v, contact, q_des, v_des, force_des, control, qp_sol, status):
        self._chk(self.L.srbm_qp_control_dev(self.h, *[C.c_void_p(p) for p in (q, v, contact, q_des, v_des, force_des, control, qp_sol, status)]))

    def get_targets_from_traj(self, time, q_des):
        """MPCController::GetTargetsFromTraj on the current trajectories -> (q_des, v_des [batch][18], force_des [batch][4][3], status)"""
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(time, dtype=np.float64), (self.batch,)))
        q = self._bcast(q_des, 19).copy(); v = np.zeros((self.batch, 18)); f = np.zeros((self.batch, 4, 3)); st = np.zeros(self.batch, np.int32)
        self._chk(self.L.srbm_get_targets_from_traj(self.h, _d(t), _d(q), _d(v), _d(f), _i(st)))
        return q, v, f, st

    def qp_control(self, q, v, contact, q_des, v_des, force_des, dump=False):
        """QPControl::ComputeControlAction for every instance -> (control [batch][36], qp_sol [batch][30], status, iterations[, QP dump])"""
        B = self.batch
        qq = self._bcast(q, 19); vv = self._bcast(v, 18); qd = self._bcast(q_des, 19); vd = self._bcast(v_des, 18); fd = self._bcast(force_des, 12)
        con = np.ascontiguousarray(np.broadcast_to(np.asarray(contact, dtype=np.int32), (B, 4)))
        ctl = np.zeros((B, 36)); sol = np.zeros((B, 30)); st = np.zeros(B, np.int32)
        dmp = np.zeros((B, 50 * 30 + 100 + 60)) if dump else None
        self._chk(self.L.srbm_qp_control(self.h, _d(qq), _d(vv), _i(con), _d(qd), _d(vd), _d(fd), _d(ctl), _d(sol), _i(st), _d(dmp) if dump else None))
        out = (ctl, sol, st & 255, st >> 8)
        if dump:
            A = dmp[:, :1500].reshape(B, 50, 30)
            out += (dict(A=A, lb=dmp[:, 1500:1550], ub=dmp[:, 1550:1600], P=dmp[:, 1600:1630], w=dmp[:, 1630:1660]),)
        return out

    # ---- the reference's statistics log ----
    def print_stat_header(self, fh):
        """header block of MPC::PrintStatLineToFile (mpc.cpp:901-939), same field widths.  (std::left is set on the stream while
        the header is written and stays set: every column of the table is LEFT aligned in its 15 characters.)"""
        cw, tw = 15, 150
        c = self.cfg
        import time as _time
        fh.write('-' * tw + '\n' + ' ' * (tw // 2 - 7) + 'MPC Statistics\n')
        fh.write('MPC started at: ' + _time.ctime() + '\n')                     # std::ctime of the wall clock (mpc.cpp:910-915)
        fh.write('Number of nodes: %d\nMPC time step: %g\nForce bounds: %g\nEnd Effector box size: %g %g\n' %
                 (self.N, c['integrator_dt'], c['force_bound'], c['ee_box_size'][0], c['ee_box_size'][1]))
        fh.write('Force cost: %g\nFoot offset: %g\nSwing height: %g\n' % (c['force_cost'], c['foot_offset'], c['swing_height']))
        fh.write('-' * tw + '\n')
        cols = ['Solve #', 'Time (ms)', 'Constraints', 'Step Norm', 'Alpha', 'Cost', 'Merit', 'Merit dd', 'Solve Type', 'QP Cost']
        fh.write(''.join(n.ljust(cw) for n in cols) + '\n' + '-' * tw + '\n')

    def print_stat_line(self, fh, solve_number, time_ms, inst=0):
        """one table row of MPC::PrintStatLineToFile (mpc.cpp:974-989) for instance `inst` from the last solve.  Merit =
        cost + mu * L1 dynamics defect (mpc.cpp:749-757, mu = 5000), Merit dd = its directional derivative along the step."""
        cw = 15
        st, err = self.status()
        s = self.stats()[inst]
        merit, merit_dd = self.merit()
        vals = ['%d' % solve_number, '%g' % time_ms, '%g' % s[2], '%g' % s[3], '%g' % s[0], '%g' % s[1], '%g' % merit[inst], '%g' % merit_dd[inst],
                SOLVE_TYPE_NAMES.get(int(st[inst]), 'Other'), '%g' % s[1]]       # last column: cost_ = GetCostValue(prev_qp_sol), the same value as 'Cost' (mpc.cpp:809, msrb.cpp:183-184)
        fh.write(''.join(v.ljust(cw) for v in vals) + '\n')

    # ---- measurement aids ----
    def enable_kernel_timing(self, max_launches):
        self._chk(self.L.srbm_enable_kernel_timing(self.h, int(max_launches)))

    def kernel_timing(self):
        ms = C.c_double(0); n = C.c_int(0)
        self._chk(self.L.srbm_get_kernel_timing(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_timings(self, max_launches=64):
        ms = np.zeros(max_launches); n = C.c_int(0)
        self._chk(self.L.srbm_get_kernel_timings(self.h, _d(ms), int(max_launches), C.byref(n)))
        return ms[:min(n.value, max_launches)].copy()

    def work_counters(self):
        it = C.c_double(0); fl = C.c_double(0)
        self._chk(self.L.srbm_get_work_counters(self.h, C.byref(it), C.byref(fl)))
        return it.value, fl.value

    def pack_results_dev(self, ptr, ld):
        self._chk(self.L.srbm_pack_results_dev(self.h, C.c_void_p(ptr), int(ld)))

    # ---- multi-GPU: RCCL all-gather of the result records through the C-ABI (include/srbm_rti.h: srbm_allgather_results) ----
    def rccl_unique_id(self):
        """ncclGetUniqueId (one rank calls it, the 128 bytes travel to the others by the host's own rendezvous)"""
        buf = (C.c_ubyte * 128)()
        self._chk(self.L.srbm_rccl_get_unique_id(buf))
        return bytes(buf)

    def rccl_comm_init_rank(self, world, rank, id_bytes):
        """ncclCommInitRank on the batch's device; returns the ncclComm_t as an integer handle"""
        assert len(id_bytes) == 128
        comm = C.c_void_p(0)
        buf = (C.c_ubyte * 128).from_buffer_copy(id_bytes)
        self._chk(self.L.srbm_rccl_comm_init_rank(self.h, int(world), int(rank), buf, C.byref(comm)))
        return comm.value

    def rccl_comm_destroy(self, comm):
        self._chk(self.L.srbm_rccl_comm_destroy(C.c_void_p(comm)))

    def allgather_results(self, comm, out_ptr):
        """this rank's result records packed into their slot of out[world * batch][record_doubles] (device pointer) and ONE in-place ncclAllGather
        on the batch's stream; asynchronous (synchronize() before reading)"""
        self._chk(self.L.srbm_allgather_results(self.h, C.c_void_p(comm), C.c_void_p(out_ptr)))

    def pack_results(self):
        """the result records of srbm_pack_results_dev in a host array [batch][srbm_result_record_doubles(N)]"""
        ld = self.result_record_doubles()
        a = np.zeros((self.batch, ld))
        self._chk(self.L.srbm_pack_results(self.h, _d(a), ld))
        return a

    # ---- results ----
    def sizes(self):
        a = np.zeros((self.batch, 8), np.int32)
        self._chk(self.L.srbm_get_sizes(self.h, _i(a)))
        return a

    def status(self):
        s = np.zeros(self.batch, np.int32); e = np.zeros(self.batch, np.int32)
        self._chk(self.L.srbm_get_status(self.h, _i(s), _i(e)))
        return s, e

    def stats(self):
        a = np.zeros((self.batch, 8))
        self._chk(self.L.srbm_get_stats(self.h, _d(a)))
        return a

    def qp_cost(self):
        a = np.zeros(self.batch)
        self._chk(self.L.srbm_get_qp_cost(self.h, _d(a)))
        return a

    def qp_solution(self):
        ld = (self.N + 1) * 12 + self.NUMAX
        a = np.zeros((self.batch, ld))
        self._chk(self.L.srbm_get_qp_solution(self.h, _d(a), ld))
        return a

    def raw_qp_minimiser(self):
        ld = (self.N + 1) * 12 + self.NUMAX
        a = np.zeros((self.batch, ld))
        self._chk(self.L.srbm_get_raw_qp_minimiser(self.h, _d(a), ld))
        return a

    def dual_solution(self):
        ld = (self.N + 1) * 12 + 6 * self.NSMAX + 16 * (self.N - 3) + 16
        z = np.zeros((self.batch, ld)); s = np.zeros((self.batch, ld))
        self._chk(self.L.srbm_get_dual_solution(self.h, _d(z), _d(s), ld))
        return z, s

    def trajectory_states(self):
        a = np.zeros((self.batch, self.N + 1, 13))
        self._chk(self.L.srbm_get_trajectory_states(self.h, _d(a)))
        return a

    def knots(self, inst):
        t = np.zeros((4, 32)); kd = np.zeros((4, 32), np.int32); nk = np.zeros(4, np.int32)
        fv = np.zeros((4, 3, 32, 2)); pv = np.zeros((4, 2, 32)); box = np.zeros(2)
        self._chk(self.L.srbm_get_knots(self.h, int(inst), _d(t), _i(kd), _i(nk), _d(fv), _d(pv), _d(box)))
        return dict(times=t, kinds=kd, nk=nk, fvals=fv, pvals=pv, box=box)

    def export_qp(self, inst):
        sz = self.sizes()[inst]
        n, m = int(sz[0]), int(sz[1])
        A = np.zeros((m, n)); b = np.zeros(m); P = np.zeros((n, n)); q = np.zeros(n)
        self._chk(self.L.srbm_export_qp(self.h, int(inst), _d(A), _d(b), _d(P), _d(q)))
        return A, b, P, q

    def param_partials(self, inst, ee, idx):
        """MPCSingleRigidBody::ComputeParamPartialsClarabel (mpc_single_rigid_body.cpp:642-792) as dense matrices: (dA, dG, db, dh) of the QP
        of the last solve w.r.t. contact time `idx` of foot `ee`, evaluated on the current trajectory of instance `inst`."""
        sz = self.sizes()[inst]
        n, me, mi = int(sz[0]), int(sz[2]), int(sz[3])
        dA = np.zeros((me, n)); dG = np.zeros((mi, n)); db = np.zeros(me); dh = np.zeros(mi)
        self._chk(self.L.srbm_gait_get_param_partials(self.h, int(inst), int(ee), int(idx), _d(dA), _d(dG), _d(db), _d(dh)))
        return dA, dG, db, dh


class BatchGaitOptimizer:
    """mpc::GaitOptimizer (/root/reference/mpc/include/gait_optimizer.h) for every instance of a BatchMPC: same method
    names in snake_case, contact-time vectors as [batch][32] rows (foot after foot, `counts` entries per foot)."""
    NV = 32
    LS_SIZE = 10

    def __init__(self, mpc):
        self.mpc = mpc
        self.L = mpc.L
        self.g = C.c_void_p()
        mpc._chk(self.L.srbm_gait_create(mpc.h, C.byref(self.g)))
        import weakref
        if not hasattr(mpc, '_gait_handles'):
            mpc._gait_handles = weakref.WeakSet()
        mpc._gait_handles.add(self)          # BatchMPC.close() releases the optimisers that borrow it first

    def close(self):
        if self.g:
            self.L.srbm_gait_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_contact_times_from_trajectory(self):
        self.mpc._chk(self.L.srbm_gait_set_contact_times_from_trajectory(self.g))

    def contact_times(self):
        xk = np.zeros((self.mpc.batch, self.NV)); counts = np.zeros((self.mpc.batch, 4), np.int32)
        self.mpc._chk(self.L.srbm_gait_get_contact_times(self.g, _d(xk), _i(counts)))
        return xk, counts

    def compute_sensitivity(self):
        self.mpc._chk(self.L.srbm_gait_compute_sensitivity(self.g))

    def sensitivity(self):
        m = self.mpc
        ld = (m.N + 1) * 12 + m.NUMAX + 6 * m.NSMAX + 16 * (m.N - 3) + (m.N + 1) * 12 + 16
        d = np.zeros((m.batch, ld))
        m._chk(self.L.srbm_gait_get_sensitivity(self.g, _d(d), ld))
        return d

    def compute_gradient(self):
        self.mpc._chk(self.L.srbm_gait_compute_gradient(self.g))

    def gradient(self):
        m = self.mpc
        g = np.zeros((m.batch, self.NV)); valid = np.zeros(m.batch, np.int32)
        m._chk(self.L.srbm_gait_get_gradient(self.g, _d(g), _i(valid)))
        return g, valid

    def optimize_contact_times(self, time):
        m = self.mpc
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(time, dtype=np.float64), (m.batch,)))
        m._chk(self.L.srbm_gait_optimize_contact_times(self.g, _d(t)))

    def lp_result(self):
        m = self.mpc
        st = np.zeros(m.batch, np.int32); pr = np.zeros(m.batch)
        m._chk(self.L.srbm_gait_get_lp_result(self.g, _i(st), _d(pr)))
        return st, pr

    def rti_advance(self, first_run_num, steps, gait_opt_freq):
        self.mpc._chk(self.L.srbm_gait_rti_advance(self.g, int(first_run_num), int(steps), int(gait_opt_freq)))

    def set_step(self, step):
        a = np.zeros((self.mpc.batch, self.NV))
        st = np.asarray(step, dtype=np.float64)
        a[:, :st.shape[-1]] = st
        self.mpc._chk(self.L.srbm_gait_set_step(self.g, _d(a)))

    def step(self):
        a = np.zeros((self.mpc.batch, self.NV))
        self.mpc._chk(self.L.srbm_gait_get_step(self.g, _d(a)))
        return a

    def line_search(self, state, init_time, ee):
        m = self.mpc
        s = m._bcast(state, 13); e = m._bcast(ee, 12)
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(init_time, dtype=np.float64), (m.batch,)))
        imin = np.zeros(m.batch, np.int32); costs = np.zeros((m.batch, self.LS_SIZE))
        m._chk(self.L.srbm_gait_line_search(self.g, _d(s), _d(t), _d(e), _i(imin), _d(costs)))
        return imin, costs

    def candidates(self):
        """the candidate batch of the last line search as a BORROWED BatchMPC view (read-back entries only): candidate c of instance b at b * 10 + c"""
        self.L.srbm_gait_debug_candidates.restype = C.c_void_p
        v = object.__new__(BatchMPC)
        m = self.mpc
        v.N, v.large, v.L, v.NUMAX, v.NSMAX, v.cfg = m.N, m.large, m.L, m.NUMAX, m.NSMAX, m.cfg
        v.batch = m.batch * self.LS_SIZE
        v.h = C.c_void_p(self.L.srbm_gait_debug_candidates(self.g))
        v.close = lambda: None                      # not ours to destroy
        return v

    def candidate_status(self):
        n = self.mpc.batch * self.LS_SIZE
        st = np.zeros(n, np.int32); err = np.zeros(n, np.int32)
        self.mpc._chk(self.L.srbm_gait_get_candidate_status(self.g, _i(st), _i(err)))
        return st.reshape(-1, self.LS_SIZE), err.reshape(-1, self.LS_SIZE)
