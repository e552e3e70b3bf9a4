"""The synthetic workloads of SURVEY.md section 8(d) -- BASELINE.json's configurations B, C, D as seeded instance generators -- and the sharding of
a global batch over ranks (section 8e).  Used by bench.py, by the parity tests (the same seeded instances) and by the developer scripts."""
import numpy as np


def shard_range(total, rank, world):
    """contiguous block [lo, hi) of `total` instances owned by `rank`"""
    per = total // world
    rem = total % world
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)



def config_b_instance(cfg, b):
    """instance b of Config B: perturbed initial state + foot positions"""
    rng = np.random.Generator(np.random.MT19937(20240112 + b))
    u = lambda lo, hi: lo + (hi - lo) * rng.random()
    m = cfg['mass']
    p = np.array([u(-0.02, 0.02), u(-0.02, 0.02), 0.30 + u(-0.01, 0.01)])
    v = np.array([u(-0.5, 0.5), u(-0.5, 0.5), u(-0.1, 0.1)])
    rpy = np.array([u(-0.05, 0.05), u(-0.05, 0.05), u(-0.05, 0.05)])
    L = np.array([u(-0.1, 0.1), u(-0.1, 0.1), u(-0.1, 0.1)])
    th = np.linalg.norm(rpy)
    quat = np.concatenate([np.sin(th / 2) / th * rpy, [np.cos(th / 2)]])
    state = np.concatenate([p, m * v, quat, L])
    hips = np.array([[0.2055, 0.147], [0.2055, -0.147], [-0.1555, 0.147], [-0.1555, -0.147]])
    ee = np.zeros((4, 3))
    for e in range(4):
        ee[e, 0] = p[0] + hips[e, 0] + u(-0.02, 0.02)
        ee[e, 1] = p[1] + hips[e, 1] + u(-0.02, 0.02)
    return state, ee


def config_c_instance(cfg, b):
    """instance b of Config C: Config B's perturbations around the srb_init of apps/a1_gait_opt_config.yaml
    (height 0.34); the target of that file is x_des = y_des = 1"""
    state, ee = config_b_instance(cfg, b)
    state[2] += cfg['srb_init'][2] - 0.30
    return state, ee


def config_d_instance(cfg, b):
    """instance b of Config D: apps/a1_config_distr_rejection.yaml values (N=50, dt=0.02); the file's
    single push becomes a distribution -- lin-mom xy ~ N(0, 2.5^2) truncated at 3 sigma, yaw ang-mom ~ N(0, 0.2^2), seed 777 + b"""
    rng = np.random.Generator(np.random.MT19937(777 + b))
    def tnorm(sig):
        while True:
            v = rng.normal(0.0, sig)
            if abs(v) <= 3 * sig:
                return v
    state = np.array(cfg['srb_init'], float)
    state[3] += tnorm(2.5); state[4] += tnorm(2.5)
    state[12] += rng.normal(0.0, 0.2)
    ee = np.array([[0.2, 0.2, 0], [0.2, -0.2, 0], [-0.2, 0.2, 0], [-0.2, -0.2, 0]], float)      # test/simulation_mpc.cpp:104-108
    return state, ee
