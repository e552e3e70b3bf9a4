#!/usr/bin/env python3
"""Extract the hot-path keys of the reference's YAML configs (SURVEY.md section 5 "Config / flags") and the A1
constants (tests/golden/a1_constants_*.json) into small JSON files the product and the tests can read on a machine
that has no /root/reference.   usage: python oracle/tools/make_configs.py /root/reference"""
import json, os, sys, yaml
ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.join(here, '..', '..')
out_dir = os.path.join(root, 'bilevel-gait-gen_amd', 'configs')
KEYS = ['num_nodes', 'integrator_dt', 'friction_coef', 'force_bound', 'swing_height', 'foot_offset', 'ee_box_size',
        'force_cost', 'Q_srbd_diag', 'srb_init', 'srb_target', 'gait_opt_freq', 'collision_frames']
for cfg in ('a1_configuration', 'a1_gait_opt_config', 'a1_config_distr_rejection'):
    y = yaml.safe_load(open(os.path.join(ref, 'apps', cfg + '.yaml')))
    c = {k: y[k] for k in KEYS if k in y}
    if 'srb_target' not in c:   # a1_gait_opt_config.yaml predates srb_target: SURVEY.md section 8(d) Config C
        tgt = list(y['srb_init']); tgt[0] = y['x_des']; tgt[1] = y['y_des']; c['srb_target'] = tgt
    if 'gait_opt_freq' not in c:
        c['gait_opt_freq'] = 5
    k = json.load(open(os.path.join(root, 'tests', 'golden', 'a1_constants_%s.json' % cfg)))
    c['mass'] = k['mass']; c['Ir'] = k['Ir']
    c['hip_xy'] = [k['hip_xy'][n] for n in ('FL', 'FR', 'RL', 'RR')]
    c['leg_origins'] = [k['leg_origins'][n] for n in ('FL', 'FR', 'RL', 'RR')]      # row f3: hip / thigh / calf joint and foot-frame origins
    c['init_config'] = k['source']['init_config']
    c['body_model'] = k['body_model']        # row f3: trunk + 12 moving links (mass, com, inertia in the joint frame), fixed links merged
    for key in ('torque_bounds', 'base_pos_gains', 'base_ang_gains', 'kp_joint_gains', 'kd_joint_gains', 'leg_tracking_weight',
                'torso_tracking_weight', 'force_tracking_weight'):      # the whole-body QP of controllers/qp_control.cpp
        if key in y:
            c[key] = y[key]
    c['source'] = 'apps/%s.yaml + models/a1_description/urdf/a1.urdf' % cfg
    json.dump(c, open(os.path.join(out_dir, cfg + '.json'), 'w'), indent=1)
    print(cfg, {kk: c[kk] for kk in ('num_nodes', 'integrator_dt', 'friction_coef', 'force_bound')})
