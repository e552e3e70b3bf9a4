#!/usr/bin/env python3
"""ORACLE tooling (test infrastructure): derive the A1 single-rigid-body constants from the URDF *asset*.

The reference gets these numbers from pinocchio at construction time:
  * robot_mass_  = pinocchio::computeTotalMass            (/root/reference/mpc/models/model.cpp:27)
  * Ir_          = composite rigid-body rotational inertia of the whole robot at `init_config`, about the
                   whole-body COM, expressed in the floating-base frame
                   (/root/reference/mpc/models/single_rigid_body_model.cpp:33-34:
                    oMi[1].actInv(oYcrb[0]).inertia() after computeCentroidalMap)
  * hip joint origins relative to the floating base (single_rigid_body_model.cpp:258-308, GetCOMToHip)
pinocchio is not available here, so this script restates the computation (URDF kinematic tree + parallel-axis
sum).  Input data: models/a1_description/urdf/a1.urdf and the `init_config` of the YAML named on the command line.
Output: JSON written to tests/golden/a1_constants_<cfg>.json (a fixture: numbers only).

usage: python oracle/tools/a1_constants.py /root/reference
"""
import json, math, sys, os
import xml.etree.ElementTree as ET
import numpy as np
import yaml

def rpy_to_R(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx

def axis_angle_R(axis, q):
    a = np.asarray(axis, float); a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(q) * K + (1 - math.cos(q)) * (K @ K)

def quat_xyzw_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

def parse_origin(el):
    if el is None:
        return np.zeros(3), np.eye(3)
    xyz = np.array([float(v) for v in el.get('xyz', '0 0 0').split()])
    rpy = [float(v) for v in el.get('rpy', '0 0 0').split()]
    return xyz, rpy_to_R(*rpy)

def derive(urdf_path, init_config):
    root = ET.parse(urdf_path).getroot()
    links, joints = {}, []
    for l in root.findall('link'):
        inr = l.find('inertial')
        if inr is None:
            links[l.get('name')] = None
            continue
        xyz, R = parse_origin(inr.find('origin'))
        m = float(inr.find('mass').get('value'))
        i = inr.find('inertia')
        I = np.array([[float(i.get('ixx')), float(i.get('ixy')), float(i.get('ixz'))],
                      [float(i.get('ixy')), float(i.get('iyy')), float(i.get('iyz'))],
                      [float(i.get('ixz')), float(i.get('iyz')), float(i.get('izz'))]])
        links[l.get('name')] = (m, xyz, R, I)
    for j in root.findall('joint'):
        if j.find('parent') is None:
            continue   # transmission blocks reuse the <joint> tag
        xyz, R = parse_origin(j.find('origin'))
        ax = j.find('axis')
        axis = [float(v) for v in ax.get('xyz').split()] if ax is not None else [1, 0, 0]
        joints.append(dict(name=j.get('name'), type=j.get('type'), parent=j.find('parent').get('link'),
                           child=j.find('child').get('link'), xyz=xyz, R=R, axis=axis))
    children = {}
    for j in joints:
        children.setdefault(j['parent'], []).append(j)
    child_links = {j['child'] for j in joints}
    root_link = [n for n in links if n not in child_links][0]
    # actuated joints in pinocchio order: depth-first, children visited in alphabetical link order
    q_joint = {}
    order = []
    def visit(link):
        for j in sorted(children.get(link, []), key=lambda jj: jj['child']):
            if j['type'] in ('revolute', 'continuous'):
                order.append(j['name'])
            visit(j['child'])
    visit(root_link)
    assert len(order) == 12, order
    for k, name in enumerate(order):
        q_joint[name] = init_config[7 + k]
    base_p = np.array(init_config[0:3]); base_R = quat_xyzw_to_R(init_config[3:7])
    bodies = []     # (mass, com_world, I_world)
    hips = {}
    leg_origins = {}      # per leg: origins of the hip, thigh and calf joints and of the foot frame in their parent link (row f3: the IK)
    for j in joints:
        for k, suffix in enumerate(('_hip_joint', '_thigh_joint', '_calf_joint', '_foot_fixed')):
            if j['name'].endswith(suffix):
                assert np.allclose(j['R'], np.eye(3)), j['name']
                if k < 3:
                    assert np.allclose(j['axis'], [1, 0, 0] if k == 0 else [0, 1, 0]), j['name']
                leg_origins.setdefault(j['name'][:2], [None] * 4)[k] = [float(v) for v in j['xyz']]
    # rigid bodies of the whole-body model (row f3, the 1 kHz QP): the trunk and the three moving links of every leg, every link
    # hanging on a FIXED joint merged into its parent (what pinocchio's URDF parser does): mass, centre of mass and rotational
    # inertia about it, in the frame of the body's joint (trunk: the floating-base frame)
    def merged(link):
        parts = []       # (m, c, I) in this link's frame
        if links[link] is not None:
            m, c, Rl, I = links[link]
            parts.append((m, c, Rl @ I @ Rl.T))
        for j in children.get(link, []):
            if j['type'] == 'fixed':
                for m, c, I in merged(j['child']):
                    parts.append((m, j['xyz'] + j['R'] @ c, j['R'] @ I @ j['R'].T))
        return parts
    def lump(parts):
        m = sum(p[0] for p in parts)
        c = sum(p[0] * p[1] for p in parts) / m
        I = np.zeros((3, 3))
        for mi, ci, Ii in parts:
            d = ci - c
            I += Ii + mi * (d @ d * np.eye(3) - np.outer(d, d))
        return dict(mass=float(m), com=[float(v) for v in c], inertia=I.tolist())
    trunk = [jj['child'] for jj in joints if jj['name'] == 'floating_base'][0]
    body_model = [lump(merged(trunk))]
    for leg in ('FL', 'FR', 'RL', 'RR'):
        for part in ('hip', 'thigh', 'calf'):
            body_model.append(lump(merged('%s_%s' % (leg, part))))
    def walk(link, p, R):
        if links[link] is not None:
            m, c, Rl, I = links[link]
            bodies.append((m, p + R @ c, R @ Rl @ I @ Rl.T @ R.T))
        for j in children.get(link, []):
            pj = p + R @ j['xyz']; Rj = R @ j['R']
            if j['name'].endswith('_hip_joint'):
                hips[j['name'][:2]] = base_R.T @ (pj - base_p)
            if j['type'] in ('revolute', 'continuous'):
                Rj = Rj @ axis_angle_R(j['axis'], q_joint[j['name']])
            walk(j['child'], pj, Rj)
    walk(root_link, base_p, base_R)
    mass = sum(b[0] for b in bodies)
    com = sum(b[0] * b[1] for b in bodies) / mass
    Iw = np.zeros((3, 3))
    for m, c, I in bodies:
        d = c - com
        Iw += I + m * (d @ d * np.eye(3) - np.outer(d, d))
    Ir = base_R.T @ Iw @ base_R
    return dict(mass=mass, Ir=Ir.tolist(), com_in_base=(base_R.T @ (com - base_p)).tolist(),
                hip_xy={k: [float(v[0]), float(v[1])] for k, v in hips.items()}, joint_order=order, leg_origins=leg_origins, body_model=body_model)

if __name__ == '__main__':
    ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
    out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden')
    urdf = os.path.join(ref, 'models/a1_description/urdf/a1.urdf')
    for cfg in ('a1_configuration', 'a1_gait_opt_config', 'a1_config_distr_rejection'):
        y = yaml.safe_load(open(os.path.join(ref, 'apps', cfg + '.yaml')))
        res = derive(urdf, y['init_config'])
        res['source'] = dict(urdf='models/a1_description/urdf/a1.urdf', yaml='apps/%s.yaml' % cfg,
                             init_config=y['init_config'])
        with open(os.path.join(out_dir, 'a1_constants_%s.json' % cfg), 'w') as f:
            json.dump(res, f, indent=1)
        print(cfg, 'mass', res['mass'], 'Ir diag', np.diag(np.array(res['Ir'])), 'hips', res['hip_xy'])
