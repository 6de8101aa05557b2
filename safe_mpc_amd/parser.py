"""config.yaml + CLI -> ``Parameters``: host-side mirror of reference src/safe_mpc/parser.py.

Same CLI flags (parser.py:11-33), same attribute names (parser.py:84-221) and the same geometry bookkeeping
(capsules / obstacles / collision pairs, parser.py:169-315), so that controller- and script-level code written
against the reference reads unchanged.  Differences, all forced by the missing third-party pieces:

* the URDF is read by :mod:`safe_mpc_amd.urdf` (``urdf_parser_py`` is not available);
* when ``robots/<name>_description/urdf/<name>.urdf`` (parser.py:78) does not exist the packaged, build-authored
  ``assets/<name>_class.urdf`` is used;
* ``act_fun`` stays a string (the reference stores a ``torch.nn`` module, parser.py:95-102); the torch module is
  built where it is needed (safe_set.py).
"""
from __future__ import annotations

import argparse
import copy
import os

import numpy as np
import yaml

from .urdf import RobotDescription

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT_DIR = os.path.dirname(PKG_DIR)

ACTIVATIONS = ('relu', 'elu', 'tanh', 'gelu', 'silu')


def default_args():
    return {'system': 'z1', 'dofs': 4, 'controller': 'naive', 'build': False, 'alpha': 10.0, 'horizon': 45,
            'activation': 'gelu', 'back_hor': 45, 'noise': 0.0, 'control_noise': 0.0,
            'joint_bounds_margin': 0.0, 'collision_margin': 0.0}


def parse_args(argv=None):
    """Same flags and defaults as reference parser.py:9-34; unknown flags are ignored so the function can be
    called from inside other programs (pytest, torchrun)."""
    d = default_args()
    ap = argparse.ArgumentParser()
    ap.add_argument('-s', '--system', type=str, default=d['system'])
    ap.add_argument('-d', '--dofs', type=int, default=d['dofs'])
    ap.add_argument('-c', '--controller', type=str, default=d['controller'])
    ap.add_argument('-b', '--build', action='store_true')
    ap.add_argument('--alpha', type=float, default=d['alpha'])
    ap.add_argument('--horizon', type=int, default=d['horizon'])
    ap.add_argument('-a', '--activation', type=str, default=d['activation'])
    ap.add_argument('--back_hor', type=int, default=d['back_hor'])
    ap.add_argument('--noise', type=float, default=d['noise'])
    ap.add_argument('--control_noise', type=float, default=d['control_noise'])
    ap.add_argument('--joint_bounds_margin', type=float, default=d['joint_bounds_margin'])
    ap.add_argument('--collision_margin', type=float, default=d['collision_margin'])
    ns, _ = ap.parse_known_args(argv)
    return vars(ns)


def align_vectors(a, b):
    """Rotation taking direction ``a`` onto ``b`` (visualisation helper of the reference, parser.py:36-58)."""
    a = np.asarray(a, float) / np.linalg.norm(a)
    b = np.asarray(b, float) / np.linalg.norm(b)
    c = float(a @ b)
    if np.isclose(c, -1.0):
        return -np.eye(3)
    v = np.cross(a, b)
    K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    return np.eye(3) + K + K @ K / (1.0 + c)


class Parameters:
    def __init__(self, args=None, urdf_name='z1', rti=True, filename=None, urdf_path=None):
        args = {**default_args(), **(args or {})}
        self.urdf_name = urdf_name
        self.PKG_DIR = PKG_DIR
        self.ROOT_DIR = ROOT_DIR
        self.CONF_DIR = os.path.join(ROOT_DIR, 'config/')
        self.DATA_DIR = os.path.join(ROOT_DIR, 'data_noise/')
        self.GEN_DIR = os.path.join(ROOT_DIR, 'generated/')
        self.NN_DIR = os.path.join(ROOT_DIR, 'nn_models/' + urdf_name + '/')
        self.ROBOTS_DIR = os.path.join(ROOT_DIR, 'robots/')
        with open(filename or os.path.join(ROOT_DIR, 'config.yaml')) as f:
            cfg = yaml.safe_load(f)

        if urdf_path is None:
            urdf_path = f'{self.ROBOTS_DIR}/{urdf_name}_description/urdf/{urdf_name}.urdf'
            if not os.path.exists(urdf_path):
                urdf_path = os.path.join(PKG_DIR, 'assets', f'{urdf_name}_class.urdf')
        self.robot_urdf = urdf_path
        self.robot_descr = RobotDescription.from_xml_file(urdf_path)
        self.links = [l.name for l in self.robot_descr.links]
        self.joints = list(self.robot_descr.joints)

        # (environment overrides of sizes and of the data directory: quick runs and tests without editing config.yaml)
        self.DATA_DIR = os.path.join(os.environ.get('SMPC_DATA_DIR', self.DATA_DIR), '')
        self.test_num = int(os.environ.get('SMPC_TEST_NUM', cfg['test_num']))
        self.n_steps = int(os.environ.get('SMPC_N_STEPS', cfg['n_steps']))
        self.cpu_num = int(cfg['cpu_num'])
        self.build = False

        self.N = int(cfg['N'])
        self.back_hor = int(cfg['back_hor'])
        self.dt = float(cfg['dt'])
        self.alpha = float(cfg['alpha'])

        self.act = str(cfg['act_fun'])
        if self.act not in ACTIVATIONS:
            raise ValueError(f'unknown activation {self.act}')
        self.act_fun = self.act
        self.net_size = list(cfg['network_size'])
        self.use_net = bool(cfg['use_net'])
        self.n_dof_safe_set = int(cfg['n_dof_safe_set'])
        self.net_path = str(cfg['network_path'])
        self.reg_term_analytic_constr = float(cfg['reg_term'])

        self.nq = int(cfg['n_dofs'])
        self.net_size[0] = self.nq * 2                      # parser.py:111
        self.ee_ref = np.array(cfg['ee_ref'], float)
        self.ee_pos = np.array(cfg['ee_position'], float)

        self.solver_type = 'SQP_RTI' if rti else 'SQP'      # parser.py:115-117
        self.solver_mode = cfg['solver_mode']
        self.nlp_max_iter = int(cfg['rti_iter']) if rti else int(cfg['nlp_max_iter'])
        self.qp_max_iter = int(cfg['qp_max_iter'])
        self.alpha_reduction = float(cfg['alpha_reduction'])
        self.alpha_min = float(cfg['alpha_min'])
        self.levenberg_marquardt = float(cfg['levenberg_marquardt'])
        self.ext_flag = cfg['ext_flag']
        self.ipopt_opts = dict(cfg.get('ipopt_opts') or {})

        self.tol_x = float(cfg['tol_x'])
        self.tol_tau = float(cfg['tol_tau'])
        self.tol_dyn = float(cfg['tol_dyn'])
        self.tol_obs = float(cfg['tol_obs'])
        self.tol_safe_set = float(cfg['tol_safe_set'])
        self.Q_weight = float(cfg['Q_weight'])
        self.R_weight = float(cfg['R_weight'])
        self.eps = float(cfg['eps'])
        self.tol_conv = float(cfg['tol_conv'])
        self.tol_cost = float(cfg['tol_cost'])
        self.globalization = 'FIXED_STEP' if rti else 'MERIT_BACKTRACKING'   # parser.py:139

        self.q_dot_gain = float(cfg['q_dot_gain'])
        self.ws_t = float(cfg['ws_t'])
        self.ws_r = float(cfg['ws_r'])
        self.q_margin = args['joint_bounds_margin']        # the YAML value is overwritten by the CLI (parser.py:145-146)

        self.obs_flag = bool(cfg['obs_flag'])
        self.abort_flag = bool(cfg['abort_flag'])
        self.frame_name = cfg['frame_ee']
        self.ee_radius = float(cfg['ee_radius'])
        self.obs_string = cfg['obs_string']
        self.ddq_max = np.array(cfg['ddq_max'], float)
        self.ddx_max = np.array(cfg['ddx_max'], float)

        self.collision_margin = args['collision_margin']
        self.noise = args['noise']
        self.control_noise = args['control_noise']

        m = self.collision_margin
        # obstacles (parser.py:169-180)
        self.obstacles = []
        for entry in cfg.get('obstacles') or []:
            obs = {k: (np.array(v, float) if isinstance(v, list) else v) for k, v in entry.items()}
            if obs['type'] == 'plane':
                obs['bounds'][0] -= m
                obs['bounds'][1] += m
            elif obs['type'] == 'sphere-obs':
                obs['radius'] -= m
            self.obstacles.append(obs)
        # capsules on the robot / fixed in the world (parser.py:184-195)
        self.robot_capsules = []
        for entry in cfg.get('robot_capsules') or []:
            cap = self.create_moving_capsule(copy.deepcopy(entry))
            cap['radius'] -= m
            self.robot_capsules.append(cap)
        self.obst_capsules = []
        for entry in cfg.get('obstacles_capsules') or []:
            cap = self.create_fixed_capsule(copy.deepcopy(entry))
            cap['radius'] -= m
            self.obst_capsules.append(cap)
        self.spheres_robot = []
        for entry in cfg.get('spheres_robot') or []:
            sph = copy.deepcopy(entry)
            sph['radius'] -= m
            self.spheres_robot.append(sph)

        # collision pairs (parser.py:205-219)
        self.collisions_pairs = []
        pairs = cfg.get('collision_pairs')
        if pairs is None:
            for c1 in self.robot_capsules:
                for c2 in self.robot_capsules:
                    if c1['name'] != c2['name']:
                        self.collisions_pairs.append(self.assign_pairs(c1['name'], c2['name'], self.obstacles,
                                                                       self.robot_capsules, []))
                for c2 in self.obst_capsules:
                    self.collisions_pairs.append(self.assign_pairs(c1['name'], c2['name'], self.obstacles,
                                                                   self.robot_capsules + self.obst_capsules, []))
                for obs in self.obstacles:
                    self.collisions_pairs.append(self.assign_pairs(c1['name'], obs['name'], self.obstacles,
                                                                   self.robot_capsules, []))
        else:
            for a, b in pairs:
                self.collisions_pairs.append(self.assign_pairs(a, b, self.obstacles,
                                                               self.robot_capsules + self.obst_capsules,
                                                               self.spheres_robot))
        self.track_traj = bool(cfg['track_traj'])
        self.noise_mass = float(cfg.get('noise_mass', 0.0))
        self.noise_inertia = float(cfg.get('noise_inertia', 0.0))
        self.noise_cm = float(cfg.get('noise_cm', 0.0))

    # ------------------------------------------------------------------------------------------------------------
    def create_moving_capsule(self, capsule):
        """Capsule riding on a URDF link (parser.py:224-243): first end point at the link origin, second one
        ``length`` along ``link_axis``, in the direction of the joint stored at the link's own list index."""
        capsule['type'] = 'moving_capsule'
        ax = capsule['link_axis']
        e0 = np.array([0.0, 0.0, 0.0, 1.0])
        e1 = e0.copy()
        capsule['direction'] = np.sign(self.joints[self.links.index(capsule['link_name'])].origin.xyz[ax])
        e1[ax] += capsule['direction'] * capsule['length']
        capsule['end_points'] = [e0, e1]
        capsule['end_points_fk'] = [None, None]
        return capsule

    def create_fixed_capsule(self, capsule):
        """World-fixed capsule given by its two segment end points (parser.py:245-254)."""
        capsule['type'] = 'fixed_capsule'
        capsule['end_points'] = np.array([capsule['point_A'], capsule['point_B']], float)
        capsule['length'] = float(np.linalg.norm(capsule['end_points'][0] - capsule['end_points'][1]))
        capsule['end_points_fk'] = capsule['end_points']
        capsule['end_points_T_fun'] = align_vectors([0, 1, 0], capsule['end_points'][1] - capsule['end_points'][0])
        return capsule

    def assign_pairs(self, obj1_name, obj2_name, obstacles_list, capsules_list, spheres_list):
        """Classify a named pair (parser.py:256-315): capsule-capsule, capsule-sphere, capsule-plane,
        sphere-sphere or sphere-plane; element 0 is the robot-side object."""
        by_name = lambda seq, n: next((o for o in seq if o['name'] == n), None)
        pair = {'elements': [None, None], 'type': None}
        sph = by_name(spheres_list, obj1_name) or by_name(spheres_list, obj2_name)
        if sph is not None:
            other = obj2_name if sph['name'] == obj1_name else obj1_name
            obs = by_name(obstacles_list, other)
            if obs is not None:
                pair['elements'] = [sph, obs]
                pair['type'] = 'sphere-sphere' if obs['type'] == 'sphere-obs' else 'sphere-plane'
                return pair
        cap1 = by_name(capsules_list, obj1_name)
        cap2 = by_name(capsules_list, obj2_name)
        pair['elements'][0] = cap1
        if cap2 is not None:
            pair['elements'][1] = cap2
            if cap1 is not None:
                pair['type'] = 'capsule-capsule'
        obs = by_name(obstacles_list, obj2_name)
        if obs is not None:
            pair['elements'][1] = obs
            if cap1 is not None:
                pair['type'] = 'capsule-sphere' if obs['type'] == 'sphere-obs' else 'capsule-plane'
        if pair['type'] is None:
            raise ValueError(f'collision pair ({obj1_name}, {obj2_name}) does not name a supported combination')
        return pair
