"""Minimal URDF reader -> serial-chain robot table.

Replaces, for the hot path only, what the reference gets from ``urdf_parser_py`` (reference
src/safe_mpc/parser.py:80-82) and from ``adam.casadi.KinDynComputations`` (env_model.py:40-45): the kinematic
chain of the first ``nq`` non-fixed joints (env_model.py:23-32), their limits (env_model.py:107-114) and the
rigid-body inertias.  Every joint that is not actuated (fixed joints and movable joints beyond ``nq``) is locked
at zero and the links behind it are lumped into the last actuated link, which is how a reduced adam model
behaves for joints that are not in ``joint_names`` [EXT-UNVERIFIED, SURVEY A.3].

Only what the solver needs is read; meshes, collisions, transmissions etc. are ignored.
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np


def rpy_to_matrix(rpy):
    """URDF fixed-axis roll/pitch/yaw -> rotation matrix  R = Rz(yaw) Ry(pitch) Rx(roll)."""
    r, p, y = (float(v) for v in rpy)
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([
        [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
        [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
        [-sp, cp * sr, cp * cr],
    ])


def _vec(text, n=3, default=0.0):
    if text is None:
        return np.full(n, default)
    vals = [float(v) for v in text.split()]
    if len(vals) != n:
        raise ValueError(f'expected {n} numbers, got "{text}"')
    return np.array(vals)


@dataclass
class Origin:
    xyz: np.ndarray = field(default_factory=lambda: np.zeros(3))
    rpy: np.ndarray = field(default_factory=lambda: np.zeros(3))

    @property
    def R(self):
        return rpy_to_matrix(self.rpy)


@dataclass
class Limit:
    lower: float = 0.0
    upper: float = 0.0
    effort: float = 0.0
    velocity: float = 0.0


@dataclass
class Inertial:
    origin: Origin
    mass: float
    inertia: np.ndarray  # 3x3 about the COM, axes of the inertial frame


@dataclass
class Link:
    name: str
    inertial: Inertial | None = None


@dataclass
class Joint:
    name: str
    type: str
    parent: str
    child: str
    origin: Origin
    axis: np.ndarray
    limit: Limit | None = None


def _origin(elem):
    if elem is None:
        return Origin()
    return Origin(_vec(elem.get('xyz')), _vec(elem.get('rpy')))


class RobotDescription:
    """The subset of ``urdf_parser_py.urdf.URDF`` the reference touches: ``links``, ``joints`` (file order,
    parser.py:81-82), ``get_root()`` (env_model.py:40), joint ``origin.xyz`` (parser.py:237), ``limit``."""

    def __init__(self, links, joints, name='robot'):
        self.name = name
        self.links = links
        self.joints = joints
        self.link_map = {l.name: l for l in links}
        self.joint_map = {j.name: j for j in joints}
        self.parent_joint = {j.child: j for j in joints}  # link name -> joint that moves it

    @classmethod
    def from_xml_file(cls, path):
        return cls.from_xml_string(open(path, 'r', encoding='utf-8').read())

    @classmethod
    def from_xml_string(cls, text):
        root = ET.fromstring(text)
        links, joints = [], []
        for le in root.findall('link'):
            inertial = None
            ie = le.find('inertial')
            if ie is not None and ie.find('mass') is not None:
                ine = ie.find('inertia')
                I = np.zeros((3, 3))
                if ine is not None:
                    g = lambda k: float(ine.get(k, 0.0))
                    I = np.array([[g('ixx'), g('ixy'), g('ixz')],
                                  [g('ixy'), g('iyy'), g('iyz')],
                                  [g('ixz'), g('iyz'), g('izz')]])
                inertial = Inertial(_origin(ie.find('origin')), float(ie.find('mass').get('value')), I)
            links.append(Link(le.get('name'), inertial))
        for je in root.findall('joint'):
            jtype = je.get('type')
            ax = je.find('axis')
            axis = _vec(ax.get('xyz')) if ax is not None else np.array([1.0, 0.0, 0.0])
            lim = None
            le = je.find('limit')
            if le is not None:
                lim = Limit(float(le.get('lower', 0.0)), float(le.get('upper', 0.0)),
                            float(le.get('effort', 0.0)), float(le.get('velocity', 0.0)))
            joints.append(Joint(je.get('name'), jtype, je.find('parent').get('link'), je.find('child').get('link'),
                                _origin(je.find('origin')), axis, lim))
        return cls(links, joints, root.get('name', 'robot'))

    def get_root(self):
        children = {j.child for j in self.joints}
        roots = [l.name for l in self.links if l.name not in children]
        if len(roots) != 1:
            raise ValueError(f'URDF must have exactly one root link, found {roots}')
        return roots[0]


@dataclass
class ChainJoint:
    name: str
    R0: np.ndarray       # previous actuated link frame (or world) -> this joint frame, at q = 0
    p0: np.ndarray
    axis: np.ndarray     # unit, joint frame
    mass: float
    com: np.ndarray      # child-link frame
    inertia: np.ndarray  # 3x3 about com, child-link axes
    q_min: float
    q_max: float
    v_max: float
    tau_max: float


class SerialChain:
    """First ``nq`` non-fixed joints of a URDF as a serial chain with lumped inertias."""

    def __init__(self, descr: RobotDescription, nq: int):
        self.descr = descr
        movable = [j for j in descr.joints if j.type != 'fixed']
        if len(movable) < nq:
            raise ValueError(f'URDF has {len(movable)} movable joints, {nq} requested')
        act = movable[:nq]                                   # env_model.py:23-32
        for j in act:
            if j.type not in ('revolute', 'continuous'):
                raise ValueError(f'joint {j.name}: only revolute joints are supported, got {j.type}')
        self.joint_names = [j.name for j in act]
        act_index = {j.name: i for i, j in enumerate(act)}
        root = descr.get_root()

        # pose of every link relative to the movable link that carries it (index -1 = world/base)
        self._carrier = {root: (-1, np.eye(3), np.zeros(3))}
        children = {}
        for j in descr.joints:
            children.setdefault(j.parent, []).append(j)
        order = [root]
        while order:
            ln = order.pop(0)
            idx, R, p = self._carrier[ln]
            for j in children.get(ln, []):
                Rj, pj = R @ j.origin.R, p + R @ j.origin.xyz
                if j.name in act_index:
                    i = act_index[j.name]
                    if idx != i - 1:
                        raise ValueError(f'actuated joints do not form a serial chain at {j.name}')
                    act[i]._chain_R0, act[i]._chain_p0 = Rj, pj
                    self._carrier[j.child] = (i, np.eye(3), np.zeros(3))
                else:  # fixed, or movable but locked at zero
                    self._carrier[j.child] = (idx, Rj, pj)
                order.append(j.child)

        # lump inertias
        m = np.zeros(nq)
        mc = np.zeros((nq, 3))
        parts = [[] for _ in range(nq)]
        for l in descr.links:
            if l.inertial is None or l.name not in self._carrier:
                continue
            idx, R, p = self._carrier[l.name]
            if idx < 0:
                continue  # rigidly attached to the world: no dynamics
            c = p + R @ l.inertial.origin.xyz
            Ic = R @ l.inertial.origin.R @ l.inertial.inertia @ l.inertial.origin.R.T @ R.T
            m[idx] += l.inertial.mass
            mc[idx] += l.inertial.mass * c
            parts[idx].append((l.inertial.mass, c, Ic))
        self.joints = []
        for i, j in enumerate(act):
            if m[i] <= 0:
                raise ValueError(f'link moved by {j.name} has no mass')
            com = mc[i] / m[i]
            I = np.zeros((3, 3))
            for (mi, ci, Ici) in parts[i]:
                d = ci - com
                I += Ici + mi * (d @ d * np.eye(3) - np.outer(d, d))
            ax = j.axis / np.linalg.norm(j.axis)
            lim = j.limit or Limit()
            self.joints.append(ChainJoint(j.name, j._chain_R0, j._chain_p0, ax, m[i], com, I,
                                          lim.lower, lim.upper, lim.velocity, lim.effort))
        self.nq = nq

    def frame(self, link_name):
        """(carrier index, R, p): pose of a URDF link frame relative to the actuated link that carries it."""
        if link_name not in self._carrier:
            raise KeyError(f'link {link_name} not in URDF')
        return self._carrier[link_name]

    # numpy reference kinematics (used by host-side helpers and tests, never by the solver)
    def link_poses(self, q):
        R, p = np.eye(3), np.zeros(3)
        out = []
        for i, j in enumerate(self.joints):
            R = R  # noqa
            p = p + R @ j.p0
            R = R @ j.R0 @ _axis_angle(j.axis, q[i])
            out.append((R.copy(), p.copy()))
        return out


def _axis_angle(a, th):
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
