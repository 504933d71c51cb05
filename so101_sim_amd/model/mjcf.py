"""MJCF-subset model compiler: scene XML + prop XMLs -> flat numpy `Model`.

Replaces, for the SO100 hand-over scenes only, what the reference obtains from
`mjcf.from_path(...)` + MuJoCo's compiler (so101_sim/tasks/base/so100_task.py:386-406,
so101_sim/tasks/so100_hand_over.py:159-206).  The MuJoCo compile semantics that results depend on
are listed in SURVEY.md section 8a-3; each is marked below where it is applied.

This is host-side, one-time work (numpy); the compiled model is serialised by `blob.py` and
consumed by the HIP library through the C ABI (include/so101.h).
"""
from __future__ import annotations

import dataclasses
import os
import xml.etree.ElementTree as ET

import numpy as np

from . import meshes as mm

# geom type codes (same numbering as csrc/so101_model.hpp)
GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = 0, 1, 2, 3, 4, 5
_GEOM_TYPES = {"plane": GEOM_PLANE, "sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE,
               "cylinder": GEOM_CYLINDER, "box": GEOM_BOX, "mesh": GEOM_MESH}
JNT_NONE, JNT_HINGE, JNT_FREE, JNT_SLIDE = 0, 1, 2, 3

# MuJoCo built-in defaults for the attributes this subset reads
_GEOM_DEFAULTS = dict(type="sphere", contype="1", conaffinity="1", condim="3",
                      friction="1 0.005 0.0001", solref="0.02 1", solimp="0.9 0.95 0.001 0.5 2",
                      solmix="1", margin="0", gap="0", density="1000", priority="0", pos="0 0 0",
                      quat="1 0 0 0", group="0")
_JOINT_DEFAULTS = dict(type="hinge", armature="0", damping="0", frictionloss="0", pos="0 0 0",
                       axis="0 0 1", solreflimit="0.02 1", solimplimit="0.9 0.95 0.001 0.5 2",
                       solreffriction="0.02 1", solimpfriction="0.9 0.95 0.001 0.5 2")
_GENERAL_DEFAULTS = dict(gaintype="fixed", gainprm="1 0 0", biastype="none", biasprm="0 0 0")
_POSITION_DEFAULTS = dict(kp="1", kv="0")
_EQUALITY_DEFAULTS = dict(solref="0.02 1", solimp="0.9 0.95 0.001 0.5 2", polycoef="0 1 0 0 0")


def _expand_includes(root: ET.Element, base_dir: str) -> ET.Element:
    """<include file=.../> = the children of the included file's root spliced in at the include's position (MuJoCo's
    semantics; aloha/scene_pbr.xml:23 includes aloha_pbr.xml, which :302 includes joint_position_actuators.xml)."""
    out = ET.Element(root.tag, root.attrib)
    for child in root:
        if child.tag == "include":
            inc = _expand_includes(ET.parse(os.path.join(base_dir, child.attrib["file"])).getroot(), base_dir)
            out.extend(list(inc))
        else:
            out.append(child)
    return out


def _euler2quat(e):
    """intrinsic x-y-z (MuJoCo's default eulerseq "xyz"), radians"""
    q = np.array([1.0, 0, 0, 0])
    for k, ang in enumerate(e):
        h = np.zeros(4); h[0] = np.cos(0.5 * ang); h[1 + k] = np.sin(0.5 * ang)
        w1, x1, y1, z1 = q; w2, x2, y2, z2 = h
        q = np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                      w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])
    return q


def _floats(s, n=None, fill=None):
    v = [float(t) for t in str(s).split()]
    if n is not None and len(v) < n:
        v = v + list(fill[len(v):] if fill is not None else [0.0] * (n - len(v)))
    return np.asarray(v, dtype=np.float64)


class _Defaults:
    """Nested <default class=...> tree; a class inherits every attribute of its ancestors."""

    def __init__(self, root: ET.Element):
        self.classes: dict[str, dict[str, dict[str, str]]] = {}
        tops = root.findall("default")
        if not tops:
            self.classes.setdefault("main", {})
        for top in tops:       # several top-level sections (one per included file) all extend the unnamed root class
            self._walk(top, "main", self.classes.get("main", {}))

    def _walk(self, node, name, inherited):
        cur = {k: dict(v) for k, v in inherited.items()}
        for child in node:
            if child.tag != "default":
                cur.setdefault(child.tag, {}).update(child.attrib)
        self.classes[name] = cur
        for child in node:
            if child.tag == "default":
                self._walk(child, child.attrib["class"], cur)

    def resolve(self, tag: str, elem: ET.Element, childclass: str | None, builtin: dict) -> dict:
        cls = elem.attrib.get("class") or childclass or "main"
        out = dict(builtin)
        out.update(self.classes.get(cls, {}).get(tag, {}))
        out.update({k: v for k, v in elem.attrib.items() if k != "class"})
        return out


@dataclasses.dataclass
class _Geom:
    name: str
    body: int
    type: int
    pos: np.ndarray
    quat: np.ndarray
    size: np.ndarray
    contype: int
    conaffinity: int
    condim: int
    friction: np.ndarray
    solref: np.ndarray
    solimp: np.ndarray
    solmix: float
    margin: float
    gap: float
    priority: int
    density: float
    mass: float | None
    mesh: str | None
    group: int
    hull: np.ndarray | None = None      # hull vertices in the geom frame as written in the file
    hull_faces: np.ndarray | None = None


@dataclasses.dataclass
class _Body:
    name: str
    parent: int
    pos: np.ndarray
    quat: np.ndarray
    jnt_type: int = JNT_NONE
    jnt_axis: np.ndarray = dataclasses.field(default_factory=lambda: np.array([0.0, 0, 1]))
    jnt_range: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(2))
    jnt_limited: int = 0
    jnt_name: str = ""
    armature: float = 0.0
    frictionloss: float = 0.0
    damping: float = 0.0
    solreflimit: np.ndarray = None
    solimplimit: np.ndarray = None
    solreffriction: np.ndarray = None
    solimpfriction: np.ndarray = None
    actfrcrange: np.ndarray | None = None   # joint-level clamp of the total actuator force (<joint actuatorfrcrange>)
    inertial: tuple | None = None       # (ipos, iquat, mass, diaginertia)


class SceneCompiler:
    """Accumulates bodies/geoms from one scene file and any number of attached free props."""

    def __init__(self):
        self.bodies: list[_Body] = [_Body("world", 0, np.zeros(3), np.array([1.0, 0, 0, 0]))]
        self.geoms: list[_Geom] = []
        self.actuators: list[dict] = []
        self.excludes: list[tuple[str, str]] = []
        self.option = dict(timestep=0.002, gravity=np.array([0.0, 0.0, -9.81]), impratio=1.0,
                           cone="pyramidal", tolerance=1e-8, iterations=100,
                           mpr_tolerance=1e-6, mpr_iterations=50)
        self.prop_bodies: list[int] = []
        self.equalities: list[dict] = []     # <equality><joint joint1 joint2 polycoef solref solimp/>
        self.keyframes: dict[str, dict] = {}
        self.general_tree = False            # scenes outside the one-hinge-chain SO100 subset (ALOHA)

    # ---------------------------------------------------------------- parsing
    def _mesh_table(self, root, base_dir, scale_override=None):
        comp = root.find("compiler")
        meshdir = ""
        if comp is not None:
            meshdir = comp.attrib.get("meshdir", comp.attrib.get("assetdir", ""))
        table = {}
        for asset in root.findall("asset"):
            for m in asset.findall("mesh"):
                file = m.attrib["file"]
                name = m.attrib.get("name", os.path.splitext(os.path.basename(file))[0])
                scale = _floats(m.attrib.get("scale", "1 1 1"))
                if scale_override is not None:     # so100_hand_over.py:187-192 replaces mesh.scale
                    scale = np.full(3, float(scale_override))
                table[name] = (os.path.join(base_dir, meshdir, file), scale)
        return table

    def add_scene(self, xml_path: str):
        base = os.path.dirname(xml_path)
        root = _expand_includes(ET.parse(xml_path).getroot(), base)
        opt = root.find("option")
        if opt is not None:
            for k in ("timestep", "impratio", "tolerance"):
                if k in opt.attrib:
                    self.option[k] = float(opt.attrib[k])
            if "cone" in opt.attrib:
                self.option["cone"] = opt.attrib["cone"]
            if "gravity" in opt.attrib:
                self.option["gravity"] = _floats(opt.attrib["gravity"])
        defaults = _Defaults(root)
        meshes = self._mesh_table(root, base)
        for wb in root.findall("worldbody"):        # (one per included file, merged in document order)
            self._walk_body(wb, 0, None, defaults, meshes)
        for contact in root.findall("contact"):
            for ex in contact.findall("exclude"):
                self.excludes.append((ex.attrib["body1"], ex.attrib["body2"]))
        for act in root.findall("actuator"):
            for a in act:
                if a.tag == "general":
                    self.actuators.append(defaults.resolve("general", a, None, _GENERAL_DEFAULTS))
                elif a.tag == "position":
                    # <position kp kv> = general with gainprm (kp 0 0), biastype affine, biasprm (0 -kp -kv)  [MuJoCo XML reference]
                    r = defaults.resolve("position", a, None, _POSITION_DEFAULTS)
                    kp, kv = float(r["kp"]), float(r["kv"])
                    r.update(gaintype="fixed", gainprm=f"{kp!r} 0 0", biastype="affine", biasprm=f"0 {-kp!r} {-kv!r}")
                    self.actuators.append(r)
                else:
                    raise NotImplementedError(f"actuator <{a.tag}> is outside the supported subset")
        for eq in root.findall("equality"):
            for e in eq:
                if e.tag != "joint" or "joint2" not in e.attrib:
                    raise NotImplementedError(f"equality <{e.tag}> is outside the supported subset")
                self.equalities.append(defaults.resolve("equality", e, None, _EQUALITY_DEFAULTS))
        for kf in root.findall("keyframe"):
            for k in kf.findall("key"):
                self.keyframes[k.attrib.get("name", f"key{len(self.keyframes)}")] = {
                    f: _floats(k.attrib[f]) for f in ("qpos", "ctrl") if f in k.attrib}

    def add_free_prop(self, xml_path: str, name: str, mesh_scale: float | None = None) -> int:
        """Attach the single root body of a prop model as a free body at the world origin
        (so100_hand_over.py:169,199: `add_free_entity`; its own <freejoint/> is removed :201-206)."""
        root = ET.parse(xml_path).getroot()
        base = os.path.dirname(xml_path)
        defaults = _Defaults(root)
        meshes = self._mesh_table(root, base, mesh_scale)
        body_el = root.find("worldbody").find("body")
        idx = len(self.bodies)
        self.bodies.append(_Body(name, 0, np.zeros(3), np.array([1.0, 0, 0, 0]), jnt_type=JNT_FREE,
                                 jnt_name=name + "/free"))
        childclass = body_el.attrib.get("childclass")
        for g in body_el.findall("geom"):
            self._add_geom(g, idx, childclass, defaults, meshes, prefix=name + "/")
        self.prop_bodies.append(idx)
        return idx

    def _walk_body(self, node, parent, childclass, defaults, meshes):
        for el in node:
            if el.tag == "geom":
                self._add_geom(el, parent, childclass, defaults, meshes)
            elif el.tag == "body":
                cc = el.attrib.get("childclass", childclass)
                b = _Body(el.attrib.get("name", f"body{len(self.bodies)}"), parent,
                          _floats(el.attrib.get("pos", "0 0 0")),
                          _euler2quat(_floats(el.attrib["euler"])) if "euler" in el.attrib
                          else _floats(el.attrib.get("quat", "1 0 0 0")))
                b.quat = b.quat / np.linalg.norm(b.quat)
                idx = len(self.bodies)
                self.bodies.append(b)
                joints = el.findall("joint")
                if len(joints) > 1 or el.find("freejoint") is not None:
                    raise NotImplementedError("scene bodies carry at most one hinge joint")
                if joints:
                    j = defaults.resolve("joint", joints[0], cc, _JOINT_DEFAULTS)
                    if j["type"] not in ("hinge", "slide") or np.any(_floats(j["pos"]) != 0):
                        raise NotImplementedError("only hinge / slide joints at the body origin")
                    b.jnt_type = JNT_HINGE if j["type"] == "hinge" else JNT_SLIDE
                    if "actuatorfrcrange" in j:
                        b.actfrcrange = _floats(j["actuatorfrcrange"])
                    b.jnt_name = j.get("name", "")
                    ax = _floats(j["axis"])
                    b.jnt_axis = ax / np.linalg.norm(ax)
                    b.armature = float(j["armature"])
                    b.frictionloss = float(j["frictionloss"])
                    b.damping = float(j["damping"])
                    if "range" in j:   # autolimits (MuJoCo default): range given => limited
                        b.jnt_range = _floats(j["range"])
                        b.jnt_limited = 0 if j.get("limited", "auto") == "false" else 1
                    b.solreflimit = _floats(j["solreflimit"], 2, [0.02, 1])
                    b.solimplimit = _floats(j["solimplimit"], 5, [0.9, 0.95, 0.001, 0.5, 2])
                    b.solreffriction = _floats(j["solreffriction"], 2, [0.02, 1])
                    b.solimpfriction = _floats(j["solimpfriction"], 5, [0.9, 0.95, 0.001, 0.5, 2])
                inert = el.find("inertial")
                if inert is not None:
                    q = _floats(inert.attrib.get("quat", "1 0 0 0"))
                    b.inertial = (_floats(inert.attrib["pos"]), q / np.linalg.norm(q),
                                  float(inert.attrib["mass"]), _floats(inert.attrib["diaginertia"]))
                self._walk_body(el, idx, cc, defaults, meshes)

    def _add_geom(self, el, body, childclass, defaults, meshes, prefix=""):
        a = defaults.resolve("geom", el, childclass, _GEOM_DEFAULTS)
        gtype = _GEOM_TYPES[a["type"]]
        q = _floats(a["quat"])
        g = _Geom(
            name=prefix + a.get("name", a.get("mesh", f"geom{len(self.geoms)}")), body=body, type=gtype,
            pos=_floats(a["pos"]), quat=q / np.linalg.norm(q),
            size=_floats(a.get("size", "0 0 0"), 3), contype=int(a["contype"]),
            conaffinity=int(a["conaffinity"]), condim=int(a["condim"]),
            friction=_floats(a["friction"], 3, [1, 0.005, 0.0001]),
            solref=_floats(a["solref"], 2, [0.02, 1]),
            solimp=_floats(a["solimp"], 5, [0.9, 0.95, 0.001, 0.5, 2]),
            solmix=float(a["solmix"]), margin=float(a["margin"]), gap=float(a["gap"]),
            priority=int(a["priority"]), density=float(a["density"]),
            mass=float(a["mass"]) if "mass" in a else None, mesh=a.get("mesh"), group=int(a["group"]))
        if gtype == GEOM_MESH:
            path, scale = meshes[g.mesh]
            if os.path.exists(path):
                verts, _ = mm.load_mesh(path)
                g.hull, g.hull_faces = mm.convex_hull(verts * scale)   # mesh geoms collide as hulls
            elif g.contype or g.conaffinity:
                raise FileNotFoundError(path)
            else:
                g.hull = None          # missing visual mesh (.MISSING_LARGE_BLOBS): proxy later
        self.geoms.append(g)


def _spatial_inertia(m, c, ic):
    cx = np.array([[0, -c[2], c[1]], [c[2], 0, -c[0]], [-c[1], c[0], 0]])
    out = np.zeros((6, 6))
    out[:3, :3] = ic - m * cx @ cx
    out[:3, 3:] = m * cx
    out[3:, :3] = -m * cx
    out[3:, 3:] = m * np.eye(3)
    return out


def finalize(sc: SceneCompiler) -> dict:
    """Resolve frames, masses, contact filters and compile-time constants into flat arrays."""
    nb = len(sc.bodies)
    B = sc.bodies
    # ---- mass properties ----------------------------------------------------------------
    body_mass = np.zeros(nb)
    body_ipos = np.zeros((nb, 3))
    body_iquat = np.tile([1.0, 0, 0, 0], (nb, 1))
    body_inertia = np.zeros((nb, 3))
    proxy_inertia = []
    for i, b in enumerate(B):
        if b.inertial is not None:
            body_ipos[i], body_iquat[i], body_mass[i], body_inertia[i] = b.inertial
        elif b.jnt_type == JNT_FREE:
            # Inferred from geoms with density > 0 (MuJoCo).  For the props this is the visual
            # mesh at density 200 (ycb/011_banana/google_64k/model.xml:40); when that blob is
            # absent the union of the collision hulls stands in for it (labelled proxy).
            dens = [g for g in sc.geoms if g.body == i and g.density > 0 and g.type == GEOM_MESH]
            rho = dens[0].density if dens else 200.0
            if dens and all(g.hull is not None for g in dens):
                solids = [(g.hull, g.hull_faces, g) for g in dens]
            else:
                solids = [(g.hull, g.hull_faces, g) for g in sc.geoms
                          if g.body == i and g.type == GEOM_MESH and g.hull is not None and (g.contype or g.conaffinity)]
                proxy_inertia.append(b.name)
            mass, mc, second = 0.0, np.zeros(3), np.zeros((3, 3))
            parts = []
            for hv, hf, g in solids:
                R = mm.quat2mat(g.quat)
                v = hv @ R.T + g.pos
                vol, com, ic = mm.polyhedron_mass_properties(v, hf)
                parts.append((rho * vol, com, rho * ic))
                mass += rho * vol
                mc += rho * vol * com
            com = mc / mass
            itot = np.zeros((3, 3))
            for m_, c_, i_ in parts:
                d = c_ - com
                itot += i_ + m_ * (d @ d * np.eye(3) - np.outer(d, d))
            w, v = mm.principal_frame(itot)
            body_mass[i], body_ipos[i], body_inertia[i], body_iquat[i] = mass, com, w, mm.mat2quat(v)

    # ---- dof layout ----------------------------------------------------------------------
    qposadr = -np.ones(nb, dtype=np.int32)
    dofadr = -np.ones(nb, dtype=np.int32)
    nq = nv = 0
    for i, b in enumerate(B):
        if b.jnt_type in (JNT_HINGE, JNT_SLIDE):
            qposadr[i], dofadr[i] = nq, nv
            nq, nv = nq + 1, nv + 1
        elif b.jnt_type == JNT_FREE:
            qposadr[i], dofadr[i] = nq, nv
            nq, nv = nq + 7, nv + 6
    weldid = np.zeros(nb, dtype=np.int32)
    for i, b in enumerate(B):
        weldid[i] = i if b.jnt_type != JNT_NONE else (weldid[b.parent] if i else 0)

    # ---- kinematics at qpos0 and M(qpos0) for invweight0 / meaninertia --------------------
    xpos = np.zeros((nb, 3))
    xmat = np.tile(np.eye(3), (nb, 1, 1))
    for i, b in enumerate(B):
        if i == 0:
            continue
        xpos[i] = xpos[b.parent] + xmat[b.parent] @ b.pos
        xmat[i] = xmat[b.parent] @ mm.quat2mat(b.quat)
    S = np.zeros((nv, 6))                       # motion axis: [omega ; velocity of the world origin]
    dof_body = np.zeros(nv, dtype=np.int32)
    for i, b in enumerate(B):
        if b.jnt_type == JNT_HINGE:
            a = xmat[i] @ b.jnt_axis
            S[dofadr[i]] = np.concatenate([a, np.cross(xpos[i], a)])
            dof_body[dofadr[i]] = i
        elif b.jnt_type == JNT_SLIDE:
            S[dofadr[i]] = np.concatenate([np.zeros(3), xmat[i] @ b.jnt_axis])
            dof_body[dofadr[i]] = i
        elif b.jnt_type == JNT_FREE:
            for k in range(3):
                S[dofadr[i] + k, 3 + k] = 1.0
                a = xmat[i][:, k]
                S[dofadr[i] + 3 + k] = np.concatenate([a, np.cross(xpos[i], a)])
            dof_body[dofadr[i]:dofadr[i] + 6] = i
    Ic = np.zeros((nb, 6, 6))
    for i in range(nb):
        if body_mass[i] > 0:
            Ri = xmat[i] @ mm.quat2mat(body_iquat[i])
            Ic[i] = _spatial_inertia(body_mass[i], xpos[i] + xmat[i] @ body_ipos[i],
                                     Ri @ np.diag(body_inertia[i]) @ Ri.T)
    for i in range(nb - 1, 0, -1):
        Ic[B[i].parent] += Ic[i]

    def is_ancestor(a, d):        # body a is d or an ancestor of d
        while d != 0:
            if d == a:
                return True
            d = B[d].parent
        return a == 0
    M = np.zeros((nv, nv))
    for r in range(nv):
        for c in range(nv):
            br, bc = dof_body[r], dof_body[c]
            if is_ancestor(bc, br):
                M[r, c] = S[r] @ Ic[br] @ S[c]
            elif is_ancestor(br, bc):
                M[r, c] = S[r] @ Ic[bc] @ S[c]
    armature = np.zeros(nv)
    frictionloss = np.zeros(nv)
    damping = np.zeros(nv)
    for i, b in enumerate(B):
        if b.jnt_type in (JNT_HINGE, JNT_SLIDE):
            armature[dofadr[i]], frictionloss[dofadr[i]], damping[dofadr[i]] = b.armature, b.frictionloss, b.damping
    M += np.diag(armature)                      # armature sits on M's diagonal (KAT-1 pins this)
    Minv = np.linalg.inv(M)
    dof_invweight0 = np.zeros(nv)
    for i, b in enumerate(B):
        if b.jnt_type in (JNT_HINGE, JNT_SLIDE):
            dof_invweight0[dofadr[i]] = Minv[dofadr[i], dofadr[i]]
        elif b.jnt_type == JNT_FREE:
            d = dofadr[i]
            dof_invweight0[d:d + 3] = np.mean(np.diag(Minv)[d:d + 3])
            dof_invweight0[d + 3:d + 6] = np.mean(np.diag(Minv)[d + 3:d + 6])
    body_invweight0 = np.zeros((nb, 2))
    for i in range(1, nb):
        if weldid[i] == 0:
            continue                            # static bodies: zero inverse weight
        c = xpos[i] + xmat[i] @ body_ipos[i]
        Jp, Jr = np.zeros((3, nv)), np.zeros((3, nv))
        for d in range(nv):
            if is_ancestor(dof_body[d], i):
                w, vo = S[d, :3], S[d, 3:]
                Jr[:, d] = w
                Jp[:, d] = vo + np.cross(w, c)
        body_invweight0[i, 0] = np.trace(Jp @ Minv @ Jp.T) / 3
        body_invweight0[i, 1] = np.trace(Jr @ Minv @ Jr.T) / 3

    # ---- geoms -----------------------------------------------------------------------------
    coll = [g for g in sc.geoms if (g.contype or g.conaffinity)]
    if sc.general_tree:
        coll.sort(key=lambda g: g.body)        # MuJoCo numbers geoms body by body (stable: document order within a body)
    ng = len(coll)
    vert_chunks, mesh_vertadr, mesh_vertnum = [], [], []
    g_arr = dict(type=np.zeros(ng, np.int32), body=np.zeros(ng, np.int32), pos=np.zeros((ng, 3)),
                 quat=np.zeros((ng, 4)), size=np.zeros((ng, 3)), contype=np.zeros(ng, np.int32),
                 conaffinity=np.zeros(ng, np.int32), condim=np.zeros(ng, np.int32),
                 friction=np.zeros((ng, 3)), solref=np.zeros((ng, 2)), solimp=np.zeros((ng, 5)),
                 solmix=np.zeros(ng), margin=np.zeros(ng), gap=np.zeros(ng),
                 priority=np.zeros(ng, np.int32), vertadr=-np.ones(ng, np.int32),
                 vertnum=np.zeros(ng, np.int32), rbound=np.zeros(ng), center=np.zeros((ng, 3)),
                 aabb=np.zeros((ng, 6)))
    nvert = 0
    shared: dict[bytes, int] = {}
    for k, g in enumerate(coll):
        for f in ("type", "body", "pos", "quat", "size", "contype", "conaffinity", "condim", "friction",
                  "solref", "solimp", "solmix", "margin", "gap", "priority"):
            g_arr[f][k] = getattr(g, f)
        if g.type == GEOM_MESH:
            hv = g.hull
            lo, hi = hv.min(0), hv.max(0)
            key = hv.tobytes()
            if sc.general_tree and key in shared:          # the same mesh on several geoms (left / right arm, camera bodies): one copy
                g_arr["vertadr"][k], g_arr["vertnum"][k] = shared[key], len(hv)
            else:
                shared[key] = nvert
                g_arr["vertadr"][k], g_arr["vertnum"][k] = nvert, len(hv)
                vert_chunks.append(hv)
                nvert += len(hv)
            ctr = 0.5 * (lo + hi)
            g_arr["center"][k] = ctr
            g_arr["rbound"][k] = np.linalg.norm(hv - ctr, axis=1).max()
            g_arr["aabb"][k] = np.concatenate([ctr, 0.5 * (hi - lo)])
        else:
            s = g.size
            half = {GEOM_PLANE: np.array([1e10, 1e10, 0.0]) if True else None,
                    GEOM_SPHERE: np.array([s[0]] * 3),
                    GEOM_CAPSULE: np.array([s[0], s[0], s[0] + s[1]]),
                    GEOM_CYLINDER: np.array([s[0], s[0], s[1]]),
                    GEOM_BOX: s.copy()}[g.type]
            g_arr["aabb"][k] = np.concatenate([np.zeros(3), half])
            g_arr["rbound"][k] = {GEOM_PLANE: 0.0, GEOM_SPHERE: s[0], GEOM_CAPSULE: s[0] + s[1],
                                  GEOM_CYLINDER: np.hypot(s[0], s[1]), GEOM_BOX: np.linalg.norm(s)}[g.type]
    verts = np.concatenate(vert_chunks) if vert_chunks else np.zeros((0, 3))

    # ---- static contact filtering (same weld body, parent-child, exclude,
    #      contype/conaffinity) -> sorted candidate pair list ---------------------------------
    name2body = {b.name: i for i, b in enumerate(B)}
    excl = {tuple(sorted((name2body[a], name2body[b]))) for a, b in sc.excludes}
    pairs = []
    for i in range(ng):
        for j in range(i + 1, ng):
            gi, gj = coll[i], coll[j]
            if not ((gi.contype & gj.conaffinity) or (gj.contype & gi.conaffinity)):
                continue
            b1, b2 = gi.body, gj.body
            w1, w2 = weldid[b1], weldid[b2]
            if w1 == w2:
                continue
            pw1, pw2 = weldid[B[w1].parent], weldid[B[w2].parent]
            if w1 != 0 and w2 != 0 and (w1 == pw2 or w2 == pw1):
                continue
            if tuple(sorted((b1, b2))) in excl:
                continue
            pairs.append((i, j))
    pairs = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)

    # ---- body-level box in the inertial frame for the OOBB reward (oobb_utils.py:155-172) --
    body_bvh_aabb = np.zeros((nb, 6))
    for bi in sc.prop_bodies:
        Ri = mm.quat2mat(body_iquat[bi])
        lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
        members = [g for g in sc.geoms if g.body == bi]
        for g in members:
            if g.type == GEOM_MESH:
                if g.hull is None:
                    continue
                # geom frame = mesh principal frame (MuJoCo re-centres meshes); tight box there
                vol, com, ic = mm.polyhedron_mass_properties(g.hull, g.hull_faces)
                _, Rm = mm.principal_frame(ic)
                loc = (g.hull - com) @ Rm
                c_l, h_l = 0.5 * (loc.min(0) + loc.max(0)), 0.5 * (loc.max(0) - loc.min(0))
                Rg = mm.quat2mat(g.quat) @ Rm
                cg = g.pos + mm.quat2mat(g.quat) @ (com + Rm @ c_l)
            elif g.type == GEOM_BOX:
                Rg, cg, h_l = mm.quat2mat(g.quat), g.pos, g.size
            else:
                continue
            Rrel = Ri.T @ Rg
            c_i = Ri.T @ (cg - body_ipos[bi])
            h_i = np.abs(Rrel) @ h_l
            lo, hi = np.minimum(lo, c_i - h_i), np.maximum(hi, c_i + h_i)
        body_bvh_aabb[bi] = np.concatenate([0.5 * (lo + hi), 0.5 * (hi - lo)])

    nu = len(sc.actuators)
    act = dict(gain=np.zeros(nu), bias=np.zeros((nu, 3)), ctrlrange=np.zeros((nu, 2)),
               forcerange=np.zeros((nu, 2)), ctrllimited=np.zeros(nu, np.int32),
               forcelimited=np.zeros(nu, np.int32), dof=np.zeros(nu, np.int32))
    jname2dof = {b.jnt_name: dofadr[i] for i, b in enumerate(B) if b.jnt_type in (JNT_HINGE, JNT_SLIDE)}
    for k, a in enumerate(sc.actuators):
        act["gain"][k] = _floats(a["gainprm"], 3)[0]
        act["bias"][k] = _floats(a["biasprm"], 3) if a["biastype"] == "affine" else 0
        if "ctrlrange" in a:
            act["ctrlrange"][k], act["ctrllimited"][k] = _floats(a["ctrlrange"]), 1
        if "forcerange" in a:
            act["forcerange"][k], act["forcelimited"][k] = _floats(a["forcerange"]), 1
        act["dof"][k] = jname2dof[a["joint"]]

    hinge = [i for i, b in enumerate(B) if b.jnt_type in (JNT_HINGE, JNT_SLIDE)]     # bodies with a one-dof joint
    free = [i for i, b in enumerate(B) if b.jnt_type == JNT_FREE]
    model = dict(
        nq=nq, nv=nv, nu=nu, nbody=nb, ngeom=ng, nvert=len(verts), npair=len(pairs),
        narm=len(hinge), nfree=len(free),
        opt_timestep=sc.option["timestep"], opt_gravity=sc.option["gravity"],
        opt_impratio=sc.option["impratio"], opt_tolerance=sc.option["tolerance"],
        opt_iterations=sc.option["iterations"], opt_cone_elliptic=int(sc.option["cone"] == "elliptic"),
        opt_mpr_tolerance=sc.option["mpr_tolerance"], opt_mpr_iterations=sc.option["mpr_iterations"],
        stat_meaninertia=float(np.mean(np.diag(M))),
        body_parent=np.array([b.parent for b in B], np.int32),
        body_pos=np.array([b.pos for b in B]), body_quat=np.array([b.quat for b in B]),
        body_ipos=body_ipos, body_iquat=body_iquat, body_mass=body_mass, body_inertia=body_inertia,
        body_jnttype=np.array([b.jnt_type for b in B], np.int32), body_qposadr=qposadr,
        body_dofadr=dofadr, body_weldid=weldid, body_invweight0=body_invweight0,
        body_bvh_aabb=body_bvh_aabb,
        arm_body=np.array(hinge, np.int32), free_body=np.array(free, np.int32),
        jnt_axis=np.array([B[i].jnt_axis for i in hinge]).reshape(-1, 3),
        jnt_range=np.array([B[i].jnt_range for i in hinge]).reshape(-1, 2),
        jnt_limited=np.array([B[i].jnt_limited for i in hinge], np.int32),
        jnt_solref=np.array([B[i].solreflimit for i in hinge]).reshape(-1, 2),
        jnt_solimp=np.array([B[i].solimplimit for i in hinge]).reshape(-1, 5),
        dof_solref=np.array([B[i].solreffriction for i in hinge]).reshape(-1, 2),
        dof_solimp=np.array([B[i].solimpfriction for i in hinge]).reshape(-1, 5),
        dof_armature=armature, dof_frictionloss=frictionloss, dof_damping=damping,
        dof_invweight0=dof_invweight0, dof_body=dof_body,
        act_gain=act["gain"], act_bias=act["bias"], act_ctrlrange=act["ctrlrange"],
        act_forcerange=act["forcerange"], act_ctrllimited=act["ctrllimited"],
        act_forcelimited=act["forcelimited"], act_dof=act["dof"],
        geom_type=g_arr["type"], geom_body=g_arr["body"], geom_pos=g_arr["pos"], geom_quat=g_arr["quat"],
        geom_size=g_arr["size"], geom_condim=g_arr["condim"], geom_friction=g_arr["friction"],
        geom_solref=g_arr["solref"], geom_solimp=g_arr["solimp"], geom_solmix=g_arr["solmix"],
        geom_margin=g_arr["margin"], geom_gap=g_arr["gap"], geom_priority=g_arr["priority"],
        geom_vertadr=g_arr["vertadr"], geom_vertnum=g_arr["vertnum"], geom_rbound=g_arr["rbound"],
        geom_center=g_arr["center"], geom_aabb=g_arr["aabb"],
        mesh_vert=verts, pair_geom=pairs,
    )
    if sc.general_tree:
        # fields only the general-tree scenes (ALOHA) carry; the SO100 blobs stay as they were
        neq = len(sc.equalities)
        jname2q = {b.jnt_name: qposadr[i] for i, b in enumerate(B) if b.jnt_type in (JNT_HINGE, JNT_SLIDE)}
        frc = np.array([B[i].actfrcrange if B[i].actfrcrange is not None else [0.0, 0.0] for i in hinge]).reshape(-1, 2)
        model.update(
            jnt_type=np.array([B[i].jnt_type for i in hinge], np.int32),
            jnt_actfrclimited=np.array([int(B[i].actfrcrange is not None) for i in hinge], np.int32),
            jnt_actfrcrange=frc,
            neq=neq,
            eq_dof=np.array([[jname2dof[e["joint1"]], jname2dof[e["joint2"]]] for e in sc.equalities], np.int32).reshape(-1, 2),
            eq_qposadr=np.array([[jname2q[e["joint1"]], jname2q[e["joint2"]]] for e in sc.equalities], np.int32).reshape(-1, 2),
            eq_polycoef=np.array([_floats(e["polycoef"], 5) for e in sc.equalities]).reshape(-1, 5),
            eq_solref=np.array([_floats(e["solref"], 2, [0.02, 1]) for e in sc.equalities]).reshape(-1, 2),
            eq_solimp=np.array([_floats(e["solimp"], 5, [0.9, 0.95, 0.001, 0.5, 2]) for e in sc.equalities]).reshape(-1, 5),
        )
    meta = dict(body_names=[b.name for b in B], geom_names=[g.name for g in coll],
                joint_names=[B[i].jnt_name for i in hinge], proxy_inertia=proxy_inertia,
                M0_diag=np.diag(M).tolist())
    return dict(model=model, meta=meta)
