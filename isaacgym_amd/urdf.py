"""URDF -> model tables: what replaces the UNVERIFIED placeholder data once the real assets exist.

The reference loads `g1_29dof_rev_1_0_pingpong_fixed_except_right_arm.urdf` / `g1_27dof.urdf` from absolute paths on its
author's machine (tasks/humanoid_pingpong_3_actor_tilt.py:415, tasks/humanoid_pingpong_3_actor_all_dof.py:470); neither file
is in the reference.  The model data in `scene.py` are therefore recalled numbers.  This module reads a URDF (links with
inertials, revolute / fixed joints with origins, axes and limits) and produces

  * `arm_specs(robot, joint_names)`: the 7-dof chain tables in the format of `scene.G1_RIGHT_ARM` (feed `scene.use_arm_tables`,
    then regenerate the compiled-in model with `python -m isaacgym_amd.modelgen` and rebuild),
  * `ta_model(robot, dof_joint_names, body_names, ...)`: the 28-link tree of the 27-dof task as a `scene.TAModel` — welded
    bodies (fixed joints) are merged into the link that carries them, exactly what `scene.build_ta_model` does by hand
    (`_lib.build_for_ta_model` compiles the chain-wave kernel for it),
  * `table_scene(robot)` / `ball_params(robot)`: the scene's other two assets — `pingpong_table.urdf` -> slab and net boxes,
    `small_ball.urdf` -> radius, mass, inertia factor (TT:496,502) — in the format `scene.build_config(table=, ball=)` takes.

`<collision>` geometry (box / sphere / cylinder / capsule with its origin) is read too and turned into the tables that are not
inertial data: `ball_shapes` (the capsule / sphere shapes the ball collides with, link-attached end points + radius),
`paddle_blade` (a cylinder on the paddle link -> blade centre, normal, radius, half thickness) and `ground_contacts` (the points
tested against the ground plane: the bottom corners of box collisions such as the G1's feet, the low points of the others).

`<mesh>` collisions carry no analytic shape.  The real G1 asset's collision geometry is mostly meshes, so dropping them in silence
would leave the ball flying through the robot: `parse` raises and lists every link whose mesh has no stand-in, unless the caller
names one per link (`mesh_bounds`: a sphere / capsule / cylinder / box placed at the mesh's own <origin>) or asks for
`on_mesh="warn"` (the dropped links are then listed in a warning and kept in `Robot.dropped_mesh_collisions`).

`write_g1_urdf()` emits the placeholder model as a URDF; tests/golden/g1_27dof_placeholder.urdf is its output, and the tests
check that parsing it reproduces the hand-built tables.  No real asset has been seen by this code.
"""
import math
import xml.etree.ElementTree as ET

import numpy as np

from . import scene


class Collision:
    """One <collision> element: `kind` in box / sphere / cylinder / capsule; `size` = (x, y, z) for a box, (radius,) for a sphere,
    (radius, length) for a cylinder or capsule (axis = local z, URDF convention); pose in the link frame."""

    def __init__(self, kind, size, xyz=(0.0, 0.0, 0.0), rpy=(0.0, 0.0, 0.0)):
        self.kind, self.size = kind, tuple(float(v) for v in size)
        self.xyz = np.asarray(xyz, dtype=np.float64)
        self.rot = scene.rpy_to_rot(*rpy)

    def segment(self):
        """(a, b, radius) of a sphere / cylinder / capsule in the link frame (a == b for a sphere)."""
        if self.kind == "sphere":
            return self.xyz.copy(), self.xyz.copy(), self.size[0]
        if self.kind in ("cylinder", "capsule"):
            half = self.rot @ np.array([0.0, 0.0, 0.5 * self.size[1]])
            return self.xyz - half, self.xyz + half, self.size[0]
        raise ValueError(f"a {self.kind} collision has no axis segment")

    def corners(self):
        """The eight corners of a box in the link frame."""
        if self.kind != "box":
            raise ValueError(f"a {self.kind} collision has no corners")
        h = 0.5 * np.asarray(self.size)
        return [self.xyz + self.rot @ (h * np.array(s)) for s in ((-1, -1, -1), (-1, 1, -1), (1, -1, -1), (1, 1, -1), (-1, -1, 1), (-1, 1, 1), (1, -1, 1), (1, 1, 1))]


class Link:
    def __init__(self, name, mass=0.0, com=(0.0, 0.0, 0.0), com_rpy=(0.0, 0.0, 0.0), inertia=None):
        self.collisions = []
        self.name, self.mass = name, float(mass)
        self.com = np.asarray(com, dtype=np.float64)
        self.com_rot = scene.rpy_to_rot(*com_rpy)
        self.inertia = np.zeros((3, 3)) if inertia is None else np.asarray(inertia, dtype=np.float64)   # about the com, inertial-frame axes


class Joint:
    def __init__(self, name, jtype, parent, child, xyz, rpy, axis, lower, upper, effort, velocity):
        self.name, self.type, self.parent, self.child = name, jtype, parent, child
        self.xyz, self.rpy = np.asarray(xyz, dtype=np.float64), tuple(float(v) for v in rpy)
        self.axis = np.asarray(axis, dtype=np.float64)
        self.lower, self.upper, self.effort, self.velocity = float(lower), float(upper), float(effort), float(velocity)


class Robot:
    def __init__(self, name, links, joints, dropped_mesh_collisions=()):
        self.name, self.links, self.joints = name, links, joints
        self.dropped_mesh_collisions = list(dropped_mesh_collisions)   # [(link, mesh file)]: only with parse(on_mesh="warn")
        self.joint_of_child = {j.child: j for j in joints.values()}

    def root(self):
        children = set(self.joint_of_child)
        roots = [n for n in self.links if n not in children]
        if len(roots) != 1:
            raise ValueError(f"URDF must have exactly one root link, found {roots}")
        return roots[0]


def _floats(text, n, default):
    if text is None:
        return tuple(default)
    v = tuple(float(x) for x in text.split())
    if len(v) != n:
        raise ValueError(f"expected {n} numbers, got {text!r}")
    return v


def _mesh_stand_in(link_name, spec, mesh_xyz, mesh_rpy):
    """A primitive named by the caller in place of a link's <mesh> collision, placed at the mesh's own <origin>:
    dict(kind="sphere", radius=) | dict(kind="capsule" | "cylinder", radius=, length=) | dict(kind="box", size=(x, y, z)), each with
    optional xyz / rpy relative to that origin."""
    kind = spec.get("kind")
    if kind == "sphere":
        size = (float(spec["radius"]),)
    elif kind in ("capsule", "cylinder"):
        size = (float(spec["radius"]), float(spec["length"]))
    elif kind == "box":
        size = tuple(float(v) for v in spec["size"])
        if len(size) != 3:
            raise ValueError(f"mesh_bounds[{link_name!r}]: a box needs size = (x, y, z)")
    else:
        raise ValueError(f"mesh_bounds[{link_name!r}]: kind {kind!r} is not sphere / capsule / cylinder / box")
    if any(v <= 0.0 for v in size):
        raise ValueError(f"mesh_bounds[{link_name!r}]: sizes must be positive")
    c = Collision(kind, size, spec.get("xyz", (0.0, 0.0, 0.0)), spec.get("rpy", (0.0, 0.0, 0.0)))
    m_rot = scene.rpy_to_rot(*mesh_rpy)
    c.xyz = np.asarray(mesh_xyz, dtype=np.float64) + m_rot @ c.xyz
    c.rot = m_rot @ c.rot
    return c


def parse(text, mesh_bounds=None, on_mesh="error"):
    """URDF text -> Robot.  Only what the dynamics need: inertials, joint origins / axes / limits, the tree, and the collision
    primitives.  mesh_bounds: {link name: stand-in primitive (or a list of them, one per mesh collision of that link)} for <mesh>
    collisions; a mesh without one is an error that names every such link (on_mesh="error", the default) or a warning
    (on_mesh="warn") — never a silent skip."""
    if on_mesh not in ("error", "warn"):
        raise ValueError("on_mesh is 'error' or 'warn'")
    mesh_bounds = dict(mesh_bounds or {})
    root = ET.fromstring(text)
    if root.tag != "robot":
        raise ValueError("not a URDF: the root element is not <robot>")
    links, joints = {}, {}
    dropped, mesh_seen = [], {}

    def collisions_of(e):
        out = []
        for c in e.findall("collision"):
            o, g = c.find("origin"), c.find("geometry")
            xyz = _floats(o.get("xyz") if o is not None else None, 3, (0, 0, 0))
            rpy = _floats(o.get("rpy") if o is not None else None, 3, (0, 0, 0))
            if g is None or len(g) != 1:
                raise ValueError(f"link {e.get('name')}: a <collision> needs exactly one geometry")
            k = g[0]
            if k.tag == "box":
                out.append(Collision("box", _floats(k.get("size"), 3, ()), xyz, rpy))
            elif k.tag == "sphere":
                out.append(Collision("sphere", (float(k.get("radius")),), xyz, rpy))
            elif k.tag in ("cylinder", "capsule"):
                out.append(Collision(k.tag, (float(k.get("radius")), float(k.get("length"))), xyz, rpy))
            elif k.tag == "mesh":
                name = e.get("name")
                spec = mesh_bounds.get(name)
                if isinstance(spec, (list, tuple)):                      # one stand-in per mesh collision of the link, in order
                    spec = spec[mesh_seen.get(name, 0)] if mesh_seen.get(name, 0) < len(spec) else None
                mesh_seen[name] = mesh_seen.get(name, 0) + 1
                if spec is None:
                    dropped.append((name, k.get("filename", "")))     # reported below: never skipped in silence
                    continue
                out.append(_mesh_stand_in(name, spec, xyz, rpy))
            else:
                raise ValueError(f"link {e.get('name')}: collision geometry <{k.tag}> is not supported")
        return out

    for e in root.findall("link"):
        name = e.get("name")
        inert = e.find("inertial")
        if inert is None:
            links[name] = Link(name)
            links[name].collisions = collisions_of(e)
            continue
        o = inert.find("origin")
        com = _floats(o.get("xyz") if o is not None else None, 3, (0, 0, 0))
        rpy = _floats(o.get("rpy") if o is not None else None, 3, (0, 0, 0))
        mass = float(inert.find("mass").get("value"))
        i = inert.find("inertia")
        ixx, iyy, izz = (float(i.get(k, 0.0)) for k in ("ixx", "iyy", "izz"))
        ixy, ixz, iyz = (float(i.get(k, 0.0)) for k in ("ixy", "ixz", "iyz"))
        links[name] = Link(name, mass, com, rpy, [[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
        links[name].collisions = collisions_of(e)
    for e in root.findall("joint"):
        name, jtype = e.get("name"), e.get("type")
        if jtype not in ("revolute", "continuous", "fixed"):
            raise ValueError(f"joint {name}: type {jtype!r} is not supported (revolute / continuous / fixed)")
        if e.find("mimic") is not None:
            raise ValueError(f"joint {name}: <mimic> couples it to another joint; the kernels drive every dof independently (no coupled joints in the tables)")
        o = e.find("origin")
        xyz = _floats(o.get("xyz") if o is not None else None, 3, (0, 0, 0))
        rpy = _floats(o.get("rpy") if o is not None else None, 3, (0, 0, 0))
        ax = e.find("axis")
        axis = _floats(ax.get("xyz") if ax is not None else None, 3, (1, 0, 0))
        lim = e.find("limit")
        lower = float(lim.get("lower", -math.pi)) if lim is not None else -math.pi
        upper = float(lim.get("upper", math.pi)) if lim is not None else math.pi
        effort = float(lim.get("effort", 0.0)) if lim is not None else 0.0
        vel = float(lim.get("velocity", 0.0)) if lim is not None else 0.0
        joints[name] = Joint(name, jtype, e.find("parent").get("link"), e.find("child").get("link"), xyz, rpy, axis, lower, upper, effort, vel)
    for j in joints.values():
        if j.parent not in links or j.child not in links:
            raise ValueError(f"joint {j.name} names a link that does not exist")
    unknown = sorted(set(mesh_bounds) - set(links))
    if unknown:
        raise ValueError(f"mesh_bounds names links the URDF does not have: {unknown}")
    if dropped:
        what = ", ".join(f"{n} ({f})" if f else n for n, f in dropped)
        msg = (f"{len(dropped)} <mesh> collision(s) have no analytic stand-in and would be dropped — the ball would pass through these links: {what}. "
               "Give each a primitive through parse(..., mesh_bounds={link: dict(kind='capsule', radius=…, length=…)}) or accept the loss "
               "explicitly with on_mesh='warn'.")
        if on_mesh == "error":
            raise ValueError(msg)
        import warnings
        warnings.warn(msg, stacklevel=2)
    return Robot(root.get("name", ""), links, joints, dropped)


def load(path, mesh_bounds=None, on_mesh="error"):
    with open(path) as f:
        return parse(f.read(), mesh_bounds=mesh_bounds, on_mesh=on_mesh)


def perturbed(text, mass_scale=1.0, origin_shift=None, limits=None):
    """URDF text -> URDF text of a different asset: every link's mass and inertia scaled by mass_scale, joint origins moved by
    origin_shift {joint: (dx, dy, dz)} (a link length), joint ranges replaced by limits {joint: (lower, upper)}.  What the N3 tests use
    to drive the kernels from a model that is NOT the placeholder tables; also a cheap what-if tool once a real asset exists."""
    root = ET.fromstring(text)
    for link in root.findall("link"):
        inert = link.find("inertial")
        if inert is None:
            continue
        m = inert.find("mass")
        m.set("value", repr(float(m.get("value")) * mass_scale))
        i = inert.find("inertia")
        for k in ("ixx", "iyy", "izz", "ixy", "ixz", "iyz"):
            i.set(k, repr(float(i.get(k, 0.0)) * mass_scale))
    seen = set()
    for j in root.findall("joint"):
        name = j.get("name")
        if origin_shift and name in origin_shift:
            o = j.find("origin")
            xyz = np.asarray(_floats(o.get("xyz"), 3, (0, 0, 0))) + np.asarray(origin_shift[name], dtype=np.float64)
            o.set("xyz", " ".join(repr(float(v)) for v in xyz))
            seen.add(name)
        if limits and name in limits:
            lim = j.find("limit")
            lim.set("lower", repr(float(limits[name][0])))
            lim.set("upper", repr(float(limits[name][1])))
            seen.add(name)
    missing = (set(origin_shift or ()) | set(limits or ())) - seen
    if missing:
        raise ValueError(f"no such joint(s): {sorted(missing)}")
    return ET.tostring(root, encoding="unicode")


def _axis_index(axis, name):
    """The kernels take joints about a coordinate axis of the child frame (every G1 joint is one); a negative axis flips the
    sign convention of q, which the tables do not carry."""
    a = np.asarray(axis, dtype=np.float64)
    k = int(np.argmax(np.abs(a)))
    e = np.zeros(3)
    e[k] = 1.0
    if not np.allclose(a, e, atol=1e-9):
        raise ValueError(f"joint {name}: axis {tuple(a)} is not +x, +y or +z of the child frame")
    return k


def _link_parts(link, offset=np.zeros(3), rot=np.eye(3)):
    """(mass, com, I_com, R) of a link's inertial seen from a frame in which the link sits at `offset` / `rot`."""
    if link.mass <= 0.0:
        return None
    return (link.mass, offset + rot @ link.com, link.inertia, rot @ link.com_rot)


def arm_specs(robot, joint_names, body_index=None):
    """The chain moved by `joint_names` (base -> tip) in the format of scene.G1_RIGHT_ARM.  `inertia` is the diagonal when the
    link's inertia tensor is diagonal in the link axes, else the 6-vector xx yy zz xy xz yz."""
    specs = []
    for name in joint_names:
        j = robot.joints[name]
        if j.type == "fixed":
            raise ValueError(f"joint {name} is fixed: a dof of the chain must be revolute")
        link = robot.links[j.child]
        inertia = link.com_rot @ link.inertia @ link.com_rot.T
        off = [inertia[0, 1], inertia[0, 2], inertia[1, 2]]
        inert = tuple(np.diag(inertia)) if np.allclose(off, 0.0, atol=1e-12) else tuple(np.diag(inertia)) + tuple(off)
        specs.append(dict(name=j.child, body=(body_index or {}).get(j.child, -1), xyz=tuple(j.xyz), rpy=j.rpy, axis=_axis_index(j.axis, name),
                          limits=(j.lower, j.upper), mass=link.mass, com=tuple(link.com), inertia=inert, effort=j.effort, vel=j.velocity))
    return specs


def ta_model(robot, dof_joint_names, body_names, gains=None, armature=None, contacts=None, bound=None):
    """scene.TAModel of the 27-dof task from a URDF: link 0 = the root link, link k = the child of dof k's joint; every other
    body is welded (through fixed joints) to one of them and is merged into it.  `body_names`: the 40 rigid-body names in Isaac
    Gym's order (pingpong_note.txt:33).  Contacts / bound default to scene's (link indices of the G1 tree)."""
    if len(dof_joint_names) != scene.TA_NUM_DOF or len(body_names) != scene.NUM_HUMANOID_BODIES:
        raise ValueError("the 27-dof task has 27 dofs and 40 rigid bodies")
    gains = list(scene.TA_P_GAINS) if gains is None else list(gains)
    body_index = {n: i for i, n in enumerate(body_names)}
    root = robot.root()
    link_names = [root] + [robot.joints[n].child for n in dof_joint_names]
    link_index = {n: i for i, n in enumerate(link_names)}

    def carrier(name):
        """(movable link index, offset, rotation) of a body: walk up through fixed joints."""
        off, rot = np.zeros(3), np.eye(3)
        while name not in link_index:
            j = robot.joint_of_child[name]
            if j.type != "fixed":
                raise ValueError(f"link {name} hangs on movable joint {j.name}, which is not one of the 27 dofs")
            r = scene.rpy_to_rot(*j.rpy)
            off, rot = j.xyz + r @ off, r @ rot
            name = j.parent
        return link_index[name], off, rot

    m = scene.TAModel()
    merged = {i: [] for i in range(scene.TA_NUM_LINKS)}
    fixed = []
    for name in body_names:
        if name in link_index:
            p = _link_parts(robot.links[name])
            if p is not None:
                merged[link_index[name]].insert(0, p)
        else:
            li, off, rot = carrier(name)
            p = _link_parts(robot.links[name], off, rot)
            if p is not None:
                merged[li].append(p)
            fixed.append((body_index[name], li, off, rot))
    if len(fixed) != scene.TA_NUM_FIXED:
        raise ValueError(f"expected {scene.TA_NUM_FIXED} welded bodies, found {len(fixed)}")
    for i, name in enumerate(link_names):
        L = m.link[i]
        if i == 0:
            L.parent, L.axis, L.body = -1, 0, body_index[name]
            scene._set(L.origin_xyz, (0, 0, 0))
            scene._set(L.origin_rot, np.eye(3).reshape(-1))
            L.lower = L.upper = L.kp = L.kd = L.effort = L.vel_limit = L.armature = 0.0
        else:
            j = robot.joints[dof_joint_names[i - 1]]
            # the parent may be a welded body (e.g. the elbow link welded under the shoulder-roll link): fold its offset in
            pi, poff, prot = carrier(j.parent)
            if pi >= i:
                raise ValueError("dof order must list parents before children")
            L.parent, L.axis, L.body = pi, _axis_index(j.axis, j.name), body_index[name]
            scene._set(L.origin_xyz, poff + prot @ j.xyz)
            scene._set(L.origin_rot, (prot @ scene.rpy_to_rot(*j.rpy)).reshape(-1))
            L.lower, L.upper = min(j.lower, j.upper), max(j.lower, j.upper)
            L.kp, L.kd = gains[i - 1], gains[i - 1] / 40.0                      # TA:757-774
            L.effort, L.vel_limit = j.effort, j.velocity
            L.armature = scene.TA_ARMATURE if armature is None else armature
        if not merged[i]:
            raise ValueError(f"link {name} has no mass")
        mass, com, inertia = scene.composite_inertial(merged[i])
        L.mass = mass
        scene._set(L.com, com)
        scene._set(L.inertia, scene._inertia_vec(inertia))
    for k, (body, li, off, rot) in enumerate(sorted(fixed, key=lambda t: t[0])):
        f = m.fixed[k]
        f.body, f.link = body, li
        scene._set(f.xyz, off)
        scene._set(f.rot, rot.reshape(-1))
    scene.fill_ta_contacts_and_limits(m, contacts=contacts, bound=bound)
    return m


# ------------------------------------------------------------------------------------------------------------------
# <collision> geometry -> the tables that are not inertial data
def _carrier_of(robot, movable_links):
    """name -> (index into `movable_links`, offset, rotation): the movable link a body is (transitively) welded to."""
    index = {n: i for i, n in enumerate(movable_links)}

    def carrier(name):
        off, rot = np.zeros(3), np.eye(3)
        while name not in index:
            j = robot.joint_of_child.get(name)
            if j is None or j.type != "fixed":
                raise ValueError(f"link {name} is not welded to one of the movable links")
            r = scene.rpy_to_rot(*j.rpy)
            off, rot = j.xyz + r @ off, r @ rot
            name = j.parent
        return index[name], off, rot
    return carrier


def ball_shapes(robot, movable_links, link_names=None):
    """The capsule / sphere shapes the ball can hit, from the sphere / cylinder / capsule collisions of `link_names` (default: every
    link that has one): [dict(link = index into movable_links, a, b, radius)], end points in that movable link's frame.  Boxes are
    not ball shapes here (the table and the net are the scene's slabs; a box on the humanoid would need its own narrow phase)."""
    carrier = _carrier_of(robot, movable_links)
    out = []
    for name in (link_names if link_names is not None else list(robot.links)):
        for c in robot.links[name].collisions:
            if c.kind == "box":
                continue
            li, off, rot = carrier(name)
            a, b, r = c.segment()
            out.append(dict(link=li, a=tuple(off + rot @ a), b=tuple(off + rot @ b), radius=r, body=name))
    return out


def paddle_blade(robot, movable_links, paddle_link):
    """The blade from the (single) cylinder collision of `paddle_link`: centre, unit normal (the cylinder's axis), radius and half
    thickness, in the frame of the movable link that carries the paddle — the fields of ppenv_config.paddle_*."""
    cyl = [c for c in robot.links[paddle_link].collisions if c.kind == "cylinder"]
    if len(cyl) != 1:
        raise ValueError(f"link {paddle_link}: expected one cylinder collision for the blade, found {len(cyl)}")
    li, off, rot = _carrier_of(robot, movable_links)(paddle_link)
    c = cyl[0]
    n = rot @ c.rot @ np.array([0.0, 0.0, 1.0])
    return dict(link=li, center=tuple(off + rot @ c.xyz), normal=tuple(n / np.linalg.norm(n)), radius=c.size[0], half_thickness=0.5 * c.size[1])


def ground_contacts(robot, movable_links, link_names, down=(0.0, 0.0, -1.0), max_points=scene.TA_MAX_CONTACTS):
    """Points tested against the ground plane, [(movable link index, point in its frame)], from the collisions of `link_names`:
    a box (the G1's feet) gives the four corners of its face that looks along `down` at the zero pose; a sphere its lowest point;
    a cylinder / capsule the low points under its two ends."""
    carrier = _carrier_of(robot, movable_links)
    d = np.asarray(down, dtype=np.float64)
    out = []
    for name in link_names:
        li, off, rot = carrier(name)
        for c in robot.links[name].collisions:
            if c.kind == "box":
                pts = sorted((off + rot @ p for p in c.corners()), key=lambda p: -float(p @ d))[:4]
                pts = sorted(pts, key=lambda p: (round(float(p[0]), 9), round(float(p[1]), 9)))
            elif c.kind == "sphere":
                pts = [off + rot @ c.xyz + d * c.size[0]]
            else:
                a, b, r = c.segment()
                pts = [off + rot @ a + d * r, off + rot @ b + d * r]
            out += [(li, tuple(float(x) for x in p)) for p in pts]
    if len(out) > max_points:
        raise ValueError(f"{len(out)} ground-contact points, the model holds {max_points}")
    return out


# ------------------------------------------------------------------------------------------------------------------
# The other two assets of the scene (TT:496 `pingpong_table.urdf`, TT:502 `small_ball.urdf`; T3:480,486, TN:501,507, T4:505,511, TA:551,557)
def _static_frames(robot):
    """{link: (offset, rotation)} of every link in the root link's frame for an asset without movable joints (the table: loaded with
    fix_base_link = True, TT:493).  A movable joint would make the slab a mechanism, which the step kernels' axis-aligned boxes are not."""
    root = robot.root()
    frames = {root: (np.zeros(3), np.eye(3))}
    pending = [j for j in robot.joints.values()]
    while pending:
        rest = []
        for j in pending:
            if j.parent in frames:
                if j.type != "fixed":
                    raise ValueError(f"joint {j.name} is {j.type}: a table asset must be one rigid piece (fixed joints only)")
                po, pr = frames[j.parent]
                r = scene.rpy_to_rot(*j.rpy)
                frames[j.child] = (po + pr @ j.xyz, pr @ r)
            else:
                rest.append(j)
        if len(rest) == len(pending):
            raise ValueError(f"links not connected to the root: {sorted(j.child for j in rest)}")
        pending = rest
    return frames


def _aligned_box(center, rot, size, what):
    """(centre, half extents along the ROOT frame's axes) of a box whose own axes are a signed permutation of them — the kernels' slabs
    (ppenv_box: centre + half extents) are axis-aligned; anything else raises instead of being squared off in silence."""
    a = np.abs(rot)
    if not np.allclose(a, np.round(a), atol=1e-6) or not np.allclose(np.round(a).sum(axis=0), 1.0) or not np.allclose(np.round(a).sum(axis=1), 1.0):
        raise ValueError(f"{what}: the box is rotated off the table frame's axes; the step kernels' slabs are axis-aligned")
    return np.asarray(center, dtype=np.float64), 0.5 * (np.round(a) @ np.asarray(size, dtype=np.float64))


def table_scene(robot, surface_tol=5e-3):
    """A table asset -> what scene.build_config(table=...) takes (the keys of scene.TABLE_GEOM + the offsets a file may carry).

    The reference gives the table's material to `table_shape_props[0]` only (TT:580-582): shape 0 — the first <collision> of the first
    link that has one, Isaac Gym's shape order — is therefore the playing surface, the SLAB.  The NET is the box that stands on the
    slab's top face (its bottom within `surface_tol` of it) and is the thinnest such box along the table's length.  Every other box
    (legs, frame) lies below the top face; the step kernels have no shape for them, so they are returned in `ignored` — and rejected
    when one reaches above the playing surface or beyond the slab's footprint, where a ball could meet it before the slab.
    Lengths in metres, in the frame of the table actor's root (the pose TT:575 places at (1.75, 0, 0))."""
    frames = _static_frames(robot)
    boxes = []
    for name, link in robot.links.items():
        off, rot = frames[name]
        for k, c in enumerate(link.collisions):
            if c.kind != "box":
                raise ValueError(f"link {name}: a table asset's collisions must be boxes, found a {c.kind}")
            ctr, half = _aligned_box(off + rot @ c.xyz, rot @ c.rot, c.size, f"link {name}, collision {k}")
            boxes.append(dict(link=name, index=k, center=ctr, half=half))
    if not boxes:
        raise ValueError("the table asset has no <collision> box")
    slab = boxes[0]
    top = slab["center"][2] + slab["half"][2]
    area = lambda b: b["half"][0] * b["half"][1]
    if any(area(b) > area(slab) * (1 + 1e-9) for b in boxes[1:]):
        raise ValueError("shape 0 of the table asset is not its largest horizontal box: the reference puts the table material on shape 0 "
                         "(TT:580-582), so the playing surface must be the first <collision>")
    standing = [b for b in boxes[1:] if abs((b["center"][2] - b["half"][2]) - top) <= surface_tol]
    if len(standing) != 1:
        raise ValueError(f"expected exactly one box standing on the playing surface (the net), found {len(standing)}")
    net = standing[0]
    if net["half"][0] > net["half"][1]:
        raise ValueError("the net must run across the table (thin along x, the table's length)")
    ignored = []
    for b in boxes[1:]:
        if b is net:
            continue
        if b["center"][2] + b["half"][2] > top + surface_tol:
            raise ValueError(f"link {b['link']}: a second box reaches above the playing surface; only the slab and the net are ball shapes")
        if np.any(np.abs(b["center"][:2] - slab["center"][:2]) + b["half"][:2] > slab["half"][:2] + surface_tol):
            raise ValueError(f"link {b['link']}: a box below the surface sticks out of the slab's footprint; the kernels would let the ball through it")
        ignored.append((b["link"], b["index"]))
    return dict(length=float(2.0 * slab["half"][0]), width=float(2.0 * slab["half"][1]), top_z=float(top), slab=float(2.0 * slab["half"][2]),
                net_height=float(2.0 * net["half"][2]), net_overhang=float(net["half"][1] - slab["half"][1]), net_half_thickness=float(net["half"][0]),
                offset_xy=(float(slab["center"][0]), float(slab["center"][1])), net_offset_xy=(float(net["center"][0]), float(net["center"][1])),
                net_bottom_z=float(net["center"][2] - net["half"][2]), ignored=ignored)


def ball_params(robot, isotropy_tol=1e-6):
    """A ball asset -> what scene.build_config(ball=...) takes: radius of its (single) sphere collision, mass, and the inertia factor
    k = I / (m r^2) of its <inertial> (2/5 solid, 2/3 thin shell) — the three fields of ppenv_config the kernels' contact code reads
    (ball_radius, ball_mass, ball_inertia_factor).  The angular damping is an Isaac Gym asset option (AssetOptions.angular_damping,
    default 0.5; the reference leaves it untouched, TT:499-502), not URDF data: scene.BALL_GEOM keeps it."""
    root = robot.root()
    if len(robot.links) != 1 or robot.joints:
        raise ValueError("a ball asset is one free link")
    link = robot.links[root]
    spheres = [c for c in link.collisions if c.kind == "sphere"]
    if len(spheres) != 1 or len(link.collisions) != 1:
        raise ValueError(f"link {root}: expected exactly one sphere collision, found {[c.kind for c in link.collisions]}")
    if not np.allclose(spheres[0].xyz + 0.0, link.com, atol=1e-9):
        raise ValueError("the ball's centre of mass must sit at the centre of its sphere (the kernels integrate a homogeneous ball)")
    r, m = spheres[0].size[0], link.mass
    if r <= 0.0 or m <= 0.0:
        raise ValueError("the ball needs a positive radius and mass")
    inertia = link.com_rot @ link.inertia @ link.com_rot.T
    d = np.diag(inertia)
    if not np.allclose(inertia, np.diag(d), atol=isotropy_tol * d.max()) or not np.allclose(d, d[0], rtol=isotropy_tol):
        raise ValueError(f"the ball's inertia tensor is not isotropic: {inertia.tolist()}")
    return dict(radius=float(r), mass=float(m), inertia_factor=float(d[0] / (m * r * r)))


# ------------------------------------------------------------------------------------------------------------------
# The placeholder model as a URDF (used to produce the test fixture; also documents what a real asset has to provide)
G1_BODY_NAMES = ["pelvis", "imu_in_pelvis", "left_hip_pitch_link", "left_hip_roll_link", "left_hip_yaw_link", "left_knee_link",
                 "left_ankle_pitch_link", "left_ankle_roll_link", "pelvis_contour_link", "right_hip_pitch_link", "right_hip_roll_link",
                 "right_hip_yaw_link", "right_knee_link", "right_ankle_pitch_link", "right_ankle_roll_link", "waist_yaw_link", "waist_roll_link",
                 "torso_link", "d435_link", "head_link", "imu_in_torso", "left_shoulder_pitch_link", "left_shoulder_roll_link",
                 "left_shoulder_yaw_link", "left_elbow_link", "left_wrist_roll_link", "left_wrist_pitch_link", "left_wrist_yaw_link",
                 "left_rubber_hand", "logo_link", "mid360_link", "right_shoulder_pitch_link", "right_shoulder_roll_link", "right_shoulder_yaw_link",
                 "right_elbow_link", "right_wrist_roll_link", "right_wrist_pitch_link", "right_wrist_yaw_link", "right_rubber_hand",
                 "pingpong_paddle"]   # tasks/pingpong_note.txt:33


def _joint_name(link_name):
    return link_name.replace("_link", "") + "_joint" if link_name.endswith("_link") else link_name + "_joint"


def ta_dof_joint_names():
    """The 27 dof joints in the order of TA:1303-1311 (left leg, right leg, waist, left arm, the five right-arm dofs)."""
    legs = ["hip_pitch", "hip_roll", "hip_yaw", "knee", "ankle_pitch", "ankle_roll"]
    arm = ["shoulder_pitch", "shoulder_roll", "shoulder_yaw", "elbow", "wrist_roll", "wrist_pitch", "wrist_yaw"]
    names = [f"left_{n}_joint" for n in legs] + [f"right_{n}_joint" for n in legs] + ["waist_yaw_joint", "waist_roll_joint", "torso_joint"]
    names += [f"left_{n}_joint" for n in arm] + [f"right_{n}_joint" for n in ("shoulder_pitch", "shoulder_roll", "wrist_roll", "wrist_pitch", "wrist_yaw")]
    return names


def write_g1_urdf(weld_right_elbow=True):
    """The placeholder G1 + paddle of scene.py as URDF text.  weld_right_elbow: the 27-dof asset (right shoulder-yaw and elbow
    fixed); False gives the 29-dof-style right arm the 7-dof tasks use."""
    def fmt(v):
        return " ".join(repr(float(x)) for x in v)
    out = ['<?xml version="1.0"?>', '<robot name="g1_placeholder">']

    def link(name, mass=0.0, com=(0, 0, 0), inertia=(0, 0, 0)):
        if mass <= 0.0:
            out.append(f'  <link name="{name}"/>')
            return
        i6 = tuple(float(x) for x in (tuple(inertia) + (0.0, 0.0, 0.0) if len(inertia) == 3 else tuple(inertia)))
        out.append(f'  <link name="{name}"><inertial><origin xyz="{fmt(com)}" rpy="0 0 0"/><mass value="{float(mass)!r}"/>'
                   f'<inertia ixx="{i6[0]!r}" iyy="{i6[1]!r}" izz="{i6[2]!r}" ixy="{i6[3]!r}" ixz="{i6[4]!r}" iyz="{i6[5]!r}"/></inertial></link>')

    def joint(child, parent, xyz, rpy=(0, 0, 0), axis=None, limits=None, effort=0.0, vel=0.0, name=None):
        name = name or _joint_name(child)
        if axis is None:
            out.append(f'  <joint name="{name}" type="fixed"><origin xyz="{fmt(xyz)}" rpy="{fmt(rpy)}"/><parent link="{parent}"/><child link="{child}"/></joint>')
        else:
            ax = ["1 0 0", "0 1 0", "0 0 1"][axis]
            out.append(f'  <joint name="{name}" type="revolute"><origin xyz="{fmt(xyz)}" rpy="{fmt(rpy)}"/><parent link="{parent}"/><child link="{child}"/>'
                       f'<axis xyz="{ax}"/><limit lower="{float(limits[0])!r}" upper="{float(limits[1])!r}" effort="{float(effort)!r}" velocity="{float(vel)!r}"/></joint>')

    def chain(specs, parent, welded=()):
        for k, s in enumerate(specs):
            link(s["name"], s["mass"], s["com"], s["inertia"])
            if k in welded:
                joint(s["name"], parent, s["xyz"], s["rpy"])
            else:
                joint(s["name"], parent, s["xyz"], s["rpy"], s["axis"], s["limits"], s["effort"], s["vel"])
            parent = s["name"]
        return parent

    P = scene.G1_PELVIS
    link("pelvis", P["mass"], P["com"], P["inertia"])
    for w in scene.G1_PELVIS_WELDED:
        link(w["name"], w["mass"], w.get("com", (0, 0, 0)), w.get("inertia", (0, 0, 0)))
        joint(w["name"], "pelvis", w["xyz"])
    chain(scene._leg("left"), "pelvis")
    chain(scene._leg("right"), "pelvis")
    waist = [dict(s) for s in scene.G1_WAIST]
    for s in waist[:2]:
        link(s["name"], s["mass"], s["com"], s["inertia"])
    joint("waist_yaw_link", "pelvis", waist[0]["xyz"], waist[0]["rpy"], waist[0]["axis"], waist[0]["limits"], waist[0]["effort"], waist[0]["vel"])
    joint("waist_roll_link", "waist_yaw_link", waist[1]["xyz"], waist[1]["rpy"], waist[1]["axis"], waist[1]["limits"], waist[1]["effort"], waist[1]["vel"])
    t = waist[2]
    link("torso_link", t["mass"], t["com"], t["inertia"])
    joint("torso_link", "waist_roll_link", t["xyz"], t["rpy"], t["axis"], t["limits"], t["effort"], t["vel"], name="torso_joint")
    for w in scene.G1_TORSO_WELDED:
        link(w["name"], w["mass"], w.get("com", (0, 0, 0)), w.get("inertia", (0, 0, 0)))
        joint(w["name"], "torso_link", w["xyz"])
    left = [scene._mirror_arm(s) for s in scene.G1_RIGHT_ARM]
    tip = chain(left, "torso_link")
    H = scene.G1_HAND
    link("left_rubber_hand", H["mass"], (H["com"][0], -H["com"][1], H["com"][2]), H["inertia"])
    joint("left_rubber_hand", tip, (H["xyz"][0], -H["xyz"][1], H["xyz"][2]))
    tip = chain(scene.G1_RIGHT_ARM, "torso_link", welded=(2, 3) if weld_right_elbow else ())
    link("right_rubber_hand", H["mass"], H["com"], H["inertia"])
    joint("right_rubber_hand", tip, H["xyz"])
    pd = scene.PADDLE
    m, r, n = pd["mass"], pd["radius"], np.asarray(pd["normal"], dtype=np.float64)
    i_disc = 0.25 * m * r * r * np.eye(3) + 0.25 * m * r * r * np.outer(n, n)
    link("pingpong_paddle", m, (0, 0, 0), tuple(np.diag(i_disc)) + (i_disc[0, 1], i_disc[0, 2], i_disc[1, 2]))
    joint("pingpong_paddle", "right_rubber_hand", pd["xyz_from_hand"])
    out.append("</robot>")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    import sys
    sys.stdout.write(write_g1_urdf(weld_right_elbow="--29dof" not in sys.argv))
