"""Vectorised numpy-float32 restatement of the hot path — TEST INFRASTRUCTURE ONLY.

Same arithmetic as oracle/bx_math.h / physics_ref.h (the reference's src/ecs/Transform.cpp:18-36 via bx, and the
free-body step behind src/physics/PhysicsSystem.cpp:863), one IEEE binary32 rounding per operation (numpy ufuncs on
float32 arrays never fuse or widen), but over whole arrays, so that parity can be checked bit for bit at BASELINE.json's
full sizes where the hash-map C++ oracle would take minutes.  tests/test_np_oracle.py pins it against the C++ oracle.
Scope: non-spinning Dynamic bodies (the only kind the reference can create without contacts) + the transform resolve.
"""
from __future__ import annotations

import numpy as np

F = np.float32
K_PI_HALF = F(1.5707963267948966)
K_INV_PI = F(0.31830988618379067)


def _bits(u):
    return np.array(u, np.uint32).view(np.float32)[()]


_EVEN = [F(-0.5), _bits(0x3d2aaaa4), _bits(0xbab60981), _bits(0x37cfab9c), _bits(0xb48b634d)]
_ODD = [_bits(0xbe2aaaab), _bits(0x3c088898), _bits(0xb9501096), _bits(0x363938a8), _bits(0xb2d70013)]


def bx_floor(a):
    """bx::floor (int-cast based): equals floor() for |a| < 2^31 except that it returns +0 for -0."""
    out = np.floor(a)
    out[out == 0] = F(0.0)   # -0 -> +0, as a - (a - float(int(a))) produces
    return out


def bx_cos(a):
    a = np.asarray(a, F)
    scaled = (a * F(2.0)) * K_INV_PI
    real = bx_floor(scaled)
    xx = a - real * K_PI_HALF
    quadrant = real.astype(np.int32) & 3
    even = (quadrant & 1) == 0
    c2, c4, c6, c8, c10 = (np.where(even, e, o).astype(F) for e, o in zip(_EVEN, _ODD))
    c0 = np.where(even, F(1.0), xx).astype(F)
    xsq = xx * xx
    acc = c10 * xsq + c8
    acc = acc * xsq + c6
    acc = acc * xsq + c4
    acc = acc * xsq + c2
    acc = acc * xsq + F(1.0)
    result = acc * c0
    neg = (quadrant == 1) | (quadrant == 2)
    return np.where(neg, -result, result).astype(F)


def bx_sin(a):
    return bx_cos(np.asarray(a, F) - K_PI_HALF)


def mtx_srt(scale, euler, pos):
    """(N,3) x3 -> (N,16), bx::mtxSRT."""
    scale, euler, pos = (np.asarray(x, F) for x in (scale, euler, pos))
    sx, cx = bx_sin(euler[:, 0]), bx_cos(euler[:, 0])
    sy, cy = bx_sin(euler[:, 1]), bx_cos(euler[:, 1])
    sz, cz = bx_sin(euler[:, 2]), bx_cos(euler[:, 2])
    s_x, s_y, s_z = scale[:, 0], scale[:, 1], scale[:, 2]
    sxsz = sx * sz
    cycz = cy * cz
    m = np.zeros((len(pos), 16), F)
    m[:, 0] = s_x * (cycz - sxsz * sy)
    m[:, 1] = (s_x * -cx) * sz
    m[:, 2] = s_x * (sxsz * cy + cz * sy)
    m[:, 4] = s_y * ((cz * sx) * sy + sz * cy)
    m[:, 5] = (s_y * cx) * cz
    m[:, 6] = s_y * (sz * sy - cycz * sx)
    m[:, 8] = (s_z * -cx) * sy
    m[:, 9] = s_z * sx
    m[:, 10] = (s_z * cx) * cy
    m[:, 12:15] = pos
    m[:, 15] = F(1.0)
    return m


def mtx_mul(a, b):
    """(N,16) x (N,16): a*b with bx::vec4MulMtx's order ((a0*b0j + a1*b1j) + a2*b2j) + a3*b3j."""
    out = np.empty_like(a)
    for i in range(4):
        for j in range(4):
            out[:, 4 * i + j] = ((a[:, 4 * i] * b[:, j] + a[:, 4 * i + 1] * b[:, 4 + j]) + a[:, 4 * i + 2] * b[:, 8 + j]) \
                                + a[:, 4 * i + 3] * b[:, 12 + j]
    return out


def integrate(pos, vel, dynamic, ticks, dt, gravity=(0.0, -9.81, 0.0), inv_mass=None):
    """`ticks` free-body sub-steps for the rows where `dynamic` is true (in place on copies)."""
    pos, vel = np.array(pos, F), np.array(vel, F)
    dt = F(dt)
    inv_mass = np.ones(len(pos), F) if inv_mass is None else np.asarray(inv_mass, F)
    p, v = pos[dynamic], vel[dynamic]
    im = inv_mass[dynamic]
    imp = [((F(g) / im).astype(F) * im) * dt for g in gravity]   # m_gravity = g / invMass; impulse = (F * invMass) * dt
    for _ in range(ticks):
        for a in range(3):
            v[:, a] = v[:, a] + imp[a]
            p[:, a] = p[:, a] + v[:, a] * dt
    pos[dynamic], vel[dynamic] = p, v
    return pos, vel


def resolve_world(parent, pos, euler, scale):
    """TransformSystem::Update on a forest given as a parent array (0xFFFFFFFF = root), everything dirty."""
    parent = np.asarray(parent, np.uint32)
    n = len(parent)
    local = mtx_srt(scale, euler, pos)
    world = local.copy()
    is_root = parent == 0xFFFFFFFF
    depth = np.zeros(n, np.int32)
    pending = ~is_root
    # depth by repeated relaxation (forests here are shallow; parents precede children in these workloads)
    d = 0
    frontier = is_root.copy()
    known = is_root.copy()
    while pending.any():
        d += 1
        idx = np.flatnonzero(pending)
        ready = known[parent[idx]]
        now = idx[ready]
        if len(now) == 0:
            raise ValueError("cycle or dangling parent")
        depth[now] = d
        world[now] = mtx_mul(world[parent[now]], local[now])
        known[now] = True
        pending[now] = False
    return world
