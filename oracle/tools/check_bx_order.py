#!/usr/bin/env python3
"""Pin the oracle's bx arithmetic (oracle/bx_math.h) against the reference's COMPILED code, mechanically.

bx itself is not under /root/reference, but the reference's committed MSVC build is:
  build/SandboxCity.dir/RelWithDebInfo/Transform.obj   contains bx::vec4MulMtx (inline COMDAT) and the call shapes of
                                                       Transform::RecalculateLocalMatrix / UpdateWorldMatrix
  build/bin/RelWithDebInfo/SandboxCity.exe             contains bx::mtxSRT and bx::mtxInverse (statically linked)
Nothing is executed or loaded: the files are read as bytes, the functions are located by the byte patterns of their call
sites (Transform::RecalculateLocalMatrix -> mtxSRT, Renderer::BeginFrame -> mtxInverse), disassembled with binutils'
objdump (pe-x86-64 / pei-x86-64), and their straight-line scalar SSE code (movss / mulss / addss / subss / divss / xorps
with the sign mask; MSVC /fp:precise: no FMA, no reassociation) is executed SYMBOLICALLY: every xmm register and stack
slot holds an expression tree over the function's inputs.  The stored results are compared, as trees, with the trees of
this repository's restatement, written below with operator overloading in exactly the structure of oracle/bx_math.h.
Multiplication and addition are commutative bit for bit in IEEE-754 and negation commutes with multiplication and
division; trees are normalised for these identities and for nothing else (in particular not for associativity, and not
for a - b versus -(b - a)).

bx::cos has one data-dependent branch (even / odd quadrant) and a conditional sign flip: both sides are executed
symbolically (constants are read from the image's .rdata), the branch conditions are checked as instruction patterns.
bx::floor (called from cos) has three paths (a >= 0; a < 0 with -a integral / fractional), each executed symbolically with
cvttss2si + cvtdq2ps modelled as trunc.  mtxSRT's six calls go to one function
(cos), so the check also confirms sin(a) = cos(a - pi/2) with pi/2 = 0x3fc90fdb.

Run in the build container only (needs /root/reference):  python oracle/tools/check_bx_order.py
Results are recorded in DESIGN.md section 3.
"""
import re
import struct
import subprocess
import sys

REF = "/root/reference/build"
EXE = REF + "/bin/RelWithDebInfo/SandboxCity.exe"
TRANSFORM_OBJ = REF + "/SandboxCity.dir/RelWithDebInfo/Transform.obj"


# ---------------------------------------------------------------- expression trees
def norm(e):
    op = e[0]
    if op in ("in", "const", "opaque"):
        return e
    if op == "neg":
        a = norm(e[1])
        return a[1] if a[0] == "neg" else ("neg", a)
    if op in ("cos", "floor", "trunc", "abs", "cosf", "sinf", "asinf", "sqrtf"):
        return (op, norm(e[1]))
    if op in ("atan2f", "max", "min"):
        return (op, norm(e[1]), norm(e[2]))
    a, b = norm(e[1]), norm(e[2])
    if op in ("mul", "div"):
        sign = 0
        if a[0] == "neg":
            a, sign = a[1], sign ^ 1
        if b[0] == "neg":
            b, sign = b[1], sign ^ 1
        if op == "mul" and repr(b) < repr(a):
            a, b = b, a
        r = (op, a, b)
        return ("neg", r) if sign else r
    if op == "add" and repr(b) < repr(a):
        a, b = b, a
    return (op, a, b)


class E:
    def __init__(self, t):
        self.t = t

    @staticmethod
    def w(o):
        return o if isinstance(o, E) else E(("const", struct.unpack("<f", struct.pack("<f", float(o)))[0]))

    def __mul__(self, o):
        return E(("mul", self.t, E.w(o).t))

    def __rmul__(self, o):
        return E(("mul", E.w(o).t, self.t))

    def __add__(self, o):
        return E(("add", self.t, E.w(o).t))

    def __sub__(self, o):
        return E(("sub", self.t, E.w(o).t))

    def __neg__(self):
        return E(("neg", self.t))

    def __pos__(self):
        return self

    def __rtruediv__(self, o):
        return E(("div", E.w(o).t, self.t))


def cosE(x):
    return E(("cos", x.t))


# ---------------------------------------------------------------- the restatements (structure of oracle/bx_math.h)
K_PI_HALF = struct.unpack("<f", struct.pack("<I", 0x3FC90FDB))[0]


def bits(u):
    return struct.unpack("<f", struct.pack("<I", u))[0]


def restate_cos(a, odd):
    """oracle/bx_math.h::cos_ on one side of its quadrant branch (the sign flip for quadrants 1, 2 is checked apart).
    (a * 2.0f is written a + a: the two are the same float for every a.)"""
    scaled = (a + a) * bits(0x3EA2F983)                # kInvPi
    real = E(("floor", scaled.t))
    xx = a - real * K_PI_HALF
    if odd:
        c0, c2, c4, c6, c8, c10 = xx, bits(0xBE2AAAAB), bits(0x3C088898), bits(0xB9501096), bits(0x363938A8), bits(0xB2D70013)
    else:
        c0, c2, c4, c6, c8, c10 = E.w(1.0), -0.5, bits(0x3D2AAAA4), bits(0xBAB60981), bits(0x37CFAB9C), bits(0xB48B634D)
    xsq = xx * xx
    acc = xsq * c10 + c8      # (operand order inside a product or sum is immaterial: normalised away)
    acc = acc * xsq + c6
    acc = acc * xsq + c4
    acc = acc * xsq + c2
    acc = acc * xsq + 1.0
    return (acc * c0).t


def restate_vec4_mul_mtx(v, m):
    return [(((v[0] * m[j] + v[1] * m[4 + j]) + v[2] * m[8 + j]) + v[3] * m[12 + j]).t for j in range(4)]


def restate_mtx_srt(sx_, sy_, sz_, ax, ay, az, tx, ty, tz):
    def sin_(a):
        return cosE(a - K_PI_HALF)
    sx, cx, sy, cy, sz, cz = sin_(ax), cosE(ax), sin_(ay), cosE(ay), sin_(az), cosE(az)
    sxsz = sx * sz
    cycz = cy * cz
    zero, one = E(("const", 0.0)), E(("const", 1.0))
    out = [None] * 16
    out[0] = sx_ * (cycz - sxsz * sy)
    out[1] = sx_ * -cx * sz
    out[2] = sx_ * (sxsz * cy + cz * sy)
    out[3] = zero
    out[4] = sy_ * (cz * sx * sy + sz * cy)
    out[5] = sy_ * cx * cz
    out[6] = sy_ * (sz * sy - cycz * sx)
    out[7] = zero
    out[8] = sz_ * -cx * sy
    out[9] = sz_ * sx
    out[10] = sz_ * cx * cy
    out[11] = zero
    out[12], out[13], out[14], out[15] = tx, ty, tz, one
    return [x.t for x in out]


def restate_mtx_inverse(a):
    xx, xy, xz, xw = a[0], a[1], a[2], a[3]
    yx, yy, yz, yw = a[4], a[5], a[6], a[7]
    zx, zy, zz, zw = a[8], a[9], a[10], a[11]
    wx, wy, wz, ww = a[12], a[13], a[14], a[15]
    det = E(("const", 0.0))
    det = det + xx * (yy * (zz * ww - zw * wz) - yz * (zy * ww - zw * wy) + yw * (zy * wz - zz * wy))
    det = det - xy * (yx * (zz * ww - zw * wz) - yz * (zx * ww - zw * wx) + yw * (zx * wz - zz * wx))
    det = det + xz * (yx * (zy * ww - zw * wy) - yy * (zx * ww - zw * wx) + yw * (zx * wy - zy * wx))
    det = det - xw * (yx * (zy * wz - zz * wy) - yy * (zx * wz - zz * wx) + yz * (zx * wy - zy * wx))
    inv = 1.0 / det
    r = [None] * 16
    r[0] = +(yy * (zz * ww - wz * zw) - yz * (zy * ww - wy * zw) + yw * (zy * wz - wy * zz)) * inv
    r[1] = -(xy * (zz * ww - wz * zw) - xz * (zy * ww - wy * zw) + xw * (zy * wz - wy * zz)) * inv
    r[2] = +(xy * (yz * ww - wz * yw) - xz * (yy * ww - wy * yw) + xw * (yy * wz - wy * yz)) * inv
    r[3] = -(xy * (yz * zw - zz * yw) - xz * (yy * zw - zy * yw) + xw * (yy * zz - zy * yz)) * inv
    r[4] = -(yx * (zz * ww - wz * zw) - yz * (zx * ww - wx * zw) + yw * (zx * wz - wx * zz)) * inv
    r[5] = +(xx * (zz * ww - wz * zw) - xz * (zx * ww - wx * zw) + xw * (zx * wz - wx * zz)) * inv
    r[6] = -(xx * (yz * ww - wz * yw) - xz * (yx * ww - wx * yw) + xw * (yx * wz - wx * yz)) * inv
    r[7] = +(xx * (yz * zw - zz * yw) - xz * (yx * zw - zx * yw) + xw * (yx * zz - zx * yz)) * inv
    r[8] = +(yx * (zy * ww - wy * zw) - yy * (zx * ww - wx * zw) + yw * (zx * wy - wx * zy)) * inv
    r[9] = -(xx * (zy * ww - wy * zw) - xy * (zx * ww - wx * zw) + xw * (zx * wy - wx * zy)) * inv
    r[10] = +(xx * (yy * ww - wy * yw) - xy * (yx * ww - wx * yw) + xw * (yx * wy - wx * yy)) * inv
    r[11] = -(xx * (yy * zw - zy * yw) - xy * (yx * zw - zx * yw) + xw * (yx * zy - zx * yy)) * inv
    r[12] = -(yx * (zy * wz - wy * zz) - yy * (zx * wz - wx * zz) + yz * (zx * wy - wx * zy)) * inv
    r[13] = +(xx * (zy * wz - wy * zz) - xy * (zx * wz - wx * zz) + xz * (zx * wy - wx * zy)) * inv
    r[14] = -(xx * (yy * wz - wy * yz) - xy * (yx * wz - wx * yz) + xz * (yx * wy - wx * yy)) * inv
    r[15] = +(xx * (yy * zz - zy * yz) - xy * (yx * zz - zx * yz) + xz * (yx * zy - zx * yy)) * inv
    return [x.t for x in r]


# ---------------------------------------------------------------- PE helpers
class Pe:
    def __init__(self, path):
        self.b = open(path, "rb").read()
        pe = struct.unpack_from("<I", self.b, 0x3C)[0]
        nsec = struct.unpack_from("<H", self.b, pe + 6)[0]
        optsz = struct.unpack_from("<H", self.b, pe + 20)[0]
        self.base = struct.unpack_from("<Q", self.b, pe + 24 + 24)[0]
        self.secs, off = [], pe + 24 + optsz
        for _ in range(nsec):
            vsz, va, rsz, raw = struct.unpack_from("<IIII", self.b, off + 8)
            self.secs.append((va, vsz, raw, rsz))
            off += 40

    def f2r(self, fo):
        return next(va + fo - raw for va, vs, raw, rs in self.secs if raw <= fo < raw + rs)

    def r2f(self, rva):
        return next(raw + rva - va for va, vs, raw, rs in self.secs if va <= rva < va + max(vs, rs))

    def call_target(self, call_file_off):
        """RVA of the function a `call rel32` at this file offset reaches, through an incremental-link thunk if any."""
        tgt = self.f2r(call_file_off + 5) + struct.unpack_from("<i", self.b, call_file_off + 1)[0]
        fo = self.r2f(tgt)
        if self.b[fo] == 0xE9:
            tgt = tgt + 5 + struct.unpack_from("<i", self.b, fo + 1)[0]
        return tgt

    def unique(self, pattern):
        hits = [m.start() for m in re.finditer(re.escape(pattern), self.b)]
        assert len(hits) == 1, (pattern.hex(), hits)
        return hits[0]

    def bytes_at_va(self, va, n):
        fo = self.r2f(va - self.base)
        return self.b[fo: fo + n]


def objdump(path, extra):
    return subprocess.run(["objdump", "-d", "--no-show-raw-insn"] + extra + [path], capture_output=True, text=True, check=True).stdout


def parse(text, start_label=None, multi_ret=False):
    ins, on = [], start_label is None
    for line in text.splitlines():
        if start_label and line.strip().startswith("0000") and "<" in line:
            on = start_label in line
            continue
        if not on:
            continue
        m = re.match(r"\s*([0-9a-f]+):\s+(\S+)\s*(.*)$", line)
        if not m:
            continue
        if m.group(2) == "int3":
            break
        ins.append((int(m.group(1), 16), m.group(2), m.group(3).split("#")[0].split("<")[0].strip()))
        if m.group(2) == "ret" and not multi_ret:
            break
    return ins


# ---------------------------------------------------------------- symbolic execution of straight-line scalar SSE
def execute(ins, in_bases, out_base, reg0=None, stack_args=None, const_reader=None, cos_target=None, calls=None, take=(),
            named_consts=None, named_calls=None, probes=None, hooks=None):
    """probes: {pc: [registers]} -> out[('probe', pc, reg)] = tree at that point.
    hooks: {call target: fn(reg, mem, sp, gpr_alias)} for calls with side effects on memory."""
    """`calls`: {target: name} for opaque unary functions; `take`: addresses of conditional jumps that are taken."""
    """in_bases: {'%rdx': 'v', ...} memory operands through these registers are inputs ('in', name, index).
    out_base: register through which results are stored.  stack_args: {entry_rsp_offset: tree}."""
    reg = dict(reg0 or {})
    mem = dict(stack_args or {})
    out, gpr = {}, {}
    # several result pointers: {'%rdx': 'yaw', ...}; results are then keyed (name, index)
    alias = dict(out_base) if isinstance(out_base, dict) else {out_base: out_base}
    sp = 0

    def addr(op):
        m = re.match(r"(-?0x[0-9a-f]+|)\((%\w+)\)", op)
        return (m.group(2), int(m.group(1), 16) if m.group(1) else 0) if m else None

    def const_of(raw):
        if raw[:4] == b"\x00\x00\x00\x80":
            return ("signmask",)
        if raw[:4] == b"\xff\xff\xff\x7f":
            return ("absmask",)
        return ("const", struct.unpack("<f", raw[:4])[0])

    def load(op, pc_next):
        r, d = addr(op)
        if r == "%rip" and named_consts is not None:      # object file: the relocation names the constant
            return named_consts[pc_next]
        if r in in_bases:
            assert d % 4 == 0
            return ("in", in_bases[r], d // 4)
        if r == "%rsp":
            return mem.get(sp + d, ("opaque", f"stack{sp + d:#x}"))
        if r == "%rip":
            return const_of(const_reader(pc_next + d, 16))
        return ("opaque", f"{r}{d:#x}")

    skip_to = None
    for k, (pc, mn, ops) in enumerate(ins):
        if skip_to is not None:
            if pc != skip_to:
                continue
            skip_to = None
        pc_next = ins[k + 1][0] if k + 1 < len(ins) else pc + 1
        parts = [p.strip() for p in re.split(r",(?![^(]*\))", ops)] if ops else []
        if probes and pc in probes:
            for r in probes[pc]:
                out[("probe", pc, r)] = reg.get(r)
        if mn in ("jne", "ja", "jbe", "je", "jp", "jb", "jae"):
            if pc in take:
                skip_to = int(parts[0], 16)
            continue
        if mn == "cvttss2si":
            gpr[parts[1]] = ("trunc_int", reg[parts[0]])
            continue
        if mn == "movd":                                      # movd %eax,%xmm: integer bits into the register
            reg[parts[1]] = ("int", gpr.get(parts[0]))
            continue
        if mn == "cvtdq2ps":
            v = reg[parts[0]]
            assert v[0] == "int", v
            reg[parts[1]] = ("trunc", v[1][1]) if isinstance(v[1], tuple) else ("const", float(v[1]))
            continue
        if mn in ("and", "test", "dec", "cmp", "cmpl", "comiss", "ucomiss"):
            continue
        if mn == "jmp":
            skip_to = int(parts[0], 16)
            continue
        if mn == "ret":
            break
        if mn == "sub" and len(parts) == 2 and parts[1] == "%rsp":
            sp -= int(parts[0][1:], 16)
        elif mn == "push" or (mn == "rex" and ops.startswith("push")):
            sp -= 8
        elif (mn == "mov" and len(parts) == 2 and (parts[0] in alias or parts[0] in in_bases) and parts[1].startswith("%r")
              and not addr(parts[1])):
            if parts[0] in alias:
                alias[parts[1]] = alias[parts[0]]             # mov %rcx,%rbx: another name for a result pointer
            if parts[0] in in_bases:
                in_bases = dict(in_bases)
                in_bases[parts[1]] = in_bases[parts[0]]       # ... or for an input pointer (an object can be both)
        elif mn == "xor" and parts[0] == parts[1]:
            gpr[parts[0]] = 0
        elif mn == "mov" and len(parts) == 2 and parts[0].startswith("$") and parts[1].startswith("%e"):
            gpr[parts[1]] = int(parts[0][1:], 16)
        elif mn in ("mov", "movl") and len(parts) == 2 and addr(parts[1]) and addr(parts[1])[0] in alias:
            d = addr(parts[1])[1]                             # integer store of a float constant into the result
            bits = gpr.get(parts[0]) if parts[0].startswith("%") else int(parts[0][1:], 16)
            if isinstance(bits, int):                         # (a pointer store, e.g. the vtable, is not a float result)
                key = d // 4 if not isinstance(out_base, dict) else (alias[addr(parts[1])[0]], d // 4)
                out[key] = ("const", struct.unpack("<f", struct.pack("<I", bits & 0xFFFFFFFF))[0])
        elif mn in ("mov", "lea", "pop", "ret", "add", "xor", "movslq", "mul", "shr", "inc", "seta", "cmova", "nopw", "nop", "nopl", "setbe", "movq", "cmpb", "testb", "movzbl", "imul", "sub"):
            pass                                              # integer bookkeeping: not modelled
        elif mn in ("shufps", "movups", "movl"):
            pass                                              # vector assembly / packed copies: lanes are not modelled
        elif mn in ("movss", "movaps"):
            src, dst = parts
            if dst.startswith("%xmm"):
                reg[dst] = reg.get(src, ("opaque", src)) if src.startswith("%xmm") else load(src, pc_next)
            else:
                r, d = addr(dst)
                if r in alias:
                    key = d // 4 if not isinstance(out_base, dict) else (alias[r], d // 4)
                    out[key] = reg.get(src, ("opaque", src))
                elif r == "%rsp":
                    mem[sp + d] = reg.get(src, ("opaque", src))
                # saves of callee-saved registers through %rax / %r11: not modelled (restored before ret)
        elif mn in ("mulss", "addss", "subss", "divss"):
            src, dst = parts
            b = reg.get(src, ("opaque", src)) if src.startswith("%xmm") else load(src, pc_next)
            reg[dst] = ({"mulss": "mul", "addss": "add", "subss": "sub", "divss": "div"}[mn], reg[dst], b)
        elif mn == "andps":
            src, dst = parts
            m = reg.get(src) if src.startswith("%xmm") else load(src, pc_next)
            assert m == ("absmask",), (hex(pc), m)
            reg[dst] = ("abs", reg[dst])
        elif mn in ("maxss", "minss"):
            src, dst = parts
            b = reg.get(src, ("opaque", src)) if src.startswith("%xmm") else load(src, pc_next)
            reg[dst] = (mn[:3], reg[dst], b)                  # x86 semantics kept: (dst, src), not commutative
        elif mn == "xorps":
            src, dst = parts
            if src == dst:
                reg[dst] = ("const", 0.0)
            else:
                m = reg.get(src) if src.startswith("%xmm") else load(src, pc_next)
                assert m == ("signmask",), (hex(pc), m)
                reg[dst] = ("neg", reg[dst])
        elif mn == "sqrtss":
            src, dst = parts
            reg[dst] = ("sqrtf", reg.get(src, ("opaque", src)))
        elif mn == "call" and named_calls is not None:
            name = named_calls[pc_next]
            if name.startswith("__security"):
                continue
            nargs = {"atan2f": 2}.get(name, 1)
            reg["%xmm0"] = (name, reg["%xmm0"]) if nargs == 1 else (name, reg["%xmm0"], reg["%xmm1"])
            for v in ("%xmm1", "%xmm2", "%xmm3", "%xmm4", "%xmm5"):
                reg[v] = ("opaque", f"clobbered {v} @ {pc:#x}")
        elif mn == "call" and hooks and int(parts[0], 16) in hooks:
            ret = hooks[int(parts[0], 16)](reg, mem, sp, out)
            for v in ("%xmm0", "%xmm1", "%xmm2", "%xmm3", "%xmm4", "%xmm5"):
                reg[v] = ("opaque", f"clobbered {v} @ {pc:#x}")
            if ret is not None:                               # a hook may model the callee's float result
                reg["%xmm0"] = ret
        elif mn == "call":
            tgt = int(parts[0], 16)
            if calls and tgt in calls:
                reg["%xmm0"] = (calls[tgt], reg["%xmm0"])
            else:
                assert cos_target is not None and tgt == cos_target, (hex(pc), ops)
                reg["%xmm0"] = ("cos", reg["%xmm0"])
            for v in ("%xmm1", "%xmm2", "%xmm3", "%xmm4", "%xmm5"):   # volatile across a call (Windows x64)
                reg[v] = ("opaque", f"clobbered {v} @ {pc:#x}")
        else:
            raise AssertionError(f"unhandled {pc:#x}: {mn} {ops}")
    out["xmm0"] = reg.get("%xmm0")
    return out


def compare(name, got, want):
    bad = [i for i in range(len(want)) if i not in got or norm(got[i]) != norm(want[i])]
    for i in bad[:3]:
        print(f"  {name}[{i}] differs\n    compiled   : {norm(got.get(i, ('missing',)))}\n    restatement: {norm(want[i])}")
    print(f"{name}: {len(want) - len(bad)} of {len(want)} outputs identical as expression trees" + ("" if not bad else "  <-- MISMATCH"))
    return not bad


def main():
    ok = True
    # 1. bx::vec4MulMtx (Transform.obj): out = rcx, v = rdx, m = r8
    text = objdump(TRANSFORM_OBJ, [])
    ins = parse(text, "?vec4MulMtx@bx@@YAXPEAMPEBM1@Z")
    got = execute(ins, {"%rdx": "v", "%r8": "m"}, "%rcx")
    v = [E(("in", "v", k)) for k in range(4)]
    m = [E(("in", "m", k)) for k in range(16)]
    ok &= compare("vec4MulMtx", got, restate_vec4_mul_mtx(v, m))
    # call shapes in Transform.obj: UpdateWorldMatrix = 4 x vec4MulMtx(world + 16 i, parent + 16 i, local)
    rel = objdump(TRANSFORM_OBJ, ["-r"])
    uw = rel[rel.index("<?UpdateWorldMatrix@Transform@@QEAAXPEBM@Z>:"):]
    uw = re.sub(r"[ \t]+", " ", uw[: uw.index("Disassembly of section", 10)])
    shape = (uw.count("IMAGE_REL_AMD64_REL32 ?vec4MulMtx@bx@@YAXPEAMPEBM1@Z") == 4 and "lea 0x24(%rcx),%rbx" in uw and
             "lea 0x64(%rcx),%rdi" in uw and "mov %rbx,%r8" in uw and "mov %rdi,%rcx" in uw and "mov %rdx,%rsi" in uw and
             "lea 0x10(%rsi),%rdx" in uw and "lea 0x10(%rdi),%rcx" in uw)
    print("UpdateWorldMatrix: world(+0x64) row i = vec4MulMtx(parent row i, local(+0x24)) x 4, i.e. world = parent * local:", shape)
    ok &= shape

    pe = Pe(EXE)
    # 2. bx::mtxSRT: located through Transform::RecalculateLocalMatrix (its first 16 bytes, then the call at +0x5b)
    f = pe.unique(bytes.fromhex("4883ec58488bc14883c124f30f104008"))
    assert pe.b[f + 0x5B] == 0xE8
    srt = pe.call_target(f + 0x5B)
    ins = parse(objdump(EXE, [f"--start-address={pe.base + srt:#x}", f"--stop-address={pe.base + srt + 0x400:#x}"]))
    calls = {int(i[2], 16) for i in ins if i[1] == "call"}
    assert len(calls) == 1, calls
    names = ["sx", "sy", "sz", "ax", "ay", "az", "tx", "ty", "tz"]
    a = {n: E(("in", n, 0)) for n in names}
    reg0 = {"%xmm1": a["sx"].t, "%xmm2": a["sy"].t, "%xmm3": a["sz"].t}
    stack = {0x28 + 8 * k: a[n].t for k, n in enumerate(names[3:])}   # Windows x64: 5th argument onwards at [rsp + 0x28 ...]
    got = execute(ins, {}, "%rcx", reg0, stack, pe.bytes_at_va, cos_target=next(iter(calls)))
    ok &= compare("mtxSRT", got, restate_mtx_srt(*[a[n] for n in names]))
    print(f"  (bx::mtxSRT at VA {pe.base + srt:#x}: {len(ins)} instructions, 6 calls into one function = bx::cos; "
          "sin(a) is cos(a - 0x3fc90fdb))")

    # 2b. bx::cos itself (the function mtxSRT calls): two sides of the quadrant branch, executed symbolically; the call
    # inside it is bx::floor.  Sign rule: `dec eax; cmp eax, 1; ja skip; xorps sign` = negate iff quadrant in {1, 2}.
    cos_rva = next(iter(calls)) - pe.base
    fo = pe.r2f(cos_rva)
    if pe.b[fo] == 0xE9:
        cos_rva = cos_rva + 5 + struct.unpack_from("<i", pe.b, fo + 1)[0]
    cins = parse(objdump(EXE, [f"--start-address={pe.base + cos_rva:#x}", f"--stop-address={pe.base + cos_rva + 0x200:#x}"]))
    floor_t = {int(i[2], 16): "floor" for i in cins if i[1] == "call"}
    jne = [i[0] for i in cins if i[1] == "jne"]
    ja = [i[0] for i in cins if i[1] == "ja"]
    mn = [i[1] for i in cins]
    sign_rule = (len(jne) == 1 and len(ja) == 1 and len(floor_t) == 1 and "cvttss2si" in mn and "dec" in mn and
                 any(i[1] == "and" and i[2].startswith("$0x3,") for i in cins) and
                 any(i[1] == "test" and i[2].startswith("$0xfffffffd,") for i in cins) and
                 any(i[1] == "cmp" and i[2].startswith("$0x1,") for i in cins) and
                 mn.index("dec") < mn.index("cmp") < mn.index("ja") < len(mn) - 1 - mn[::-1].index("xorps"))
    a_in = E(("in", "a", 0))
    okc = True
    for odd in (False, True):
        # jne taken = quadrant is odd (the even-quadrant constants are skipped); ja taken = no sign flip
        got = execute(cins, {}, "%none", {"%xmm0": a_in.t}, {}, pe.bytes_at_va, calls=floor_t, take=tuple(jne if odd else ()) + tuple(ja))
        same = norm(got["xmm0"]) == norm(restate_cos(a_in, odd))
        print(f"cos, {'odd ' if odd else 'even'} quadrants: Horner chain, constants and reduction identical as expression trees: {same}")
        okc &= same
    print("cos: quadrant = int(floor(scaled)) & 3, even iff (q & ~2) == 0, negated iff q - 1 <= 1 (unsigned), i.e. q in {1, 2}:", sign_rule)
    ok &= okc and sign_rule

    # 2c. bx::floor (the function cos calls): three paths
    fl_rva = next(iter(floor_t)) - pe.base
    fo = pe.r2f(fl_rva)
    if pe.b[fo] == 0xE9:
        fl_rva = fl_rva + 5 + struct.unpack_from("<i", pe.b, fo + 1)[0]
    fins = parse(objdump(EXE, [f"--start-address={pe.base + fl_rva:#x}", f"--stop-address={pe.base + fl_rva + 0x100:#x}"]), multi_ret=True)
    jbe = [i[0] for i in fins if i[1] == "jbe"]
    jpne = [i[0] for i in fins if i[1] in ("jp", "jne")]
    assert len(jbe) == 1 and len(jpne) == 2, (jbe, jpne)
    t_neg = ("trunc", ("neg", a_in.t))
    want_floor = {
        "a >= 0 (or NaN)": (tuple(jbe), ("sub", a_in.t, ("sub", a_in.t, ("trunc", a_in.t)))),              # a - fract(a)
        "a < 0, -a integral": ((), ("sub", ("neg", t_neg), ("const", 0.0))),                               # -trunc(-a) - 0
        "a < 0, -a fractional": (tuple(jpne), ("sub", ("neg", t_neg), ("const", 1.0))),                    # -trunc(-a) - 1
    }
    okf = True
    for label, (take, want) in want_floor.items():
        got = execute(fins, {}, "%none", {"%xmm0": a_in.t}, {}, pe.bytes_at_va, take=take)
        same = norm(got["xmm0"]) == norm(want)
        print(f"floor, {label}: {same}")
        okf &= same
    print("  (oracle/bx_math.h::floor_ writes the negative side as -((-a - fract(-a)) [+ 1]); -a - fract(-a) == trunc(-a) exactly "
          "for |a| < 2^31, and -(t + 1) == -t - 1, -(t) == -t - 0 bit for bit, so the two forms agree on every input)")
    ok &= okf

    # 3. bx::mtxInverse: located through Renderer::BeginFrame (lea 0x64(%rax),%rdx ; lea 0(%rbp),%rcx ; call)
    f = pe.unique(bytes.fromhex("488d5064488d4d00e8"))
    inv = pe.call_target(f + 8)
    ins = parse(objdump(EXE, [f"--start-address={pe.base + inv:#x}", f"--stop-address={pe.base + inv + 0x1000:#x}"]))
    assert not any(i[1].startswith("j") or i[1] == "call" for i in ins)
    got = execute(ins, {"%rdx": "a"}, "%rcx", const_reader=pe.bytes_at_va)
    ok &= compare("mtxInverse", got, restate_mtx_inverse([E(("in", "a", k)) for k in range(16)]))
    print(f"  (bx::mtxInverse at VA {pe.base + inv:#x}: {len(ins)} instructions, "
          f"{sum(1 for i in ins if i[1] in ('mulss', 'addss', 'subss', 'divss'))} scalar float operations, no branches)")
    print("RESULT:", "the restatement has the compiled code's operation order" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
