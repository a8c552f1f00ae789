#!/usr/bin/env python3
"""A small symbolic executor for MSVC x64 scalar-SSE code, for the Bullet functions of the reference's exe that have branches,
pointer arguments and 16-byte moves (check_solver_setup.py).  TEST INFRASTRUCTURE (oracle/): it interprets a disassembly
listing; nothing of the reference is loaded or run.

State: every xmm register is four lanes of expression trees; general registers hold symbolic pointers ("ptr", space, offset),
integers ("int", v), raw float bits ("f32", tree) or None; memory is a map (space, byte offset) -> tree per 4-byte cell, with
store forwarding, plus a map of 8-byte pointer cells.  A cell never written reads as the input ("in", space, offset).  The stack
is the space "stk", offsets relative to %rsp at entry.  Control flow follows `decisions` {pc of a conditional jump: taken?};
a conditional jump without a decision, an instruction that touches an xmm register and is not modelled, or a call without a hook
raises — nothing is skipped silently.  Every float comparison executed is recorded in `self.compares`.
"""
import re
import struct

_ALIAS = {}
for _r in ("ax", "bx", "cx", "dx", "si", "di", "bp", "sp"):
    _ALIAS["%e" + _r] = "%r" + _r
    _ALIAS["%" + _r] = "%r" + _r
for _r, _l in (("ax", "al"), ("bx", "bl"), ("cx", "cl"), ("dx", "dl"), ("si", "sil"), ("di", "dil")):
    _ALIAS["%" + _l] = "%r" + _r
for _i in range(8, 16):
    for _s in ("d", "w", "b"):
        _ALIAS[f"%r{_i}{_s}"] = f"%r{_i}"

CONDJ = {"je", "jne", "ja", "jae", "jb", "jbe", "jp", "jnp", "jg", "jge", "jl", "jle", "js", "jns"}
ZERO = ("const", 0.0)


def f32(bits):
    return struct.unpack("<f", struct.pack("<I", bits & 0xFFFFFFFF))[0]


def split_ops(ops):
    return [p.strip() for p in re.split(r",(?![^(]*\))", ops)] if ops else []


class Unmodelled(Exception):
    pass


class Machine:
    def __init__(self, pe, ins, gpr=None, stack_ptrs=None, decisions=None, ptr_loads=None, hooks=None, pc_hooks=None):
        self.pe, self.ins = pe, ins
        self.index = {pc: i for i, (pc, _, _) in enumerate(ins)}
        self.gpr = dict(gpr or {})
        self.gpr["%rsp"] = ("ptr", "stk", 0)
        self.xmm = {}
        self.mem = {}
        self.memp = {("stk", off): v for off, v in (stack_ptrs or {}).items()}
        self.decisions = dict(decisions or {})
        self.ptr_loads = dict(ptr_loads or {})
        self.hooks = dict(hooks or {})
        self.pc_hooks = dict(pc_hooks or {})
        self.compares = []
        self.trace = []
        self.last_flags = None
        self.decider = None
        self.concrete_ints = False   # integer registers, flags and integer memory cells carry values (loop counters, indices)
        self.memi = {}
        self.iflags = None
        self.fflags = None          # '>' '<' '=' : how the last comiss came out (fcmp_decider), for every consumer of its flags
        self.fcmp_decider = None    # fn(machine, pc, a, b) -> '>' | '<' | '=' per executed comiss (concrete_ints mode)

    # ---------------------------------------------------------------- operands
    def reg64(self, r):
        return _ALIAS.get(r, r)

    def is32(self, r):
        return r.startswith("%e") or re.fullmatch(r"%r\d+d", r) is not None

    def addr(self, op, pc_next):
        m = re.fullmatch(r"(-?0x[0-9a-f]+|)\((%\w+)\)", op)
        if not m:
            # base + index: modelled when one of the two holds the integer 0 (a pool base folded into the other's symbolic pointer)
            mi = re.fullmatch(r"(-?0x[0-9a-f]+|)\((%\w+),(%\w+),(\d)\)", op)
            if not mi:
                raise Unmodelled(f"addressing {op}")
            a, b = self.gpr.get(mi.group(2)), self.gpr.get(mi.group(3))
            if isinstance(a, tuple) and a[0] == "int" and isinstance(b, tuple) and b[0] == "ptr" and mi.group(4) == "1":
                a, b = b, a                                                                         # (offset register first, pointer second)
            for x, y in ((a, b), (b, a)):
                if isinstance(x, tuple) and x[0] == "pdiff" and isinstance(y, tuple) and y[0] == "ptr" and y[1] == x[2][0] and mi.group(4) == "1":
                    d0 = int(mi.group(1), 16) if mi.group(1) else 0
                    return (x[1][0], x[1][1] + (y[2] - x[2][1]) + d0)                               # (p - q) + (q + k) = p + k
            if isinstance(b, tuple) and b[0] == "int" and isinstance(a, tuple) and a[0] == "ptr":   # a known integer index
                d0 = int(mi.group(1), 16) if mi.group(1) else 0
                return (a[1], a[2] + d0 + b[1] * int(mi.group(4)))
            if mi.group(4) != "1":
                raise Unmodelled(f"addressing {op}: {a}, {b}")
            if b == ("int", 0):
                return self.addr(f"{mi.group(1)}({mi.group(2)})", pc_next)
            if a == ("int", 0):
                return self.addr(f"{mi.group(1)}({mi.group(3)})", pc_next)
            raise Unmodelled(f"addressing {op}: {a}, {b}")
        d = int(m.group(1), 16) if m.group(1) else 0
        r = m.group(2)
        if r == "%rip":
            return ("rip", pc_next + d)
        v = self.gpr.get(r)
        if not (isinstance(v, tuple) and v[0] == "ptr"):
            raise Unmodelled(f"{op}: {r} holds {v}")
        return (v[1], v[2] + d)

    def const_at(self, va, lanes=1):
        raw = self.pe.b[self.pe.r2f(va - self.pe.base): self.pe.r2f(va - self.pe.base) + 16]
        out = []
        for k in range(lanes):
            w = raw[4 * k: 4 * k + 4]
            out.append(("signmask",) if w == b"\x00\x00\x00\x80" else ("absmask",) if w == b"\xff\xff\xff\x7f" else ("const", struct.unpack("<f", w)[0]))
        return out

    def cell(self, key):
        return self.mem.get(key, ("in", key[0], key[1]))

    def load_lanes(self, op, pc_next, lanes):
        if op.startswith("%xmm"):
            return list(self.xmm.get(op, [("opaque", op + str(k)) for k in range(4)]))[:lanes] + [ZERO] * (4 - lanes)
        a = self.addr(op, pc_next)
        if a[0] == "rip":
            return self.const_at(a[1], lanes) + [ZERO] * (4 - lanes)
        return [self.cell((a[0], a[1] + 4 * k)) for k in range(lanes)] + [ZERO] * (4 - lanes)

    def store_lanes(self, op, pc_next, vals):
        a = self.addr(op, pc_next)
        for k, v in enumerate(vals):
            self.mem[(a[0], a[1] + 4 * k)] = v
            self.memp.pop((a[0], a[1] + 4 * k), None)

    def get(self, r):
        return self.xmm.setdefault(r, [("opaque", r + str(k)) for k in range(4)])

    # ---------------------------------------------------------------- run
    def run(self, start=None, limit=20000):
        i = 0 if start is None else self.index[start]
        steps = 0
        while True:
            steps += 1
            if steps > limit:
                raise Unmodelled("step limit")
            pc, mn, ops = self.ins[i]
            pc_next = self.ins[i + 1][0] if i + 1 < len(self.ins) else pc + 8
            if pc in self.pc_hooks:
                self.pc_hooks[pc](self)
            p = split_ops(ops)
            self.trace.append(pc)
            if mn == "ret":
                return
            if mn == "jmp":
                i = self.index[int(p[0], 16)]
                continue
            if mn in CONDJ:
                known = self.eval_cond(mn) if self.concrete_ints else None
                if known is not None:
                    if known:
                        i = self.index[int(p[0], 16)]
                        continue
                    i += 1
                    continue
                if pc not in self.decisions or self.concrete_ints:
                    d = self.decider(self, pc, mn, self.last_flags) if self.decider else None
                    if d is None:
                        raise Unmodelled(f"conditional jump without a decision at {pc:#x}: {mn} {ops} after {self.last_flags}")
                    self.decisions[pc] = d
                if self.decisions[pc]:
                    i = self.index[int(p[0], 16)]
                    continue
                i += 1
                continue
            self.step(pc, mn, p, ops, pc_next)
            i += 1

    # ---------------------------------------------------------------- concrete integers (opt-in)
    @staticmethod
    def _sx(v, bits):
        v &= (1 << bits) - 1
        return v - (1 << bits) if v >> (bits - 1) else v

    def ival(self, op, pc_next, bits=64):
        """The integer an operand holds, or None."""
        if op.startswith("$"):
            return self._sx(int(op[1:], 16), bits)
        if op.startswith("%"):
            v = self.gpr.get(self.reg64(op))
            return self._sx(v[1], bits) if isinstance(v, tuple) and v[0] == "int" else None
        try:
            a = self.addr(op, pc_next)
        except Unmodelled:
            return None
        v = self.memi.get(a)
        return self._sx(v[1], bits) if v else None

    def eval_cond(self, mn):
        """Which way a conditional jump / move / set goes when the flags come from integers; None when they do not."""
        f = self.iflags
        if f is None:
            if self.fflags is None:
                return None
            cc = mn[1:] if mn.startswith("j") else mn[4:] if mn.startswith("cmov") else mn[3:]
            r = self.fflags
            return {"a": r == ">", "ae": r in (">", "="), "b": r == "<", "be": r in ("<", "="), "e": r == "=", "ne": r != "=", "p": False, "np": True}.get(cc)
        a, b, kind, bits = f
        cc = mn[1:] if mn.startswith("j") else mn[4:] if mn.startswith("cmov") else mn[3:]
        r = (a - b) if kind == "cmp" else (a & b)
        r = self._sx(r, bits)
        ua, ub = a & ((1 << bits) - 1), b & ((1 << bits) - 1)
        if kind == "test":
            table = {"e": r == 0, "ne": r != 0, "s": r < 0, "ns": r >= 0, "le": r <= 0, "g": r > 0, "l": r < 0, "ge": r >= 0}
        else:
            table = {"e": a == b, "ne": a != b, "l": a < b, "le": a <= b, "g": a > b, "ge": a >= b, "b": ua < ub, "be": ua <= ub, "a": ua > ub, "ae": ua >= ub,
                     "s": r < 0, "ns": r >= 0}
        return table.get(cc)

    def call_function(self, va):
        """A direct call into code that is executed with the same state: the callee's instructions run on this machine's registers
        and memory (its frame lies below the caller's in the one stack space), then the caller continues."""
        saved = (self.ins, self.index)
        sp = self.gpr["%rsp"]
        self.gpr["%rsp"] = ("ptr", "stk", sp[2] - 8)
        self.ins = function_listing(self.pe, va, 0x3000)
        if self.ins and self.ins[0][1] == "jmp" and self.ins[0][2].startswith("*") and getattr(self, "import_hook", None):
            # an import thunk (memcpy, atan2f ...): the driver says what the imported function does to the state
            self.ins = saved[0]
            self.gpr["%rsp"] = sp
            self.import_hook(self, self.ins and va)
            return
        self.index = {pc: i for i, (pc, _, _) in enumerate(self.ins)}
        self.depth = getattr(self, "depth", 0) + 1
        if self.depth > 6:
            raise Unmodelled("call depth")
        self.run()
        self.depth -= 1
        self.ins, self.index = saved
        self.gpr["%rsp"] = sp

    def step_int(self, pc, mn, p, ops, pc_next):
        """Integer instructions with known operands (concrete_ints).  True when the instruction was handled here."""
        g = self.gpr
        if mn in ("comiss", "ucomiss"):
            self.iflags = None
            self.fflags = None
            if self.fcmp_decider is not None:
                src, dst = p
                a, b = self.get(dst)[0], self.load_lanes(src, pc_next, 1)[0]
                self.compares.append((pc, a, b))
                self.last_flags = (pc, mn, ops)
                self.fflags = self.fcmp_decider(self, pc, a, b)
                if self.fflags is None:
                    raise Unmodelled(f"{pc:#x}: {mn} {ops} without a decision")
                return True
            return False
        if any(x.startswith("%xmm") for x in p):
            return False
        bits = 32 if (p and p[-1].startswith("%") and self.is32(p[-1])) or mn.endswith("l") and mn not in ("shl", "sal", "jl") else 64
        if mn == "rep" and ops.startswith("stos %eax"):
            cnt, val, dst = g.get("%rcx"), g.get("%rax"), g.get("%rdi")
            if not (isinstance(cnt, tuple) and cnt[0] == "int" and isinstance(val, tuple) and val[0] == "int" and isinstance(dst, tuple) and dst[0] == "ptr"):
                raise Unmodelled(f"{pc:#x}: rep stos with {cnt}, {val}, {dst}")
            for k in range(cnt[1]):
                self.memi[(dst[1], dst[2] + 4 * k)] = ("int", self._sx(val[1], 32))
                self.mem[(dst[1], dst[2] + 4 * k)] = ("const", f32(val[1]))
            g["%rdi"] = ("ptr", dst[1], dst[2] + 4 * cnt[1])
            g["%rcx"] = ("int", 0)
            return True
        if mn in ("cmp", "cmpl", "cmpq", "test", "testl", "testq", "testb", "cmpb"):
            if mn.endswith("b"):
                bits = 8
            elif p[1].startswith("%") and self.is32(p[1]):
                bits = 32
            b, a = self.ival(p[0], pc_next, bits), self.ival(p[1], pc_next, bits)
            if mn == "test" and p[0] == p[1] and isinstance(g.get(self.reg64(p[0])), tuple) and g[self.reg64(p[0])][0] == "ptr":
                a = b = 1                    # (a symbolic pointer is not null)
            if mn.startswith("cmp") and a is None and b is None and all(x.startswith("%") for x in p):
                va, vb = g.get(self.reg64(p[1])), g.get(self.reg64(p[0]))
                if isinstance(va, tuple) and isinstance(vb, tuple) and va[0] == vb[0] == "ptr" and va[1] == vb[1]:
                    a, b = va[2], vb[2]      # (two pointers into the same object)
            self.iflags = (a, b, "cmp" if mn.startswith("cmp") else "test", bits) if a is not None and b is not None else None
            self.fflags = None
            self.last_flags = (pc, mn, ops)
            return True
        if mn.startswith("cmov") and len(p) == 2:
            k = self.eval_cond(mn)
            if k is None:
                return False
            if k:
                g[self.reg64(p[1])] = g.get(self.reg64(p[0])) if p[0].startswith("%") else (self.memi.get(self.addr(p[0], pc_next)) or self.memp.get(self.addr(p[0], pc_next)))
            return True
        if mn.startswith("set") and len(p) == 1 and p[0].startswith("%"):
            k = self.eval_cond(mn)
            if k is None:
                k = self.decider(self, pc, mn, self.last_flags) if self.decider else None
                if k is None:
                    raise Unmodelled(f"{pc:#x}: {mn} without a decision")
            g[self.reg64(p[0])] = ("int", 1 if k else 0)
            return True
        if mn in ("movslq", "movzbl", "movzwl", "movsbl", "cltq", "cdqe"):
            if mn in ("cltq", "cdqe"):
                v = self.ival("%eax", pc_next, 32)
                g["%rax"] = ("int", v) if v is not None else None
                return True
            v = self.ival(p[0], pc_next, 32 if mn == "movslq" else 8 if mn.endswith("bl") else 16)
            if v is None:
                return False
            if mn.startswith("movz"):
                v &= 0xFF if mn == "movzbl" else 0xFFFF
            g[self.reg64(p[1])] = ("int", v)
            return True
        if mn in ("mov", "movl", "movq") and len(p) == 2 and not p[1].startswith("%"):
            v = self.ival(p[0], pc_next)
            a = self.addr(p[1], pc_next)
            if v is not None:
                self.memi[a] = ("int", v)
            else:
                self.memi.pop(a, None)
            return False                  # (the float view of the cell is kept by the general code)
        if mn in ("mov", "movl") and len(p) == 2 and p[1].startswith("%") and not p[0].startswith(("%", "$")):
            try:
                a = self.addr(p[0], pc_next)
            except Unmodelled:
                return False
            if a in self.memi and a not in self.memp:
                g[self.reg64(p[1])] = self.memi[a]
                return True
            return False
        arith = {"add": lambda a, b: a + b, "sub": lambda a, b: a - b, "imul": lambda a, b: a * b, "and": lambda a, b: a & b, "or": lambda a, b: a | b,
                 "xor": lambda a, b: a ^ b, "shl": lambda a, b: a << (b & 63), "sal": lambda a, b: a << (b & 63), "sar": lambda a, b: a >> (b & 63),
                 "shr": lambda a, b: (a & ((1 << bits) - 1)) >> (b & 63)}
        base = mn[:-1] if mn in ("addl", "subl", "addq", "subq", "andl", "orl") else mn
        if base in ("xor", "sub") and len(p) == 2 and p[0] == p[1] and p[0].startswith("%"):
            g[self.reg64(p[0])] = ("int", 0)
            self.iflags = (0, 0, "cmp", 64)
            return True
        if base in arith and len(p) == 2:
            dst = p[1]
            if base == "sub" and dst.startswith("%") and p[0].startswith("%"):
                va, vb = g.get(self.reg64(dst)), g.get(self.reg64(p[0]))
                if isinstance(va, tuple) and isinstance(vb, tuple) and va[0] == vb[0] == "ptr":
                    # pointer - pointer: an integer inside one object, a "difference" between two (resolved when it is added back)
                    g[self.reg64(dst)] = ("int", va[2] - vb[2]) if va[1] == vb[1] else ("pdiff", (va[1], va[2]), (vb[1], vb[2]))
                    self.iflags = None
                    return True
            if dst.startswith("%") and isinstance(g.get(self.reg64(dst)), tuple) and g[self.reg64(dst)][0] == "ptr" and base in ("add", "sub"):
                k = self.ival(p[0], pc_next)
                if k is None:
                    g[self.reg64(dst)] = None
                    return True
                v = g[self.reg64(dst)]
                g[self.reg64(dst)] = ("ptr", v[1], v[2] + (k if base == "add" else -k))
                self.iflags = None
                return True
            a, b = self.ival(dst, pc_next, bits), self.ival(p[0], pc_next, bits)
            if dst == "%rsp":
                return False
            if a is None or b is None:
                if dst.startswith("%"):
                    g[self.reg64(dst)] = None
                else:
                    self.memi.pop(self.addr(dst, pc_next), None)
                self.iflags = None
                return True
            r = self._sx(arith[base](a, b), bits)
            if dst.startswith("%"):
                g[self.reg64(dst)] = ("int", r)
            else:
                self.memi[self.addr(dst, pc_next)] = ("int", r)
            self.iflags = (r, 0, "cmp", bits)
            return True
        if mn == "imul" and len(p) == 1:
            a, b = self.ival("%eax", pc_next, 32), self.ival(p[0], pc_next, 32)
            if a is None or b is None:
                g["%rax"] = g["%rdx"] = None
                return True
            r = a * b
            g["%rax"] = ("int", self._sx(r, 32))
            g["%rdx"] = ("int", self._sx(r >> 32, 32))
            self.iflags = None
            return True
        if mn in ("inc", "dec", "incl", "decl", "incq", "decq", "neg", "not") and len(p) == 1:
            a = self.ival(p[0], pc_next, bits)
            if a is None:
                if p[0].startswith("%"):
                    g[self.reg64(p[0])] = None
                self.iflags = None
                return True
            r = {"inc": a + 1, "dec": a - 1, "neg": -a, "not": ~a}[mn.rstrip("lq") if mn not in ("neg", "not") else mn]
            r = self._sx(r, bits)
            if p[0].startswith("%"):
                g[self.reg64(p[0])] = ("int", r)
            else:
                self.memi[self.addr(p[0], pc_next)] = ("int", r)
            if mn not in ("not",):
                self.iflags = (r, 0, "cmp", bits)
            return True
        if mn == "lea" and len(p) == 2:
            m = re.fullmatch(r"(-?0x[0-9a-f]+|)\((%\w+)?(?:,(%\w+),(\d))?\)", p[0])
            if m:
                d = int(m.group(1), 16) if m.group(1) else 0
                bv = g.get(m.group(2)) if m.group(2) else ("int", 0)
                iv = g.get(m.group(3)) if m.group(3) else ("int", 0)
                sc = int(m.group(4)) if m.group(4) else 1
                if isinstance(bv, tuple) and bv[0] == "int" and isinstance(iv, tuple) and iv[0] == "int":
                    g[self.reg64(p[1])] = ("int", self._sx(bv[1] + iv[1] * sc + d, 32 if self.is32(p[1]) else 64))
                    return True
            return False
        return False

    def step(self, pc, mn, p, ops, pc_next):
        g = self.gpr
        if mn in ("nop", "nopw", "nopl", "xchg", "data16", "cltq", "cdqe") or mn.startswith("nop"):
            return
        if mn == "rex" and ops.startswith("push"):
            mn, p = "push", split_ops(ops.split(None, 1)[1])
        if self.concrete_ints and self.step_int(pc, mn, p, ops, pc_next):
            return
        if mn == "push":
            sp = g["%rsp"]
            g["%rsp"] = ("ptr", "stk", sp[2] - 8)
            self.saved_regs = getattr(self, "saved_regs", {})
            self.saved_regs[sp[2] - 8] = g.get(self.reg64(p[0])) if p and p[0].startswith("%") else None
            return
        if mn == "pop":
            sp = g["%rsp"]
            g["%rsp"] = ("ptr", "stk", sp[2] + 8)
            g[self.reg64(p[0])] = getattr(self, "saved_regs", {}).pop(sp[2], None)
            return
        if mn in ("sub", "add") and len(p) == 2 and p[1] == "%rsp" and p[0].startswith("$"):
            sp = g["%rsp"]
            k = int(p[0][1:], 16)
            g["%rsp"] = ("ptr", "stk", sp[2] - k if mn == "sub" else sp[2] + k)
            return
        if mn in ("test", "cmp", "testb", "cmpl", "cmpq", "cmpb", "testl"):
            self.last_flags = (pc, mn, ops)
            return                       # (integer flags: the decisions say which way the jump goes)
        if mn == "lea":
            try:
                a = self.addr(p[0], pc_next)
                g[self.reg64(p[1])] = ("ptr", a[0], a[1])
            except Unmodelled:
                g[self.reg64(p[1])] = None          # (integer arithmetic through lea)
            return
        if mn in ("mov", "movl", "movq", "movslq", "movzbl", "movzwl", "movabs") and not any(x.startswith("%xmm") for x in p):
            src, dst = p
            if dst.startswith("%"):
                d64 = self.reg64(dst)
                if src.startswith("$"):
                    g[d64] = ("int", int(src[1:], 16))
                elif src.startswith("%"):
                    g[d64] = g.get(self.reg64(src))
                else:
                    a = self.addr(src, pc_next)
                    if a in self.memp:
                        g[d64] = self.memp[a]
                    elif a in self.ptr_loads:
                        g[d64] = self.ptr_loads[a]
                    elif self.is32(dst) and mn == "mov":
                        g[d64] = ("f32", self.cell(a))
                    else:
                        g[d64] = None
                return
            a = self.addr(dst, pc_next)
            wide = mn == "movq" or (mn == "mov" and src.startswith("%r") and not self.is32(src) and not src.endswith(("d", "w", "b")))
            if src.startswith("$"):
                v = int(src[1:], 16)
                self.mem[a] = ("const", f32(v))
                self.memp.pop(a, None)
                if wide:
                    self.mem[(a[0], a[1] + 4)] = ("const", f32(v >> 32))
                return
            v = g.get(self.reg64(src))
            if isinstance(v, tuple) and v[0] == "ptr":
                self.memp[a] = v
                self.mem.pop(a, None)
                self.mem.pop((a[0], a[1] + 4), None)
            elif isinstance(v, tuple) and v[0] == "int":
                self.memp.pop(a, None)
                self.mem[a] = ("const", f32(v[1])) if v[1] else ZERO
                if wide:
                    self.mem[(a[0], a[1] + 4)] = ("const", f32(v[1] >> 32)) if v[1] >> 32 else ZERO
            elif isinstance(v, tuple) and v[0] == "f32":
                self.memp.pop(a, None)
                self.mem[a] = v[1]
            else:
                self.memp[a] = None            # (a callee-saved register's unknown content going to its home slot)
                self.mem[a] = ("opaque", f"gpr {src}@{pc:#x}")
                if wide:
                    self.mem[(a[0], a[1] + 4)] = ("opaque", f"gpr {src}@{pc:#x}+4")
            return
        if mn in ("xor",) and len(p) == 2 and p[0] == p[1]:
            g[self.reg64(p[0])] = ("int", 0)
            return
        if mn in ("incl", "decl", "incq", "decq", "addl", "subl", "addq", "subq", "andl", "orl") and not p[-1].startswith("%"):
            a = self.addr(p[-1], pc_next)
            self.mem[a] = ("opaque", f"int@{pc:#x}")
            return
        if mn.startswith("cmov") and len(p) == 2 and p[1].startswith("%") and self.decider is not None:
            d = self.decisions.get(pc)
            if d is None:
                d = self.decider(self, pc, mn, self.last_flags)
            if d is not None:
                self.decisions[pc] = d
                if d:
                    g[self.reg64(p[1])] = g.get(self.reg64(p[0])) if p[0].startswith("%") else self.memp.get(self.addr(p[0], pc_next))
                return
        if mn in ("shl", "shr", "sar", "add", "sub", "and", "or", "inc", "dec", "imul", "neg", "not", "xor", "setne", "sete", "seta", "setb", "setbe", "setae",
                  "cmovne", "cmove", "cmova", "cmovb", "cmovbe", "cmovae", "cmovg", "cmovl", "cmovge", "cmovle", "movsbl", "movswl") \
                and not any(x.startswith("%xmm") for x in p):
            if p and p[-1].startswith("%"):
                g[self.reg64(p[-1])] = None
            return
        if mn == "call":
            if ops in self.hooks:
                self.hooks[ops](self)
                return
            if re.fullmatch(r"0x[0-9a-f]+", ops):
                from check_bullet_order import _resolve
                target = _resolve(self.pe, int(ops, 16))
                if target in self.hooks:
                    self.hooks[target](self)
                    return
                self.call_function(target)
                return
            raise Unmodelled(f"call {ops} at {pc:#x}")
        # ---------------------------------------------------------------- SSE
        if mn == "movss":
            src, dst = p
            if dst.startswith("%xmm"):
                if src.startswith("%xmm"):
                    d = list(self.get(dst))
                    d[0] = self.get(src)[0]
                    self.xmm[dst] = d
                else:
                    self.xmm[dst] = self.load_lanes(src, pc_next, 1)
            else:
                self.store_lanes(dst, pc_next, [self.get(src)[0]])
            return
        if mn in ("movaps", "movups", "movdqa", "movdqu"):
            src, dst = p
            if dst.startswith("%xmm"):
                self.xmm[dst] = self.load_lanes(src, pc_next, 4)
            else:
                self.store_lanes(dst, pc_next, list(self.get(src)))
            return
        if mn == "cvtdq2ps":
            src, dst = p
            v = self.get(src)
            self.xmm[dst] = [("const", float(x[1])) if isinstance(x, tuple) and x[0] == "ibits" else ZERO if x == ZERO else ("opaque", f"cvtdq2ps@{pc:#x}") for x in v]
            return
        if mn == "movd" or (mn == "movq" and any(x.startswith("%xmm") for x in p)):
            src, dst = p
            if dst.startswith("%xmm") and src.startswith("%"):
                v = g.get(self.reg64(src))
                if self.concrete_ints and isinstance(v, tuple) and v[0] == "int" and v[1] != 0:
                    self.xmm[dst] = [("ibits", self._sx(v[1], 32)), ZERO, ZERO, ZERO]      # integer bits, for cvtdq2ps
                    return
                self.xmm[dst] = [v[1] if isinstance(v, tuple) and v[0] == "f32" else ZERO if v == ("int", 0) else ("opaque", f"movd@{pc:#x}"), ZERO, ZERO, ZERO]
            elif src.startswith("%xmm") and dst.startswith("%"):
                g[self.reg64(dst)] = ("f32", self.get(src)[0])
            else:
                raise Unmodelled(f"{mn} {ops}")
            return
        if mn in ("mulss", "addss", "subss", "divss", "maxss", "minss"):
            src, dst = p
            b = self.load_lanes(src, pc_next, 1)[0]
            d = list(self.get(dst))
            d[0] = ({"mulss": "mul", "addss": "add", "subss": "sub", "divss": "div", "maxss": "max", "minss": "min"}[mn], d[0], b)
            self.xmm[dst] = d
            return
        if mn in ("mulps", "addps", "subps", "divps", "maxps", "minps"):
            src, dst = p
            b = self.load_lanes(src, pc_next, 4)
            d = self.get(dst)
            self.xmm[dst] = [(mn[:3], d[k], b[k]) for k in range(4)]
            return
        if mn == "sqrtss":
            src, dst = p
            d = list(self.get(dst))
            d[0] = ("sqrtf", self.load_lanes(src, pc_next, 1)[0])
            self.xmm[dst] = d
            return
        if mn in ("xorps", "pxor", "xorpd"):
            src, dst = p
            if src == dst:
                self.xmm[dst] = [ZERO] * 4
                return
            m = self.load_lanes(src, pc_next, 4)
            d = self.get(dst)
            self.xmm[dst] = [("neg", d[k]) if m[k] == ("signmask",) else d[k] if m[k] == ZERO else ("opaque", f"xorps@{pc:#x}") for k in range(4)]
            return
        if mn in ("andps", "pand"):
            src, dst = p
            m = self.load_lanes(src, pc_next, 4)
            d = self.get(dst)
            self.xmm[dst] = [("abs", d[k]) if m[k] == ("absmask",) else ZERO if m[k] == ZERO else ("opaque", f"andps@{pc:#x}") for k in range(4)]
            return
        if mn == "shufps":
            imm, src, dst = p
            k = int(imm[1:], 16)
            d = self.get(dst)
            s = self.load_lanes(src, pc_next, 4)
            self.xmm[dst] = [d[k & 3], d[(k >> 2) & 3], s[(k >> 4) & 3], s[(k >> 6) & 3]]
            return
        if mn in ("unpcklps",):
            src, dst = p
            d = self.get(dst)
            s = self.load_lanes(src, pc_next, 4)
            self.xmm[dst] = [d[0], s[0], d[1], s[1]]
            return
        if mn in ("comiss", "ucomiss"):
            src, dst = p
            self.compares.append((pc, self.get(dst)[0], self.load_lanes(src, pc_next, 1)[0]))
            self.last_flags = (pc, mn, ops)
            return
        raise Unmodelled(f"{pc:#x}: {mn} {ops}")


def function_listing(pe, va, size=0x2000):
    """The instructions of the function at `va`, up to the first int3 after its last ret."""
    from check_bullet_order import _disasm
    ins = _disasm(pe, va, size, multi_ret=True)
    return ins


def show(t, names=None, leaf=None):
    """Compact infix text of a (normalised) tree.  names: {tree: label} for sub-trees to print by name; leaf: fn(space, off) -> text."""
    if names and t in names:
        return names[t]
    op = t[0]
    if op == "in":
        return leaf(t[1], t[2]) if leaf else f"{t[1]}[{t[2]:#x}]"
    if op == "const":
        return repr(t[1])
    if op == "opaque":
        return f"<{t[1]}>"
    if op in ("neg",):
        return "-" + show(t[1], names, leaf)
    if op in ("abs", "sqrtf", "floor", "trunc"):
        return f"{op}({show(t[1], names, leaf)})"
    if op in ("max", "min", "atan2f"):
        return f"{op}({show(t[1], names, leaf)}, {show(t[2], names, leaf)})"
    sym = {"add": " + ", "sub": " - ", "mul": "*", "div": " / "}[op]
    return "(" + show(t[1], names, leaf) + sym + show(t[2], names, leaf) + ")"


def norm2(t):
    """check_bx_order.norm, then a + (-b) -> a - b and a - (-b) -> a + b (exact identities in IEEE-754), re-normalised."""
    from check_bx_order import norm

    def fix(e):
        if not isinstance(e, tuple) or e[0] in ("in", "const", "opaque"):
            return e
        e = (e[0],) + tuple(fix(x) if isinstance(x, tuple) else x for x in e[1:])
        if e[0] in ("div", "mul"):
            for k in (1, 2):
                if e[k][0] == "const" and e[k][1] < 0.0:          # (-c) * x and (-c) / x are -(c * x), -(c / x)
                    args = list(e[1:])
                    args[k - 1] = ("const", -e[k][1])
                    return ("neg", (e[0],) + tuple(args))
        if e[0] == "add":
            for x, y in ((e[1], e[2]), (e[2], e[1])):
                if y[0] == "const" and y[1] < 0.0:
                    return ("sub", x, ("const", -y[1]))          # x + (-c) is x - c
            if e[2][0] == "neg":
                return ("sub", e[1], e[2][1])
            if e[1][0] == "neg":
                return ("sub", e[2], e[1][1])
        if e[0] == "sub" and e[2][0] == "neg":
            return ("add", e[1], e[2][1])
        if e[0] == "sub" and e[2][0] == "const" and e[2][1] < 0.0:
            return ("add", e[1], ("const", -e[2][1]))
        return e
    t = norm(t)
    for _ in range(6):                    # (to a fixed point: a rewrite can expose another one level up)
        u = norm(fix(t))
        if u == t:
            break
        t = u
    return t
