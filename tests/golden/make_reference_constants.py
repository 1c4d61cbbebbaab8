"""Generator of tests/golden/reference_constants.npz -- what the reference's KERNEL sources pin without being compiled.

The reference's .cu files cannot be built here (they include cuda.h / cuda_runtime.h / cuda_fp16.h, which this image lacks,
and stand-ins are not allowed), so nothing the reference's kernels computed can be recorded.  What their TEXT fixes can:

* shencoder/src/shencoder.cu:50-356 -- the 64 spherical-harmonics polynomials and their 3 x 64 derivatives are literal
  arithmetic expressions.  A small expression parser (below, ours) reads each `outputs[i] = <expr> ;` / `dx[i]` / `dy[i]` /
  `dz[i]` statement and (a) EVALUATES it in float32, operation by operation in the source's own order (C++ left-to-right
  associativity, `f`-suffixed literals rounded once from decimal) on seeded directions -> golden input/output vectors of
  kernel_sh's arithmetic (up to nvcc's fma contraction); (b) EXPANDS it into monomials c * x^i y^j z^k with exact float64
  coefficient products -> the per-term coefficient table.
* gridencoder/src/gridencoder.cu:42 -- the hash primes; raymarching/src/pcg32.h:32-34,66-72,111 -- PCG32 multiplier, default
  state / stream, output-function shifts, next_float's mantissa trick; raymarching/src/raymarching.cu:21-24 -- SQRT3 & co,
  :886 the early-termination threshold of composite_rays.
* raymarching/src/raymarching.h:7-19, gridencoder/src/gridencoder.h:12-13, shencoder/src/shencoder.h:9-12 and the three
  bindings.cpp -- the names and argument order of the 15 operator entry points (the drop-in boundary's signatures).

Only numbers, exponent tables and identifier lists are stored: data, no source text.  Run here (the reference exists only
in this container):  python tests/golden/make_reference_constants.py
"""
import os
import re
import sys

import numpy as np

REF = os.environ.get("PVD_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_constants.npz")

# ---------------------------------------------------------------------------------------------- expression parser
_TOK = re.compile(r"\s*(?:(\d+\.\d*(?:[eE][-+]?\d+)?f?|\d+f?)|([A-Za-z_]\w*)|(.))")


def tokenize(s):
    out = []
    for num, name, op in _TOK.findall(s):
        if num:
            out.append(("num", num))
        elif name:
            out.append(("var", name))
        elif op.strip():
            out.append(("op", op))
    return out


class Parser:
    """expr := term (('+'|'-') term)* ; term := unary ('*' unary)* ; unary := '-' unary | atom ; atom := num | var | '(' expr ')'"""

    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else (None, None)

    def take(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def expr(self):
        node = self.term()
        while self.peek() in (("op", "+"), ("op", "-")):
            op = self.take()[1]
            node = (op, node, self.term())
        return node

    def term(self):
        node = self.unary()
        while self.peek() == ("op", "*"):
            self.take()
            node = ("*", node, self.unary())
        return node

    def unary(self):
        if self.peek() == ("op", "-"):
            self.take()
            return ("neg", self.unary())
        return self.atom()

    def atom(self):
        kind, v = self.take()
        if kind == "num":
            return ("num", v)
        if kind == "var":
            if v == "pow" and self.peek() == ("op", "("):  # pow(<var>, <int>): a handful of derivative terms
                self.take()
                base = self.expr()
                assert self.take() == ("op", ",")
                kind_n, n = self.take()
                assert kind_n == "num" and n.isdigit(), n
                assert self.take() == ("op", ")")
                return ("pow", base, int(n))
            return ("var", v)
        assert (kind, v) == ("op", "("), (kind, v)
        node = self.expr()
        assert self.take() == ("op", ")")
        return node


def parse(s):
    p = Parser(tokenize(s))
    node = p.expr()
    assert p.i == len(p.t), (s, p.t[p.i:])
    return node


def lit32(txt):
    """an f-suffixed C literal: decimal -> float32, one rounding"""
    assert txt.endswith("f"), txt
    return np.float32(txt[:-1])


def eval32(node, env):
    """float32 evaluation in the source's own operation order (arrays of float32)"""
    k = node[0]
    if k == "num":
        return lit32(node[1])
    if k == "var":
        return env[node[1]]
    if k == "neg":
        return -eval32(node[1], env)
    if k == "pow":  # powf(float, int): correctly rounded power of the float32 base (CUDA's is within an ulp of it)
        return np.power(eval32(node[1], env).astype(np.float64), node[2]).astype(np.float32)
    a, b = eval32(node[1], env), eval32(node[2], env)
    r = a + b if k == "+" else a - b if k == "-" else a * b
    return np.float32(r) if np.isscalar(r) else r.astype(np.float32)


# monomial expansion: {(i, j, k): coefficient}
_VARS = {"x": (1, 0, 0), "y": (0, 1, 0), "z": (0, 0, 1), "xy": (1, 1, 0), "xz": (1, 0, 1), "yz": (0, 1, 1), "xyz": (1, 1, 1),
         "x2": (2, 0, 0), "y2": (0, 2, 0), "z2": (0, 0, 2), "x4": (4, 0, 0), "y4": (0, 4, 0), "z4": (0, 0, 4),
         "x6": (6, 0, 0), "y6": (0, 6, 0), "z6": (0, 0, 6)}


def expand(node):
    k = node[0]
    if k == "num":
        return {(0, 0, 0): float(node[1].rstrip("f"))}
    if k == "var":
        return {_VARS[node[1]]: 1.0}
    if k == "neg":
        return {e: -c for e, c in expand(node[1]).items()}
    if k == "pow":
        out = {(0, 0, 0): 1.0}
        for _ in range(node[2]):
            out = expand(("*", ("poly", out), node[1]))
        return out
    if k == "poly":
        return node[1]
    a, b = expand(node[1]), expand(node[2])
    if k in "+-":
        out = dict(a)
        for e, c in b.items():
            out[e] = out.get(e, 0.0) + (c if k == "+" else -c)
        return out
    out = {}
    for ea, ca in a.items():
        for eb, cb in b.items():
            e = (ea[0] + eb[0], ea[1] + eb[1], ea[2] + eb[2])
            out[e] = out.get(e, 0.0) + ca * cb
    return out


# ---------------------------------------------------------------------------------------------- shencoder.cu
def sh_statements():
    txt = open(os.path.join(REF, "shencoder/src/shencoder.cu")).read()
    body = txt[txt.index("kernel_sh("):txt.index("kernel_sh_backward")]
    stm = {"outputs": {}, "dx": {}, "dy": {}, "dz": {}}
    for name, idx, rhs in re.findall(r"\b(outputs|dx|dy|dz)\[(\d+)\]\s*=\s*([^;]+);", body):
        assert int(idx) not in stm[name], (name, idx)
        stm[name][int(idx)] = parse(rhs)
    for name in stm:
        assert sorted(stm[name]) == list(range(64)), (name, len(stm[name]))
    # the temporaries, in the reference's own definitions (shencoder.cu:45-48): parsed, not assumed
    defs = re.findall(r"scalar_t\s+((?:\w+\s*=\s*[^,;]+[,;]\s*)+)", body)
    temps = []
    for d in defs:
        for name, rhs in re.findall(r"(\w+)\s*=\s*([^,;]+)[,;]", d):
            if name in ("x", "y", "z"):
                continue
            temps.append((name, parse(rhs)))
    assert [t[0] for t in temps] == ["xy", "xz", "yz", "x2", "y2", "z2", "xyz", "x4", "y4", "z4", "x6", "y6", "z6"], temps
    return stm, temps


def sh_fixture(out):
    stm, temps = sh_statements()
    rng = np.random.default_rng(20260927)
    d = rng.standard_normal((509, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    extra = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1], [0, 0, 0],
                      [0.3, -0.2, 0.5], [1.5, -0.7, 0.25]], dtype=np.float64)  # axis-aligned, zero and non-unit inputs
    dirs = np.concatenate([d, extra]).astype(np.float32)
    env = {"x": dirs[:, 0].copy(), "y": dirs[:, 1].copy(), "z": dirs[:, 2].copy()}
    for name, node in temps:  # the temporaries' own exponents must be what the table above says
        env[name] = eval32(node, env)
        assert expand(node) == {_VARS[name]: 1.0}, name
    for key, name in (("sh_out", "outputs"), ("sh_dx", "dx"), ("sh_dy", "dy"), ("sh_dz", "dz")):
        vals = np.zeros((len(dirs), 64), np.float32)
        for i in range(64):
            vals[:, i] = eval32(stm[name][i], env)
        out[key] = vals
    out["sh_dirs"] = dirs
    # per-term coefficient table
    t_out, t_exp, t_coef, lead = [], [], [], []
    for fam, name in enumerate(("outputs", "dx", "dy", "dz")):
        for i in range(64):
            poly = {e: c for e, c in expand(stm[name][i]).items() if c != 0.0}
            for e in sorted(poly):
                t_out.append(fam * 64 + i)
                t_exp.append(e)
                t_coef.append(poly[e])
    out["sh_term_output"] = np.array(t_out, np.int32)          # family * 64 + index; family 0 = value, 1..3 = d/dx, d/dy, d/dz
    out["sh_term_exponents"] = np.array(t_exp, np.int8)        # [T, 3] powers of x, y, z
    out["sh_term_coefficient"] = np.array(t_coef, np.float64)  # products of the source's literals
    # the first literal of every value statement (the normalisation constant K_l^m as the reference spells it)
    txt = open(os.path.join(REF, "shencoder/src/shencoder.cu")).read()
    body = txt[txt.index("kernel_sh("):txt.index("kernel_sh_backward")]
    for idx, rhs in re.findall(r"\boutputs\[(\d+)\]\s*=\s*([^;]+);", body):
        lead.append(float(re.search(r"-?\d+\.\d+", rhs).group(0)))
    out["sh_lead_literal"] = np.array(lead, np.float64)


# ---------------------------------------------------------------------------------------------- scalars
def scalar_fixture(out):
    g = open(os.path.join(REF, "gridencoder/src/gridencoder.cu")).read()
    m = re.search(r"primes\[7\]\s*=\s*\{([^}]+)\}", g)
    out["hash_primes"] = np.array([int(v) for v in m.group(1).replace(" ", "").split(",")], np.uint64).astype(np.uint32)
    p = open(os.path.join(REF, "raymarching/src/pcg32.h")).read()
    for key, name in (("pcg32_default_state", "PCG32_DEFAULT_STATE"), ("pcg32_default_stream", "PCG32_DEFAULT_STREAM"),
                      ("pcg32_mult", "PCG32_MULT")):
        out[key] = np.array([int(re.search(name + r"\s+0x([0-9a-fA-F]+)ULL", p).group(1), 16)], np.uint64)
    nu = p[p.index("uint32_t next_uint()"):p.index("uint32_t next_uint(uint32_t bound)")]
    shifts = [int(v) for v in re.findall(r">>\s*(\d+)u", nu)]
    assert len(shifts) == 3, shifts
    out["pcg32_output_shifts"] = np.array(shifts, np.uint32)  # xorshift, truncation, rotation
    nf = p[p.index("float next_float()"):p.index("double next_double()")]
    out["pcg32_float_shift"] = np.array([int(re.search(r">>\s*(\d+)\)", nf).group(1))], np.uint32)
    out["pcg32_float_exponent_bits"] = np.array([int(re.search(r"0x([0-9a-fA-F]+)u", nf).group(1), 16)], np.uint32)
    r = open(os.path.join(REF, "raymarching/src/raymarching.cu")).read()
    for key, name in (("rm_sqrt3", "SQRT3"), ("rm_rsqrt3", "RSQRT3"), ("rm_pi", "PI"), ("rm_rpi", "RPI")):
        out[key] = np.array([lit32(re.search(r"float " + name + r"\(\)\s*\{\s*return\s+([0-9.]+f)", r).group(1))], np.float32)
    comp = r[r.index("kernel_composite_rays("):]
    out["rm_composite_rays_T_threshold"] = np.array([float(re.search(r"if \(T < ([0-9e.-]+)\) break", comp).group(1))], np.float64)


# ---------------------------------------------------------------------------------------------- signatures
def signature_fixture(out):
    names, args = [], []
    for mod, hdr, bind in (("raymarching", "raymarching/src/raymarching.h", "raymarching/src/bindings.cpp"),
                           ("gridencoder", "gridencoder/src/gridencoder.h", "gridencoder/src/bindings.cpp"),
                           ("shencoder", "shencoder/src/shencoder.h", "shencoder/src/bindings.cpp")):
        h = open(os.path.join(REF, hdr)).read()
        b = open(os.path.join(REF, bind)).read()
        bound = re.findall(r'm\.def\("(\w+)",\s*&(\w+)', b)
        decl = {}
        for fn, arglist in re.findall(r"^void (\w+)\(([^)]*)\);", h, flags=re.M):
            decl[fn] = [a.strip().split()[-1] for a in arglist.split(",")]
        for pyname, cname in bound:
            names.append("%s.%s" % (mod, pyname))
            args.append(",".join(decl[cname]))
    out["signature_names"] = np.array(names)   # "<module>.<python name>" in binding order
    out["signature_args"] = np.array(args)     # comma-joined argument names in declaration order


def main():
    out = {}
    sh_fixture(out)
    scalar_fixture(out)
    signature_fixture(out)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    sys.exit(main())
