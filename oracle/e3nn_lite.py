"""Clean-room restatement of the e3nn 0.5.1 arithmetic used by the reference.

TEST INFRASTRUCTURE (see oracle/__init__.py).  e3nn is an un-vendored
third-party dependency of the reference (env.yaml:51, e3nn==0.5.1); it is not
installed here and there is no network, so the algorithm is restated from its
published definition.  **Parity unpinned**: the reference holds no golden
vectors at this boundary; correctness is established by property tests
(tests/test_e3nn_lite.py): rotation invariance of every Wigner-3j tensor,
equivariance of the tensor products, known-answer values.

Call sites in the reference: druglib/models/Docking/interaction/tpscore.py
:7,25,163,226,373,598,620,680,708,717,728,729,742,753,755.

Conventions reproduced (e3nn 0.5.1):
* Irreps keep their written order; a feature vector is the concatenation of
  blocks ``[mul, 2l+1]`` (mul outer).
* Real spherical harmonics with y as the polar axis, ``component``
  normalisation (|Y_l|^2 = 2l+1), components ordered m=-l..l.
* ``wigner_3j`` = real-basis change of the SU(2) Clebsch-Gordan coefficients
  (Racah formula), Frobenius-normalised to 1.
* ``FullyConnectedTensorProduct``: instructions enumerated i1 outer, i2 middle,
  i_out inner, mode ``uvw``, per-sample weights laid out as consecutive
  ``(mul1, mul2, mul_out)`` row-major blocks, ``irrep_normalization=component``,
  ``path_normalization=element``.
* ``FullTensorProduct``: mode ``uvuv``, weight-free, output irreps sorted by
  (l, p), coefficient sqrt(2 l_out + 1).
"""
import math
import re
from fractions import Fraction
from functools import lru_cache
from math import factorial

import torch


# --------------------------------------------------------------------------- irreps
class Irrep(tuple):
    def __new__(cls, l, p=None):
        if p is None:
            if isinstance(l, Irrep):
                return l
            if isinstance(l, str):
                m = re.fullmatch(r"\s*(\d+)([eo])\s*", l)
                assert m, f"bad irrep {l!r}"
                l, p = int(m.group(1)), (1 if m.group(2) == "e" else -1)
            else:
                l, p = l
        assert p in (-1, 1) and l >= 0
        return super().__new__(cls, (int(l), int(p)))

    @property
    def l(self):
        return self[0]

    @property
    def p(self):
        return self[1]

    @property
    def dim(self):
        return 2 * self[0] + 1

    def __repr__(self):
        return f"{self.l}{'e' if self.p == 1 else 'o'}"

    def __mul__(self, other):
        """Selection rule: |l1-l2| <= l <= l1+l2, parity product."""
        other = Irrep(other)
        p = self.p * other.p
        return [Irrep(l, p) for l in range(abs(self.l - other.l), self.l + other.l + 1)]


class _MulIr(tuple):
    def __new__(cls, mul, ir):
        return super().__new__(cls, (int(mul), Irrep(ir)))

    @property
    def mul(self):
        return self[0]

    @property
    def ir(self):
        return self[1]

    @property
    def dim(self):
        return self[0] * self[1].dim


class Irreps(tuple):
    def __new__(cls, irreps):
        if isinstance(irreps, Irreps):
            return irreps
        out = []
        if isinstance(irreps, Irrep):
            out.append(_MulIr(1, irreps))
        elif isinstance(irreps, str):
            for term in irreps.split("+"):
                term = term.strip()
                if not term:
                    continue
                if "x" in term:
                    mul, ir = term.split("x")
                    out.append(_MulIr(int(mul), Irrep(ir)))
                else:
                    out.append(_MulIr(1, Irrep(term)))
        else:
            for item in irreps:
                if isinstance(item, (str, Irrep)) and not isinstance(item, _MulIr):
                    out.append(_MulIr(1, Irrep(item)))
                else:
                    mul, ir = item
                    out.append(_MulIr(mul, Irrep(ir)))
        return super().__new__(cls, out)

    @staticmethod
    def spherical_harmonics(lmax, p=-1):
        return Irreps([(1, (l, p ** l)) for l in range(lmax + 1)])

    @property
    def dim(self):
        return sum(mi.dim for mi in self)

    @property
    def num_irreps(self):
        return sum(mi.mul for mi in self)

    @property
    def ls(self):
        return [mi.ir.l for mi in self for _ in range(mi.mul)]

    def slices(self):
        out, i = [], 0
        for mi in self:
            out.append(slice(i, i + mi.dim))
            i += mi.dim
        return out

    def sort(self):
        """Stable sort by (l, p); returns (irreps, p, inv) like e3nn."""
        order = sorted(range(len(self)), key=lambda i: (self[i].ir, i))  # inv
        p = [0] * len(self)
        for new, old in enumerate(order):
            p[old] = new
        return Irreps([self[i] for i in order]), tuple(p), tuple(order)

    def __repr__(self):
        return "+".join(f"{mi.mul}x{mi.ir}" for mi in self)


# --------------------------------------------------------------------------- SH
def _sh_component(lmax, x, y, z):
    """Real SH l=0..lmax (<=2), 'component' normalised, m=-l..l, y polar."""
    out = [torch.ones_like(x)]
    if lmax >= 1:
        s3 = math.sqrt(3.0)
        out += [s3 * x, s3 * y, s3 * z]
    if lmax >= 2:
        s15, s5 = math.sqrt(15.0), math.sqrt(5.0)
        y2 = y * y
        x2z2 = x * x + z * z
        out += [s15 * x * z, s15 * x * y, s5 * (y2 - 0.5 * x2z2), s15 * y * z,
                (s15 / 2.0) * (z * z - x * x)]
    assert lmax <= 2, "e3nn_lite: only lmax<=2 is needed by the reference config"
    return torch.stack(out, dim=-1)


def spherical_harmonics(l, x, normalize, normalization="integral"):
    """o3.spherical_harmonics(irreps | 'le' | int, x, normalize, normalization)."""
    if isinstance(l, int):
        ls = [l]
    else:
        ls = [mi.ir.l for mi in Irreps(l) for _ in range(mi.mul)]
    lmax = max(ls)
    if normalize:
        x = torch.nn.functional.normalize(x, dim=-1)  # zero stays zero
    sh = _sh_component(lmax, x[..., 0], x[..., 1], x[..., 2])
    sh = torch.cat([sh[..., l * l:(l + 1) * (l + 1)] for l in ls], dim=-1)
    if normalization == "integral":
        sh = sh / math.sqrt(4 * math.pi)
    elif normalization == "norm":
        sh = sh / torch.cat([math.sqrt(2 * l + 1) * torch.ones(2 * l + 1, dtype=sh.dtype) for l in ls])
    else:
        assert normalization == "component"
    return sh


# --------------------------------------------------------------------------- Wigner 3j
def _su2_cg_coeff(j1, m1, j2, m2, j3, m3):
    if m3 != m1 + m2:
        return 0.0
    vmin = int(max(-j1 + j2 + m3, -j1 + m1, 0))
    vmax = int(min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3))

    def f(n):
        assert n == round(n)
        return factorial(round(n))

    c = ((2.0 * j3 + 1.0) * Fraction(
        f(j3 + j1 - j2) * f(j3 - j1 + j2) * f(j1 + j2 - j3) * f(j3 + m3) * f(j3 - m3),
        f(j1 + j2 + j3 + 1) * f(j1 - m1) * f(j1 + m1) * f(j2 - m2) * f(j2 + m2))) ** 0.5
    s = 0
    for v in range(vmin, vmax + 1):
        s += (-1) ** int(v + j2 + m2) * Fraction(
            f(j2 + j3 + m1 - v) * f(j1 - m1 + v),
            f(v) * f(j3 - j1 + j2 - v) * f(j3 + m3 - v) * f(v + j1 - j2 - m3))
    return float(c * s)


def _su2_cg(j1, j2, j3):
    mat = torch.zeros(2 * j1 + 1, 2 * j2 + 1, 2 * j3 + 1, dtype=torch.float64)
    if abs(j1 - j2) <= j3 <= j1 + j2:
        for m1 in range(-j1, j1 + 1):
            for m2 in range(-j2, j2 + 1):
                if abs(m1 + m2) <= j3:
                    mat[j1 + m1, j2 + m2, j3 + m1 + m2] = _su2_cg_coeff(j1, m1, j2, m2, j3, m1 + m2)
    return mat


def _real_to_complex(l):
    q = torch.zeros(2 * l + 1, 2 * l + 1, dtype=torch.complex128)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = 1 / 2 ** 0.5
        q[l + m, l - abs(m)] = -1j / 2 ** 0.5
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m / 2 ** 0.5
        q[l + m, l - abs(m)] = 1j * (-1) ** m / 2 ** 0.5
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def wigner_3j(l1, l2, l3):
    """Real-basis Wigner-3j, float64, Frobenius norm 1, shape [2l1+1,2l2+1,2l3+1]."""
    assert abs(l2 - l3) <= l1 <= l2 + l3
    q1, q2, q3 = _real_to_complex(l1), _real_to_complex(l2), _real_to_complex(l3)
    c = _su2_cg(l1, l2, l3).to(torch.complex128)
    c = torch.einsum("ij,kl,mn,ikn->jlm", q1, q2, torch.conj(q3.T), c)
    assert torch.all(torch.abs(c.imag) < 1e-5)
    c = c.real
    return c / c.norm()


# --------------------------------------------------------------------------- tensor products
class _Instr:
    __slots__ = ("i1", "i2", "io", "mode", "has_weight", "coeff", "w_off", "w_shape", "_n_el")

    def __init__(self, i1, i2, io, mode, has_weight):
        self.i1, self.i2, self.io, self.mode, self.has_weight = i1, i2, io, mode, has_weight
        self.coeff, self.w_off, self.w_shape = 1.0, 0, None


class _TensorProduct(torch.nn.Module):
    def __init__(self, in1, in2, out, instr):
        super().__init__()
        self.irreps_in1, self.irreps_in2, self.irreps_out = Irreps(in1), Irreps(in2), Irreps(out)
        self.instructions = instr
        off = 0
        for ins in instr:
            m1, m2, mo = self.irreps_in1[ins.i1].mul, self.irreps_in2[ins.i2].mul, self.irreps_out[ins.io].mul
            if ins.mode == "uvw":
                n_el = m1 * m2
                ins.w_shape = (m1, m2, mo)
            elif ins.mode == "uvuv":
                n_el = 1
                assert mo == m1 * m2
            else:
                raise NotImplementedError(ins.mode)
            ins._n_el = n_el
            if ins.has_weight:
                ins.w_off = off
                off += m1 * m2 * mo
        self.weight_numel = off
        for ins in instr:
            alpha = self.irreps_out[ins.io].ir.dim  # irrep_normalization='component'
            x = sum(i._n_el for i in instr if i.io == ins.io)  # path_normalization='element'
            if x > 0:
                alpha /= x
            ins.coeff = math.sqrt(alpha)

    def forward(self, x1, x2, weight=None):
        z = x1.shape[0]
        s1, s2 = self.irreps_in1.slices(), self.irreps_in2.slices()
        outs = [None] * len(self.irreps_out)
        for ins in self.instructions:
            mi1, mi2, mio = self.irreps_in1[ins.i1], self.irreps_in2[ins.i2], self.irreps_out[ins.io]
            a = x1[:, s1[ins.i1]].reshape(z, mi1.mul, mi1.ir.dim)
            b = x2[:, s2[ins.i2]].reshape(z, mi2.mul, mi2.ir.dim)
            c = wigner_3j(mi1.ir.l, mi2.ir.l, mio.ir.l).to(x1.dtype)
            if ins.mode == "uvw":
                w = weight[:, ins.w_off:ins.w_off + mi1.mul * mi2.mul * mio.mul].reshape(z, *ins.w_shape)
                # y[z,u,v,k] = sum_ij C[i,j,k] a[z,u,i] b[z,v,j]; out[z,w,k] = sum_uv w[z,u,v,w] y[z,u,v,k]
                y = torch.einsum("ijk,zui,zvj->zuvk", c, a, b)
                r = torch.einsum("zuvw,zuvk->zwk", w, y)
            else:  # uvuv
                r = torch.einsum("ijk,zui,zvj->zuvk", c, a, b).reshape(z, mi1.mul * mi2.mul, mio.ir.dim)
            r = (ins.coeff * r).reshape(z, mio.dim)
            outs[ins.io] = r if outs[ins.io] is None else outs[ins.io] + r
        for i, mio in enumerate(self.irreps_out):
            if outs[i] is None:
                outs[i] = x1.new_zeros(z, mio.dim)
        return torch.cat(outs, dim=1)


class FullyConnectedTensorProduct(_TensorProduct):
    def __init__(self, irreps_in1, irreps_in2, irreps_out, shared_weights=False, **_):
        assert shared_weights is False, "the reference only uses shared_weights=False"
        in1, in2, out = Irreps(irreps_in1), Irreps(irreps_in2), Irreps(irreps_out)
        instr = [
            _Instr(i1, i2, io, "uvw", True)
            for i1, mi1 in enumerate(in1)
            for i2, mi2 in enumerate(in2)
            for io, mio in enumerate(out)
            if mio.ir in mi1.ir * mi2.ir
        ]
        super().__init__(in1, in2, out, instr)


class FullTensorProduct(_TensorProduct):
    def __init__(self, irreps_in1, irreps_in2, **_):
        in1, in2 = Irreps(irreps_in1), Irreps(irreps_in2)
        out, raw = [], []
        for i1, mi1 in enumerate(in1):
            for i2, mi2 in enumerate(in2):
                for ir_out in mi1.ir * mi2.ir:
                    raw.append((i1, i2, len(out)))
                    out.append((mi1.mul * mi2.mul, ir_out))
        out, p, _ = Irreps(out).sort()
        instr = [_Instr(i1, i2, p[io], "uvuv", False) for i1, i2, io in raw]
        super().__init__(in1, in2, out, instr)
