// Host-side SO(3) bookkeeping of libdbfr: real Wigner-3j tensors (Racah formula +
// real<->complex basis change, the published e3nn 0.5.1 construction), tensor-product
// path enumeration (FullyConnectedTensorProduct: i1 outer, i2 middle, i_out inner,
// mode uvw, irrep_normalization=component, path_normalization=element), and the
// self-check that the closed forms baked into the kernels equal those tensors.
// Replaces: e3nn.o3.FullyConnectedTensorProduct.__init__ as called at
// druglib/models/Docking/interaction/tpscore.py:163 (instruction/weight layout only).
#include <cmath>
#include <complex>
#include <cstdio>

#include "common.h"

typedef std::complex<double> cd;

static double fact(int n) {
  double r = 1;
  for (int i = 2; i <= n; ++i) r *= i;
  return r;
}

static double su2_cg(int j1, int m1, int j2, int m2, int j3, int m3) {
  if (m3 != m1 + m2) return 0.0;
  int vmin = std::max(std::max(-j1 + j2 + m3, -j1 + m1), 0);
  int vmax = std::min(std::min(j2 + j3 + m1, j3 - j1 + j2), j3 + m3);
  double c = std::sqrt((2.0 * j3 + 1.0) * fact(j3 + j1 - j2) * fact(j3 - j1 + j2) * fact(j1 + j2 - j3) *
                       fact(j3 + m3) * fact(j3 - m3) /
                       (fact(j1 + j2 + j3 + 1) * fact(j1 - m1) * fact(j1 + m1) * fact(j2 - m2) * fact(j2 + m2)));
  double s = 0;
  for (int v = vmin; v <= vmax; ++v) {
    double sign = ((v + j2 + m2) & 1) ? -1.0 : 1.0;
    s += sign * fact(j2 + j3 + m1 - v) * fact(j1 - m1 + v) /
         (fact(v) * fact(j3 - j1 + j2 - v) * fact(j3 + m3 - v) * fact(v + j1 - j2 - m3));
  }
  return c * s;
}

// q[row = complex m index][col = real index]
static std::vector<cd> real_to_complex(int l) {
  int d = 2 * l + 1;
  std::vector<cd> q(d * d, cd(0, 0));
  const double is2 = 1.0 / std::sqrt(2.0);
  for (int m = -l; m < 0; ++m) {
    q[(l + m) * d + (l - m)] = cd(is2, 0);
    q[(l + m) * d + (l + m)] = cd(0, -is2);
  }
  q[l * d + l] = cd(1, 0);
  for (int m = 1; m <= l; ++m) {
    double sg = (m & 1) ? -1.0 : 1.0;
    q[(l + m) * d + (l + m)] = cd(sg * is2, 0);
    q[(l + m) * d + (l - m)] = cd(0, sg * is2);
  }
  cd ph = std::pow(cd(0, -1), l);
  for (auto& x : q) x *= ph;
  return q;
}

void wigner3j_real(int l1, int l2, int l3, std::vector<double>& out) {
  int d1 = 2 * l1 + 1, d2 = 2 * l2 + 1, d3 = 2 * l3 + 1;
  out.assign(d1 * d2 * d3, 0.0);
  if (l3 < std::abs(l1 - l2) || l3 > l1 + l2) return;
  std::vector<double> C(d1 * d2 * d3, 0.0);
  for (int m1 = -l1; m1 <= l1; ++m1)
    for (int m2 = -l2; m2 <= l2; ++m2)
      if (std::abs(m1 + m2) <= l3)
        C[((l1 + m1) * d2 + (l2 + m2)) * d3 + (l3 + m1 + m2)] = su2_cg(l1, m1, l2, m2, l3, m1 + m2);
  auto Q1 = real_to_complex(l1), Q2 = real_to_complex(l2), Q3 = real_to_complex(l3);
  // einsum('ij,kl,mn,ikn->jlm', Q1, Q2, conj(Q3^T), C): Q3c[m][n] = conj(Q3[n][m])
  double nrm = 0;
  for (int j = 0; j < d1; ++j)
    for (int l = 0; l < d2; ++l)
      for (int m = 0; m < d3; ++m) {
        cd s(0, 0);
        for (int i = 0; i < d1; ++i)
          for (int k = 0; k < d2; ++k)
            for (int n = 0; n < d3; ++n) {
              double c = C[(i * d2 + k) * d3 + n];
              if (c == 0.0) continue;
              s += Q1[i * d1 + j] * Q2[k * d2 + l] * std::conj(Q3[n * d3 + m]) * c;
            }
        out[(j * d2 + l) * d3 + m] = s.real();
        nrm += s.real() * s.real();
      }
  nrm = std::sqrt(nrm);
  for (auto& x : out) x /= nrm;
}

static int offset_of(const std::vector<Irr>& v, int idx) {
  int o = 0;
  for (int i = 0; i < idx; ++i) o += v[i].mul * v[i].dim();
  return o;
}

ConvSpec make_conv_spec(int kind) {
  ConvSpec s;
  const Irr e0{NS, 0, 1}, o1{NV, 1, -1}, e1{NV, 1, 1}, o0{NS, 0, -1};
  std::vector<std::vector<Irr>> seq = {{e0}, {e0, o1}, {e0, o1, e1}, {e0, o1, e1, o0}};
  std::vector<Irr> sh = {{1, 0, 1}, {1, 1, -1}, {1, 2, 1}};
  s.K = 3 * NS;
  if (kind >= 0 && kind <= 3) {
    s.in = seq[kind];
    s.out = seq[std::min(kind + 1, 3)];
    s.sh = sh;
  } else if (kind == 4) {  // final_conv: tpscore.py:348-356
    s.in = seq[3];
    s.out = {{2, 1, -1}, {2, 1, 1}};
    s.sh = sh;
    s.K = 2 * NS;
  } else {                 // tor_bond_conv / sc_tor_bond_conv: tpscore.py:374-382,396-404
    // sh irreps = FullTensorProduct(sh,"2e").irreps_out sorted; only its leading 0e,1o,1e
    // slots (record offsets 0,1,4) can reach scalar outputs from l<=1 inputs.
    s.in = seq[3];
    s.out = {{NS, 0, -1}, {NS, 0, 1}};
    s.sh = {{1, 0, 1}, {1, 1, -1}, {1, 1, 1}};
  }
  s.D_in = offset_of(s.in, (int)s.in.size());
  s.D_out = offset_of(s.out, (int)s.out.size());
  int woff = 0;
  for (int i1 = 0; i1 < (int)s.in.size(); ++i1)
    for (int i2 = 0; i2 < (int)s.sh.size(); ++i2)
      for (int io = 0; io < (int)s.out.size(); ++io) {
        const Irr &a = s.in[i1], &b = s.sh[i2], &c = s.out[io];
        if (a.p * b.p != c.p) continue;
        if (c.l < std::abs(a.l - b.l) || c.l > a.l + b.l) continue;
        PathDesc p{};
        p.i1 = i1; p.i2 = i2; p.io = io; p.l1 = a.l; p.l2 = b.l; p.lo = c.l;
        p.mul1 = a.mul; p.mulo = c.mul; p.w_off = woff;
        woff += a.mul * b.mul * c.mul;
        p.in_off = offset_of(s.in, i1);
        p.sh_off = offset_of(s.sh, i2);
        p.out_off = offset_of(s.out, io);
        s.paths.push_back(p);
      }
  s.W = woff;
  for (auto& p : s.paths) {
    double fan = 0;
    for (auto& q : s.paths)
      if (q.io == p.io) fan += (double)q.mul1 * 1.0;  // mul2 == 1 always
    p.coeff = (float)std::sqrt((2.0 * p.lo + 1.0) / fan);
    double cg;
    if (p.l1 == 0 && p.l2 == 0 && p.lo == 0) { p.type = PT_SS; cg = 1.0; }
    else if (p.l1 == 0 && p.l2 == 1 && p.lo == 1) { p.type = PT_SV; cg = 1.0 / std::sqrt(3.0); }
    else if (p.l1 == 1 && p.l2 == 0 && p.lo == 1) { p.type = PT_VS; cg = 1.0 / std::sqrt(3.0); }
    else if (p.l1 == 1 && p.l2 == 1 && p.lo == 0) { p.type = PT_VVS; cg = 1.0 / std::sqrt(3.0); }
    else if (p.l1 == 1 && p.l2 == 1 && p.lo == 1) { p.type = PT_VVV; cg = 1.0 / std::sqrt(6.0); }
    else if (p.l1 == 1 && p.l2 == 2 && p.lo == 1) { p.type = PT_VTV; cg = 1.0 / std::sqrt(30.0); }
    else { p.type = -1; cg = 0; }
    p.fold = (float)((double)p.coeff * cg);
  }
  return s;
}

// closed forms used by the kernels (must mirror conv.hip / graph.hip exactly)
static void closed_form(int type, const double* z, const double* s, double* out) {
  const double r3 = std::sqrt(3.0);
  switch (type) {
    case PT_SS: out[0] = z[0] * s[0]; break;
    case PT_SV: for (int k = 0; k < 3; ++k) out[k] = z[0] * s[k]; break;
    case PT_VS: for (int k = 0; k < 3; ++k) out[k] = z[k] * s[0]; break;
    case PT_VVS: out[0] = z[0] * s[0] + z[1] * s[1] + z[2] * s[2]; break;
    case PT_VVV:
      out[0] = z[1] * s[2] - z[2] * s[1];
      out[1] = z[2] * s[0] - z[0] * s[2];
      out[2] = z[0] * s[1] - z[1] * s[0];
      break;
    case PT_VTV: {
      double m00 = -s[2] - r3 * s[4], m01 = r3 * s[1], m02 = r3 * s[0];
      double m11 = 2 * s[2], m12 = r3 * s[3], m22 = -s[2] + r3 * s[4];
      out[0] = m00 * z[0] + m01 * z[1] + m02 * z[2];
      out[1] = m01 * z[0] + m11 * z[1] + m12 * z[2];
      out[2] = m02 * z[0] + m12 * z[1] + m22 * z[2];
      break;
    }
  }
}

int so3_selftest(std::string& err) {
  struct T { int type, l1, l2, lo; double scale; };
  const T tests[] = {{PT_SS, 0, 0, 0, 1.0}, {PT_SV, 0, 1, 1, 1 / std::sqrt(3.0)}, {PT_VS, 1, 0, 1, 1 / std::sqrt(3.0)},
                     {PT_VVS, 1, 1, 0, 1 / std::sqrt(3.0)}, {PT_VVV, 1, 1, 1, 1 / std::sqrt(6.0)},
                     {PT_VTV, 1, 2, 1, 1 / std::sqrt(30.0)}};
  for (const T& t : tests) {
    std::vector<double> C;
    wigner3j_real(t.l1, t.l2, t.lo, C);
    int d1 = 2 * t.l1 + 1, d2 = 2 * t.l2 + 1, d3 = 2 * t.lo + 1;
    // probe with fixed pseudo-random vectors
    double z[3] = {0.37, -1.21, 0.58}, s[5] = {0.91, -0.44, 0.27, 1.13, -0.72};
    double ref[3] = {0, 0, 0}, got[3] = {0, 0, 0};
    for (int i = 0; i < d1; ++i)
      for (int j = 0; j < d2; ++j)
        for (int k = 0; k < d3; ++k) ref[k] += C[(i * d2 + j) * d3 + k] * z[i] * s[j];
    closed_form(t.type, z, s, got);
    for (int k = 0; k < d3; ++k)
      if (std::fabs(got[k] * t.scale - ref[k]) > 1e-12) {
        char buf[160];
        snprintf(buf, sizeof buf, "closed form of w3j(%d,%d,%d) disagrees with the Racah tensor (k=%d: %g vs %g)",
                 t.l1, t.l2, t.lo, k, got[k] * t.scale, ref[k]);
        err = buf;
        return DBFR_ERR_SELFTEST;
      }
  }
  return DBFR_OK;
}
