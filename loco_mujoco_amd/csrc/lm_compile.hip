// lm_compile.hip — the model compiler on the device: a freshly randomised model per environment and episode.
//
// The reference re-compiles its MuJoCo model at every reset() from a randomised XML (loco_mujoco/environments/base.py:183-185,
// utils/domain_randomization.py:219-227,386-514). On this path a re-compile changes the per-environment tables of
// lm_set_model_variants and nothing else (lowering.variant_tables): the inertial record (per link mass / centre of mass / inertia
// tensor, armature, dof_invweight0, friction-loss regulariser; the solver's scale 1 / (meaninertia nv)) and the contact constants of
// the geom table and the geom-pair table (friction, and the regulariser's body_invweight0 sums). This kernel writes them for one
// environment from that environment's own draws, in float64, following the host compiler line by line:
//   draws                      utils/domain_randomization.py JointRandomization.model_draw_ops (the reference's rules, :386-514)
//   inertial numbers           mjcf.inertia_from_spec (the engine compiler's boundmass / boundinertia / balanceinertia)
//   M(qpos0), its inverse      mjcf.mass_matrix, mjcf._set_const (dof_invweight0, body_invweight0, meaninertia)
//   record / contact constants lowering.lower: merged_inertial, fill_dof, _fill_contact_params, the pair records
// The constants of the program (Jacobians at qpos0, the bodies without a rule summed once, where every value lands) come from
// lowering.model_compiler_tables. One workgroup of 64 lanes per environment that needs a model; a launch where nobody restarted
// costs one byte read per workgroup. Not on the step's critical path: restarts are rare (one per episode).
#include "lm_compile.h"
#include "../../include/lm_layout.h"

namespace lmc {
namespace {

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// eigen decomposition of a symmetric 3x3 (cyclic Jacobi): a -> diag(ev), columns of v = eigenvectors
__device__ void jacobi3(double a[3][3], double ev[3], double v[3][3]) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) v[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 32; sweep++) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    const double dia = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
    if (off <= 1e-60 + 1e-34 * dia) break;
    for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
      if (a[p][q] == 0.0) continue;
      const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
      for (int k = 0; k < 3; k++) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
      for (int k = 0; k < 3; k++) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
      for (int k = 0; k < 3; k++) { const double vkp = v[k][p], vkq = v[k][q]; v[k][p] = c * vkp - s * vkq; v[k][q] = s * vkp + c * vkq; }
    }
  }
  for (int i = 0; i < 3; i++) ev[i] = a[i][i];
}

// r a r^T for a symmetric a given as (xx yy zz xy xz yz), r row-major 3x3; result in the same six-number form
__device__ void rotate_sym(const double* r, const double* a6, double* out6) {
  const double a[3][3] = {{a6[0], a6[3], a6[4]}, {a6[3], a6[1], a6[5]}, {a6[4], a6[5], a6[2]}};
  double t[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[i][j] = r[i * 3 + 0] * a[0][j] + r[i * 3 + 1] * a[1][j] + r[i * 3 + 2] * a[2][j];
  auto e = [&](int i, int j) { return t[i][0] * r[j * 3 + 0] + t[i][1] * r[j * 3 + 1] + t[i][2] * r[j * 3 + 2]; };
  out6[0] = e(0, 0); out6[1] = e(1, 1); out6[2] = e(2, 2); out6[3] = e(0, 1); out6[4] = e(0, 2); out6[5] = e(1, 2);
}

__global__ __launch_bounds__(64) void compile_models_kernel(Args a) {
  const int e = blockIdx.x, t = threadIdx.x;
  if (e >= a.N) return;
  const bool want = a.all || (a.mask ? a.mask[e] != 0 : a.dirty[e] != 0);
  if (!want) return;
  const int* ih = a.ib;
  const double* dh = a.db;
  const int nv = ih[1], nrb = ih[2], ngs = ih[3], nd = ih[4], nslot = ih[5], nrec = ih[6], ncon = ih[7], nbody = ih[8];
  const bool pyramidal = ih[9] != 0, balance = ih[10] != 0;
  const double impratio = dh[0], boundmass = dh[1], boundinertia = dh[2], minval = dh[3];
  const int* i_draw = ih + kIntHead;
  const int* i_rb = i_draw + nd * kDrawInts;
  const int* i_rec = i_rb + nrb * kRbInts;
  const int* i_con = i_rec + nrec * 2;
  const double* d_draw = dh + kDblHead;
  const double* d_rb = d_draw + nd * kDrawDbls;
  const double* d_jac = d_rb + nrb * kRbDbls;
  const double* d_mbase = d_jac + (long long)nbody * 6 * nv;
  const double* d_arm = d_mbase + nv * nv;
  const double* d_fd = d_arm + nv;
  const double* d_slot = d_fd + nv;
  const double* d_fric = d_slot + nslot * 10;

  __shared__ double M[kMaxNv * kMaxNv], Mi[kMaxNv * kMaxNv];
  __shared__ double drawn[kMaxDraw];
  __shared__ double arm[kMaxNv], rb_mass[kMaxRbody], rb_vals[kMaxRbody * 6], rb_sv[kMaxRbody * 3];
  __shared__ double rb_in[kMaxRbody * 6], rb_iw[kMaxRbody * 6];       // inertia tensor in the body frame / in the world at qpos0
  __shared__ double fric[kMaxGslot * 3], biw[kMaxBody];
  __shared__ double V[3 * kMaxNv + 1 + kMaxSlot * 10];
  __shared__ unsigned gen_s;

  if (t == 0) { gen_s = a.gen[e] + 1u; a.gen[e] = gen_s; a.dirty[e] = 0; }
  __syncthreads();
  const unsigned long long gid = (unsigned long long)(a.env_offset + e);
  const unsigned long long key = a.seed ^ mix64(gid * 2ull + 1ull) ^ ((unsigned long long)gen_s << 32) ^ 0x6A09E667F3BCC909ull;

  // ---- the draws (reference utils/domain_randomization.py:299-383,386-514: uniform / normal, the normals clipped at 0)
  for (int i = t; i < nd; i += 64) {
    const int kind = i_draw[i * kDrawInts];
    const double pa = d_draw[i * 2], pb = d_draw[i * 2 + 1];
    const unsigned long long r1 = mix64(key ^ (unsigned long long)(2 * i + 1) * 0xD6E8FEB86659FD93ull);
    const unsigned long long r2 = mix64(key ^ (unsigned long long)(2 * i + 2) * 0xD6E8FEB86659FD93ull);
    const double u1 = ((double)(r1 >> 11) + 0.5) * (1.0 / 9007199254740992.0), u2 = (double)(r2 >> 11) * (1.0 / 9007199254740992.0);
    double v;
    if (kind == 2) v = pa + (pb - pa) * u1;
    else v = fmax(pa + pb * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2), 0.0);
    drawn[i] = v;
    a.draws[(long long)e * nd + i] = v;
  }
  for (int i = t; i < nv; i += 64) arm[i] = d_arm[i];
  for (int j = t; j < nrb; j += 64) {
    rb_mass[j] = d_rb[j * kRbDbls];
    for (int k = 0; k < 6; k++) rb_vals[j * 6 + k] = d_rb[j * kRbDbls + 1 + k];
    for (int k = 0; k < 3; k++) rb_sv[j * 3 + k] = 0.0;
  }
  for (int i = t; i < ngs * 3; i += 64) fric[i] = d_fric[i];
  __syncthreads();
  if (t == 0)
    for (int i = 0; i < nd; i++) {
      const int target = i_draw[i * kDrawInts + 1], idx = i_draw[i * kDrawInts + 2], comp = i_draw[i * kDrawInts + 3];
      const double v = drawn[i];
      if (target == 0) arm[idx] = v;
      else if (target == 1) rb_mass[idx] = v;
      else if (target == 2) { rb_vals[idx * 6 + comp] = v; if (comp == 0) { rb_vals[idx * 6 + 3] = 0.0; rb_vals[idx * 6 + 4] = 0.0; rb_vals[idx * 6 + 5] = 0.0; } }
      else if (target == 3) rb_sv[idx * 3 + comp] = v;
      else fric[idx * 3 + comp] = v;
    }
  __syncthreads();

  // ---- inertial numbers of the drawn bodies (mjcf.inertia_from_spec)
  for (int j = t; j < nrb; j += 64) {
    const double* rd = d_rb + j * kRbDbls;
    const int kind = i_rb[j * kRbInts + 1], has_sv = i_rb[j * kRbInts + 3];
    double vals[6];
    for (int k = 0; k < 6; k++) vals[k] = rb_vals[j * 6 + k];
    if (has_sv) {
      // fullinertia rule (:500-513): the singular values of the upper-triangular matrix redrawn, the six numbers read back from U diag(s') V^T
      const double* U = rd + 16; const double* Vt = rd + 25;
      auto el = [&](int r, int c) { return U[r * 3 + 0] * rb_sv[j * 3 + 0] * Vt[0 * 3 + c] + U[r * 3 + 1] * rb_sv[j * 3 + 1] * Vt[1 * 3 + c] + U[r * 3 + 2] * rb_sv[j * 3 + 2] * Vt[2 * 3 + c]; };
      vals[0] = el(0, 0); vals[1] = el(1, 1); vals[2] = el(2, 2); vals[3] = el(0, 1); vals[4] = el(0, 2); vals[5] = el(1, 2);
    }
    double in6[6];
    if (kind == 2) for (int k = 0; k < 6; k++) in6[k] = vals[k];
    else { const double d6[6] = {vals[0], vals[1], vals[2], 0.0, 0.0, 0.0}; rotate_sym(rd + 7, d6, in6); }
    if (boundinertia > 0.0 || balance) {
      // the compiler's bounds act on the principal moments: lower bound first, then the triangle-inequality repair
      double am[3][3] = {{in6[0], in6[3], in6[4]}, {in6[3], in6[1], in6[5]}, {in6[4], in6[5], in6[2]}}, ev[3], vec[3][3];
      jacobi3(am, ev, vec);
      if (boundinertia > 0.0) for (int k = 0; k < 3; k++) ev[k] = fmax(ev[k], boundinertia);
      const double big = fmax(ev[0], fmax(ev[1], ev[2])), sum = ev[0] + ev[1] + ev[2];
      if (balance && (sum - big < big)) ev[0] = ev[1] = ev[2] = sum / 3.0;
      auto el = [&](int r, int c) { return vec[r][0] * ev[0] * vec[c][0] + vec[r][1] * ev[1] * vec[c][1] + vec[r][2] * ev[2] * vec[c][2]; };
      in6[0] = el(0, 0); in6[1] = el(1, 1); in6[2] = el(2, 2); in6[3] = el(0, 1); in6[4] = el(0, 2); in6[5] = el(1, 2);
    }
    if (boundmass > 0.0) rb_mass[j] = fmax(rb_mass[j], boundmass);
    for (int k = 0; k < 6; k++) rb_in[j * 6 + k] = in6[k];
    rotate_sym(rd + 46, in6, rb_iw + j * 6);                 // in the world at qpos0
  }
  __syncthreads();

  // ---- M(qpos0) = bodies without a rule + the drawn bodies + armature (mjcf.mass_matrix)
  for (int p = t; p < nv * nv; p += 64) {
    const int i = p / nv, k = p % nv;
    double s = d_mbase[p] + (i == k ? arm[i] : 0.0);
    for (int j = 0; j < nrb; j++) {
      const double* J = d_jac + (long long)i_rb[j * kRbInts] * 6 * nv;
      const double* w = rb_iw + j * 6;
      const double jp = J[0 * nv + i] * J[0 * nv + k] + J[1 * nv + i] * J[1 * nv + k] + J[2 * nv + i] * J[2 * nv + k];
      const double r0 = J[3 * nv + k], r1 = J[4 * nv + k], r2 = J[5 * nv + k];
      const double jr = J[3 * nv + i] * (w[0] * r0 + w[3] * r1 + w[4] * r2) + J[4 * nv + i] * (w[3] * r0 + w[1] * r1 + w[5] * r2) +
                        J[5 * nv + i] * (w[4] * r0 + w[5] * r1 + w[2] * r2);
      s += rb_mass[j] * jp + jr;
    }
    M[p] = s; Mi[p] = s;
  }
  __syncthreads();
  // Cholesky in place (lower triangle of Mi), one lane: 36^3 / 6 multiply-adds
  if (t == 0)
    for (int j = 0; j < nv; j++) {
      double d = Mi[j * nv + j];
      for (int k = 0; k < j; k++) d -= Mi[j * nv + k] * Mi[j * nv + k];
      d = sqrt(fmax(d, 1e-300));
      Mi[j * nv + j] = d;
      for (int i = j + 1; i < nv; i++) {
        double s = Mi[i * nv + j];
        for (int k = 0; k < j; k++) s -= Mi[i * nv + k] * Mi[j * nv + k];
        Mi[i * nv + j] = s / d;
      }
    }
  __syncthreads();
  // inverse, one column per lane, written into the UPPER part's storage of a second matrix: M is needed no more but for its trace
  double trace = 0.0;
  for (int i = 0; i < nv; i++) trace += M[i * nv + i];
  __syncthreads();
  if (t < nv) {
    double x[kMaxNv];
    for (int i = 0; i < nv; i++) {                         // L y = e_t
      double s = (i == t) ? 1.0 : 0.0;
      for (int k = 0; k < i; k++) s -= Mi[i * nv + k] * x[k];
      x[i] = s / Mi[i * nv + i];
    }
    for (int i = nv - 1; i >= 0; i--) {                    // L^T x = y
      double s = x[i];
      for (int k = i + 1; k < nv; k++) s -= Mi[k * nv + i] * x[k];
      x[i] = s / Mi[i * nv + i];
    }
    for (int i = 0; i < nv; i++) M[i * nv + t] = x[i];      // M <- M^-1
  }
  __syncthreads();

  // ---- derived constants (mjcf._set_const) and the values the record is gathered from
  const int V_ARM = 0, V_INVW = nv, V_RFL = 2 * nv, V_SCALE = 3 * nv, V_LINK = 3 * nv + 1;
  for (int i = t; i < nv; i += 64) {
    const double iw = M[i * nv + i];
    V[V_ARM + i] = arm[i]; V[V_INVW + i] = iw;
    V[V_RFL + i] = d_fd[i] >= 0.0 ? fmax(minval, d_fd[i] * iw) : 0.0;
  }
  if (t == 0) V[V_SCALE] = 1.0 / ((trace / nv) * nv);
  for (int b = t; b < nbody; b += 64) {
    const double* J = d_jac + (long long)b * 6 * nv;
    double acc = 0.0;
    for (int r = 0; r < 3; r++)
      for (int i = 0; i < nv; i++) {
        const double ji = J[r * nv + i];
        if (ji == 0.0) continue;
        double s = 0.0;
        for (int k = 0; k < nv; k++) s += M[i * nv + k] * J[r * nv + k];
        acc += ji * s;
      }
    biw[b] = acc / 3.0;
  }
  // inertial slots: the bodies without a rule (summed on the host) + the drawn bodies, then mass / centre of mass / inertia about it
  for (int s = t; s < nslot; s += 64) {
    double acc[10];
    for (int k = 0; k < 10; k++) acc[k] = d_slot[s * 10 + k];
    for (int j = 0; j < nrb; j++) {
      if (i_rb[j * kRbInts + 2] != s) continue;
      const double* rd = d_rb + j * kRbDbls;
      const double m = rb_mass[j], cx = rd[34], cy = rd[35], cz = rd[36];
      double io[6];
      rotate_sym(rd + 37, rb_in + j * 6, io);
      const double cc = cx * cx + cy * cy + cz * cz;
      acc[0] += m; acc[1] += m * cx; acc[2] += m * cy; acc[3] += m * cz;
      acc[4] += io[0] + m * (cc - cx * cx); acc[5] += io[1] + m * (cc - cy * cy); acc[6] += io[2] + m * (cc - cz * cz);
      acc[7] += io[3] - m * cx * cy; acc[8] += io[4] - m * cx * cz; acc[9] += io[5] - m * cy * cz;
    }
    double* o = V + V_LINK + s * 10;
    if (acc[0] <= 0.0) { for (int k = 0; k < 10; k++) o[k] = 0.0; continue; }
    const double m = acc[0], cx = acc[1] / m, cy = acc[2] / m, cz = acc[3] / m, cc = cx * cx + cy * cy + cz * cz;
    o[0] = m; o[1] = cx; o[2] = cy; o[3] = cz;
    o[4] = acc[4] - m * (cc - cx * cx); o[5] = acc[5] - m * (cc - cy * cy); o[6] = acc[6] - m * (cc - cz * cz);
    o[7] = acc[7] + m * cx * cy; o[8] = acc[8] + m * cx * cz; o[9] = acc[9] + m * cy * cz;
  }
  __syncthreads();

  // ---- the environment's inertial record
  float* rec = a.vrec + (long long)e * (LM_IR_SIZE * LM_NCHAIN);
  for (int i = t; i < nrec; i += 64) rec[i_rec[2 * i]] = (float)V[i_rec[2 * i + 1]];
  // ---- contact constants of its geom table and geom-pair table (lowering._fill_contact_params and the pair records)
  float* gt = a.vgt + (long long)e * LM_GT_SIZE;
  float* gpt = a.vgpt ? a.vgpt + (long long)e * a.gpt_floats : nullptr;
  for (int i = t; i < ncon; i += 64) {
    const int* op = i_con + i * kConInts;
    float* tab = op[0] == 0 ? gt : gpt;
    if (!tab) continue;
    const int at = op[1], st = op[2], dim = op[3] & 15, mix = (op[3] >> 4) & 15, pair = (op[3] >> 8) & 1;
    const double* fa = fric + op[4] * 3; const double* fb = fric + op[5] * 3;
    double f3[3];
    for (int k = 0; k < 3; k++) f3[k] = fmax(1e-5, mix == 0 ? fmax(fa[k], fb[k]) : (mix == 1 ? fa[k] : fb[k]));      // (mjMINMU: a friction drawn at exactly 0
                                                                                                                  // would make rr = 0 / 0 below; lowering.py clamps alike)
    const double fr[5] = {f3[0], f3[0], f3[1], f3[2], f3[2]};
    const double tran = biw[op[6]] + biw[op[7]];
    double vt, vmu, rr[5];
    if (pyramidal) {
      const double mu = (pair && dim != 3) ? 0.0 : fr[0];
      vmu = mu;
      vt = dim == 3 ? 2.0 * mu * mu * (1.0 + mu * mu) * tran : (pair ? 4.0 * tran : tran);
      for (int k = 0; k < 5; k++) rr[k] = 1.0;
    } else {
      vt = tran;
      vmu = fr[0] / sqrt(fmax(minval, impratio));
      const double rr1 = 1.0 / fmax(minval, impratio);
      rr[0] = rr1; rr[1] = rr1 * fr[0] * fr[0] / (fr[1] * fr[1]);
      for (int k = 2; k < 5; k++) rr[k] = rr1 * fr[0] * fr[0] / (fr[k] * fr[k]);
    }
    tab[at] = (float)vt; tab[at + 2 * st] = (float)vmu;
    for (int k = 0; k < 5; k++) { tab[at + (3 + k) * st] = (float)fr[k]; tab[at + (8 + k) * st] = (float)rr[k]; }
  }
  if (a.slack && t < 12) a.slack[(long long)t * a.N + e] = 0.0f;
}

__global__ void replicate_kernel(float* dst, const float* src, long long n_per, int N) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_per) return;
  const float v = src[i];
  for (int e = blockIdx.y; e < N; e += gridDim.y) dst[(long long)e * n_per + i] = v;
}

__global__ void iota_kernel(int* dst, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) dst[i] = i;
}

}  // namespace

void launch(const Args& a, hipStream_t stream) { hipLaunchKernelGGL(compile_models_kernel, dim3(a.N), dim3(64), 0, stream, a); }

void replicate(float* dst, const float* src, long long n_per, int N, hipStream_t stream) {
  if (n_per <= 0 || N <= 0) return;
  hipLaunchKernelGGL(replicate_kernel, dim3((unsigned)((n_per + 255) / 256), N < 64 ? N : 64), dim3(256), 0, stream, dst, src, n_per, N);
}

void iota(int* dst, int N, hipStream_t stream) { hipLaunchKernelGGL(iota_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, dst, N); }

}  // namespace lmc
