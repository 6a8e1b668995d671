/*
 * apd_oracle.c -- CPU restatement of Go-RIO's APD-GICP scan matching (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the parity oracle for the HIP path.  It is NOT part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product library
 * (go-rio_amd/csrc) never links, includes or calls anything in oracle/.
 *
 * It restates, in plain C (no Eigen / PCL / FLANN, which are absent from the build image), the algorithm in
 *   APD  = /root/reference/fast_apdgicp/include/fast_gicp/gicp/impl/fast_apdgicp_impl.hpp
 *   LSQ  = /root/reference/fast_apdgicp/include/fast_gicp/gicp/impl/lsq_registration_impl.hpp
 *   SO3  = /root/reference/fast_apdgicp/include/fast_gicp/so3/so3.hpp
 * Each function cites the lines it follows.
 *
 * PARITY PINNING: the reference ships no test, golden vector or fixture that instantiates FastAPDGICP
 *  and it cannot be compiled here (Eigen, PCL, FLANN missing).  The oracle is
 * therefore pinned by (i) an independent NumPy/SciPy restatement of the same lines (oracle/apd_numpy.py,
 * different code, same maths), (ii) known-transform recovery in the acceptance shape of the reference's
 * gicp_test.cpp (0.05 m / 1 deg, forward / backward / swap), and (iii) analytic identities (identical clouds at
 * identity => b = 0, correspondences = identity permutation).  Third-party arithmetic (PCL 1.10 kd-tree,
 * Eigen 3.3.7 JacobiSVD / inverse / LDLT) is restated from its published algorithm: "parity unpinned" for
 * those pieces, as recorded in DESIGN.md.
 *
 * Third-party behaviour restated here:
 *  - pcl::search::KdTree::nearestKSearch == exact k-NN under FLANN L2_Simple<float>:
 *        d = ((dx*dx) + dy*dy) + dz*dz   in float, no FMA (reference builds with -msse4.2 only).
 *    Ties are resolved to the LOWEST index (FLANN's order is tree-dependent; we define it).
 *  - Eigen::Isometry3f * Vector4f == coefficient product accumulated k = 0..3:
 *        q_r = ((m_r0*x + m_r1*y) + m_r2*z) + m_r3*1   in float, no FMA.
 *  - Eigen::JacobiSVD<Matrix3d> of a symmetric PSD matrix == symmetric eigen-decomposition with U == V,
 *    singular values sorted descending (cyclic Jacobi here).
 *  - Eigen::Matrix4d::inverse == adjugate / determinant (cofactor expansion).
 *  - Eigen::LDLT<6x6> == LDL^T with symmetric diagonal pivoting.
 *  - atan2(float,float) / sqrt(float) under `using namespace std` (APD:7, 198-199) are the float overloads.
 *    They are restated as the correctly rounded float results (float)atan2((double)y,(double)x) and
 *    (float)sqrt((double)x): glibc's own atan2f is within 1 ulp of that but is not the same function on every
 *    libm, and the GPU has no glibc; the correctly rounded value is the one definition both sides can meet.
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -msse4.2 -ffp-contract=off, the reference's flags
 * fast_apdgicp/CMakeLists.txt:11-16 plus contraction explicitly off).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_PI_2
#define M_PI_2 1.57079632679489661923
#endif

/* enum order of fast_gicp::RegularizationMethod, gicp_settings.hpp:6 */
enum { APDO_REG_NONE = 0, APDO_REG_MIN_EIG = 1, APDO_REG_NORMALIZED_MIN_EIG = 2, APDO_REG_PLANE = 3, APDO_REG_FROBENIUS = 4 };
/* enum order of fast_gicp::LSQ_OPTIMIZER_TYPE, lsq_registration.hpp:13 */
enum { APDO_OPT_GN = 0, APDO_OPT_LM = 1 };

typedef struct {
  int k_correspondences;          /* APD:21  (20) */
  int regularization;             /* APD:25  (PLANE) */
  double dist_var;                /* APDH:118 (0.86) */
  double azimuth_var;             /* APDH:116 (0.5) */
  double elevation_var;           /* APDH:117 (1.0) */
  double corr_dist_threshold;     /* APD:23 float max; launch 2.0 */
  int max_iterations;             /* LSQ:13 (64) */
  double rotation_epsilon;        /* LSQ:14 (2e-3) */
  double transformation_epsilon;  /* LSQ:15 (5e-4); launch 0.1 */
  int optimizer;                  /* LSQ:17 (LM) */
  int lm_max_iterations;          /* LSQ:19 (10) */
  double lm_init_lambda_factor;   /* LSQ:20 (1e-9) */
  int num_threads;                /* APD:34-42; 0 => omp max */
  int search;                     /* 0 = exhaustive scan (bit-exact index oracle), 1 = exact kd-tree (what pcl::search::KdTree does, APD:178, 364) */
} apdo_params;

void apdo_default_params(apdo_params* p) {
  p->k_correspondences = 20;
  p->regularization = APDO_REG_PLANE;
  p->dist_var = 0.86;
  p->azimuth_var = 0.5;
  p->elevation_var = 1.0;
  p->corr_dist_threshold = (double)FLT_MAX;
  p->max_iterations = 64;
  p->rotation_epsilon = 2e-3;
  p->transformation_epsilon = 5e-4;
  p->optimizer = APDO_OPT_LM;
  p->lm_max_iterations = 10;
  p->lm_init_lambda_factor = 1e-9;
  p->num_threads = 0;
  p->search = 0;
}

static int nthreads(const apdo_params* p) {
#ifdef _OPENMP
  return p->num_threads > 0 ? p->num_threads : omp_get_max_threads();
#else
  (void)p;
  return 1;
#endif
}

/* ---------------------------------------------------------------- small dense helpers (row-major) */

static void mat3_mul(const double* A, const double* B, double* C) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
      t[i * 3 + j] = s;
    }
  memcpy(C, t, sizeof(t));
}

static void mat4_mul(const double* A, const double* B, double* C) {
  double t[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0.0;
      for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j];
      t[i * 4 + j] = s;
    }
  memcpy(C, t, sizeof(t));
}

static void mat4_transpose(const double* A, double* At) {
  double t[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) t[j * 4 + i] = A[i * 4 + j];
  memcpy(At, t, sizeof(t));
}

/* Matrix4d::inverse (APD:217): adjugate / determinant by cofactor expansion. Returns 0 when singular. */
static int mat4_inverse(const double* m, double* inv_out) {
  double inv[16];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  if (det == 0.0) return 0;
  double r = 1.0 / det;
  for (int i = 0; i < 16; i++) inv_out[i] = inv[i] * r;
  return 1;
}

static void mat3_inverse(const double* m, double* out) {
  double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  double r = 1.0 / det;
  double t[9];
  t[0] = c00 * r;
  t[1] = (m[2] * m[7] - m[1] * m[8]) * r;
  t[2] = (m[1] * m[5] - m[2] * m[4]) * r;
  t[3] = c01 * r;
  t[4] = (m[0] * m[8] - m[2] * m[6]) * r;
  t[5] = (m[2] * m[3] - m[0] * m[5]) * r;
  t[6] = c02 * r;
  t[7] = (m[1] * m[6] - m[0] * m[7]) * r;
  t[8] = (m[0] * m[4] - m[1] * m[3]) * r;
  memcpy(out, t, sizeof(t));
}

/*
 * Symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations; eigenvalues returned sorted DESCENDING with
 * matching eigenvector columns in V (row-major 3x3, column j = j-th eigenvector).  Stands in for
 * Eigen::JacobiSVD<Matrix3d>(sym PSD, FullU|FullV) at APD:266 and APD:385: for a symmetric PSD matrix the SVD has
 * U == V == eigenvectors and singular values == eigenvalues (sorted descending by Eigen).
 */
static void sym3_eigen(const double* A_in, double* evals, double* V) {
  double A[9];
  memcpy(A, A_in, sizeof(A));
  for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    double diag = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-32 * diag || off == 0.0) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double apq = A[p * 3 + q];
        if (apq == 0.0) continue;
        double app = A[p * 3 + p], aqq = A[q * 3 + q];
        double theta = (aqq - app) / (2.0 * apq);
        double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        /* A <- J^T A J with J = rotation in the (p,q) plane */
        for (int k = 0; k < 3; k++) {
          double akp = A[k * 3 + p], akq = A[k * 3 + q];
          A[k * 3 + p] = c * akp - s * akq;
          A[k * 3 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {
          double apk = A[p * 3 + k], aqk = A[q * 3 + k];
          A[p * 3 + k] = c * apk - s * aqk;
          A[q * 3 + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
          V[k * 3 + p] = c * vkp - s * vkq;
          V[k * 3 + q] = s * vkp + c * vkq;
        }
      }
  }
  double w[3] = {A[0], A[4], A[8]};
  int order[3] = {0, 1, 2};
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2 - i; j++)
      if (w[order[j]] < w[order[j + 1]]) {
        int t = order[j];
        order[j] = order[j + 1];
        order[j + 1] = t;
      }
  double Vs[9];
  for (int j = 0; j < 3; j++) {
    evals[j] = w[order[j]];
    for (int k = 0; k < 3; k++) Vs[k * 3 + j] = V[k * 3 + order[j]];
  }
  memcpy(V, Vs, sizeof(Vs));
}

/* geo_weight of APD:266-269 / APD:330-333: sigma_3 / sigma_1 of the (regularised) source covariance. */
static double geo_weight_of(const double* cov4) {
  double C[9] = {cov4[0], cov4[1], cov4[2], cov4[4], cov4[5], cov4[6], cov4[8], cov4[9], cov4[10]};
  double w[3], V[9];
  sym3_eigen(C, w, V);
  double s0 = fabs(w[0]), s1 = fabs(w[1]), s2 = fabs(w[2]);
  double smax = fmax(s0, fmax(s1, s2)), smin = fmin(s0, fmin(s1, s2));
  return smin / smax;
}

/* ---------------------------------------------------------------- exact float metric (FLANN L2_Simple) */

static inline float sqdist3f(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  float r = dx * dx;
  r = r + dy * dy;
  r = r + dz * dz;
  return r;
}

/* ---------------------------------------------------------------- exact kd-tree (the reference's search structure)
 * pcl::search::KdTree wraps FLANN's KDTreeSingleIndex: an EXACT kd-tree (eps = 0) over the float xyz, L2_Simple metric.  This is a
 * plain restatement of that idea (median split on the widest axis, leaves of <= 15 points, branch-and-bound descent); candidate
 * distances use the same float expression as the exhaustive scan and ties are resolved on (distance, index), so both searches of
 * this file return identical results -- the tree only makes the CPU baseline algorithmically faithful to the reference. */
typedef struct {
  int left, right;   /* children, -1 for leaves */
  int begin, end;    /* point range in `order` (leaves) */
  int axis;
  float split;
  float lo[3], hi[3];
} kd_node;
typedef struct {
  const float* xyz;
  int n;
  int* order;
  kd_node* nodes;
  int n_nodes, cap_nodes;
} kd_tree;

static int kd_build_rec(kd_tree* t, int begin, int end) {
  if (t->n_nodes == t->cap_nodes) {
    t->cap_nodes = t->cap_nodes * 2 + 64;
    t->nodes = (kd_node*)realloc(t->nodes, (size_t)t->cap_nodes * sizeof(kd_node));
  }
  const int id = t->n_nodes++;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int q = begin; q < end; q++) {
    const float* p = t->xyz + 3 * (size_t)t->order[q];
    for (int a = 0; a < 3; a++) {
      if (p[a] < lo[a]) lo[a] = p[a];
      if (p[a] > hi[a]) hi[a] = p[a];
    }
  }
  kd_node nd;
  nd.left = nd.right = -1;
  nd.begin = begin;
  nd.end = end;
  nd.axis = 0;
  nd.split = 0;
  memcpy(nd.lo, lo, sizeof(lo));
  memcpy(nd.hi, hi, sizeof(hi));
  if (end - begin > 15) {
    int ax = 0;
    for (int a = 1; a < 3; a++)
      if (hi[a] - lo[a] > hi[ax] - lo[ax]) ax = a;
    if (hi[ax] > lo[ax]) {
      /* median split by nth_element (quickselect) on the chosen axis */
      int l = begin, r = end - 1;
      const int mid = (begin + end) / 2;
      while (l < r) {
        const float pv = t->xyz[3 * (size_t)t->order[(l + r) / 2] + ax];
        int i = l, j = r;
        while (i <= j) {
          while (t->xyz[3 * (size_t)t->order[i] + ax] < pv) i++;
          while (t->xyz[3 * (size_t)t->order[j] + ax] > pv) j--;
          if (i <= j) {
            int tmp = t->order[i];
            t->order[i] = t->order[j];
            t->order[j] = tmp;
            i++;
            j--;
          }
        }
        if (j < mid) l = i;
        if (mid < i) r = j;
      }
      nd.axis = ax;
      nd.split = t->xyz[3 * (size_t)t->order[mid] + ax];
      t->nodes[id] = nd;
      const int L = kd_build_rec(t, begin, mid);
      const int R = kd_build_rec(t, mid, end);
      t->nodes[id].left = L;
      t->nodes[id].right = R;
      return id;
    }
  }
  t->nodes[id] = nd;
  return id;
}

static kd_tree* kd_build(const float* xyz, int n) {
  kd_tree* t = (kd_tree*)calloc(1, sizeof(kd_tree));
  t->xyz = xyz;
  t->n = n;
  t->order = (int*)malloc((size_t)n * sizeof(int));
  for (int i = 0; i < n; i++) t->order[i] = i;
  kd_build_rec(t, 0, n);
  return t;
}
static void kd_free(kd_tree* t) {
  if (!t) return;
  free(t->order);
  free(t->nodes);
  free(t);
}

/* lower bound of sqdist3(q, p) for p in the node's box: same un-fused float expression on the clamped differences => monotone */
static inline float kd_box_bound(const kd_node* nd, const float* q) {
  float d[3];
  for (int a = 0; a < 3; a++) {
    float c = q[a] < nd->lo[a] ? nd->lo[a] : (q[a] > nd->hi[a] ? nd->hi[a] : q[a]);
    d[a] = q[a] - c;
  }
  float r = d[0] * d[0];
  r = r + d[1] * d[1];
  r = r + d[2] * d[2];
  return r;
}

/* k best (distance, index) pairs, ascending, lexicographic ties */
static void kd_search(const kd_tree* t, int node, const float* q, int k, float* bd, int* bi, int* cnt) {
  const kd_node* nd = &t->nodes[node];
  if (*cnt == k && kd_box_bound(nd, q) > bd[k - 1]) return;
  if (nd->left < 0) {
    for (int s = nd->begin; s < nd->end; s++) {
      const int j = t->order[s];
      const float d = sqdist3f(q[0], q[1], q[2], t->xyz[3 * (size_t)j], t->xyz[3 * (size_t)j + 1], t->xyz[3 * (size_t)j + 2]);
      if (*cnt == k && !(d < bd[k - 1] || (d == bd[k - 1] && j < bi[k - 1]))) continue;
      int pos = *cnt < k ? *cnt : k - 1;
      while (pos > 0 && (d < bd[pos - 1] || (d == bd[pos - 1] && j < bi[pos - 1]))) {
        bd[pos] = bd[pos - 1];
        bi[pos] = bi[pos - 1];
        pos--;
      }
      bd[pos] = d;
      bi[pos] = j;
      if (*cnt < k) (*cnt)++;
    }
    return;
  }
  const int first = q[nd->axis] < nd->split ? nd->left : nd->right;
  const int second = first == nd->left ? nd->right : nd->left;
  kd_search(t, first, q, k, bd, bi, cnt);
  kd_search(t, second, q, k, bd, bi, cnt);
}

static kd_tree* g_align_tree = NULL; /* target tree of the align() in progress (single align at a time, like one FastAPDGICP object) */

/* self k-NN through the kd-tree; identical output to apdo_knn_self */
int apdo_knn_self_kdtree(const float* xyz, int n, int k, int* idx_out, float* sqd_out, int num_threads) {
  if (n < k || k <= 0 || k > 64) return -1;
  kd_tree* t = kd_build(xyz, n);
#ifdef _OPENMP
  int nt = num_threads > 0 ? num_threads : omp_get_max_threads();
#else
  int nt = 1;
#endif
  (void)nt;
#pragma omp parallel for num_threads(nt) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    float bd[64];
    int bi[64];
    int cnt = 0;
    kd_search(t, 0, xyz + 3 * (size_t)i, k, bd, bi, &cnt);
    for (int j = 0; j < k; j++) {
      if (idx_out) idx_out[(size_t)i * k + j] = bi[j];
      if (sqd_out) sqd_out[(size_t)i * k + j] = bd[j];
    }
  }
  kd_free(t);
  return 0;
}

/*
 * Exact brute-force k-NN of every point in its own cloud (APD:364: kdtree.nearestKSearch(cloud->at(i), k, ...)),
 * sorted ascending by (distance, index); includes the point itself.  idx_out: n*k int32, sqd_out: n*k float
 * (either may be NULL).  Requires n >= k (reference quirk APD:366-369: fewer points leave columns uninitialised).
 */
int apdo_knn_self(const float* xyz, int n, int k, int* idx_out, float* sqd_out, int num_threads) {
  if (n < k || k <= 0 || k > 64) return -1;
#ifdef _OPENMP
  int nt = num_threads > 0 ? num_threads : omp_get_max_threads();
#else
  int nt = 1;
#endif
  (void)nt;
#pragma omp parallel for num_threads(nt) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    float bd[64];
    int bi[64];
    int cnt = 0;
    const float qx = xyz[3 * i], qy = xyz[3 * i + 1], qz = xyz[3 * i + 2];
    for (int j = 0; j < n; j++) {
      float d = sqdist3f(qx, qy, qz, xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2]);
      if (cnt == k && !(d < bd[k - 1])) continue; /* j ascending: an equal distance never displaces a lower index */
      int pos = cnt < k ? cnt : k - 1;
      while (pos > 0 && d < bd[pos - 1]) {
        bd[pos] = bd[pos - 1];
        bi[pos] = bi[pos - 1];
        pos--;
      }
      bd[pos] = d;
      bi[pos] = j;
      if (cnt < k) cnt++;
    }
    for (int j = 0; j < k; j++) {
      if (idx_out) idx_out[(size_t)i * k + j] = bi[j];
      if (sqd_out) sqd_out[(size_t)i * k + j] = bd[j];
    }
  }
  return 0;
}

/*
 * calculate_covariances, APD:351-411.  cov_out: n * 16 doubles, each a row-major 4x4 (Matrix4d is symmetric here so
 * the storage order does not matter).  knn_idx: n*k neighbour indices (from apdo_knn_self).
 */
int apdo_covariances_from_knn(const float* xyz, int n, const int* knn_idx, int k, int regularization, double* cov_out, int num_threads) {
#ifdef _OPENMP
  int nt = num_threads > 0 ? num_threads : omp_get_max_threads();
#else
  int nt = 1;
#endif
  (void)nt;
  int bad = 0;
#pragma omp parallel for num_threads(nt) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    /* APD:366-372: neighbours as doubles, subtract the row mean, cov = X X^T / k (4th row/col are exactly 0) */
    double mean[3] = {0, 0, 0};
    for (int j = 0; j < k; j++) {
      const float* p = xyz + 3 * (size_t)knn_idx[(size_t)i * k + j];
      mean[0] += (double)p[0];
      mean[1] += (double)p[1];
      mean[2] += (double)p[2];
    }
    mean[0] /= (double)k;
    mean[1] /= (double)k;
    mean[2] /= (double)k;
    double C[9] = {0};
    for (int j = 0; j < k; j++) {
      const float* p = xyz + 3 * (size_t)knn_idx[(size_t)i * k + j];
      double d[3] = {(double)p[0] - mean[0], (double)p[1] - mean[1], (double)p[2] - mean[2]};
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) C[a * 3 + b] += d[a] * d[b];
    }
    for (int a = 0; a < 9; a++) C[a] /= (double)k;

    double R[9];
    if (regularization == APDO_REG_NONE) { /* APD:374-376 */
      memcpy(R, C, sizeof(R));
    } else if (regularization == APDO_REG_FROBENIUS) { /* APD:377-383 */
      double Cl[9], Ci[9], N[9];
      memcpy(Cl, C, sizeof(Cl));
      Cl[0] += 1e-3;
      Cl[4] += 1e-3;
      Cl[8] += 1e-3;
      mat3_inverse(Cl, Ci);
      double fro = 0.0;
      for (int a = 0; a < 9; a++) fro += Ci[a] * Ci[a];
      fro = sqrt(fro);
      for (int a = 0; a < 9; a++) N[a] = Ci[a] / fro;
      mat3_inverse(N, R);
    } else { /* APD:384-406 */
      double w[3], V[9], vals[3];
      sym3_eigen(C, w, V);
      double s[3] = {fabs(w[0]), fabs(w[1]), fabs(w[2])};
      if (regularization == APDO_REG_PLANE) {
        vals[0] = 1.0;
        vals[1] = 1.0;
        vals[2] = 1e-3;
      } else if (regularization == APDO_REG_MIN_EIG) {
        for (int a = 0; a < 3; a++) vals[a] = fmax(s[a], 1e-3);
      } else if (regularization == APDO_REG_NORMALIZED_MIN_EIG) {
        double smax = fmax(s[0], fmax(s[1], s[2]));
        for (int a = 0; a < 3; a++) vals[a] = fmax(s[a] / smax, 1e-3);
      } else {
#pragma omp atomic write
        bad = 1; /* APD:389-391 abort() */
        continue;
      }
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) R[a * 3 + b] = V[a * 3 + 0] * vals[0] * V[b * 3 + 0] + V[a * 3 + 1] * vals[1] * V[b * 3 + 1] + V[a * 3 + 2] * vals[2] * V[b * 3 + 2];
    }
    double* o = cov_out + (size_t)i * 16;
    memset(o, 0, 16 * sizeof(double));
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) o[a * 4 + b] = R[a * 3 + b];
  }
  return bad ? -2 : 0;
}

int apdo_calculate_covariances(const float* xyz, int n, const apdo_params* p, double* cov_out) {
  int k = p->k_correspondences;
  int* idx = (int*)malloc((size_t)n * k * sizeof(int));
  if (!idx) return -3;
  int rc = p->search == 1 ? apdo_knn_self_kdtree(xyz, n, k, idx, NULL, p->num_threads) : apdo_knn_self(xyz, n, k, idx, NULL, p->num_threads);
  if (rc == 0) rc = apdo_covariances_from_knn(xyz, n, idx, k, p->regularization, cov_out, p->num_threads);
  free(idx);
  return rc;
}

/* ---------------------------------------------------------------- correspondences + Mahalanobis (APD:160-220) */

/* trans.cast<float>() then `trans_f * p` (APD:164, 176): Eigen coefficient product, float, no FMA. */
static inline void transform_point_f(const float* Tf, float x, float y, float z, float* q) {
  for (int r = 0; r < 3; r++) {
    float a = Tf[r * 4 + 0] * x;
    a = a + Tf[r * 4 + 1] * y;
    a = a + Tf[r * 4 + 2] * z;
    a = a + Tf[r * 4 + 3]; /* * 1.0f is exact */
    q[r] = a;
  }
}

/* sensor covariance cov_r at the transformed point, APD:194-210 */
static void sensor_cov(const apdo_params* p, const float* q, double* cov_r) {
  double px = (double)q[0], py = (double)q[1], pz = (double)q[2];
  double dist = sqrt(px * px + py * py + pz * pz);                    /* APD:194 */
  double s_x = dist * p->dist_var / 400;                              /* APD:195 */
  double s_y = dist * sin(p->azimuth_var / 180 * M_PI);               /* APD:196 */
  double s_z = dist * sin(p->elevation_var / 180 * M_PI);             /* APD:197 */
  float rxy = (float)sqrt((double)(q[0] * q[0] + q[1] * q[1]));       /* sqrt(float): float overload, argument evaluated in float */
  double elevation = (double)(float)atan2((double)rxy, (double)q[2]); /* APD:198 atan2(float,float) */
  double azimuth = (double)(float)atan2((double)q[1], (double)q[0]);  /* APD:199 */
  double ce = cos(elevation), se = sin(elevation), ca = cos(azimuth), sa = sin(azimuth);
  /* R = yaw(Z, azimuth) * pitch(Y, elevation), APD:200-203 */
  double Rz[9] = {ca, -sa, 0, sa, ca, 0, 0, 0, 1};
  double Ry[9] = {ce, 0, se, 0, 1, 0, -se, 0, ce};
  double R[9], A[9];
  mat3_mul(Rz, Ry, R);
  double S[9] = {s_x, 0, 0, 0, s_y, 0, 0, 0, s_z}; /* APD:204-205 */
  mat3_mul(R, S, A);                               /* APD:207 */
  for (int a = 0; a < 3; a++)                      /* APD:208 cov_r = A A^T */
    for (int b = 0; b < 3; b++) cov_r[a * 3 + b] = A[a * 3 + 0] * A[b * 3 + 0] + A[a * 3 + 1] * A[b * 3 + 1] + A[a * 3 + 2] * A[b * 3 + 2];
}

/*
 * update_correspondences, APD:160-220.  T: row-major 4x4 double (Isometry3d).  Brute-force exact 1-NN with ties to
 * the lowest index.  Outputs: corr[n] (-1 when rejected), sqd[n], maha[n*16] (row-major 4x4; untouched for rejected).
 */
int apdo_update_correspondences(const double* T, const float* src_xyz, int n, const float* tgt_xyz, int m, const double* src_cov, const double* tgt_cov, const apdo_params* p, int* corr, float* sqd, double* maha) {
  float Tf[16];
  for (int i = 0; i < 16; i++) Tf[i] = (float)T[i]; /* APD:164 */
  double Tt[16];
  mat4_transpose(T, Tt);
  int nt = nthreads(p);
  (void)nt;
  const double thr2 = p->corr_dist_threshold * p->corr_dist_threshold;
  /* the reference builds the target tree once in setInputTarget (APD:132); apdo_align does the same and lends it through g_align_tree */
  const int borrowed = (p->search == 1 && g_align_tree && g_align_tree->xyz == tgt_xyz && g_align_tree->n == m);
  kd_tree* tree = borrowed ? g_align_tree : ((p->search == 1 && m > 0) ? kd_build(tgt_xyz, m) : NULL);
#pragma omp parallel for num_threads(nt) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    float q[3];
    transform_point_f(Tf, src_xyz[3 * i], src_xyz[3 * i + 1], src_xyz[3 * i + 2], q); /* APD:176 */
    float best = INFINITY;
    int bj = -1;
    if (tree) {
      int cnt = 0;
      kd_search(tree, 0, q, 1, &best, &bj, &cnt);
    } else {
      for (int j = 0; j < m; j++) { /* APD:178 */
        float d = sqdist3f(q[0], q[1], q[2], tgt_xyz[3 * j], tgt_xyz[3 * j + 1], tgt_xyz[3 * j + 2]);
        if (d < best) {
          best = d;
          bj = j;
        }
      }
    }
    sqd[i] = best;                                  /* APD:180 */
    corr[i] = ((double)best < thr2) ? bj : -1;      /* APD:183 */
    if (corr[i] < 0) continue;                      /* APD:185-187 */
    const double* cA = src_cov + (size_t)i * 16;
    const double* cB = tgt_cov + (size_t)corr[i] * 16;
    double cr[9];
    sensor_cov(p, q, cr);
    double A4[16], B4[16];
    for (int a = 0; a < 16; a++) {
      A4[a] = cA[a];
      B4[a] = cB[a];
    }
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) { /* cov + cov_dist, APD:209-214 */
        A4[a * 4 + b] += cr[a * 3 + b];
        B4[a * 4 + b] += cr[a * 3 + b];
      }
    double TA[16], TAT[16], RCR[16];
    mat4_mul(T, A4, TA);
    mat4_mul(TA, Tt, TAT);
    for (int a = 0; a < 16; a++) RCR[a] = B4[a] + TAT[a]; /* APD:214 */
    RCR[15] = 1.0;                                         /* APD:215 */
    double* M = maha + (size_t)i * 16;
    if (!mat4_inverse(RCR, M)) {
      for (int a = 0; a < 16; a++) M[a] = NAN;
    }
    M[15] = 0.0; /* APD:218 */
  }
  if (!borrowed) kd_free(tree);
  return 0;
}

/* error term shared by linearize / compute_error: APD:255-276 and APD:320-341 */
static inline double point_error(const double* T, const float* a, const float* b, const double* M, double weight, double* Ta_out, double* e_out) {
  double A4[4] = {(double)a[0], (double)a[1], (double)a[2], 1.0};
  double B4[4] = {(double)b[0], (double)b[1], (double)b[2], 1.0};
  double Ta[4], e[4];
  for (int r = 0; r < 4; r++) Ta[r] = T[r * 4 + 0] * A4[0] + T[r * 4 + 1] * A4[1] + T[r * 4 + 2] * A4[2] + T[r * 4 + 3] * A4[3];
  for (int r = 0; r < 4; r++) e[r] = B4[r] - Ta[r];
  double q = 0.0;
  for (int r = 0; r < 4; r++) {
    double s = 0.0;
    for (int c = 0; c < 4; c++) s += M[r * 4 + c] * e[c];
    q += e[r] * s;
  }
  if (Ta_out) memcpy(Ta_out, Ta, sizeof(Ta));
  if (e_out) memcpy(e_out, e, sizeof(e));
  return weight * q;
}

/*
 * linearize, APD:224-307 (calls update_correspondences first, APD:226).  H: 36 doubles row-major, b: 6 doubles;
 * either may be NULL (then only the error is returned, APD:278-280).  corr / sqd / maha are the object's state.
 * geo_w[n] = per-source-point sigma3/sigma1 of the regularised covariance (recomputed by the reference in every call,
 * APD:266-269; a pure function of src_cov, precomputed by apdo_geo_weights).
 */
double apdo_linearize(const double* T, const float* src_xyz, const float* src_label, int n, const float* tgt_xyz, const float* tgt_label, int m, const double* src_cov, const double* tgt_cov, const double* geo_w, const apdo_params* p, int* corr, float* sqd, double* maha, double* H, double* b) {
  apdo_update_correspondences(T, src_xyz, n, tgt_xyz, m, src_cov, tgt_cov, p, corr, sqd, maha);
  int nt = nthreads(p);
  double* Hs = (double*)calloc((size_t)nt * 36, sizeof(double));
  double* bs = (double*)calloc((size_t)nt * 6, sizeof(double));
  double sum_errors = 0.0;
  const double cl = 1.0 / (double)n; /* 1.0 / correspondences_.size(), APD:273 */
#pragma omp parallel for num_threads(nt) reduction(+ : sum_errors) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    int j = corr[i];
    if (j < 0) continue;
    const double* M = maha + (size_t)i * 16;
    double w = 1.0 + geo_w[i] + ((tgt_label[j] == src_label[i]) ? cl : 0.0); /* APD:266-276 */
    double Ta[4], e[4];
    sum_errors += point_error(T, src_xyz + 3 * (size_t)i, tgt_xyz + 3 * (size_t)j, M, w, Ta, e);
    if (!H || !b) continue;
    /* J = [skew(Ta) | -I ; 0], APD:284-287 */
    double J[4][6] = {{0}};
    J[0][1] = -Ta[2];
    J[0][2] = Ta[1];
    J[1][0] = Ta[2];
    J[1][2] = -Ta[0];
    J[2][0] = -Ta[1];
    J[2][1] = Ta[0];
    J[0][3] = -1.0;
    J[1][4] = -1.0;
    J[2][5] = -1.0;
    double MJ[4][6], Me[4];
    for (int r = 0; r < 4; r++) {
      for (int c = 0; c < 6; c++) {
        double s = 0.0;
        for (int kk = 0; kk < 4; kk++) s += M[r * 4 + kk] * J[kk][c];
        MJ[r][c] = s;
      }
      double s = 0.0;
      for (int kk = 0; kk < 4; kk++) s += M[r * 4 + kk] * e[kk];
      Me[r] = s;
    }
#ifdef _OPENMP
    int tid = omp_get_thread_num();
#else
    int tid = 0;
#endif
    double* Ht = Hs + (size_t)tid * 36;
    double* bt = bs + (size_t)tid * 6;
    for (int r = 0; r < 6; r++) { /* APD:289-293 (H, b NOT weighted) */
      for (int c = 0; c < 6; c++) {
        double s = 0.0;
        for (int kk = 0; kk < 4; kk++) s += J[kk][r] * MJ[kk][c];
        Ht[r * 6 + c] += s;
      }
      double s = 0.0;
      for (int kk = 0; kk < 4; kk++) s += J[kk][r] * Me[kk];
      bt[r] += s;
    }
  }
  if (H && b) { /* APD:297-304 */
    memset(H, 0, 36 * sizeof(double));
    memset(b, 0, 6 * sizeof(double));
    for (int t = 0; t < nt; t++) {
      for (int a = 0; a < 36; a++) H[a] += Hs[(size_t)t * 36 + a];
      for (int a = 0; a < 6; a++) b[a] += bs[(size_t)t * 6 + a];
    }
  }
  free(Hs);
  free(bs);
  return sum_errors;
}

/* compute_error, APD:310-346: stale correspondences and Mahalanobis matrices of the last linearize. */
double apdo_compute_error(const double* T, const float* src_xyz, const float* src_label, int n, const float* tgt_xyz, const float* tgt_label, const double* geo_w, const apdo_params* p, const int* corr, const double* maha) {
  int nt = nthreads(p);
  (void)nt;
  double sum_errors = 0.0;
  const double cl = 1.0 / (double)n;
#pragma omp parallel for num_threads(nt) reduction(+ : sum_errors) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    int j = corr[i];
    if (j < 0) continue;
    double w = 1.0 + geo_w[i] + ((tgt_label[j] == src_label[i]) ? cl : 0.0);
    sum_errors += point_error(T, src_xyz + 3 * (size_t)i, tgt_xyz + 3 * (size_t)j, maha + (size_t)i * 16, w, NULL, NULL);
  }
  return sum_errors;
}

void apdo_geo_weights(const double* cov, int n, double* geo_w) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) geo_w[i] = geo_weight_of(cov + (size_t)i * 16);
}

/* ---------------------------------------------------------------- optimiser shell (LSQ) */

/* Eigen::LDLT<6x6>(A).solve(rhs): LDL^T with symmetric diagonal pivoting (largest |diagonal|). */
static void ldlt6_solve(const double* A_in, const double* rhs, double* x) {
  const int N = 6;
  double A[36];
  memcpy(A, A_in, sizeof(A));
  int perm[6];
  for (int i = 0; i < N; i++) perm[i] = i;
  for (int k = 0; k < N; k++) {
    int piv = k;
    double best = fabs(A[k * N + k]);
    for (int i = k + 1; i < N; i++)
      if (fabs(A[i * N + i]) > best) {
        best = fabs(A[i * N + i]);
        piv = i;
      }
    if (piv != k) {
      for (int c = 0; c < N; c++) {
        double t = A[k * N + c];
        A[k * N + c] = A[piv * N + c];
        A[piv * N + c] = t;
      }
      for (int r = 0; r < N; r++) {
        double t = A[r * N + k];
        A[r * N + k] = A[r * N + piv];
        A[r * N + piv] = t;
      }
      int t = perm[k];
      perm[k] = perm[piv];
      perm[piv] = t;
    }
    double d = A[k * N + k];
    if (d == 0.0) continue;
    double col[6];
    for (int i = k + 1; i < N; i++) col[i] = A[i * N + k];
    for (int i = k + 1; i < N; i++) {
      double l = col[i] / d;
      for (int j = k + 1; j <= i; j++) A[i * N + j] -= l * col[j];
      A[i * N + k] = l;
    }
    for (int i = k + 1; i < N; i++)
      for (int j = i + 1; j < N; j++) A[i * N + j] = A[j * N + i];
  }
  double y[6];
  for (int i = 0; i < N; i++) y[i] = rhs[perm[i]];
  for (int i = 0; i < N; i++)
    for (int j = 0; j < i; j++) y[i] -= A[i * N + j] * y[j];
  for (int i = 0; i < N; i++) y[i] = (A[i * N + i] != 0.0) ? y[i] / A[i * N + i] : 0.0;
  for (int i = N - 1; i >= 0; i--)
    for (int j = i + 1; j < N; j++) y[i] -= A[j * N + i] * y[j];
  for (int i = 0; i < N; i++) x[perm[i]] = y[i];
}

/* so3_exp (SO3:59-78) -> Quaterniond -> toRotationMatrix; delta (row-major 4x4) = [R | d[3:6]]  (LSQ:117-119, 140-142) */
static void delta_from_d(const double* d, double* delta) {
  double theta_sq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  double imag, real;
  if (theta_sq < 1e-10) {
    double theta_quad = theta_sq * theta_sq;
    imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
    real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
  } else {
    double theta = sqrt(theta_sq);
    double half = 0.5 * theta;
    imag = sin(half) / theta;
    real = cos(half);
  }
  double w = real, x = imag * d[0], y = imag * d[1], z = imag * d[2];
  /* Eigen QuaternionBase::toRotationMatrix (no normalisation) */
  double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  double R[9] = {1.0 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.0 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1.0 - (txx + tyy)};
  memset(delta, 0, 16 * sizeof(double));
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) delta[r * 4 + c] = R[r * 3 + c];
    delta[r * 4 + 3] = d[3 + r];
  }
  delta[15] = 1.0;
}

/* Isometry3d product delta * x0 (LSQ:119, 144) */
static void isom_mul(const double* A, const double* B, double* C) {
  double t[16];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) t[r * 4 + c] = A[r * 4 + 0] * B[0 * 4 + c] + A[r * 4 + 1] * B[1 * 4 + c] + A[r * 4 + 2] * B[2 * 4 + c];
    t[r * 4 + 3] = A[r * 4 + 0] * B[3] + A[r * 4 + 1] * B[7] + A[r * 4 + 2] * B[11] + A[r * 4 + 3];
  }
  t[12] = t[13] = t[14] = 0.0;
  t[15] = 1.0;
  memcpy(C, t, sizeof(t));
}

/* is_converged, LSQ:83-92 */
static int is_converged(const double* delta, const apdo_params* p) {
  double rmax = 0.0, tmax = 0.0;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) {
      double v = fabs(delta[r * 4 + c] - (r == c ? 1.0 : 0.0)) * (1.0 / p->rotation_epsilon);
      if (v > rmax) rmax = v;
    }
    double v = fabs(delta[r * 4 + 3]) * (1.0 / p->transformation_epsilon);
    if (v > tmax) tmax = v;
  }
  return fmax(rmax, tmax) < 1.0;
}

/* Everything an align() needs; mirrors the members of FastAPDGICP (APDH:101-121) */
typedef struct {
  const float *src_xyz, *src_label, *tgt_xyz, *tgt_label;
  int n, m;
  const double *src_cov, *tgt_cov, *geo_w;
  int* corr;
  float* sqd;
  double* maha;
} apdo_state;

typedef struct {
  int n_linearize;      /* number of linearize() calls == GN/LM outer iterations executed */
  int n_compute_error;  /* number of LM trial evaluations */
} apdo_counters;

/*
 * LsqRegistration::computeTransformation, LSQ:55-80 with step_lm LSQ:127-173 and step_gn LSQ:107-123.
 * guess: row-major 4x4 float (pcl Matrix4f).  Outputs: final_T (row-major float 4x4, LSQ:78), final_H (36),
 * converged, nr_iterations (LSQ:68: index of the last iteration started).
 * trace (optional, may be NULL): per outer iteration 16 doubles of x0 AFTER the step, up to max_iterations entries;
 * trace_corr (optional): n ints per outer iteration = correspondences used by that iteration's linearize.
 */
int apdo_align(const float* guess, const float* src_xyz, const float* src_label, int n, const float* tgt_xyz, const float* tgt_label, int m, const double* src_cov, const double* tgt_cov, const apdo_params* p, float* final_T, double* final_H, int* converged_out, int* nr_iterations_out, apdo_counters* counters, double* trace, int* trace_corr) {
  double x0[16];
  for (int i = 0; i < 16; i++) x0[i] = (double)guess[i]; /* LSQ:56 */
  x0[12] = x0[13] = x0[14] = 0.0;
  x0[15] = 1.0;
  double lm_lambda = -1.0; /* LSQ:58 */
  int converged = 0;       /* LSQ:59 */
  int nr_iterations = 0;
  double Hfin[36];
  for (int i = 0; i < 36; i++) Hfin[i] = (i % 7 == 0) ? 1.0 : 0.0; /* LSQ:23 */

  int* corr = (int*)malloc((size_t)n * sizeof(int));
  float* sqd = (float*)malloc((size_t)n * sizeof(float));
  double* maha = (double*)malloc((size_t)n * 16 * sizeof(double));
  double* geo_w = (double*)malloc((size_t)n * sizeof(double));
  if (!corr || !sqd || !maha || !geo_w) return -3;
  apdo_geo_weights(src_cov, n, geo_w);
  apdo_counters cnt = {0, 0};
  if (p->search == 1 && m > 0) g_align_tree = kd_build(tgt_xyz, m);

  for (int it = 0; it < p->max_iterations && !converged; it++) { /* LSQ:67 */
    nr_iterations = it;                                           /* LSQ:68 */
    double H[36], b[6], delta[16], nb[6], d[6];
    int ok = 0;
    double y0 = apdo_linearize(x0, src_xyz, src_label, n, tgt_xyz, tgt_label, m, src_cov, tgt_cov, geo_w, p, corr, sqd, maha, H, b);
    cnt.n_linearize++;
    if (trace_corr) memcpy(trace_corr + (size_t)it * n, corr, (size_t)n * sizeof(int));
    for (int i = 0; i < 6; i++) nb[i] = -b[i];
    if (p->optimizer == APDO_OPT_GN) { /* LSQ:107-123 */
      ldlt6_solve(H, nb, d);
      delta_from_d(d, delta);
      isom_mul(delta, x0, x0);
      memcpy(Hfin, H, sizeof(H));
      ok = 1;
    } else { /* LSQ:127-173 */
      if (lm_lambda < 0.0) {
        double mx = 0.0;
        for (int i = 0; i < 6; i++) mx = fmax(mx, fabs(H[i * 6 + i]));
        lm_lambda = p->lm_init_lambda_factor * mx; /* LSQ:131-133 */
      }
      double nu = 2.0;
      for (int j = 0; j < p->lm_max_iterations; j++) {
        double Hl[36];
        memcpy(Hl, H, sizeof(H));
        for (int i = 0; i < 6; i++) Hl[i * 6 + i] += lm_lambda;
        ldlt6_solve(Hl, nb, d); /* LSQ:137-138 */
        delta_from_d(d, delta);
        double xi[16];
        isom_mul(delta, x0, xi); /* LSQ:144 */
        double yi = apdo_compute_error(xi, src_xyz, src_label, n, tgt_xyz, tgt_label, geo_w, p, corr, maha);
        cnt.n_compute_error++;
        double den = 0.0;
        for (int i = 0; i < 6; i++) den += d[i] * (lm_lambda * d[i] - b[i]);
        double rho = (y0 - yi) / den; /* LSQ:146 */
        if (rho < 0) {                /* LSQ:156-164 */
          if (is_converged(delta, p)) {
            ok = 1;
            break;
          }
          lm_lambda = nu * lm_lambda;
          nu = 2 * nu;
          continue;
        }
        memcpy(x0, xi, sizeof(xi)); /* LSQ:166 */
        double f = 1 - pow(2 * rho - 1, 3);
        lm_lambda = lm_lambda * fmax(1.0 / 3.0, f); /* LSQ:167 */
        memcpy(Hfin, H, sizeof(H));                 /* LSQ:168 */
        ok = 1;
        break;
      }
    }
    if (trace) memcpy(trace + (size_t)it * 16, x0, 16 * sizeof(double));
    if (!ok) break; /* LSQ:71-74 "lm not converged!!" */
    converged = is_converged(delta, p); /* LSQ:75 */
  }
  for (int i = 0; i < 16; i++) final_T[i] = (float)x0[i]; /* LSQ:78 */
  if (final_H) memcpy(final_H, Hfin, sizeof(Hfin));
  if (converged_out) *converged_out = converged;
  if (nr_iterations_out) *nr_iterations_out = nr_iterations;
  if (counters) *counters = cnt;
  kd_free(g_align_tree);
  g_align_tree = NULL;
  free(corr);
  free(sqd);
  free(maha);
  free(geo_w);
  return 0;
}

/* ---------------------------------------------------------------- helpers exported for tests */
void apdo_ldlt6_solve(const double* A, const double* rhs, double* x) { ldlt6_solve(A, rhs, x); }
void apdo_delta_from_d(const double* d, double* delta) { delta_from_d(d, delta); }
void apdo_sym3_eigen(const double* A, double* w, double* V) { sym3_eigen(A, w, V); }
void apdo_sensor_cov(const apdo_params* p, const float* q, double* cov_r) { sensor_cov(p, q, cov_r); }
void apdo_transform_point_f(const double* T, const float* p, float* q) {
  float Tf[16];
  for (int i = 0; i < 16; i++) Tf[i] = (float)T[i];
  transform_point_f(Tf, p[0], p[1], p[2], q);
}

/* ------------------------------------------------------------------------------------------------ scan-to-submap target assembly
 *
 * SMO = /root/reference/4DRadarSLAM/apps/scan_matching_odometry_nodelet.cpp.  SMO:602-618: for the last max_submap_frames - 1
 * keyframes before the newest, rel_pose = odom_i^-1 * odom_newest (double), pcl::transformPointCloud(*cloud_i, out, rel_pose),
 * submap += out; then downsample(submap) (SMO:405-415) and setInputTarget.  The caller computes the rel_pose matrices.
 *
 * Third-party behaviour restated (PCL 1.10, NOT under /root/reference: "parity unpinned"):
 *  - pcl::transformPointCloud with a double matrix (common/impl/transforms.hpp, pcl::detail::Transformer<double>::se3):
 *        out_r = (float)(tf(r,0) * x + tf(r,1) * y + tf(r,2) * z + tf(r,3))   in double, left to right; all other fields copied.
 *  - downsample_method NONE (launch/ntu_loop3.launch:82) = pcl::PassThrough without a filter field: drops non-finite points.
 *  - downsample_method VOXELGRID = pcl::VoxelGrid<PointXYZINormal> (filters/impl/voxel_grid.hpp), downsample_all_data = true:
 *        float inverse leaf = 1 / leaf; min_b = floor(min * inv), div_b = max_b - min_b + 1; voxel index of a point
 *        idx = (floor(x inv) - min_b.x) + (floor(y inv) - min_b.y) div_b.x + (floor(z inv) - min_b.z) div_b.x div_b.y;
 *        points sorted by idx, one output point per occupied voxel in ascending idx order: xyz = float sum / count
 *        (AccumulatorXYZ), the "normal" (normal_x = cluster label here, normal_y = normal_z = 0) summed and NORMALISED
 *        (AccumulatorNormal: Eigen 3.3 normalize() leaves a zero vector alone) -- so a voxel's label becomes sign(sum).
 *        PCL orders the points of one voxel by an unstable std::sort; here they are added in ascending input order.
 *        When the voxel count overflows int32 PCL warns and returns the input unchanged; so does this.
 * Returns the number of output points (<= cap), or -1 when cap is too small.
 */
typedef struct {
  int64_t idx;
  int point;
} apdo_vox_ref;

static int vox_cmp(const void* a, const void* b) {
  const apdo_vox_ref* x = (const apdo_vox_ref*)a;
  const apdo_vox_ref* y = (const apdo_vox_ref*)b;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
  return x->point < y->point ? -1 : (x->point > y->point ? 1 : 0);
}

int apdo_submap_assemble(const float* xyz /* all frames, packed n x 3 */, const float* label, const int* frame_n, const double* rel_pose /* count x 16, row-major */,
                         int count, double voxel_leaf, float* out_xyz, float* out_label, int cap) {
  int total = 0;
  for (int k = 0; k < count; k++) total += frame_n[k];
  float* tx = (float*)malloc(sizeof(float) * 3 * (size_t)(total > 0 ? total : 1));
  float* tl = (float*)malloc(sizeof(float) * (size_t)(total > 0 ? total : 1));
  int m = 0, off = 0;
  for (int k = 0; k < count; k++) {
    const double* T = rel_pose + (size_t)k * 16;
    for (int i = 0; i < frame_n[k]; i++) {
      const float* p = xyz + 3 * (size_t)(off + i);
      if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue; /* PassThrough / VoxelGrid both skip non-finite points */
      const double x = (double)p[0], y = (double)p[1], z = (double)p[2];
      tx[3 * (size_t)m + 0] = (float)(T[0] * x + T[1] * y + T[2] * z + T[3]);
      tx[3 * (size_t)m + 1] = (float)(T[4] * x + T[5] * y + T[6] * z + T[7]);
      tx[3 * (size_t)m + 2] = (float)(T[8] * x + T[9] * y + T[10] * z + T[11]);
      tl[m] = label ? label[off + i] : 0.0f;
      m++;
    }
    off += frame_n[k];
  }
  int n_out = -1;
  int passthrough = !(voxel_leaf > 0.0);
  apdo_vox_ref* refs = NULL;
  if (!passthrough && m > 0) {
    const float inv = 1.0f / (float)voxel_leaf;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < m; i++)
      for (int a = 0; a < 3; a++) {
        if (tx[3 * (size_t)i + a] < mn[a]) mn[a] = tx[3 * (size_t)i + a];
        if (tx[3 * (size_t)i + a] > mx[a]) mx[a] = tx[3 * (size_t)i + a];
      }
    const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1, dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)INT32_MAX) {
      passthrough = 1; /* "Leaf size is too small for the input dataset. Integer indices would overflow." -> output = input */
    } else {
      int min_b[3], max_b[3], div_b[3];
      for (int a = 0; a < 3; a++) {
        min_b[a] = (int)floorf(mn[a] * inv);
        max_b[a] = (int)floorf(mx[a] * inv);
        div_b[a] = max_b[a] - min_b[a] + 1;
      }
      refs = (apdo_vox_ref*)malloc(sizeof(apdo_vox_ref) * (size_t)m);
      for (int i = 0; i < m; i++) {
        const int i0 = (int)floorf(tx[3 * (size_t)i] * inv) - min_b[0];
        const int i1 = (int)floorf(tx[3 * (size_t)i + 1] * inv) - min_b[1];
        const int i2 = (int)floorf(tx[3 * (size_t)i + 2] * inv) - min_b[2];
        refs[i].idx = (int64_t)i0 + (int64_t)i1 * div_b[0] + (int64_t)i2 * div_b[0] * div_b[1];
        refs[i].point = i;
      }
      qsort(refs, (size_t)m, sizeof(apdo_vox_ref), vox_cmp);
      n_out = 0;
      int i = 0;
      while (i < m) {
        int j = i;
        float sx = 0.f, sy = 0.f, sz = 0.f, sn = 0.f;
        while (j < m && refs[j].idx == refs[i].idx) {
          const int q = refs[j].point;
          sx += tx[3 * (size_t)q];
          sy += tx[3 * (size_t)q + 1];
          sz += tx[3 * (size_t)q + 2];
          sn += tl[q];
          j++;
        }
        if (n_out >= cap) {
          n_out = -1;
          break;
        }
        const float cnt = (float)(j - i);
        out_xyz[3 * (size_t)n_out] = sx / cnt;
        out_xyz[3 * (size_t)n_out + 1] = sy / cnt;
        out_xyz[3 * (size_t)n_out + 2] = sz / cnt;
        const float z2 = sn * sn;
        out_label[n_out] = z2 > 0.f ? sn / sqrtf(z2) : sn;
        n_out++;
        i = j;
      }
    }
  }
  if (passthrough || m == 0) {
    if (m > cap) {
      n_out = -1;
    } else {
      memcpy(out_xyz, tx, sizeof(float) * 3 * (size_t)m);
      memcpy(out_label, tl, sizeof(float) * (size_t)m);
      n_out = m;
    }
  }
  free(refs);
  free(tx);
  free(tl);
  return n_out;
}

/*
 * pcl::RadiusOutlierRemoval as the preprocessing nodelet uses it (PREP:163-171, 626-634; radius_radius / radius_min_neighbors of the
 * launch files).  PCL 1.10 (dense input): nearestKSearch(point, min_pts + 1) on the cloud itself -- the query is its own nearest
 * neighbour at distance 0 -- and the point is kept unless fewer than min_pts + 1 neighbours exist or radius^2 < (float) squared
 * distance of the last one (compared in double).  Restated without the tree: keep <=> the number of points j, i included, with
 * (double) d2(i, j) <= radius * radius exceeds min_pts.  "Parity unpinned": PCL is not under /root/reference.  keep_out[n]; returns
 * the number of points kept.
 */
int apdo_radius_outlier_mask(const float* xyz, int n, double radius, int min_pts, unsigned char* keep_out) {
  const double r2 = radius * radius;
  int kept = 0;
#pragma omp parallel for schedule(static) reduction(+ : kept)
  for (int i = 0; i < n; ++i) {
    int cnt = 0;
    for (int j = 0; j < n; ++j)
      if ((double)sqdist3f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2]) <= r2) ++cnt;
    keep_out[i] = cnt > min_pts ? 1 : 0;
    kept += keep_out[i];
  }
  return kept;
}

/*
 * pcl::StatisticalOutlierRemoval as the preprocessing nodelet uses it by default (PREP:153-162, 626-634; statistical_mean_k 20,
 * statistical_stddev 1.0 -- the launch files carry 30 / 1.2).  PCL 1.10 filters/impl/statistical_outlier_removal.hpp, restated from
 * memory of that source ("parity unpinned": PCL is not under /root/reference): for every point nearestKSearch(point, mean_k + 1) on
 * the cloud itself (sorted float squared distances, the first being the query at 0), dist_sum (double) += sqrt(nn_dists[k]) for
 * k = 1 .. mean_k with the global-namespace sqrt, i.e. in double, distances[i] = (float)(dist_sum / mean_k); then over all points in
 * index order: sum += d, sq_sum += d * d (double, d float), mean = sum / n, variance = (sq_sum - sum * sum / n) / (n - 1),
 * threshold = mean + stddev_mul * sqrt(variance); a point stays when distances[i] <= threshold.
 * Restated without the tree: the mean_k + 1 smallest float squared distances of every point by a bounded insertion.
 * keep_out[n], dist_out[n] (may be NULL); returns the number of points kept, -1 when n <= mean_k or mean_k > 63.
 */
int apdo_statistical_outlier_mask(const float* xyz, int n, int mean_k, double stddev_mul, unsigned char* keep_out, float* dist_out) {
  const int k = mean_k + 1;
  if (mean_k < 1 || mean_k > 63 || n < k) return -1;
  float* dist = (float*)malloc(sizeof(float) * (size_t)n);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    float best[64];
    int m = 0;
    for (int j = 0; j < n; ++j) {
      const float d = sqdist3f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2]);
      if (m == k && !(d < best[k - 1])) continue;
      int t = m < k ? m++ : k - 1;
      while (t > 0 && best[t - 1] > d) {
        best[t] = best[t - 1];
        --t;
      }
      best[t] = d;
    }
    double dist_sum = 0.0;
    for (int t = 1; t < k; ++t) dist_sum += sqrt((double)best[t]);
    dist[i] = (float)(dist_sum / mean_k);
  }
  double sum = 0.0, sq_sum = 0.0;
  for (int i = 0; i < n; ++i) {
    sum += dist[i];
    sq_sum += dist[i] * dist[i];
  }
  const double mean = sum / (double)n;
  const double variance = (sq_sum - sum * sum / (double)n) / ((double)n - 1);
  const double threshold = mean + stddev_mul * sqrt(variance);
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    keep_out[i] = dist[i] <= threshold ? 1 : 0;
    kept += keep_out[i];
    if (dist_out) dist_out[i] = dist[i];
  }
  free(dist);
  return kept;
}

/* ------------------------------------------------------------------------------------------------ preprocessing: DBSCAN cluster labels
 *
 * PREP = /root/reference/4DRadarSLAM/apps/preprocessing_nodelet_ntu.cpp, DBS = /root/reference/4DRadarSLAM/include/dbscan/DBSCAN_simple.h.
 * PREP:518-568: DBSCANKdtreeCluster (core min points 10, tolerance 0.9, cluster size 20..25000) over the whole scan; clusters ranked by
 * the distance of their centroid from the sensor; normal_x = rank + 1 for their points, 0 elsewhere -- the cluster label APD:272 compares.
 * DBS:28-100 is an order-dependent queue: points are visited in index order; an unprocessed point whose neighbourhood (radius
 * |norm - 1| / 50 + eps, DBS:39) holds at least minPts points (itself included) seeds a cluster; queue members are expanded with the
 * radius (norm - 1) / 100 + eps (DBS:65-68) and claim every still unprocessed neighbour; a point already claimed is never re-claimed.
 * Third-party behaviour restated ("parity unpinned": PCL / FLANN are not under /root/reference): pcl::search::KdTree::radiusSearch
 * (index, radius) = exact FLANN radius search on float L2_Simple squared distances, a neighbour when d2 < (float)(radius * radius)
 * (flann RadiusResultSet::addPoint), the query point itself included.  The order of the neighbours does not influence the result.
 * std::hypot(float x3) is restated as the correctly rounded float of the double norm; ties between centroid distances keep discovery order.
 * labels_out[n]; returns the number of clusters.
 */
int apdo_dbscan_labels(const float* xyz, int n, double eps, int min_pts, int min_cluster, int max_cluster, float* labels_out) {
  enum { UN = 0, PROCESSING = 1, PROCESSED = 2 };
  const size_t nn_ = (size_t)(n > 0 ? n : 1);
  unsigned char* types = (unsigned char*)calloc(nn_, 1);
  unsigned char* noise = (unsigned char*)calloc(nn_, 1);
  int* queue = (int*)malloc(sizeof(int) * nn_);
  int* nn = (int*)malloc(sizeof(int) * nn_);
  /* accepted clusters: member lists packed one after the other (a point can be in several: DBS:50-54 re-queue seed neighbours that an
   * earlier cluster already holds) */
  size_t mem_cap = nn_ * 2, mem_n = 0;
  int* members = (int*)malloc(sizeof(int) * mem_cap);
  int clu_cap = 64, n_clusters = 0;
  size_t* clu_begin = (size_t*)malloc(sizeof(size_t) * (size_t)(clu_cap + 1));
  clu_begin[0] = 0;
  for (int i = 0; i < n; i++) labels_out[i] = 0.0f;
#define APDO_NORM(q) ((double)sqrtf(xyz[3 * (size_t)(q)] * xyz[3 * (size_t)(q)] + xyz[3 * (size_t)(q) + 1] * xyz[3 * (size_t)(q) + 1] + xyz[3 * (size_t)(q) + 2] * xyz[3 * (size_t)(q) + 2]))
#define APDO_RADIUS_SEARCH(q, radius, count)                                            \
  do {                                                                                   \
    const float r2_ = (float)((radius) * (radius));                                      \
    const float qx_ = xyz[3 * (size_t)(q)], qy_ = xyz[3 * (size_t)(q) + 1], qz_ = xyz[3 * (size_t)(q) + 2]; \
    (count) = 0;                                                                         \
    for (int j_ = 0; j_ < n; j_++) {                                                     \
      const float dx_ = qx_ - xyz[3 * (size_t)j_], dy_ = qy_ - xyz[3 * (size_t)j_ + 1], dz_ = qz_ - xyz[3 * (size_t)j_ + 2]; \
      float d_ = dx_ * dx_;                                                              \
      d_ = d_ + dy_ * dy_;                                                               \
      d_ = d_ + dz_ * dz_;                                                               \
      if (d_ < r2_) nn[(count)++] = j_;                                                  \
    }                                                                                    \
  } while (0)
  for (int i = 0; i < n; i++) {
    if (types[i] == PROCESSED) continue;
    const double r_seed = fabs(APDO_NORM(i) - 1) / 50 + eps; /* DBS:36-39: float products summed in float, std::sqrt(float) -> float, then double arithmetic */
    int cnt = 0;
    APDO_RADIUS_SEARCH(i, r_seed, cnt);
    if (cnt < min_pts) {
      noise[i] = 1;
      continue;
    }
    int qn = 0;
    queue[qn++] = i;
    types[i] = PROCESSED;
    for (int j = 0; j < cnt; j++)
      if (nn[j] != i) {
        queue[qn++] = nn[j]; /* DBS:50-54: every neighbour of the seed joins the queue, whatever its state */
        types[nn[j]] = PROCESSING;
      }
    int sq = 1;
    while (sq < qn) {
      const int c = queue[sq];
      if (noise[c] || types[c] == PROCESSED) {
        types[c] = PROCESSED;
        sq++;
        continue;
      }
      /* DBS:65-67: (std::sqrt(float expression) - 1) / 100 is evaluated in FLOAT (float - int, float / int); only the sum with the
       * double eps_ is double.  (The seed radius above is different: DBS:36-39 first stores the float root in a double.) */
      const float ef = ((float)APDO_NORM(c) - 1.0f) / 100.0f;
      const double r_exp = (double)ef + eps;
      APDO_RADIUS_SEARCH(c, r_exp, cnt);
      if (cnt >= min_pts)
        for (int j = 0; j < cnt; j++)
          if (types[nn[j]] == UN) {
            queue[qn++] = nn[j];
            types[nn[j]] = PROCESSING;
          }
      types[c] = PROCESSED;
      sq++;
    }
    if (qn >= min_cluster && qn <= max_cluster) { /* DBS:83-95 (the queue holds no duplicate, so sort + unique only orders it) */
      if (mem_n + (size_t)qn > mem_cap) {
        mem_cap = (mem_n + (size_t)qn) * 2;
        members = (int*)realloc(members, sizeof(int) * mem_cap);
      }
      if (n_clusters + 1 > clu_cap) {
        clu_cap *= 2;
        clu_begin = (size_t*)realloc(clu_begin, sizeof(size_t) * (size_t)(clu_cap + 1));
      }
      memcpy(members + mem_n, queue, sizeof(int) * (size_t)qn);
      mem_n += (size_t)qn;
      clu_begin[++n_clusters] = mem_n;
    }
  }
#undef APDO_RADIUS_SEARCH
#undef APDO_NORM
  /* PREP:533-568: centroid (float sums over the SORTED member list, DBS:91), distance from the sensor, labels written cluster by cluster in
   * ascending distance, so a point two clusters share ends with the label of the farther one */
  if (n_clusters > 0) {
    float* dist = (float*)malloc(sizeof(float) * (size_t)n_clusters);
    int* order = (int*)malloc(sizeof(int) * (size_t)n_clusters);
    unsigned char* in = (unsigned char*)calloc(nn_, 1);
    for (int c = 0; c < n_clusters; c++) {
      const int* m = members + clu_begin[c];
      const int cnt = (int)(clu_begin[c + 1] - clu_begin[c]);
      for (int j = 0; j < cnt; j++) in[m[j]] = 1;
      float sx = 0.f, sy = 0.f, sz = 0.f;
      for (int i = 0; i < n; i++) /* ascending index = the sorted indices vector */
        if (in[i]) {
          sx += xyz[3 * (size_t)i];
          sy += xyz[3 * (size_t)i + 1];
          sz += xyz[3 * (size_t)i + 2];
          in[i] = 0;
        }
      const float cx = sx / cnt, cy = sy / cnt, cz = sz / cnt;
      dist[c] = (float)sqrt((double)cx * cx + (double)cy * cy + (double)cz * cz);
      order[c] = c;
    }
    for (int a = 1; a < n_clusters; a++) { /* stable insertion sort by distance */
      const int v = order[a];
      int b = a - 1;
      while (b >= 0 && dist[order[b]] > dist[v]) {
        order[b + 1] = order[b];
        b--;
      }
      order[b + 1] = v;
    }
    for (int r = 0; r < n_clusters; r++) {
      const int c = order[r];
      for (size_t j = clu_begin[c]; j < clu_begin[c + 1]; j++) labels_out[members[j]] = (float)(r + 1);
    }
    free(in); free(order); free(dist);
  }
  free(clu_begin); free(members); free(nn); free(queue); free(noise); free(types);
  return n_clusters;
}

/* ------------------------------------------------------------------------------------------------ preprocessing: REVE Doppler ego-velocity
 *
 * REVE = /root/reference/4DRadarSLAM/src/radar_ego_velocity_estimator.cpp, REVEH = .../include/radar_ego_velocity_estimator.h.
 * REVE:60-170 estimate(): per target range / azimuth / elevation gates, zero-velocity test on the |doppler| order statistic, else
 * REVE:172-250 solve3DFullRansac + REVE:252-303 solve3DFull (normal equations, LDLT, sigma from the residual).
 * The reference draws its RANSAC samples from std::shuffle with a std::random_device seed (REVE:186-193): they cannot be reproduced,
 * so the samples are an INPUT here (sample_idx[n_iter][n_sample]: indices into the list of valid targets, what idx[0..N) holds after
 * the shuffle).  Restated third-party arithmetic ("parity unpinned"): Eigen::LDLT<3x3>, H^T H products (summed row by row here),
 * atan2 / sqrt float overloads as correctly rounded floats (as everywhere in this file).
 */
typedef struct {
  float min_dist, max_dist, min_db, elevation_thresh_deg, azimuth_thresh_deg, doppler_velocity_correction_factor;
  float thresh_zero_velocity, allowed_outlier_percentage, sigma_zero_velocity_x, sigma_zero_velocity_y, sigma_zero_velocity_z;
  float sigma_offset_radar_x, sigma_offset_radar_y, sigma_offset_radar_z, max_sigma_x, max_sigma_y, max_sigma_z;
  float inlier_thresh;
  int use_ransac, n_ransac_points;
} apdo_reve_config;

void apdo_reve_default_config(apdo_reve_config* c) { /* REVEH:30-60 */
  c->min_dist = 1; c->max_dist = 400; c->min_db = 0; c->elevation_thresh_deg = 22.5f; c->azimuth_thresh_deg = 56.5f; c->doppler_velocity_correction_factor = 1;
  c->thresh_zero_velocity = 0.05f; c->allowed_outlier_percentage = 0.30f; c->sigma_zero_velocity_x = 1.0e-03f; c->sigma_zero_velocity_y = 3.2e-03f; c->sigma_zero_velocity_z = 1.0e-02f;
  c->sigma_offset_radar_x = 0; c->sigma_offset_radar_y = 0; c->sigma_offset_radar_z = 0; c->max_sigma_x = 0.2f; c->max_sigma_y = 0.2f; c->max_sigma_z = 0.2f;
  c->inlier_thresh = 0.5f; c->use_ransac = 1; c->n_ransac_points = 5;
}

static void ldlt3_solve(const double* A_in, const double* rhs, double* x) { /* Eigen::LDLT with symmetric diagonal pivoting, n = 3 */
  double A[9];
  int perm[3] = {0, 1, 2};
  memcpy(A, A_in, sizeof(A));
  for (int k = 0; k < 3; k++) {
    int piv = k;
    double best = fabs(A[k * 3 + k]);
    for (int i = k + 1; i < 3; i++)
      if (fabs(A[i * 3 + i]) > best) {
        best = fabs(A[i * 3 + i]);
        piv = i;
      }
    if (piv != k) {
      for (int c = 0; c < 3; c++) { double t = A[k * 3 + c]; A[k * 3 + c] = A[piv * 3 + c]; A[piv * 3 + c] = t; }
      for (int r = 0; r < 3; r++) { double t = A[r * 3 + k]; A[r * 3 + k] = A[r * 3 + piv]; A[r * 3 + piv] = t; }
      int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    const double d = A[k * 3 + k];
    if (d == 0.0) continue;
    double col[3];
    for (int i = k + 1; i < 3; i++) col[i] = A[i * 3 + k];
    for (int i = k + 1; i < 3; i++) {
      const double l = col[i] / d;
      for (int j = k + 1; j <= i; j++) A[i * 3 + j] -= l * col[j];
      A[i * 3 + k] = l;
    }
    for (int i = k + 1; i < 3; i++)
      for (int j = i + 1; j < 3; j++) A[i * 3 + j] = A[j * 3 + i];
  }
  double y[3];
  for (int i = 0; i < 3; i++) y[i] = rhs[perm[i]];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < i; j++) y[i] -= A[i * 3 + j] * y[j];
  for (int i = 0; i < 3; i++) y[i] = (A[i * 3 + i] != 0.0) ? y[i] / A[i * 3 + i] : 0.0;
  for (int i = 2; i >= 0; i--)
    for (int j = i + 1; j < 3; j++) y[i] -= A[j * 3 + i] * y[j];
  for (int i = 0; i < 3; i++) x[perm[i]] = y[i];
}

/* solve3DFull, REVE:252-303, on the rows listed in `rows` of the feature table f[.][4] = (nx, ny, nz, doppler).  Returns what the
 * reference returns (always true: REVE:301) and, in *sigma_ok, whether the sigma test of REVE:282-289 passed. */
static void reve_solve(const double* f, const int* rows, int m, const apdo_reve_config* c, int estimate_sigma, double* v, double* sigma, int* sigma_ok) {
  double HTH[9] = {0}, HTy[3] = {0};
  for (int q = 0; q < m; q++) {
    const double* r = f + 4 * (size_t)rows[q];
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) HTH[a * 3 + b] += r[a] * r[b];
      HTy[a] += r[a] * r[3];
    }
  }
  ldlt3_solve(HTH, HTy, v); /* use_cholesky_instead_of_bdcsvd = true (REVEH:53) */
  if (sigma_ok) *sigma_ok = 0;
  if (!estimate_sigma) return;
  double ee = 0.0;
  for (int q = 0; q < m; q++) {
    const double* r = f + 4 * (size_t)rows[q];
    const double e = (r[0] * v[0] + r[1] * v[1] + r[2] * v[2]) - r[3];
    ee += e * e;
  }
  /* (HTH)^-1 diagonal by cofactors (Eigen's 3x3 inverse) */
  const double c00 = HTH[4] * HTH[8] - HTH[5] * HTH[7], c01 = HTH[5] * HTH[6] - HTH[3] * HTH[8], c02 = HTH[3] * HTH[7] - HTH[4] * HTH[6];
  const double det = HTH[0] * c00 + HTH[1] * c01 + HTH[2] * c02;
  const double i00 = c00 / det, i11 = (HTH[0] * HTH[8] - HTH[2] * HTH[6]) / det, i22 = (HTH[0] * HTH[4] - HTH[1] * HTH[3]) / det;
  const double s = ee / (double)(m - 3);
  double sg[3] = {s * i00, s * i11, s * i22};
  sigma[0] = sg[0]; sigma[1] = sg[1]; sigma[2] = sg[2];
  if (sg[0] >= 0.0 && sg[1] >= 0.0 && sg[2] >= 0.0) {
    sigma[0] = sqrt(sg[0]) + c->sigma_offset_radar_x;
    sigma[1] = sqrt(sg[1]) + c->sigma_offset_radar_y;
    sigma[2] = sqrt(sg[2]) + c->sigma_offset_radar_z;
    if (sigma_ok) *sigma_ok = sigma[0] < c->max_sigma_x && sigma[1] < c->max_sigma_y && sigma[2] < c->max_sigma_z;
  }
}

/* features of the VALID targets (REVE:75-90): returns their number; valid_idx[k] = target index, f[k][4] = nx, ny, nz, doppler */
int apdo_reve_features(const float* t /* n x 5: x y z intensity doppler */, int n, const apdo_reve_config* c, int* valid_idx, double* f) {
  int m = 0;
  const double az_lim = (double)c->azimuth_thresh_deg * M_PI / 180.0, el_lim = (double)c->elevation_thresh_deg * M_PI / 180.0;
  for (int i = 0; i < n; i++) {
    const float x = t[5 * (size_t)i], y = t[5 * (size_t)i + 1], z = t[5 * (size_t)i + 2], inten = t[5 * (size_t)i + 3], dop = t[5 * (size_t)i + 4];
    const double r = sqrt((double)x * (double)x + (double)y * (double)y + (double)z * (double)z);
    const double azimuth = (double)(float)atan2((double)y, (double)x);
    float rxy2 = x * x;
    rxy2 = rxy2 + y * y;
    const float rxy = (float)sqrt((double)rxy2);
    const double elevation = (double)(float)atan2((double)rxy, (double)z) - M_PI_2;
    if (r > c->min_dist && r < c->max_dist && inten > c->min_db && fabs(azimuth) < az_lim && fabs(elevation) < el_lim) {
      const float d = -dop * c->doppler_velocity_correction_factor;
      valid_idx[m] = i;
      f[4 * (size_t)m] = x / r;
      f[4 * (size_t)m + 1] = y / r;
      f[4 * (size_t)m + 2] = z / r;
      f[4 * (size_t)m + 3] = (double)d;
      m++;
    }
  }
  return m;
}

static int dbl_cmp(const void* a, const void* b) {
  const double x = *(const double*)a, y = *(const double*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* estimate(), REVE:60-170.  inlier_mask / outlier_mask: n bytes (by target index).  Returns success (0 / 1); *n_valid, *zero_velocity. */
int apdo_reve_estimate(const float* t, int n, const apdo_reve_config* c, const unsigned int* sample_idx, int n_iter, double* v_r, double* sigma_v_r,
                       unsigned char* inlier_mask, unsigned char* outlier_mask, int* n_valid, int* zero_velocity) {
  int* valid = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  double* f = (double*)malloc(sizeof(double) * 4 * (size_t)(n > 0 ? n : 1));
  const int m = apdo_reve_features(t, n, c, valid, f);
  memset(inlier_mask, 0, (size_t)n);
  memset(outlier_mask, 0, (size_t)n);
  v_r[0] = v_r[1] = v_r[2] = 0.0;
  sigma_v_r[0] = sigma_v_r[1] = sigma_v_r[2] = 0.0;
  *n_valid = m;
  *zero_velocity = 0;
  int success = 0;
  if (m > 2) {
    double* vd = (double*)malloc(sizeof(double) * (size_t)m);
    for (int k = 0; k < m; k++) vd[k] = fabs(f[4 * (size_t)k + 3]);
    const size_t nth = (size_t)((double)m * (1.0 - (double)c->allowed_outlier_percentage));
    qsort(vd, (size_t)m, sizeof(double), dbl_cmp); /* std::nth_element: only v_dopplers[n] is read */
    const double median = vd[nth < (size_t)m ? nth : (size_t)m - 1];
    free(vd);
    if (median < c->thresh_zero_velocity) { /* REVE:111-121 */
      *zero_velocity = 1;
      sigma_v_r[0] = c->sigma_zero_velocity_x; sigma_v_r[1] = c->sigma_zero_velocity_y; sigma_v_r[2] = c->sigma_zero_velocity_z;
      for (int k = 0; k < m; k++)
        if (fabs(f[4 * (size_t)k + 3]) < c->thresh_zero_velocity) inlier_mask[valid[k]] = 1;
      success = 1;
    } else if (!c->use_ransac) {
      int* rows = (int*)malloc(sizeof(int) * (size_t)m);
      for (int k = 0; k < m; k++) { rows[k] = k; inlier_mask[valid[k]] = 1; }
      reve_solve(f, rows, m, c, 1, v_r, sigma_v_r, NULL);
      free(rows);
      success = 1;
    } else { /* solve3DFullRansac, REVE:172-250 */
      int* best_in = (int*)malloc(sizeof(int) * (size_t)m);
      int* best_out = (int*)malloc(sizeof(int) * (size_t)m);
      int* cur_in = (int*)malloc(sizeof(int) * (size_t)m);
      int* cur_out = (int*)malloc(sizeof(int) * (size_t)m);
      int nbi = 0, nbo = 0;
      if (m >= c->n_ransac_points) {
        for (int k = 0; k < n_iter; k++) {
          int rows[64];
          for (int q = 0; q < c->n_ransac_points && q < 64; q++) rows[q] = (int)sample_idx[(size_t)k * c->n_ransac_points + q];
          double v[3], sg[3];
          reve_solve(f, rows, c->n_ransac_points, c, 0, v, sg, NULL);
          v_r[0] = v[0]; v_r[1] = v[1]; v_r[2] = v[2]; /* the reference solves into v_r itself */
          int ni = 0, no = 0;
          for (int j = 0; j < m; j++) {
            const double* r = f + 4 * (size_t)j;
            const double err = fabs(r[3] - (r[0] * v[0] + r[1] * v[1] + r[2] * v[2]));
            if (err < c->inlier_thresh) cur_in[ni++] = j; else cur_out[no++] = j;
          }
          if ((float)no / (float)(ni + no) > 0.05) { /* REVE:215-220: too many outliers -> all of them count as inliers */
            for (int q = 0; q < no; q++) cur_in[ni++] = cur_out[q];
            no = 0;
          }
          if (ni > nbi) { memcpy(best_in, cur_in, sizeof(int) * (size_t)ni); nbi = ni; }
          if (no > nbo) { memcpy(best_out, cur_out, sizeof(int) * (size_t)no); nbo = no; }
        }
      }
      if (nbi > 0) {
        reve_solve(f, best_in, nbi, c, 1, v_r, sigma_v_r, NULL);
        success = 1; /* REVE:301 returns true whatever the sigma test said */
      }
      for (int q = 0; q < nbi; q++) inlier_mask[valid[best_in[q]]] = 1;
      for (int q = 0; q < nbo; q++) outlier_mask[valid[best_out[q]]] = 1;
      free(cur_out); free(cur_in); free(best_out); free(best_in);
    }
  }
  free(f);
  free(valid);
  return success;
}
