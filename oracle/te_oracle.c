/*
 * te_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See te_oracle.h.
 *
 * Plain C99, no dependencies.  Build: see oracle/Makefile (-O2 -ffp-contract=off so that the
 * double arithmetic is reproducible and is not fused differently from a generic x86-64 build of
 * the reference).  Every function cites the reference lines (relative to /root/reference) or the
 * un-vendored upstream algorithm it restates.
 */
#include "te_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;

void teo_set_threads(int n) { g_threads = n < 1 ? 1 : n; }

/* The degenerate-plane rule of the filter that WROTE the reference's bag (2018).  The bag's golden layers hold UnitZ at the
 * two cells -- (row 99, col 117) and (99, 118) -- whose 6-point border discs are exactly planar and tilted; today's area
 * method (and this oracle by default) returns the plane's normal there.  grid_map_filters up to 1.6 -- from memory, the
 * library is not vendored -- formed the scatter matrix of the CENTRED points, NN * NN^T, and ran the eigen-solver only if
 * covarianceMatrix.fullPivHouseholderQr().rank() >= 3; otherwise eigenvectors = Identity, eigenvalues = (1, 1, 0): UnitZ.
 * With this switch on the oracle applies that rule (rank by full pivoting, a pivot counting if it exceeds 3 * DBL_EPSILON of
 * the largest) and reproduces the bag on 13 300 of 13 300 cells in every layer, bit for bit (tests/test_oracle_kat.py): on
 * the bag the third pivot is exactly 0 on 1 419 cells (1 417 of them flat, where UnitZ is the normal anyway) and at least
 * 1.4e-3 of the first everywhere else, so the classification does not hang on the threshold.  Off by default: the product
 * follows the current filter, and on float32 terrain an exactly planar tilted disc does not occur. */
static int g_rank_rule = 0;
void teo_set_normals_rank_rule(int on) { g_rank_rule = on != 0; }

int teo_get_max_threads(void) {
#ifdef _OPENMP
  return omp_get_num_procs();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------- */
/* grid_map_core geometry (un-vendored; GridMapMath.cpp)                                         */
/* ------------------------------------------------------------------------------------------- */

/* GridMap::setGeometry: size = round(length/res); length = size * res. */
void teo_geom_init(teo_geom* g, int rows, int cols, double res, double pos_x, double pos_y) {
  g->rows = rows;
  g->cols = cols;
  g->res = res;
  g->len_x = (double)rows * res;
  g->len_y = (double)cols * res;
  g->pos_x = pos_x;
  g->pos_y = pos_y;
}

/* TE/config/robot_filter_parameter.yaml:1-37, robot_footprint_parameter.yaml:3-8, robot.yaml:10 */
void teo_params_default(teo_params* p) {
  p->normals_radius = 0.05;
  p->normals_axis = 2;
  p->slope_critical = 1.0;
  p->step_critical = 0.12;
  p->step_radius1 = 0.04;
  p->step_radius2 = 0.04;
  p->step_ncrit = 4;
  p->rough_critical = 0.05;
  p->rough_radius = 0.05;
  p->w_scale = 1.0f / 3.0f;
  p->w_slope = 1.0f;
  p->w_step = 1.0f;
  p->w_rough = 1.0f;
  p->fp_radius = 0.30;
  p->fp_offset = 0.15;
  p->fp_default = 0.3;
  p->fp_max_gap = 0.3;
  p->fp_critical_step = 0.12;
  p->fp_check_roughness = 0;
}

#define IDX(g, i, j) ((size_t)(j) * (size_t)(g)->rows + (size_t)(i))

/* getPositionFromIndex: position = mapPosition + (0.5*length - 0.5*res) + res * (-index)
 * (start index (0,0); evaluated left to right like the Eigen expression). */
static inline double cell_x(const teo_geom* g, int i) {
  return (g->pos_x + (0.5 * g->len_x - 0.5 * g->res)) + g->res * (double)(-i);
}
static inline double cell_y(const teo_geom* g, int j) {
  return (g->pos_y + (0.5 * g->len_y - 0.5 * g->res)) + g->res * (double)(-j);
}

/* checkIfPositionWithinMap: transformed = -(position - mapPosition - 0.5*length) in [0, length) */
static inline int pos_inside(const teo_geom* g, double x, double y) {
  const double tx = -((x - g->pos_x) - 0.5 * g->len_x);
  const double ty = -((y - g->pos_y) - 0.5 * g->len_y);
  return tx >= 0.0 && ty >= 0.0 && tx < g->len_x && ty < g->len_y;
}

/* getIndexFromPosition: indexVector = (position - 0.5*length - mapPosition) / res; index = (int)(-v) */
static inline int pos_to_index(const teo_geom* g, double x, double y, int* i, int* j) {
  const double vx = ((x - 0.5 * g->len_x) - g->pos_x) / g->res;
  const double vy = ((y - 0.5 * g->len_y) - g->pos_y) / g->res;
  *i = (int)(-vx);
  *j = (int)(-vy);
  return pos_inside(g, x, y) && *i >= 0 && *j >= 0 && *i < g->rows && *j < g->cols;
}

static inline int finitef(float v) { return isfinite(v); }

/* Conservative integer half-width of the bounding box of a circle; the exact membership test is
 * the double test of CircleIterator::isInside below, so a larger box never changes the set. */
static inline int box_halfwidth(const teo_geom* g, double radius) {
  double k = ceil(radius / g->res) + 1.0;
  if (k > 1e6) k = 1e6;
  return (int)k;
}

/* CircleIterator (un-vendored): SubmapIterator over the clamped bounding box, row index outer,
 * column index inner (incrementIndexForSubmap), keeping cells whose centre satisfies
 * (position - center).array().square().sum() <= radius*radius. */
#define CIRCLE_FOREACH(g, ci, cj, radius, II, JJ, ...)                                \
  do {                                                                                \
    const double cx__ = cell_x((g), (ci)), cy__ = cell_y((g), (cj));                  \
    const double r2__ = (radius) * (radius);                                          \
    const int k__ = box_halfwidth((g), (radius));                                     \
    const int i0__ = (ci)-k__ < 0 ? 0 : (ci)-k__;                                     \
    const int i1__ = (ci) + k__ > (g)->rows - 1 ? (g)->rows - 1 : (ci) + k__;         \
    const int j0__ = (cj)-k__ < 0 ? 0 : (cj)-k__;                                     \
    const int j1__ = (cj) + k__ > (g)->cols - 1 ? (g)->cols - 1 : (cj) + k__;         \
    for (int II = i0__; II <= i1__; ++II) {                                           \
      const double dx__ = cell_x((g), II) - cx__;                                     \
      for (int JJ = j0__; JJ <= j1__; ++JJ) {                                         \
        const double dy__ = cell_y((g), JJ) - cy__;                                   \
        if (dx__ * dx__ + dy__ * dy__ <= r2__) {                                      \
          __VA_ARGS__                                                                 \
        }                                                                             \
      }                                                                               \
    }                                                                                 \
  } while (0)

int teo_circle_count(const teo_geom* g, int i, int j, double radius) {
  int n = 0;
  CIRCLE_FOREACH(g, i, j, radius, a, b, { ++n; });
  return n;
}

/* ------------------------------------------------------------------------------------------- */
/* 3x3 symmetric eigen-decomposition (stands in for Eigen::SelfAdjointEigenSolver)              */
/* ------------------------------------------------------------------------------------------- */
/* Upstream calls SelfAdjointEigenSolver<MatrixXd>::computeDirect; for a dynamic-size matrix type
 * Eigen 3.3 dispatches that to the iterative compute().  Any backward-stable symmetric solver
 * returns the same eigenpairs to a few ulp; cyclic Jacobi is used here.  Eigenvalues ascending
 * (first-minimum selection sort, like Eigen), eigenvectors in the columns of V. */
static void eig3_sym(const double Ain[3][3], double w[3], double V[3][3]) {
  double A[3][3];
  memcpy(A, Ain, sizeof(A));
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) V[a][b] = (a == b) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; ++sweep) {
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    if (off == 0.0) break;
    for (int p = 0; p < 2; ++p) {
      for (int q = p + 1; q < 3; ++q) {
        const double apq = A[p][q];
        if (apq == 0.0) continue;
        const double g100 = 100.0 * fabs(apq);
        /* after a few sweeps, drop elements that no longer change the diagonal */
        if (sweep > 3 && fabs(A[p][p]) + g100 == fabs(A[p][p]) && fabs(A[q][q]) + g100 == fabs(A[q][q])) {
          A[p][q] = A[q][p] = 0.0;
          continue;
        }
        const double h = A[q][q] - A[p][p];
        double t;
        if (fabs(h) + g100 == fabs(h)) {
          t = apq / h;
        } else {
          const double theta = 0.5 * h / apq;
          t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
          if (theta < 0.0) t = -t;
        }
        const double c = 1.0 / sqrt(1.0 + t * t);
        const double s = t * c;
        const double tau = s / (1.0 + c);
        A[p][p] -= t * apq;
        A[q][q] += t * apq;
        A[p][q] = A[q][p] = 0.0;
        const int r = 3 - p - q; /* the third index */
        {
          const double arp = A[r][p], arq = A[r][q];
          A[r][p] = A[p][r] = arp - s * (arq + arp * tau);
          A[r][q] = A[q][r] = arq + s * (arp - arq * tau);
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = vkp - s * (vkq + vkp * tau);
          V[k][q] = vkq + s * (vkp - vkq * tau);
        }
      }
    }
  }
  w[0] = A[0][0];
  w[1] = A[1][1];
  w[2] = A[2][2];
  for (int a = 0; a < 2; ++a) {
    int m = a;
    for (int b = a + 1; b < 3; ++b)
      if (w[b] < w[m]) m = b;
    if (m != a) {
      const double tw = w[a];
      w[a] = w[m];
      w[m] = tw;
      for (int k = 0; k < 3; ++k) {
        const double tv = V[k][a];
        V[k][a] = V[k][m];
        V[k][m] = tv;
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------- */
/* a1  NormalVectorsFilter::computeWithAreaSerial / areaSingleNormalComputation (un-vendored)    */
/* ------------------------------------------------------------------------------------------- */
static void normals_cell(const teo_geom* g, const float* elev, double radius, int axis, int ci, int cj, float* nx,
                         float* ny, float* nz) {
  size_t n = 0;
  double sum[3] = {0.0, 0.0, 0.0};
  double ss[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  CIRCLE_FOREACH(g, ci, cj, radius, a, b, {
    const float zf = elev[IDX(g, a, b)];
    if (finitef(zf)) { /* GridMap::getPosition3 returns false for non-finite values */
      const double p[3] = {cell_x(g, a), cell_y(g, b), (double)zf};
      ++n;
      sum[0] += p[0];
      sum[1] += p[1];
      sum[2] += p[2];
      for (int u = 0; u < 3; ++u)
        for (int v = 0; v < 3; ++v) ss[u][v] += p[u] * p[v]; /* sumSquared.noalias() += point * point^T */
    }
  });
  double nv[3] = {0.0, 0.0, 1.0}; /* nPoints < 3 -> UnitZ */
  if (n >= 3) {
    const double dn = (double)n;
    const double mean[3] = {sum[0] / dn, sum[1] / dn, sum[2] / dn};
    double cov[3][3];
    for (int u = 0; u < 3; ++u)
      for (int v = 0; v < 3; ++v) cov[u][v] = ss[u][v] / dn - mean[u] * mean[v];
    int full_rank = 1;
    if (g_rank_rule) { /* see teo_set_normals_rank_rule: scatter matrix of the centred points, rank by full pivoting */
      double S[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
      CIRCLE_FOREACH(g, ci, cj, radius, a, b, {
        const float zf = elev[IDX(g, a, b)];
        if (finitef(zf)) {
          const double q[3] = {cell_x(g, a) - mean[0], cell_y(g, b) - mean[1], (double)zf - mean[2]};
          for (int u = 0; u < 3; ++u)
            for (int v = 0; v < 3; ++v) S[u][v] += q[u] * q[v];
        }
      });
      double first = 0.0;
      int rank = 0;
      for (int k = 0; k < 3; ++k) {
        int pr = k, pc = k;
        double best = -1.0;
        for (int u = k; u < 3; ++u)
          for (int v = k; v < 3; ++v)
            if (fabs(S[u][v]) > best) {
              best = fabs(S[u][v]);
              pr = u;
              pc = v;
            }
        if (k == 0) first = best;
        if (!(best > 3.0 * 2.220446049250313e-16 * first) || best == 0.0) break;
        ++rank;
        for (int v = 0; v < 3; ++v) {
          const double t = S[k][v];
          S[k][v] = S[pr][v];
          S[pr][v] = t;
        }
        for (int u = 0; u < 3; ++u) {
          const double t = S[u][k];
          S[u][k] = S[u][pc];
          S[u][pc] = t;
        }
        for (int u = k + 1; u < 3; ++u) {
          const double f = S[u][k] / S[k][k];
          for (int v = k; v < 3; ++v) S[u][v] -= f * S[k][v];
        }
      }
      full_rank = rank >= 3;
    }
    double w[3], V[3][3];
    eig3_sym(cov, w, V);
    if (full_rank && w[1] > 1e-8) { /* second eigenvalue zero -> normal undefined -> UnitZ */
      nv[0] = V[0][0];
      nv[1] = V[1][0];
      nv[2] = V[2][0];
    }
  }
  if (nv[axis] < 0.0) { /* unitaryNormalVector.dot(normalVectorPositiveAxis_) < 0 */
    nv[0] = -nv[0];
    nv[1] = -nv[1];
    nv[2] = -nv[2];
  }
  const size_t o = IDX(g, ci, cj);
  nx[o] = (float)nv[0];
  ny[o] = (float)nv[1];
  nz[o] = (float)nv[2];
}

int teo_normals(const teo_geom* g, const float* elev, double radius, int axis, float* nx, float* ny, float* nz) {
  if (axis < 0 || axis > 2) return -1;
  const size_t N = (size_t)g->rows * g->cols;
  for (size_t k = 0; k < N; ++k) nx[k] = ny[k] = nz[k] = NAN; /* GridMap::add fills NaN */
#pragma omp parallel for schedule(dynamic, 4) num_threads(g_threads)
  for (int j = 0; j < g->cols; ++j)
    for (int i = 0; i < g->rows; ++i)
      if (finitef(elev[IDX(g, i, j)])) normals_cell(g, elev, radius, axis, i, j, nx, ny, nz);
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* a2  SlopeFilter::update   TEF/src/SlopeFilter.cpp:59-88                                       */
/* ------------------------------------------------------------------------------------------- */
int teo_slope(const teo_geom* g, const float* nz, double crit, float* out) {
  const size_t N = (size_t)g->rows * g->cols;
#pragma omp parallel for num_threads(g_threads)
  for (size_t k = 0; k < N; ++k) {
    out[k] = NAN;                        /* :63 mapOut.add(type_) */
    if (!finitef(nz[k])) continue;       /* :71 */
    const double slope = acos((double)nz[k]); /* :74 */
    if (slope < crit)                    /* :76 */
      out[k] = (float)(1.0 - slope / crit);
    else
      out[k] = (float)0.0;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* a4+a5  StepFilter::update   TEF/src/StepFilter.cpp:102-182                                    */
/* ------------------------------------------------------------------------------------------- */
int teo_step(const teo_geom* g, const float* elev, double crit, double r1, double r2, int ncrit, float* out,
             float* step_height_out) {
  const size_t N = (size_t)g->rows * g->cols;
  float* sh = step_height_out ? step_height_out : (float*)malloc(N * sizeof(float));
  if (!sh) return -2;
  for (size_t k = 0; k < N; ++k) out[k] = sh[k] = NAN; /* :106-107 */

  /* first iteration :112-144 */
#pragma omp parallel for schedule(dynamic, 4) num_threads(g_threads)
  for (int j = 0; j < g->cols; ++j) {
    for (int i = 0; i < g->rows; ++i) {
      if (!finitef(elev[IDX(g, i, j)])) continue; /* :113 */
      double hmax = 0.0, hmin = 0.0;
      int init = 0;
      CIRCLE_FOREACH(g, i, j, r1, a, b, {
        const float zf = elev[IDX(g, a, b)];
        if (finitef(zf)) { /* :126 */
          const double h = (double)zf;
          if (!init) {
            hmax = hmin = h;
            init = 1;
          } else {
            if (h > hmax) hmax = h;
            if (h < hmin) hmin = h;
          }
        }
      });
      if (init) sh[IDX(g, i, j)] = (float)(hmax - hmin); /* :142-143 */
    }
  }

  /* second iteration :147-178 (every cell, valid elevation or not) */
#pragma omp parallel for schedule(dynamic, 4) num_threads(g_threads)
  for (int j = 0; j < g->cols; ++j) {
    for (int i = 0; i < g->rows; ++i) {
      int ncells = 0, valid = 0;
      double smax = 0.0; /* :149 */
      CIRCLE_FOREACH(g, i, j, r2, a, b, {
        const float s = sh[IDX(g, a, b)];
        if (finitef(s)) { /* :159 */
          valid = 1;
          if ((double)s > smax) smax = (double)s;  /* :162-164 */
          if ((double)s > crit) ++ncells;          /* :165 */
        }
      });
      if (valid) {
        const double a1 = (double)ncells / (double)ncrit * smax;
        const double step = smax < a1 ? smax : a1; /* std::min(stepMax, nCells/nCrit*stepMax) :170 */
        out[IDX(g, i, j)] = step < crit ? (float)(1.0 - step / crit) : (float)0.0; /* :172-176 */
      }
    }
  }
  if (!step_height_out) free(sh); /* :180 erase("step_height") */
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* a7  RoughnessFilter::update   TEF/src/RoughnessFilter.cpp:73-132                              */
/* ------------------------------------------------------------------------------------------- */
int teo_roughness(const teo_geom* g, const float* elev, const float* nx, const float* ny, const float* nz,
                  double crit, double radius, float* out) {
  const size_t N = (size_t)g->rows * g->cols;
  for (size_t k = 0; k < N; ++k) out[k] = NAN; /* :77 */
  const int kb = box_halfwidth(g, radius);
  const size_t cap = (size_t)(2 * kb + 1) * (size_t)(2 * kb + 1);
  int err = 0;
#pragma omp parallel num_threads(g_threads)
  {
    double* pts = (double*)malloc(cap * 3 * sizeof(double)); /* :87-88 MatrixXd points(3, max) */
    if (!pts) {
#pragma omp atomic write
      err = 1;
    }
#pragma omp for schedule(dynamic, 4)
    for (int j = 0; j < g->cols; ++j) {
      if (!pts) continue;
      for (int i = 0; i < g->rows; ++i) {
        const size_t o = IDX(g, i, j);
        if (!finitef(nx[o])) continue; /* :84 */
        size_t n = 0;
        CIRCLE_FOREACH(g, i, j, radius, a, b, {
          const float zf = elev[IDX(g, a, b)];
          if (finitef(zf)) { /* :98 */
            pts[3 * n + 0] = cell_x(g, a);
            pts[3 * n + 1] = cell_y(g, b);
            pts[3 * n + 2] = (double)zf;
            ++n;
          }
        });
        double m[3] = {0.0, 0.0, 0.0}; /* :105 rowwise().sum() / nPoints */
        for (size_t k = 0; k < n; ++k) {
          m[0] += pts[3 * k + 0];
          m[1] += pts[3 * k + 1];
          m[2] += pts[3 * k + 2];
        }
        m[0] /= (double)n;
        m[1] /= (double)n;
        m[2] /= (double)n;
        const double ax = (double)nx[o], ay = (double)ny[o], az = (double)nz[o]; /* :108-110 */
        const double plane = m[0] * ax + m[1] * ay + m[2] * az;                  /* :111 */
        double sum = 0.0;
        for (size_t k = 0; k < n; ++k) { /* :113-116 */
          const double dist = ax * pts[3 * k + 0] + ay * pts[3 * k + 1] + az * pts[3 * k + 2] - plane;
          sum += dist * dist;
        }
        const double rough = sqrt(sum / (double)(n - 1)); /* :117; n==1 -> 0/0 = NaN -> score 0 */
        out[o] = rough < crit ? (float)(1.0 - rough / crit) : (float)0.0; /* :119-124 */
      }
    }
    free(pts);
  }
  return err ? -2 : 0;
}

/* ------------------------------------------------------------------------------------------- */
/* a9  MathExpressionFilter, fixed form (un-vendored EigenLab on MatrixXf => float32, left to right) */
/* ------------------------------------------------------------------------------------------- */
int teo_combine(long n, const float* slope, const float* step, const float* rough, float w_scale, float w_slope,
                float w_step, float w_rough, float* out) {
  for (long k = 0; k < n; ++k) {
    const float a = w_slope * slope[k];
    const float b = w_step * step[k];
    const float c = w_rough * rough[k];
    const float ab = a + b;
    const float abc = ab + c;
    out[k] = w_scale * abc;
  }
  return 0;
}

/* a1..a10 in the YAML order (robot_filter_parameter.yaml:1-37) */
int teo_chain(const teo_geom* g, const teo_params* p, const float* elev, float* slope, float* step, float* rough,
              float* trav, float* nx_out, float* ny_out, float* nz_out) {
  const size_t N = (size_t)g->rows * g->cols;
  float* nx = nx_out ? nx_out : (float*)malloc(N * sizeof(float));
  float* ny = ny_out ? ny_out : (float*)malloc(N * sizeof(float));
  float* nz = nz_out ? nz_out : (float*)malloc(N * sizeof(float));
  int rc = -2;
  if (nx && ny && nz) {
    rc = teo_normals(g, elev, p->normals_radius, p->normals_axis, nx, ny, nz);
    if (!rc) rc = teo_slope(g, nz, p->slope_critical, slope);
    if (!rc) rc = teo_step(g, elev, p->step_critical, p->step_radius1, p->step_radius2, p->step_ncrit, step, NULL);
    if (!rc) rc = teo_roughness(g, elev, nx, ny, nz, p->rough_critical, p->rough_radius, rough);
    if (!rc) rc = teo_combine((long)N, slope, step, rough, p->w_scale, p->w_slope, p->w_step, p->w_rough, trav);
  }
  if (!nx_out) free(nx); /* DeletionFilter */
  if (!ny_out) free(ny);
  if (!nz_out) free(nz);
  return rc;
}

/* ------------------------------------------------------------------------------------------- */
/* SpiralIterator (un-vendored, grid_map_core/src/iterators/SpiralIterator.cpp)                  */
/* ------------------------------------------------------------------------------------------- */
static inline int signum_i(int v) { return (v > 0) - (v < 0); }

typedef struct spiral_it {
  const teo_geom* g;
  double cx, cy, r2;
  int ci, cj;
  unsigned distance, nrings;
  int* bi; /* ring buffer (pointsRing_), consumed from the back */
  int* bj;
  int cnt, cap;
} spiral_it;

static inline int spiral_past_end(const spiral_it* s) { return s->distance == s->nrings && s->cnt == 0; }

static void spiral_generate_ring(spiral_it* s) {
  s->distance++;
  int px = (int)s->distance, py = 0;
  const teo_geom* g = s->g;
  do {
    const int mi = px + s->ci, mj = py + s->cj;
    if (mi >= 0 && mj >= 0 && mi < g->rows && mj < g->cols) { /* checkIfIndexInRange */
      int keep = 1;
      if (s->distance == s->nrings || s->distance == s->nrings - 1) { /* only the outer rings are tested */
        const double dx = cell_x(g, mi) - s->cx, dy = cell_y(g, mj) - s->cy;
        keep = (dx * dx + dy * dy <= s->r2);
      }
      if (keep && s->cnt < s->cap) {
        s->bi[s->cnt] = mi;
        s->bj[s->cnt] = mj;
        s->cnt++;
      }
    }
    const int nx = -signum_i(py), ny = signum_i(px);
    if (nx != 0 && (unsigned)sqrt((double)(px + nx) * (px + nx) + (double)py * py) == s->distance)
      px += nx;
    else if (ny != 0 && (unsigned)sqrt((double)px * px + (double)(py + ny) * (py + ny)) == s->distance)
      py += ny;
    else {
      px += nx;
      py += ny;
    }
  } while ((unsigned)px != s->distance || py != 0);
}

static void spiral_init(spiral_it* s, const teo_geom* g, int ci, int cj, double radius, int* bi, int* bj, int cap) {
  s->g = g;
  s->ci = ci; /* gridMap.getIndex(center, indexCenter): centre of a cell maps to that cell */
  s->cj = cj;
  s->cx = cell_x(g, ci);
  s->cy = cell_y(g, cj);
  s->r2 = radius * radius;
  s->distance = 0;
  s->nrings = (unsigned)ceil(radius / g->res);
  s->bi = bi;
  s->bj = bj;
  s->cap = cap;
  s->cnt = 0;
  s->bi[0] = ci; /* centre is always in range here */
  s->bj[0] = cj;
  s->cnt = 1;
}

static inline void spiral_next(spiral_it* s) {
  s->cnt--;
  if (s->cnt == 0 && !spiral_past_end(s)) spiral_generate_ring(s);
  /* upstream generates one ring per increment; an empty ring leaves the iterator on an empty
   * vector only if it is also past the end, otherwise the caller's next ++ pops again.  Rings of
   * a centre inside the map are never empty before the map is exhausted, but guard anyway. */
  while (s->cnt == 0 && !spiral_past_end(s)) spiral_generate_ring(s);
}

static int ring_capacity(const teo_geom* g, double radius) {
  const double k = ceil(radius / g->res) + 2.0;
  return (int)(8.0 * k + 16.0);
}

int teo_spiral_offsets(const teo_geom* g, int ci, int cj, double radius, int* di, int* dj, int* ring, int cap) {
  const int rc = ring_capacity(g, radius);
  int* bi = (int*)malloc(sizeof(int) * 2 * rc);
  if (!bi) return -2;
  spiral_it s;
  spiral_init(&s, g, ci, cj, radius, bi, bi + rc, rc);
  int n = 0;
  while (!spiral_past_end(&s)) {
    if (n < cap) {
      di[n] = s.bi[s.cnt - 1] - ci;
      dj[n] = s.bj[s.cnt - 1] - cj;
      ring[n] = (int)sqrt((double)(di[n] * di[n] + dj[n] * dj[n])); /* getCurrentRadius()/res: integer norm */
    }
    ++n;
    spiral_next(&s);
  }
  free(bi);
  return n;
}

/* ------------------------------------------------------------------------------------------- */
/* a14  isTraversableForFilters: checkForSlope / checkForStep / checkForRoughness                */
/*      TE/src/TraversabilityMap.cpp:774-921  (pure per-cell functions; the reference memoises   */
/*      them in slope_footprint / step_footprint / roughness_footprint)                          */
/* ------------------------------------------------------------------------------------------- */

/* checkForSlope :867-893 and checkForRoughness :895-921 share the same shape (factor 2 vs 1.5). */
static int check_count_zero(const teo_geom* g, const float* layer, int i, int j, double factor, double max_gap,
                            float* memo) {
  const size_t o = IDX(g, i, j);
  if (layer[o] == 0.0) {
    const double wr = 3.0 * g->res;
    const double crit_len = max_gap / 3.0;
    const int ncrit = (int)floor(factor * wr * crit_len / pow(g->res, 2));
    int n = 0, bad = 0;
    CIRCLE_FOREACH(g, i, j, wr, a, b, {
      if (!bad) {
        if (layer[IDX(g, a, b)] == 0.0) n++;
        if (n > ncrit) bad = 1;
      }
    });
    if (memo) memo[o] = bad ? 0.0f : 1.0f;
    return !bad;
  }
  return 1;
}

/* LineIterator (un-vendored; Bresenham) */
typedef struct line_it {
  int i, j, inc1i, inc1j, inc2i, inc2j, den, num, numadd, ncells, icell;
} line_it;

static void line_init(line_it* L, int si, int sj, int ei, int ej) {
  L->icell = 0;
  L->i = si;
  L->j = sj;
  const int dx = abs(ei - si), dy = abs(ej - sj);
  L->inc1i = L->inc2i = (ei >= si) ? 1 : -1;
  L->inc1j = L->inc2j = (ej >= sj) ? 1 : -1;
  if (dx >= dy) {
    L->inc1i = 0;
    L->inc2j = 0;
    L->den = dx;
    L->num = dx / 2;
    L->numadd = dy;
    L->ncells = dx + 1;
  } else {
    L->inc2i = 0;
    L->inc1j = 0;
    L->den = dy;
    L->num = dy / 2;
    L->numadd = dx;
    L->ncells = dy + 1;
  }
}
static inline void line_next(line_it* L) {
  L->num += L->numadd;
  if (L->num >= L->den) {
    L->num -= L->den;
    L->i += L->inc1i;
    L->j += L->inc1j;
  }
  L->i += L->inc2i;
  L->j += L->inc2j;
  L->icell++;
}

/* boundPositionToRange (GridMapMath.cpp) for one axis */
static inline double bound_axis(double position, double len, double mappos) {
  double shifted = position - mappos + 0.5 * len;
  double eps = 10.0 * 2.220446049250313e-16;
  if (fabs(position) > 1.0) eps *= fabs(position);
  if (shifted <= 0)
    shifted = eps;
  else if (shifted >= len)
    shifted = len - eps;
  return shifted + mappos - 0.5 * len;
}

/* checkForStep :794-865 */
static int check_step(const teo_geom* g, const float* elev, const float* step, int ci, int cj, double crit_step,
                      double max_gap, float* memo) {
  const size_t oc = IDX(g, ci, cj);
  if (!(step[oc] == 0.0)) return 1;
  const double wr = 2.5 * g->res;
  const double cx = cell_x(g, ci), cy = cell_y(g, cj);
  double height = (double)elev[oc];
  int candi[64], candj[64];
  int ncand = 0;
  CIRCLE_FOREACH(g, ci, cj, wr, a, b, {
    const size_t o = IDX(g, a, b);
    if ((double)elev[o] > crit_step + height && step[o] == 0.0 && ncand < 64) {
      candi[ncand] = a;
      candj[ncand] = b;
      ++ncand;
    }
  });
  if (ncand == 0) {
    candi[0] = ci;
    candj[0] = cj;
    ncand = 1;
  }
  for (int c = 0; c < ncand; ++c) {
    const int ii = candi[c], ij = candj[c];
    const double sl = 2.5 * g->res; /* subMapLength */
    const double sx = cell_x(g, ii), sy = cell_y(g, ij); /* subMapPos */
    const double tcx = cx - sx, tcy = cy - sy;           /* toCenter */
    /* GridMap::getSubmap -> getSubmapInformation */
    double tlx = bound_axis(sx + 0.5 * sl, g->len_x, g->pos_x), tly = bound_axis(sy + 0.5 * sl, g->len_y, g->pos_y);
    int ti, tj, bi, bj;
    if (!pos_to_index(g, tlx, tly, &ti, &tj)) {
      if (memo) memo[oc] = 0.0f;
      return 0;
    }
    double brx = bound_axis(sx - 0.5 * sl, g->len_x, g->pos_x), bry = bound_axis(sy - 0.5 * sl, g->len_y, g->pos_y);
    if (!pos_to_index(g, brx, bry, &bi, &bj)) {
      if (memo) memo[oc] = 0.0f;
      return 0;
    }
    const double tcornx = cell_x(g, ti) + 0.5 * g->res, tcorny = cell_y(g, tj) + 0.5 * g->res; /* topLeftCorner */
    const int sr = bi - ti + 1, sc = bj - tj + 1;                                               /* submap size */
    const double slx = (double)sr * g->res, sly = (double)sc * g->res;                          /* submap length */
    const double spx = tcornx - 0.5 * slx, spy = tcorny - 0.5 * sly;                            /* submap position */
    /* getSubmapInformation's final getIndexFromPosition(requested position in submap) always succeeds here */
    height = (double)elev[IDX(g, ii, ij)];
    for (int lin = 0; lin < sr * sc; ++lin) { /* GridMapIterator over the submap: row index fastest */
      const int a = lin % sr, b = lin / sr;
      const size_t o = IDX(g, ti + a, tj + b);
      if (step[o] == 0.0 && (double)elev[o] < height - crit_step) {
        /* subMap.getPosition: the submap's own geometry (length re-derived by setGeometry) */
        const double px = (spx + (0.5 * slx - 0.5 * g->res)) + g->res * (double)(-a);
        const double py = (spy + (0.5 * sly - 0.5 * g->res)) + g->res * (double)(-b);
        const double vx = px - sx, vy = py - sy;
        if (sqrt(vx * vx + vy * vy) < 0.025) continue;
        if (sqrt(tcx * tcx + tcy * tcy) > 0.025) {
          if (tcx * vx + tcy * vy < 0.0) continue;
        }
        double qx = sx + vx, qy = sy + vy;
        for (;;) {
          const double ex = (qx - sx) + vx, ey = (qy - sy) + vy;
          if (!(sqrt(ex * ex + ey * ey) < max_gap && pos_inside(g, qx + vx, qy + vy))) break;
          qx += vx;
          qy += vy;
        }
        int ei, ej;
        pos_to_index(g, qx, qy, &ei, &ej);
        if (ei < 0) ei = 0;
        if (ej < 0) ej = 0;
        if (ei > g->rows - 1) ei = g->rows - 1;
        if (ej > g->cols - 1) ej = g->cols - 1;
        int gap_start = 0, gap_end = 0;
        line_it L;
        for (line_init(&L, ii, ij, ei, ej); L.icell < L.ncells; line_next(&L)) {
          const float ef = elev[IDX(g, L.i, L.j)];
          if ((double)ef > height + crit_step) {
            if (memo) memo[oc] = 0.0f;
            return 0;
          }
          if ((double)ef < height - crit_step || !finitef(ef)) {
            gap_start = 1;
          } else if (gap_start) {
            gap_end = 1;
            break;
          }
        }
        if (gap_start && !gap_end) {
          if (memo) memo[oc] = 0.0f;
          return 0;
        }
      }
    }
  }
  if (memo) memo[oc] = 1.0f;
  return 1;
}

/* isTraversableForFilters :774-792 for every cell, starting from all-NaN memo layers (slope_fp / step_fp /
 * rough_fp may be NULL): untrav[cell] = 1 where it returns false. */
static void untraversable_cells(const teo_geom* g, const teo_params* p, const float* elev, const float* slope, const float* step,
                                const float* rough, unsigned char* untrav, float* slope_fp, float* step_fp, float* rough_fp) {
  const size_t N = (size_t)g->rows * g->cols;
  for (size_t k = 0; k < N; ++k) {
    if (slope_fp) slope_fp[k] = NAN;
    if (step_fp) step_fp[k] = NAN;
    if (rough_fp) rough_fp[k] = NAN;
  }
#pragma omp parallel for schedule(dynamic, 4) num_threads(g_threads)
  for (int j = 0; j < g->cols; ++j) {
    for (int i = 0; i < g->rows; ++i) {
      int ok = check_count_zero(g, slope, i, j, 2.0, p->fp_max_gap, slope_fp);
      if (ok) ok = check_step(g, elev, step, i, j, p->fp_critical_step, p->fp_max_gap, step_fp);
      if (ok && p->fp_check_roughness) ok = check_count_zero(g, rough, i, j, 1.5, p->fp_max_gap, rough_fp);
      untrav[IDX(g, i, j)] = (unsigned char)!ok;
    }
  }
}

/* ------------------------------------------------------------------------------------------- */
/* a13  traversabilityFootprint(radius, offset) :307-318 -> isTraversable :654-746               */
/* ------------------------------------------------------------------------------------------- */
int teo_footprint(const teo_geom* g, const teo_params* p, const float* elev, const float* slope, const float* step,
                  const float* rough, const float* trav, float* footprint, float* slope_fp, float* step_fp,
                  float* rough_fp) {
  const size_t N = (size_t)g->rows * g->cols;
  const double rmin = p->fp_radius, rmax = p->fp_radius + p->fp_offset;
  unsigned char* untrav = (unsigned char*)malloc(N);
  if (!untrav) return -2;
  for (size_t k = 0; k < N; ++k) footprint[k] = NAN;
  untraversable_cells(g, p, elev, slope, step, rough, untrav, slope_fp, step_fp, rough_fp);
  const int rc = ring_capacity(g, rmax);
  int err = 0;
#pragma omp parallel num_threads(g_threads)
  {
    int* buf = (int*)malloc(sizeof(int) * 2 * rc);
    if (!buf) {
#pragma omp atomic write
      err = 1;
    }
#pragma omp for schedule(dynamic, 4)
    for (int j = 0; j < g->cols; ++j) {
      if (!buf) continue;
      for (int i = 0; i < g->rows; ++i) {
        int ncells = 0;
        double t = 0.0;
        int done = 0;
        spiral_it s;
        for (spiral_init(&s, g, i, j, rmax, buf, buf + rc, rc); !spiral_past_end(&s); spiral_next(&s)) {
          const int a = s.bi[s.cnt - 1], b = s.bj[s.cnt - 1];
          const size_t o = IDX(g, a, b);
          if (untrav[o]) {
            const int da = a - i, db = b - j;
            const double ru = (double)(int)sqrt((double)(da * da + db * db)) * g->res; /* getCurrentRadius */
            if (rmin == 0.0 || ru <= rmin) { /* :694-704 */
              footprint[IDX(g, i, j)] = 0.0f;
            } else { /* :705-711 */
              const double factor = ((ru - rmin) / (rmax - rmin) + 1.0) / 2.0;
              t *= factor / ncells;
              footprint[IDX(g, i, j)] = (float)t;
            }
            done = 1;
            break; /* :714-717 computeUntraversablePolygon == false */
          }
          ncells++;
          t += finitef(trav[o]) ? (double)trav[o] : p->fp_default; /* :719-724 */
        }
        if (!done) footprint[IDX(g, i, j)] = (float)(t / ncells); /* :732-735 */
      }
    }
    free(buf);
  }
  free(untrav);
  return err ? -2 : 0;
}

/* ------------------------------------------------------------------------------------------- */
/* N2  checkInclination :748-762 (footprint/check_robot_inclination == true, :114) on a layer   */
/*     robot_slope.  start == end: atPosition(robot_slope, start) == 0 -> false; otherwise a     */
/*     LineIterator from the start index to the end index, cells that are not valid (not finite) */
/*     skipped, any 0 -> false.  *outside = 1 when a position is outside the map: atPosition     */
/*     throws there and getIndex()'s failure is ignored (undefined indices); callers report      */
/*     status 1.                                                                                  */
/* ------------------------------------------------------------------------------------------- */
static int inclination_ok(const teo_geom* g, const float* robot_slope, double sx, double sy, double ex, double ey,
                          int* outside) {
  int si, sj, ei, ej;
  *outside = 0;
  if (ex == sx && ey == sy) { /* Eigen operator== on the two positions */
    if (!pos_inside(g, sx, sy) || !pos_to_index(g, sx, sy, &si, &sj)) {
      *outside = 1;
      return 0;
    }
    return !((double)robot_slope[IDX(g, si, sj)] == 0.0);
  }
  if (!pos_to_index(g, sx, sy, &si, &sj) || !pos_to_index(g, ex, ey, &ei, &ej)) {
    *outside = 1;
    return 0;
  }
  line_it L;
  for (line_init(&L, si, sj, ei, ej); L.icell < L.ncells; line_next(&L)) {
    const float v = robot_slope[IDX(g, L.i, L.j)];
    if (!finitef(v)) continue;
    if ((double)v == 0.0) return 0;
  }
  return 1;
}

/* batched: segment k = start_end_xy[4k .. 4k+4) = sx sy ex ey; ok[k] = checkInclination's result, status[k] 0 / 1 outside */
int teo_check_inclination(const teo_geom* g, const float* robot_slope, int n, const double* start_end_xy, unsigned char* ok,
                          int* status) {
  for (int k = 0; k < n; ++k) {
    const double* q = start_end_xy + 4 * (size_t)k;
    int outside;
    ok[k] = (unsigned char)inclination_ok(g, robot_slope, q[0], q[1], q[2], q[3], &outside);
    status[k] = outside;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* N2  checkCircularFootprintPath :344-462 for a batch of paths, with publishPolygons == false,   */
/*     compute_untraversable_polygon == false and footprint/check_robot_inclination == false     */
/*     (robot_footprint_parameter.yaml:10), on a map whose traversability_footprint layer is     */
/*     complete: isTraversable(center, ...) then takes its memo branch (:672-677) for every      */
/*     centre -- value = layer, traversable = value != 0.                                         */
/*     Poses outside the map: the reference calls getIndex() and ignores its result (undefined   */
/*     start/end index); here such a path gets status 1 and is reported unsafe.                   */
/*     A single pose outside the map uses traversabilityDefault_ (:663-665).                      */
/* ------------------------------------------------------------------------------------------- */
int teo_check_circular_paths_incl(const teo_geom* g, const float* footprint, double fp_default, const float* robot_slope,
                                  int n_paths, const int* pose_offset, const double* pose_xy, unsigned char* is_safe,
                                  double* traversability, int* status) {
  for (int k = 0; k < n_paths; ++k) {
    const int n = pose_offset[k + 1] - pose_offset[k];
    const double* xy = pose_xy + 2 * (size_t)pose_offset[k];
    is_safe[k] = 0;
    traversability[k] = 0.0;
    status[k] = 0;
    if (n <= 0) { /* :330-334 "This path has no poses to check" */
      status[k] = 2;
      continue;
    }
    double res_trav = 0.0, length_path = 0.0;
    double ex = 0.0, ey = 0.0, sx, sy;
    int ok = 1;
    for (int i = 0; i < n && ok; ++i) {
      sx = ex;
      sy = ey;
      ex = xy[2 * i];
      ey = xy[2 * i + 1];
      if (robot_slope && (n == 1 || i > 0)) { /* :366-370, :390-394 checkRobotInclination_ */
        int outside;
        const int good = n == 1 ? inclination_ok(g, robot_slope, ex, ey, ex, ey, &outside)
                                : inclination_ok(g, robot_slope, sx, sy, ex, ey, &outside);
        if (!good) {
          status[k] = outside;
          ok = 0;
          break;
        }
      }
      if (n == 1) { /* :365-385 */
        double t;
        int trav;
        int ci, cj;
        if (!pos_inside(g, ex, ey)) {
          t = fp_default;
          trav = fp_default != 0.0;
        } else {
          pos_to_index(g, ex, ey, &ci, &cj);
          t = (double)footprint[IDX(g, ci, cj)];
          trav = t != 0.0;
        }
        if (!trav) {
          ok = 0;
          break;
        }
        res_trav = t;
      }
      if (n > 1 && i > 0) { /* :388-456 */
        int si, sj, ei, ej;
        if (!pos_to_index(g, sx, sy, &si, &sj) || !pos_to_index(g, ex, ey, &ei, &ej)) {
          status[k] = 1;
          ok = 0;
          break;
        }
        double sum = 0.0;
        int nline = 0;
        line_it L;
        for (line_init(&L, ei, ej, si, sj); L.icell < L.ncells; line_next(&L)) { /* from the end index to the start index */
          const double t = (double)footprint[IDX(g, L.i, L.j)];
          if (!(t != 0.0)) {
            ok = 0;
            break;
          }
          sum += t;
          nline++;
          for (int s = 0; s < 3; ++s) /* nSkip :396 */
            if (L.icell < L.ncells) line_next(&L);
        }
        if (!ok) break;
        const double t = sum / (double)nline;
        const double dx = ex - sx, dy = ey - sy;
        const double length_segment = sqrt(dx * dx + dy * dy);
        if (i > 1) { /* :443-447 (lengthPath keeps its value between iterations) */
          const double length_previous = length_path;
          length_path += length_segment;
          res_trav = (length_segment * t + length_previous * res_trav) / length_path;
        } else {
          length_path = length_segment;
          res_trav = t;
        }
      }
    }
    if (ok) {
      is_safe[k] = 1;
      traversability[k] = res_trav;
    }
  }
  return 0;
}

int teo_check_circular_paths(const teo_geom* g, const float* footprint, double fp_default, int n_paths,
                             const int* pose_offset, const double* pose_xy, unsigned char* is_safe,
                             double* traversability, int* status) {
  return teo_check_circular_paths_incl(g, footprint, fp_default, NULL, n_paths, pose_offset, pose_xy, is_safe, traversability,
                                       status);
}

/* ------------------------------------------------------------------------------------------- */
/* N3  polygon footprints.  grid_map_core is not vendored by the reference: Polygon::isInside,    */
/*     PolygonIterator (findSubmapParameters + SubmapIterator) are restated from the published   */
/*     grid_map 1.6 sources; parity unpinned (the reference has no fixture for this path).       */
/* ------------------------------------------------------------------------------------------- */
/* grid_map::Polygon::isInside: crossing-number test, edge (i, j = i-1) */
static int polygon_inside(int n, const double* v, double px, double py) {
  int cross = 0;
  for (int i = 0, j = n - 1; i < n; j = i++) {
    const double xi = v[2 * i], yi = v[2 * i + 1], xj = v[2 * j], yj = v[2 * j + 1];
    if (((yi > py) != (yj > py)) && (px < (xj - xi) * (py - yi) / (yj - yi) + xi)) cross++;
  }
  return cross % 2;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* isTraversable(polygon, traversability) :586-645 with computeUntraversablePolygon == false; *value = 0 when the
 * polygon is not traversable (every caller discards the partial sum).  PolygonIterator: bounding box of the vertices,
 * both corners bound to the map (boundPositionToRange), converted to indices (a corner that lands exactly on the map
 * border would give an index one past the end in the reference: clamped here), SubmapIterator order (row index outer). */
static int polygon_traversable(const teo_geom* g, const unsigned char* untrav, const float* trav, double def, int n,
                               const double* v, double* value) {
  double tlx = v[0], tly = v[1], brx = v[0], bry = v[1];
  for (int k = 1; k < n; ++k) {
    tlx = tlx < v[2 * k] ? v[2 * k] : tlx;
    tly = tly < v[2 * k + 1] ? v[2 * k + 1] : tly;
    brx = v[2 * k] < brx ? v[2 * k] : brx;
    bry = v[2 * k + 1] < bry ? v[2 * k + 1] : bry;
  }
  tlx = bound_axis(tlx, g->len_x, g->pos_x);
  tly = bound_axis(tly, g->len_y, g->pos_y);
  brx = bound_axis(brx, g->len_x, g->pos_x);
  bry = bound_axis(bry, g->len_y, g->pos_y);
  int ti, tj, bi, bj;
  pos_to_index(g, tlx, tly, &ti, &tj);
  pos_to_index(g, brx, bry, &bi, &bj);
  ti = clampi(ti, 0, g->rows - 1);
  bi = clampi(bi, 0, g->rows - 1);
  tj = clampi(tj, 0, g->cols - 1);
  bj = clampi(bj, 0, g->cols - 1);
  unsigned ncells = 0;
  double t = 0.0;
  for (int a = ti; a <= bi; ++a) {
    const double px = cell_x(g, a);
    for (int b = tj; b <= bj; ++b) {
      if (!polygon_inside(n, v, px, cell_y(g, b))) continue;
      const size_t o = IDX(g, a, b);
      if (untrav[o]) { /* :603-611 */
        *value = 0.0;
        return 0;
      }
      ncells++;
      t += finitef(trav[o]) ? (double)trav[o] : def; /* :613-618 */
    }
  }
  if (ncells == 0) { /* :626-629 */
    *value = def;
    return def != 0.0;
  }
  *value = t / ncells;
  return 1;
}

int teo_polygons_traversable(const teo_geom* g, const teo_params* p, const float* elev, const float* slope, const float* step,
                             const float* rough, const float* trav, int n_polygons, const int* vertex_offset,
                             const double* vertex_xy, unsigned char* is_traversable, double* traversability) {
  unsigned char* untrav = (unsigned char*)malloc((size_t)g->rows * g->cols);
  if (!untrav) return -2;
  untraversable_cells(g, p, elev, slope, step, rough, untrav, NULL, NULL, NULL);
  for (int k = 0; k < n_polygons; ++k) {
    const int n = vertex_offset[k + 1] - vertex_offset[k];
    if (n < 1) {
      free(untrav);
      return -1;
    }
    is_traversable[k] = (unsigned char)polygon_traversable(g, untrav, trav, p->fp_default, n,
                                                           vertex_xy + 2 * (size_t)vertex_offset[k], &traversability[k]);
  }
  free(untrav);
  return 0;
}

/* The rotation part of  toPosition * orientation * positionToVertex  (:270-283) for a yaw-only orientation:
 * kindr AngleAxis(yaw, 0, 0, 1) * identity -> quaternion (w, 0, 0, z) = (cos(yaw/2), 0, 0, sin(yaw/2)); Eigen's
 * Quaternion::toRotationMatrix; linear * v.  The translation is added per cell. */
void teo_rotate_footprint(int n_points, const double* points_xy, double yaw, double* out_xy) {
  const double w = cos(yaw / 2.0), z = sin(yaw / 2.0);
  const double tz = 2.0 * z, twz = tz * w, tzz = tz * z;
  const double r00 = 1.0 - (0.0 + tzz), r01 = 0.0 - twz, r10 = 0.0 + twz, r11 = 1.0 - (0.0 + tzz);
  for (int k = 0; k < n_points; ++k) {
    const double px = points_xy[2 * k], py = points_xy[2 * k + 1];
    out_xy[2 * k] = r00 * px + r01 * py;
    out_xy[2 * k + 1] = r10 * px + r11 * py;
  }
}

/* traversabilityFootprint(footprintYaw) :239-305: layers traversability_x (footprint as given) and traversability_rot
 * (footprint turned by yaw) -- for every cell the footprint polygon centred on it. */
int teo_polygon_footprint(const teo_geom* g, const teo_params* p, const float* elev, const float* slope, const float* step,
                          const float* rough, const float* trav, int n_points, const double* points_xy, double yaw,
                          float* trav_x, float* trav_rot) {
  if (n_points < 1 || n_points > 64) return -1;
  unsigned char* untrav = (unsigned char*)malloc((size_t)g->rows * g->cols);
  if (!untrav) return -2;
  untraversable_cells(g, p, elev, slope, step, rough, untrav, NULL, NULL, NULL);
  double off[2][128];
  teo_rotate_footprint(n_points, points_xy, 0.0, off[0]);
  teo_rotate_footprint(n_points, points_xy, yaw, off[1]);
#pragma omp parallel for schedule(dynamic, 4) num_threads(g_threads)
  for (int j = 0; j < g->cols; ++j) {
    for (int i = 0; i < g->rows; ++i) {
      const double cx = cell_x(g, i), cy = cell_y(g, j);
      for (int which = 0; which < 2; ++which) {
        double v[128], t;
        for (int k = 0; k < n_points; ++k) {
          v[2 * k] = off[which][2 * k] + cx;
          v[2 * k + 1] = off[which][2 * k + 1] + cy;
        }
        const int ok = polygon_traversable(g, untrav, trav, p->fp_default, n_points, v, &t);
        (which ? trav_rot : trav_x)[IDX(g, i, j)] = ok ? (float)t : 0.0f; /* :293-300 */
      }
    }
  }
  free(untrav);
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* N2 (polygon half)  checkPolygonalFootprintPath :464-584 with publishPolygons == false,         */
/*     compute_untraversable_polygon == false, check_robot_inclination == false.                 */
/*     grid_map::Polygon::convexHull / monotoneChainConvexHullOfPoints / getArea and Eigen's     */
/*     Translation * Quaternion * Vector3 are restated from the published sources (unpinned).    */
/* ------------------------------------------------------------------------------------------- */
typedef struct pt2 {
  double x, y;
} pt2;

static int pt_less(const void* a, const void* b) { /* sortVertices: x, then y */
  const pt2 *p = (const pt2*)a, *q = (const pt2*)b;
  if (p->x < q->x || (p->x == q->x && p->y < q->y)) return -1;
  if (q->x < p->x || (q->x == p->x && q->y < p->y)) return 1;
  return 0;
}

/* vectorsMakeClockwiseTurn(pivot, v1, v2): cross(v1 - pivot, v2 - pivot) <= 0 */
static int clockwise(pt2 o, pt2 a, pt2 b) {
  const double ax = a.x - o.x, ay = a.y - o.y, bx = b.x - o.x, by = b.y - o.y;
  return ax * by - bx * ay <= 0.0;
}

/* monotoneChainConvexHullOfPoints; hull must hold 2*n points; returns the number of hull vertices */
static int convex_hull(int n, const pt2* pts, pt2* sorted, pt2* hull) {
  if (n <= 3) {
    memcpy(hull, pts, sizeof(pt2) * (size_t)n);
    return n;
  }
  memcpy(sorted, pts, sizeof(pt2) * (size_t)n);
  qsort(sorted, (size_t)n, sizeof(pt2), pt_less);
  int k = 0;
  for (int i = 0; i < n; ++i) { /* lower hull */
    while (k >= 2 && clockwise(hull[k - 2], hull[k - 1], sorted[i])) k--;
    hull[k++] = sorted[i];
  }
  for (int i = n - 2, t = k + 1; i >= 0; i--) { /* upper hull */
    while (k >= t && clockwise(hull[k - 2], hull[k - 1], sorted[i])) k--;
    hull[k++] = sorted[i];
  }
  return k - 1;
}

static double polygon_area(int n, const pt2* v) { /* Polygon::getArea */
  double area = 0.0;
  int j = n - 1;
  for (int i = 0; i < n; i++) {
    area += (v[j].x + v[i].x) * (v[j].y - v[i].y);
    j = i;
  }
  return fabs(area / 2.0);
}

/* isTraversable(polygon, computeUntraversablePolygon = true, traversability, untraversablePolygon) :592-645
 * (FootprintPath.compute_untraversable_polygon): every cell of the polygon is visited, the positions of the untraversable
 * ones are collected in PolygonIterator order, and the untraversable polygon is their monotoneChainConvexHullOfPoints
 * (no vertices when the polygon is traversable).  *traversability as in polygon_traversable (0 when untraversable: the
 * reference leaves the partial sum, which every caller discards).  Returns -3 when the hull has more than cap vertices. */
int teo_polygon_untraversable_hull(const teo_geom* g, const teo_params* p, const float* elev, const float* slope,
                                   const float* step, const float* rough, const float* trav, int n, const double* v,
                                   unsigned char* is_traversable, double* traversability, int cap, int* n_hull,
                                   double* hull_xy) {
  if (n < 1) return -1;
  const size_t N = (size_t)g->rows * g->cols;
  unsigned char* untrav = (unsigned char*)malloc(N);
  if (!untrav) return -2;
  untraversable_cells(g, p, elev, slope, step, rough, untrav, NULL, NULL, NULL);
  double tlx = v[0], tly = v[1], brx = v[0], bry = v[1];
  for (int k = 1; k < n; ++k) {
    tlx = tlx < v[2 * k] ? v[2 * k] : tlx;
    tly = tly < v[2 * k + 1] ? v[2 * k + 1] : tly;
    brx = v[2 * k] < brx ? v[2 * k] : brx;
    bry = v[2 * k + 1] < bry ? v[2 * k + 1] : bry;
  }
  tlx = bound_axis(tlx, g->len_x, g->pos_x);
  tly = bound_axis(tly, g->len_y, g->pos_y);
  brx = bound_axis(brx, g->len_x, g->pos_x);
  bry = bound_axis(bry, g->len_y, g->pos_y);
  int ti, tj, bi, bj;
  pos_to_index(g, tlx, tly, &ti, &tj);
  pos_to_index(g, brx, bry, &bi, &bj);
  ti = clampi(ti, 0, g->rows - 1);
  bi = clampi(bi, 0, g->rows - 1);
  tj = clampi(tj, 0, g->cols - 1);
  bj = clampi(bj, 0, g->cols - 1);
  const size_t box = (size_t)(bi - ti + 1) * (size_t)(bj - tj + 1);
  pt2* pts = (pt2*)malloc(sizeof(pt2) * (4 * box + 4));
  if (!pts) {
    free(untrav);
    return -2;
  }
  pt2 *sorted = pts + box, *hull = pts + 2 * box;
  unsigned ncells = 0;
  int nbad = 0;
  double t = 0.0;
  for (int a = ti; a <= bi; ++a) {
    const double px = cell_x(g, a);
    for (int b = tj; b <= bj; ++b) {
      if (!polygon_inside(n, v, px, cell_y(g, b))) continue;
      const size_t o = IDX(g, a, b);
      if (untrav[o]) { /* :603-609 getPosition of the cell */
        pts[nbad].x = px;
        pts[nbad].y = cell_y(g, b);
        nbad++;
      } else {
        ncells++;
        t += finitef(trav[o]) ? (double)trav[o] : p->fp_default;
      }
    }
  }
  int ok = nbad == 0;
  double value = 0.0;
  if (ok) { /* :624-632 */
    if (ncells == 0) {
      value = p->fp_default;
      ok = p->fp_default != 0.0;
    } else {
      value = t / ncells;
    }
  }
  *is_traversable = (unsigned char)ok;
  *traversability = ok ? value : 0.0;
  int rc = 0;
  *n_hull = 0;
  if (!ok) { /* :634-640 */
    const int nh = convex_hull(nbad, pts, sorted, hull);
    if (nh > cap) {
      rc = -3;
    } else {
      *n_hull = nh;
      for (int k = 0; k < nh; ++k) {
        hull_xy[2 * k] = hull[k].x;
        hull_xy[2 * k + 1] = hull[k].y;
      }
    }
  }
  free(pts);
  free(untrav);
  return rc;
}

#define TEO_MAX_PATH_VERTS 1024 /* vertices of one (conservative) pose polygon */

int teo_check_polygon_paths_incl(const teo_geom* g, const teo_params* p, const float* elev, const float* slope,
                                 const float* step, const float* rough, const float* trav, const float* robot_slope,
                                 int n_paths, const int* pose_offset, const double* poses, int n_points,
                                 const double* points_xyz, const unsigned char* conservative, unsigned char* is_safe,
                                 double* traversability, double* area, int* status) {
  if (n_points < 1 || n_points > 32) return -1;
  unsigned char* untrav = (unsigned char*)malloc((size_t)g->rows * g->cols);
  if (!untrav) return -2;
  untraversable_cells(g, p, elev, slope, step, rough, untrav, NULL, NULL, NULL);
  for (int k = 0; k < n_paths; ++k) {
    const int n = pose_offset[k + 1] - pose_offset[k];
    is_safe[k] = 0;
    traversability[k] = 0.0;
    area[k] = 0.0;
    status[k] = 0;
    if (n <= 0) { /* :330-334 */
      status[k] = 2;
      continue;
    }
    pt2 poly1[TEO_MAX_PATH_VERTS], poly2[TEO_MAX_PATH_VERTS], all[2 * TEO_MAX_PATH_VERTS], sorted[2 * TEO_MAX_PATH_VERTS],
        hull[4 * TEO_MAX_PATH_VERTS];
    int n1 = 0, n2 = 0, ok = 1;
    double sx, sy, ex = 0.0, ey = 0.0, t = 0.0;
    for (int i = 0; i < n && ok; ++i) {
      const double* q = poses + 7 * (size_t)(pose_offset[k] + i);
      memcpy(poly1, poly2, sizeof(pt2) * (size_t)n2); /* polygon1 = polygon2 */
      n1 = n2;
      sx = ex;
      sy = ey;
      ex = q[0];
      ey = q[1];
      { /* toPosition * orientation * positionToVertex: Quaternion::toRotationMatrix, linear * v + translation */
        const double x = q[3], y = q[4], z = q[5], w = q[6];
        const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
        const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
                     tyz = tz * y, tzz = tz * z;
        const double r00 = 1.0 - (tyy + tzz), r01 = txy - twz, r02 = txz + twy;
        const double r10 = txy + twz, r11 = 1.0 - (txx + tzz), r12 = tyz - twx;
        for (int m = 0; m < n_points; ++m) {
          const double px = points_xyz[3 * m], py = points_xyz[3 * m + 1], pz = points_xyz[3 * m + 2];
          poly2[m].x = ((r00 * px + r01 * py) + r02 * pz) + q[0];
          poly2[m].y = ((r10 * px + r11 * py) + r12 * pz) + q[1];
        }
        n2 = n_points;
      }
      if (conservative && conservative[k] && i > 0) { /* :512-522 */
        const double dx = ex - sx, dy = ey - sy;
        const int m1 = n1, m2 = n2;
        for (int m = 0; m < m1; ++m) {
          poly2[n2].x = poly1[m].x + dx;
          poly2[n2].y = poly1[m].y + dy;
          n2++;
        }
        for (int m = 0; m < m2; ++m) {
          poly1[n1].x = poly2[m].x - dx;
          poly1[n1].y = poly2[m].y - dy;
          n1++;
        }
      }
      if (robot_slope && (n == 1 || i > 0)) { /* :526-528, :553-557 checkRobotInclination_ */
        int outside;
        const int good = n == 1 ? inclination_ok(g, robot_slope, ex, ey, ex, ey, &outside)
                                : inclination_ok(g, robot_slope, sx, sy, ex, ey, &outside);
        if (!good) {
          status[k] = outside;
          ok = 0;
          break;
        }
      }
      if (n == 1) { /* :524-546 */
        if (!polygon_traversable(g, untrav, trav, p->fp_default, n2, (const double*)poly2, &t)) {
          ok = 0;
          break;
        }
        traversability[k] = t;
        area[k] = polygon_area(n2, poly2);
      }
      if (n > 1 && i > 0) { /* :548-579 */
        memcpy(all, poly1, sizeof(pt2) * (size_t)n1);
        memcpy(all + n1, poly2, sizeof(pt2) * (size_t)n2);
        const int nh = convex_hull(n1 + n2, all, sorted, hull);
        if (!polygon_traversable(g, untrav, trav, p->fp_default, nh, (const double*)hull, &t)) {
          ok = 0;
          break;
        }
        if (i > 1) {
          const double area_previous = area[k];
          const double area_polygon = polygon_area(nh, hull) - polygon_area(n1, poly1);
          area[k] += area_polygon;
          traversability[k] = (area_polygon * t + area_previous * traversability[k]) / area[k];
        } else {
          area[k] = polygon_area(nh, hull);
          traversability[k] = t;
        }
      }
      /* the conservative vertex lists grow by n_points with every pose; the reference grows without bound, here the
       * path is refused once the next pose would not fit */
      if (conservative && conservative[k] && n2 + n_points > TEO_MAX_PATH_VERTS && i + 1 < n) {
        status[k] = 3;
        ok = 0;
      }
    }
    if (ok) is_safe[k] = 1; /* on failure the partial traversability / area stay, like result in the reference */
  }
  free(untrav);
  return 0;
}

int teo_check_polygon_paths(const teo_geom* g, const teo_params* p, const float* elev, const float* slope, const float* step,
                            const float* rough, const float* trav, int n_paths, const int* pose_offset, const double* poses,
                            int n_points, const double* points_xyz, const unsigned char* conservative, unsigned char* is_safe,
                            double* traversability, double* area, int* status) {
  return teo_check_polygon_paths_incl(g, p, elev, slope, step, rough, trav, NULL, n_paths, pose_offset, poses, n_points,
                                      points_xyz, conservative, is_safe, traversability, area, status);
}
