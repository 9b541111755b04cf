/*
 * sgym_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see sgym_oracle.h).
 *
 * Plain scalar C restatement of the reference's per-step path.  Every function cites the
 * reference file:line (relative to /root/reference) it follows.  Compiled with
 * -ffp-contract=off: every fused multiply-add below is an explicit fma() that mirrors a place
 * where the reference's numpy/OpenBLAS stack fuses (np.linalg.norm of a short vector).
 */
#include "sgym_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXE 16384

/* ------------------------------------------------------------------------------------------
 * Counter build (-DSGO_COUNT_FLOPS -> _build/libsgym_oracle_count.so; tools/count_flops.py): the ALGORITHMIC fp64
 * operations of the path, per category, for bench.py's vector-ALU roofline (SURVEY 8d "Algorithmic flops").
 * Convention: add / sub / mul / div / sqrt / rint / compare-select (min, max, an ordered compare that decides) = 1, fma = 2;
 * index arithmetic, searches and copies = 0.  Each site adds the count of the MINIMAL formulation of what it computes --
 * slopes of a knot segment, and terms that depend on one entity only, are prepared once (per segment / per entity-step),
 * not per use -- times nothing: the number of calls is what the run measures.  The all-pairs searches (collision broad
 * phase, pedestrian neighbour search) are charged SURVEY 8d's 6 flops (dx, dy, two products, sum, compare) per UNORDERED
 * pair of present entities, and are reported apart so that a reader can leave them out.
 * ---------------------------------------------------------------------------------------- */
#ifdef SGO_COUNT_FLOPS
enum { FL_LERP, FL_STATS, FL_SINCOS, FL_CORNERS, FL_BROAD, FL_SAT, FL_CTRL, FL_METRIC, FL_PED_GOAL, FL_PED_ENTITY,
       FL_PED_PAIR, FL_PED_MOVE, FL_ENTITY_STEPS, FL_NCAT };
static unsigned long long sgo_flops[FL_NCAT];
#define FLOPS(c, n) (sgo_flops[c] += (unsigned long long)(n))
/* out[FL_NCAT]; the last entry counts entity-steps (entities x executed steps); reset != 0 clears the counters */
int sgo_flops_read(unsigned long long *out, int reset)
{
    for (int i = 0; i < FL_NCAT; ++i) { out[i] = sgo_flops[i]; if (reset) sgo_flops[i] = 0; }
    return FL_NCAT;
}
#else
#define FLOPS(c, n) ((void)0)
#endif
/* per-call constants (derivations beside each site) */
#define FLN_LERP 13    /* dq = t - x_lo; 6 x (slope * dq + y_lo) */
#define FLN_STATS 19   /* 6 sub, 6 div, norm3 = mul + 2 fma + sqrt, distance += */
#define FLN_SINCOS 49  /* sgo_sincos: Cody-Waite reduction 12, sin kernel 18, cos kernel 19 */
#define FLN_CORNERS 40 /* 4 half-extent offsets x 2, 8 coordinates x (2 mul + 2 add) */
#define FLN_PAIR_SEARCH 6
#define FLN_PID 29     /* errors 8, gain 4, derivative / integral terms 6, steer 3, |e_lon| test 1, accel 5, gains 2 */
#define FLN_VEHICLE 48 /* clips 4, dx dy 2, tan polynomial 30, dh 2, pose += 6, speed 2 + 2 clamps */
#define FLN_METRIC 12  /* norm3 6, w 1, running mean 4, max 1 */
#define FLN_PED_FORCE_GOAL 16 /* g 2, |g| 4, test 1, 1/tau 1, 2 x (div, mul, sub, mul) */
#define FLN_PED_ENTITY 8      /* per pedestrian that is somebody's neighbour: |v| + 1e-10 (5), unit velocity 2, step 1 */
#define FLN_PED_PAIR 91       /* r 2, |r| 4, q 4, |q| 5, sum 1, b 5, k1 3, db 8, -b/sigma 1, exp 25, k2 1, rep 2, att 2,
                                 two sight weights 2 x 10, 2 x (w * f) 4, force += 4 */
#define FLN_PED_MOVE 44       /* |F| 4 + 1, speed cap 2, atan2 30, clip 2, sd 1, pose += 4 */

/* ------------------------------------------------------------------------------------------
 * log for the Box-Muller transform of noise_mode 2: the classic Sun fdlibm __ieee754_log restated
 * (argument reduction to [sqrt(2)/2, sqrt(2)], degree-14 minimax in s = f / (2 + f)); domain: finite
 * normal x > 0.  Shared by restatement with the device (sg_log): plain add / mul / div only.
 * ---------------------------------------------------------------------------------------- */
double sgo_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int32_t hx = (int32_t)(bits >> 32);
    int k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int32_t i0 = (hx + 0x95f64) & 0x100000;
    bits = ((uint64_t)(uint32_t)(hx | (i0 ^ 0x3ff00000)) << 32) | (bits & 0xffffffffu); /* normalize x or x/2 */
    memcpy(&x, &bits, 8);
    k += i0 >> 20;
    const double f = x - 1.0, dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) { /* |f| < 2**-20 */
        if (f == 0.0) return k == 0 ? 0.0 : dk * ln2_hi + dk * ln2_lo;
        const double R = f * f * (0.5 - 0.33333333333333333 * f);
        return k == 0 ? f - R : dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    const double s = f / (2.0 + f), z = s * s, w = z * z;
    const int32_t i = (hx - 0x6147a) | (0x6b851 - hx);
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6)), t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    if (i > 0) {
        const double hfsq = 0.5 * f * f;
        return k == 0 ? f - (hfsq - s * (hfsq + R)) : dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    return k == 0 ? f - s * (f - R) : dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

/* Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): counter (entity, step, 0, 0),
 * key (seed_lo ^ scenario, seed_hi).  Two 53-bit uniforms in (0, 1) from the four output words, Box-Muller. */
void sgo_noise_pair(uint64_t seed, uint32_t scenario, uint32_t entity, uint32_t step, double *z2)
{
    uint32_t c0 = entity, c1 = step, c2 = 0, c3 = 0, k0 = (uint32_t)seed ^ scenario, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const double u1 = ((double)((((uint64_t)c0 << 32) | c1) >> 11) + 0.5) * 0x1p-53;
    const double u2 = ((double)((((uint64_t)c2 << 32) | c3) >> 11) + 0.5) * 0x1p-53;
    const double r = sqrt(-2.0 * sgo_log(u1));
    double sn, cs;
    sgo_sincos(6.28318530717958623200e+00 * u2, &sn, &cs);
    z2[0] = r * cs;
    z2[1] = r * sn;
}

/* ------------------------------------------------------------------------------------------
 * sin/cos.  numpy's np.cos/np.sin (entity/base.py:113, controller.py:126-127) are platform
 * SIMD routines accurate to <1 ulp but not bit-reproducible across libms.  The oracle and the
 * HIP kernels therefore both use this fixed plain-fp64 algorithm (two-step Cody-Waite
 * reduction by pi/2 and the classic Sun fdlibm minimax kernels, restated), which is within
 * 1 ulp of the reference and identical bit-for-bit wherever IEEE fp64 add/mul is.
 * ---------------------------------------------------------------------------------------- */
static const double INV_PIO2 = 6.36619772367581382433e-01;
static const double PIO2_1 = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
static const double PIO2_2 = 6.07710050630396597660e-11; /* next 33 bits */
static const double PIO2_2T = 2.02226624879595063154e-21;
static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                    S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                    S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                    C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                    C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;

void sgo_sincos(double x, double *s, double *c)
{
    FLOPS(FL_SINCOS, FLN_SINCOS);
    if (!(fabs(x) < 1.0e5)) { /* huge / inf / nan headings: defer to libm */
        *s = sin(x);
        *c = cos(x);
        return;
    }
    double fn = rint(x * INV_PIO2);
    int n = (int)fn;
    double t = x - fn * PIO2_1;
    double w = fn * PIO2_2;
    double r = t - w;
    w = fn * PIO2_2T - ((t - r) - w);
    double y0 = r - w;
    double y1 = (r - y0) - w;

    double z = y0 * y0;
    double v = z * y0;
    double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);

    double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double ay = fabs(y0);
    double kc;
    if (ay < 0.3) {
        kc = 1.0 - (0.5 * z - (z * rc - y0 * y1));
    } else {
        double qx;
        if (ay > 0.78125) {
            qx = 0.28125;
        } else {
            uint64_t b;
            memcpy(&b, &ay, 8);
            b = (b - ((uint64_t)0x00200000 << 32)) & 0xFFFFFFFF00000000ULL;
            memcpy(&qx, &b, 8);
        }
        double hz = 0.5 * z - qx;
        double a = 1.0 - qx;
        kc = a - (hz - (z * rc - y0 * y1));
    }
    switch (n & 3) {
    case 0: *s = ks; *c = kc; break;
    case 1: *s = kc; *c = -ks; break;
    case 2: *s = -ks; *c = -kc; break;
    default: *s = -kc; *c = ks; break;
    }
}

/* tan(steer) of VehicleController._step (controller.py:128, np.tan).  |x| < 0.67434: the classic odd
 * minimax polynomial on the primary interval (fdlibm kernel, restated; < 1 ulp); otherwise sin/cos. */
static const double TN[13] = {
    3.33333333333334091986e-01, 1.33333333333201242699e-01, 5.39682539762260521377e-02,
    2.18694882948595424599e-02, 8.86323982359930005737e-03, 3.59207910759131235356e-03,
    1.45620945432529025516e-03, 5.88041240820264096874e-04, 2.46463134818469906812e-04,
    7.81794442939557092300e-05, 7.14072491382608190305e-05, -1.85586374855275456654e-05,
    2.59073051863633712884e-05,
};

double sgo_tan(double x)
{
    if (!(fabs(x) < 0.67434)) {
        double s, c;
        sgo_sincos(x, &s, &c);
        return s / c;
    }
    double z = x * x;
    double w = z * z;
    double r = TN[1] + w * (TN[3] + w * (TN[5] + w * (TN[7] + w * (TN[9] + w * TN[11]))));
    double v = z * (TN[2] + w * (TN[4] + w * (TN[6] + w * (TN[8] + w * (TN[10] + w * TN[12])))));
    double s = z * x;
    r = z * (s * (r + v));
    r = r + TN[0] * s;
    return x + r;
}

/* exp / atan2 used by the social force model (np.exp pedestrian/social_force.py:172, np.arctan2 :113).
 * Fixed plain-fp64 algorithms (classic fdlibm reductions and minimax polynomials, restated) shared with
 * the HIP kernels so that CPU and GPU agree bit-for-bit; < 1 ulp from numpy's. */
double sgo_exp(double x)
{
    static const double LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10,
                        INVLN2 = 1.44269504088896338700e+00;
    static const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                        P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                        P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 709.782712893383973096) return INFINITY;
    if (x < -745.13321910194110842) return 0.0;
    double k = rint(x * INVLN2);
    double hi = x - k * LN2HI;
    double lo = k * LN2LO;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    return ldexp(y, (int)k);
}

static double atan_pos(double ax) /* atan of a non-negative finite or infinite argument */
{
    static const double HI[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01,
                                 9.82793723247329054082e-01, 1.57079632679489655800e+00};
    static const double LO[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17,
                                 1.39033110312309984516e-17, 6.12323399573676603587e-17};
    static const double A[11] = {3.33333333333329318027e-01,  -1.99999999998764832476e-01,
                                 1.42857142725034663711e-01,  -1.11111104054623557880e-01,
                                 9.09088713343650656196e-02,  -7.69187620504482999495e-02,
                                 6.66107313738753120669e-02,  -5.83357013379057348645e-02,
                                 4.97687799461593236017e-02,  -3.65315727442169155270e-02,
                                 1.62858201153657823623e-02};
    if (ax >= 7.378697629483821e19) return HI[3] + LO[3]; /* 2^66 */
    int id;
    double x;
    if (ax < 0.4375) { id = -1; x = ax; }
    else if (ax < 0.6875) { id = 0; x = (2.0 * ax - 1.0) / (2.0 + ax); }
    else if (ax < 1.1875) { id = 1; x = (ax - 1.0) / (ax + 1.0); }
    else if (ax < 2.4375) { id = 2; x = (ax - 1.5) / (1.0 + 1.5 * ax); }
    else { id = 3; x = -1.0 / ax; }
    double z = x * x, w = z * z;
    double s1 = z * (A[0] + w * (A[2] + w * (A[4] + w * (A[6] + w * (A[8] + w * A[10])))));
    double s2 = w * (A[1] + w * (A[3] + w * (A[5] + w * (A[7] + w * A[9]))));
    if (id < 0) return x - x * (s1 + s2);
    return HI[id] - ((x * (s1 + s2) - LO[id]) - x);
}

double sgo_atan2(double y, double x)
{
    static const double PI = 3.1415926535897931160E+00, PI_LO = 1.2246467991473531772E-16;
    if (x != x || y != y) return x + y;
    if (y == 0.0) return (x < 0.0 || (x == 0.0 && signbit(x))) ? copysign(PI, y) : y;
    if (x == 0.0) return copysign(0.5 * PI, y);
    double z = atan_pos(fabs(y / x));
    if (x > 0.0) return y > 0.0 ? z : -z;
    z = PI - (z - PI_LO);
    return y > 0.0 ? z : -z;
}

/* np.linalg.norm of a 2-/3-vector = sqrt(x.dot(x)); the OpenBLAS ddot tail loop on x86-64 is an
 * FMA chain (probed against numpy 2.2.6 / OpenBLAS 0.3.29: 20000/20000 bitwise matches). */
static double norm2(double a, double b) { return sqrt(fma(b, b, a * a)); }
static double norm3(double a, double b, double c) { return sqrt(fma(c, c, fma(b, b, a * a))); }

/* ------------------------------------------------------------------------------------------
 * scipy.interpolate.interp1d(kind="linear")._call_linear
 * (scipy/interpolate/_interpolate.py:457-481): idx = clip(searchsorted_left(x, xn), 1, n-1),
 * slope = (y_hi - y_lo)/(x_hi - x_lo), y = slope*(xn - x_lo) + y_lo.
 * y is [n][stride] row-major; m channels are evaluated.
 * ---------------------------------------------------------------------------------------- */
static int searchsorted_left(const double *x, int n, double v)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (x[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

static void call_linear(const double *x, const double *y, int n, int stride, int m, double xn,
                        double *out)
{
    int idx = searchsorted_left(x, n, xn);
    if (idx < 1) idx = 1;
    if (idx > n - 1) idx = n - 1;
    double x_lo = x[idx - 1], x_hi = x[idx];
    const double *y_lo = y + (size_t)(idx - 1) * stride, *y_hi = y + (size_t)idx * stride;
    for (int c = 0; c < m; ++c) {
        double slope = (y_hi[c] - y_lo[c]) / (x_hi - x_lo);
        out[c] = slope * (xn - x_lo) + y_lo[c];
    }
}

/* Trajectory.position_at_t, scenario_gym/trajectory.py:142-205 (scalar-t branch :188-196).
 * knots: [n][7].  none_outside <=> extrapolate=False.  Returns 0 for None. */
int sgo_position_at_t(const double *knots, int n, double t, int ext_bck, int ext_fwd,
                      int none_outside, double *out)
{
    double min_t = knots[0], max_t = knots[(size_t)(n - 1) * 7];
    if (none_outside && (t < min_t || t > max_t)) return 0;
    if (t < min_t && !ext_bck) { memcpy(out, knots + 1, 48); return 1; }
    if (t > max_t && !ext_fwd) { memcpy(out, knots + (size_t)(n - 1) * 7 + 1, 48); return 1; }
    if (n == 1) { /* trajectory.py:175-177: duplicate the knot at t + 1e-3 */
        double x[2] = {knots[0], knots[0] + 1e-3};
        double y[12];
        memcpy(y, knots + 1, 48);
        memcpy(y + 6, knots + 1, 48);
        call_linear(x, y, 2, 6, 6, t, out);
        return 1;
    }
    /* x = column 0 (stride 7): gather times */
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (knots[(size_t)mid * 7] < t) lo = mid + 1; else hi = mid;
    }
    int idx = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
    const double *a = knots + (size_t)(idx - 1) * 7, *b = knots + (size_t)idx * 7;
    for (int c = 0; c < 6; ++c) {
        double slope = (b[1 + c] - a[1 + c]) / (b[0] - a[0]);
        out[c] = slope * (t - a[0]) + a[1 + c];
    }
    FLOPS(FL_LERP, FLN_LERP);
    return 1;
}

/* Trajectory.velocity_at_t, scenario_gym/trajectory.py:243-273 (eps = 1e-4) */
void sgo_velocity_at_t(const double *knots, int n, double t, double *out)
{
    const double eps = 1e-4;
    double min_t = knots[0], max_t = knots[(size_t)(n - 1) * 7];
    double a[6], b[6];
    sgo_position_at_t(knots, n, t + eps / 2, 1, 1, 0, a);
    sgo_position_at_t(knots, n, t - eps / 2, 1, 1, 0, b);
    int inside = (min_t <= t) && (t <= max_t);
    for (int c = 0; c < 6; ++c) out[c] = inside ? (a[c] - b[c]) / eps : 0.0;
}

/* ------------------------------------------------------------------------------------------
 * BatchReplayEntity.add_entities, scenario_gym/entity/batch.py:55-128: union knot grid `ts`,
 * stage-1 resample of every trajectory onto it (fill = first/last knot), stage-2 interp1d on ts.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int n_ent;  /* batched entities */
    int n_grid; /* len(ts) */
    double *ts; /* [N] */
    double *X;  /* [N][n_ent*6] */
} batch_t;

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

static void batch_build(batch_t *B, const int64_t *knot_off, const double *knots, const int *ids,
                        int n_ent)
{
    B->n_ent = n_ent;
    B->n_grid = 0;
    B->ts = NULL;
    B->X = NULL;
    if (n_ent == 0) return;
    size_t cap = 0;
    for (int i = 0; i < n_ent; ++i) {
        int n = (int)(knot_off[ids[i] + 1] - knot_off[ids[i]]);
        cap += n == 1 ? 2 : n;
    }
    double *all = (double *)malloc(cap * sizeof(double));
    size_t k = 0;
    for (int i = 0; i < n_ent; ++i) {
        const double *d = knots + knot_off[ids[i]] * 7;
        int n = (int)(knot_off[ids[i] + 1] - knot_off[ids[i]]);
        for (int j = 0; j < n; ++j) all[k++] = d[(size_t)j * 7];
        if (n == 1) all[k++] = d[0] + 1e-1; /* batch.py:85-88 */
    }
    qsort(all, k, sizeof(double), cmp_double);
    size_t N = 0;
    for (size_t i = 0; i < k; ++i)
        if (i == 0 || all[i] != all[N - 1]) all[N++] = all[i];
    B->n_grid = (int)N;
    B->ts = all;
    B->X = (double *)malloc(N * n_ent * 6 * sizeof(double));
    for (int i = 0; i < n_ent; ++i) {
        const double *d = knots + knot_off[ids[i]] * 7;
        int n = (int)(knot_off[ids[i] + 1] - knot_off[ids[i]]);
        double two[14];
        if (n == 1) {
            memcpy(two, d, 56);
            memcpy(two + 7, d, 56);
            two[7] += 1e-1;
            d = two;
            n = 2;
        }
        /* x = d[:,0], y = d[:,1:] -> copy to contiguous arrays for call_linear */
        double *x = (double *)malloc(n * sizeof(double));
        for (int j = 0; j < n; ++j) x[j] = d[(size_t)j * 7];
        for (size_t g = 0; g < N; ++g) {
            double *o = B->X + (g * n_ent + i) * 6;
            double tq = all[g];
            if (tq < x[0]) memcpy(o, d + 1, 48);                       /* fill below: d[0,1:] */
            else if (tq > x[n - 1]) memcpy(o, d + (size_t)(n - 1) * 7 + 1, 48); /* above */
            else call_linear(x, d + 1, n, 7, 6, tq, o);
        }
        free(x);
    }
}

static void batch_free(batch_t *B)
{
    free(B->ts);
    free(B->X);
}

/* self.fn(t), batch.py:122-128: interp1d(ts, X.T, fill=(X[0], X[-1])) */
static void batch_eval(const batch_t *B, double t, double *out /*[n_ent][6]*/)
{
    int m = B->n_ent * 6;
    if (t < B->ts[0]) memcpy(out, B->X, m * sizeof(double));
    else if (t > B->ts[B->n_grid - 1])
        memcpy(out, B->X + (size_t)(B->n_grid - 1) * m, m * sizeof(double));
    else call_linear(B->ts, B->X, B->n_grid, m, m, t, out);
    FLOPS(FL_LERP, (unsigned long long)FLN_LERP * (unsigned)B->n_ent); /* (clamped rows: charged alike, the GPU lane interpolates a constant piece) */
}

void sgo_batch_eval(const int64_t *knot_off, const double *knots, int n_ent, int persist,
                    const double *ts, int n_t, double *out, uint8_t *present)
{
    int *ids = (int *)malloc(n_ent * sizeof(int));
    for (int i = 0; i < n_ent; ++i) ids[i] = i;
    batch_t B;
    batch_build(&B, knot_off, knots, ids, n_ent);
    for (int q = 0; q < n_t; ++q) {
        batch_eval(&B, ts[q], out + (size_t)q * n_ent * 6);
        for (int i = 0; i < n_ent; ++i) { /* BatchReplayEntity.step, batch.py:34-53 */
            const double *d = knots + knot_off[i] * 7;
            int n = (int)(knot_off[i + 1] - knot_off[i]);
            present[(size_t)q * n_ent + i] =
                persist || n == 1 || (ts[q] >= d[0] && ts[q] <= d[(size_t)(n - 1) * 7]);
        }
    }
    batch_free(&B);
    free(ids);
}

/* ------------------------------------------------------------------------------------------
 * Entity.get_bounding_box_points, scenario_gym/entity/base.py:100-138: corners RR, FR, FL, RL.
 * out[i] = xy + points[i] @ [[c, s], [-s, c]]
 * ---------------------------------------------------------------------------------------- */
void sgo_corners(const double *pose, const double *bbox, double *out)
{
    double W = bbox[0], L = bbox[1], cx = bbox[2], cy = bbox[3];
    double s, c;
    sgo_sincos(pose[3], &s, &c);
    FLOPS(FL_CORNERS, FLN_CORNERS);
    double px[4] = {cx - 0.5 * L, cx + 0.5 * L, cx + 0.5 * L, cx - 0.5 * L};
    double py[4] = {cy + 0.5 * W, cy + 0.5 * W, cy - 0.5 * W, cy - 0.5 * W};
    for (int i = 0; i < 4; ++i) {
        out[2 * i] = pose[0] + (px[i] * c + py[i] * (-s));
        out[2 * i + 1] = pose[1] + (px[i] * s + py[i] * c);
    }
}

/* Closed-set intersection of two convex quads = shapely `intersects` predicate used by
 * detect_geom_collisions (scenario_gym/utils.py:52-59).  Separating-axis test over the 8 edge
 * lines: separated iff some edge has every vertex of the other quad strictly outside.
 * Orientation-agnostic (sign of the quad's own doubled area picks the outer side). */
static double cross2(double ex, double ey, double dx, double dy) { return ex * dy - ey * dx; }

#ifdef SGO_COUNT_FLOPS
static int sgo_sat_count = 1; /* detect_collisions evaluates both (i, j) and (j, i): the predicate is symmetric, one is charged */
#endif
int sgo_quads_intersect(const double *A, const double *B)
{
    /* counter build: as executed, with the early exits -- 7 per pass (orientation), 2 per edge, 6 per vertex tested */
    for (int pass = 0; pass < 2; ++pass) {
        const double *P = pass ? B : A, *Q = pass ? A : B;
        /* orientation from the diagonal cross product (doubled area) */
        double o = cross2(P[4] - P[0], P[5] - P[1], P[6] - P[2], P[7] - P[3]);
        FLOPS(FL_SAT, sgo_sat_count ? 7 : 0);
        for (int i = 0; i < 4; ++i) {
            int j = (i + 1) & 3;
            double ax = P[2 * i], ay = P[2 * i + 1];
            double ex = P[2 * j] - ax, ey = P[2 * j + 1] - ay;
            int all_out = 1;
            FLOPS(FL_SAT, sgo_sat_count ? 2 : 0);
            for (int k = 0; k < 4 && all_out; ++k) {
                double cr = cross2(ex, ey, Q[2 * k] - ax, Q[2 * k + 1] - ay);
                FLOPS(FL_SAT, sgo_sat_count ? 6 : 0);
                /* outside = opposite side to the interior; interior has sign(o) */
                if (o > 0 ? !(cr < 0) : !(cr > 0)) all_out = 0;
            }
            if (all_out) return 0;
        }
    }
    return 1;
}

static int corners_equal(const double *a, const double *b)
{
    for (int i = 0; i < 8; ++i)
        if (a[i] != b[i]) return 0;
    return 1;
}

/* FutureCollisionDetector._step (scenario_gym/sensor/common.py:87-106): the ego's box at
 * trajectory.position_at_t(t_j) against every other entity's box at ITS position_at_t(t_j) (clamped outside the
 * trajectory, presence is not consulted), t_j = np.linspace(t, t + horizon, n): any closed-set overlap whose geometry
 * differs from the ego's (utils.py:59).  np.linspace (numpy/_core/function_base.py): step = (stop - start) / (n - 1),
 * y_j = j * step + start, y_{n-1} = stop. */
int sgo_future_collision(const sgo_scenario *sc, double t, double horizon, int n)
{
    const int E = sc->n_entities, ego = sc->ego;
    const double start = t, stop = t + horizon;
    const double step = n > 1 ? (stop - start) / (double)(n - 1) : 0.0;
    for (int j = 0; j < n; ++j) {
        double tj = (double)j * step + start;
        if (n > 1 && j == n - 1) tj = stop;
        double pe[6], ce[8];
        const double *ke = sc->knots + sc->knot_off[ego] * 7;
        sgo_position_at_t(ke, (int)(sc->knot_off[ego + 1] - sc->knot_off[ego]), tj, 0, 0, 0, pe);
        sgo_corners(pe, sc->bbox + (size_t)ego * 4, ce);
        for (int i = 0; i < E; ++i) {
            if (i == ego || sc->kind[i] == 0) continue;
            double pi[6], ci[8];
            const double *ki = sc->knots + sc->knot_off[i] * 7;
            sgo_position_at_t(ki, (int)(sc->knot_off[i + 1] - sc->knot_off[i]), tj, 0, 0, 0, pi);
            sgo_corners(pi, sc->bbox + (size_t)i * 4, ci);
            if (!corners_equal(ce, ci) && sgo_quads_intersect(ce, ci)) return 1;
        }
    }
    return 0;
}

static double linspace_at(double start, double stop, int n, int j)
{ /* np.linspace: step = (stop - start) / (n - 1); y_j = j * step + start; y_{n-1} = stop */
    if (n > 1 && j == n - 1) return stop;
    const double step = n > 1 ? (stop - start) / (double)(n - 1) : 0.0;
    return (double)j * step + start;
}

/* RasterizedMapSensor, "entity" layer (scenario_gym/sensor/map.py:120-192): out[i][j] = 1 iff the grid point
 *   X[i][j] = (linspace(-width/2, width/2, nw)[j], linspace(-height/2, height/2, nh)[i])
 * rotated by theta = ego heading + pi/2 and moved to the ego (`(X @ R.T) + xy`; numpy's matmul evaluates each
 * coordinate as fma(X1, R.T[1][k], X0 * R.T[0][k]) -- probed 11537/11537) lies strictly inside the bounding box of a
 * present entity (the ego included; shapely `contains`: boundary points are outside).
 * poses [E][6] with NaN x for absent entities. */
void sgo_raster_entities(const double *poses, const double *bbox, int E, int ego, double width, double height, int nw,
                         int nh, uint8_t *out)
{
    const double *pe = poses + (size_t)ego * 6;
    double s, c;
    sgo_sincos(pe[3] + 3.14159265358979311600e+00 / 2, &s, &c);
    double *cor = (double *)malloc((size_t)E * 8 * sizeof(double));
    for (int e = 0; e < E; ++e)
        if (poses[(size_t)e * 6] == poses[(size_t)e * 6]) sgo_corners(poses + (size_t)e * 6, bbox + (size_t)e * 4, cor + (size_t)e * 8);
    for (int i = 0; i < nh; ++i)
        for (int j = 0; j < nw; ++j) {
            const double x0 = linspace_at(-width / 2, width / 2, nw, j), x1 = linspace_at(-height / 2, height / 2, nh, i);
            const double px = fma(x1, -s, x0 * c) + pe[0], py = fma(x1, c, x0 * s) + pe[1];
            int hit = 0;
            for (int e = 0; e < E && !hit; ++e) {
                if (poses[(size_t)e * 6] != poses[(size_t)e * 6]) continue;
                const double *P = cor + (size_t)e * 8;
                const double o = (P[4] - P[0]) * (P[7] - P[3]) - (P[5] - P[1]) * (P[6] - P[2]); /* ring orientation */
                int inside = o != 0;
                for (int k = 0; k < 4 && inside; ++k) {
                    const int m = (k + 1) & 3;
                    const double cr = (P[2 * m] - P[2 * k]) * (py - P[2 * k + 1]) - (P[2 * m + 1] - P[2 * k + 1]) * (px - P[2 * k]);
                    if (o > 0 ? !(cr > 0) : !(cr < 0)) inside = 0;
                }
                hit = inside;
            }
            out[(size_t)i * nw + j] = (uint8_t)hit;
        }
    free(cor);
}

/* ---- road surfaces: point in the union of polygons ------------------------------------------------------------
 * RoadNetwork.driveable_surface & co are unary_union()s of the boundary polygons (road_network.py:306-328) and the
 * callers ask `contains(Point)` (state.py:401-407, sensor/map.py:198-271).  shapely/GEOS is not part of the reference
 * sources; its published algorithm for this predicate is JTS/GEOS RayCrossingCounter with the robust orientation
 * index: a point is inside a polygon iff a ray towards +x crosses its rings an odd number of times, and a point ON a
 * ring is not contained.  Restated here with an exact orientation sign (fp64 filter, then an exact expansion sum),
 * so the answer is the mathematical one for the given fp64 coordinates.  A point is in the union iff it is strictly
 * inside one of the polygons (points on an edge shared by two polygons of the same union are the measure-zero
 * exception: GEOS dissolves such edges, this restatement reports them outside). */
static void two_sum(double a, double b, double *s, double *e)
{
    double x = a + b, bb = x - a;
    *s = x;
    *e = (a - (x - bb)) + (b - bb);
}

/* sign of (ax - px) * (by - py) - (ay - py) * (bx - px), exactly */
static int orient_sign(double ax, double ay, double bx, double by, double px, double py)
{
    const double dl = (ax - px) * (by - py), dr = (ay - py) * (bx - px), det = dl - dr;
    const double bound = 1e-15 * (fabs(dl) + fabs(dr)); /* > (3 + 16 eps) eps of Shewchuk's stage-A bound */
    if (det > bound) return 1;
    if (det < -bound) return -1;
    /* = ax*by - ax*py - px*by - ay*bx + ay*px + py*bx  (px*py cancels): six exact products, summed exactly */
    const double fa[6] = {ax, -ax, -px, -ay, ay, py}, fb[6] = {by, py, by, bx, px, bx};
    double e[12];
    int n = 0;
    for (int k = 0; k < 6; ++k) {
        const double hi = fa[k] * fb[k], lo = fma(fa[k], fb[k], -hi);
        const double term[2] = {lo, hi};
        for (int u = 0; u < 2; ++u) { /* grow-expansion: e stays non-overlapping, increasing magnitude */
            double q = term[u];
            for (int i = 0; i < n; ++i) two_sum(q, e[i], &q, &e[i]);
            e[n++] = q;
        }
    }
    for (int i = n - 1; i >= 0; --i)
        if (e[i] != 0.0) return e[i] > 0 ? 1 : -1;
    return 0;
}

/* RayCrossingCounter.countSegment for one edge: *cross toggles on a crossing; returns 1 if the point is ON the edge */
static int ray_edge(double x1, double y1, double x2, double y2, double px, double py, int *cross)
{
    if (x1 < px && x2 < px) return 0; /* strictly to the left of the point */
    if (px == x2 && py == y2) return 1;
    if (y1 == py && y2 == py) { /* horizontal edge at the ray's height */
        const double lo = x1 < x2 ? x1 : x2, hi = x1 < x2 ? x2 : x1;
        return px >= lo && px <= hi;
    }
    if ((y1 > py && y2 <= py) || (y2 > py && y1 <= py)) {
        int o = orient_sign(x1, y1, x2, y2, px, py);
        if (o == 0) return 1;
        if (y2 < y1) o = -o;
        if (o > 0) *cross ^= 1;
    }
    return 0;
}

static int polygon_contains(const sgo_road_network *net, int k, double x, double y)
{
    int cross = 0;
    for (int64_t r = net->ring_off[k]; r < net->ring_off[k + 1]; ++r) {
        const int64_t a = net->vert_off[r], b = net->vert_off[r + 1];
        for (int64_t i = a; i < b; ++i) {
            const int64_t j = i + 1 < b ? i + 1 : a;
            if (ray_edge(net->verts[2 * i], net->verts[2 * i + 1], net->verts[2 * j], net->verts[2 * j + 1], x, y, &cross))
                return 0; /* on the boundary */
        }
    }
    return cross;
}

int sgo_surface_contains(const sgo_road_network *net, uint32_t layer, double x, double y)
{
    if (!net) return 0;
    for (int k = 0; k < net->n_polygons; ++k)
        if ((net->layers[k] & layer) && polygon_contains(net, k, x, y)) return 1;
    return 0;
}

/* the same for many points; polygons whose bounding box misses the point are skipped (test speed only) */
void sgo_surface_contains_points(const sgo_road_network *net, uint32_t layer, int n, const double *xs, const double *ys,
                                 uint8_t *out)
{
    memset(out, 0, (size_t)n);
    if (!net) return;
    double *bb = (double *)malloc((size_t)(net->n_polygons > 0 ? net->n_polygons : 1) * 4 * sizeof(double));
    for (int k = 0; k < net->n_polygons; ++k) {
        double *b = bb + (size_t)k * 4;
        b[0] = b[1] = INFINITY;
        b[2] = b[3] = -INFINITY;
        for (int64_t i = net->vert_off[net->ring_off[k]]; i < net->vert_off[net->ring_off[k + 1]]; ++i) {
            b[0] = fmin(b[0], net->verts[2 * i]); b[1] = fmin(b[1], net->verts[2 * i + 1]);
            b[2] = fmax(b[2], net->verts[2 * i]); b[3] = fmax(b[3], net->verts[2 * i + 1]);
        }
    }
    for (int q = 0; q < n; ++q)
        for (int k = 0; k < net->n_polygons && !out[q]; ++k) {
            const double *b = bb + (size_t)k * 4;
            if (!(net->layers[k] & layer) || xs[q] < b[0] || xs[q] > b[2] || ys[q] < b[1] || ys[q] > b[3]) continue;
            out[q] = (uint8_t)polygon_contains(net, k, xs[q], ys[q]);
        }
    free(bb);
}

void sgo_raster_map(const double *poses, const double *bbox, int E, int ego, double width, double height, int nw, int nh,
                    const sgo_road_network *net, int n_layers, const int32_t *layers, uint8_t *out)
{
    const double *pe = poses + (size_t)ego * 6;
    double s, c;
    sgo_sincos(pe[3] + 3.14159265358979311600e+00 / 2, &s, &c);
    for (int l = 0; l < n_layers; ++l) {
        uint8_t *o = out + (size_t)l * nw * nh;
        if (layers[l] == 0) {
            sgo_raster_entities(poses, bbox, E, ego, width, height, nw, nh, o);
            continue;
        }
        double *xs = (double *)malloc((size_t)nw * nh * 2 * sizeof(double)), *ys = xs + (size_t)nw * nh;
        for (int i = 0; i < nh; ++i)
            for (int j = 0; j < nw; ++j) {
                const double x0 = linspace_at(-width / 2, width / 2, nw, j), x1 = linspace_at(-height / 2, height / 2, nh, i);
                xs[(size_t)i * nw + j] = fma(x1, -s, x0 * c) + pe[0];
                ys[(size_t)i * nw + j] = fma(x1, c, x0 * s) + pe[1];
            }
        sgo_surface_contains_points(net, (uint32_t)layers[l], nw * nh, xs, ys, o);
        free(xs);
    }
}

/* State.collisions -> detect_collisions -> detect_geom_collisions
 * (scenario_gym/state/state.py:306-310, state/utils.py:10-49, utils.py:28-62).
 * rows[i] bit j set <=> entity j is listed for entity i.  Geometry-equality quirks:
 * equal geometries never list each other (utils.py:59) and a geometry maps back to the LAST
 * entity that owns it (state/utils.py:32-40).  mult[i*E+j] = how often j is listed for i. */
static void detect_collisions(int E, int W, const uint8_t *present, const double *poses,
                              const double *bbox, uint64_t *rows, uint8_t *mult, double *scratch)
{
    double *cor = scratch;                 /* [E][8] */
    double *aabb = scratch + (size_t)E * 8; /* [E][4] */
    int *last = (int *)(scratch + (size_t)E * 12); /* [E] */
    memset(rows, 0, (size_t)E * W * sizeof(uint64_t));
    if (mult) memset(mult, 0, (size_t)E * E);
    for (int i = 0; i < E; ++i) {
        if (!present[i]) continue;
        sgo_corners(poses + (size_t)i * 6, bbox + (size_t)i * 4, cor + (size_t)i * 8);
        double *c = cor + (size_t)i * 8, *b = aabb + (size_t)i * 4;
        b[0] = b[2] = c[0];
        b[1] = b[3] = c[1];
        for (int k = 1; k < 4; ++k) {
            if (c[2 * k] < b[0]) b[0] = c[2 * k];
            if (c[2 * k] > b[2]) b[2] = c[2 * k];
            if (c[2 * k + 1] < b[1]) b[1] = c[2 * k + 1];
            if (c[2 * k + 1] > b[3]) b[3] = c[2 * k + 1];
        }
    }
#ifdef SGO_COUNT_FLOPS
    {
        unsigned long long np = 0;
        for (int i = 0; i < E; ++i) np += present[i] != 0;
        FLOPS(FL_BROAD, FLN_PAIR_SEARCH * (np * (np - (np > 0)) / 2)); /* all unordered pairs of present entities */
    }
#endif
    for (int i = 0; i < E; ++i) {
        if (!present[i]) continue;
        last[i] = i;
        for (int j = i + 1; j < E; ++j)
            if (present[j] && corners_equal(cor + (size_t)i * 8, cor + (size_t)j * 8)) last[i] = j;
    }
    for (int i = 0; i < E; ++i) {
        if (!present[i]) continue;
        for (int j = 0; j < E; ++j) {
            if (j == i || !present[j]) continue;
            const double *a = aabb + (size_t)i * 4, *b = aabb + (size_t)j * 4;
            if (a[2] < b[0] || b[2] < a[0] || a[3] < b[1] || b[3] < a[1]) continue; /* STRtree envelope */
            if (corners_equal(cor + (size_t)i * 8, cor + (size_t)j * 8)) continue;  /* g != g_prime */
#ifdef SGO_COUNT_FLOPS
            sgo_sat_count = j > i;
            const int hit_ = sgo_quads_intersect(cor + (size_t)i * 8, cor + (size_t)j * 8);
            sgo_sat_count = 1;
            if (!hit_) continue;
#else
            if (!sgo_quads_intersect(cor + (size_t)i * 8, cor + (size_t)j * 8)) continue;
#endif
            int o = last[j];
            rows[(size_t)i * W + (o >> 6)] |= (uint64_t)1 << (o & 63);
            if (mult) mult[(size_t)i * E + o]++;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * The rollout: ScenarioGym.reset_scenario/step/rollout (scenario_gym/scenario_gym.py:217-267),
 * State.reset/step/update_poses/update_statistics (scenario_gym/state/state.py:106-239).
 * ---------------------------------------------------------------------------------------- */
typedef struct { double speed, e_lon_prev, e_lat_prev, e_lon_int; } ctrl_state;

/* VehicleController._step, scenario_gym/controller.py:105-140 */
static void vehicle_step(ctrl_state *cs, const double *ctrl, double l, double dt, double accel,
                         double steer, double *pose)
{
    double max_accel = ctrl[SGO_C_MAX_ACCEL], max_steer = ctrl[SGO_C_MAX_STEER];
    FLOPS(FL_CTRL, FLN_VEHICLE);
    accel = fmin(fmax(accel, -max_accel), max_accel); /* np.clip */
    steer = fmin(fmax(steer, -max_steer), max_steer);
    double h = pose[3], s, c;
    sgo_sincos(h, &s, &c);
    double dx = cs->speed * c;
    double dy = cs->speed * s;
    double dh = cs->speed * sgo_tan(steer) / l;
    pose[0] += dx * dt;
    pose[1] += dy * dt;
    pose[3] += dh * dt;
    double speed = cs->speed + accel * dt;
    if (ctrl[SGO_C_ALLOW_REVERSE] == 0.0) speed = fmax(0.0, speed);
    if (!isnan(ctrl[SGO_C_MAX_SPEED])) speed = fmin(ctrl[SGO_C_MAX_SPEED], speed);
    cs->speed = speed;
}

/* PIDController._step, scenario_gym/controller.py:205-258 */
static void pid_step(ctrl_state *cs, const double *ctrl, double l, double state_dt, double dt,
                     const double *target, double *pose)
{
    double h = pose[3], s, c;
    sgo_sincos(h, &s, &c);
    FLOPS(FL_CTRL, FLN_PID);
    double e0 = target[0] - pose[0], e1 = target[1] - pose[1];
    double e_lon = c * e0 + s * e1;
    double e_lat = -s * e0 + c * e1;
    double speed = cs->speed, gain;
    if (speed > 5.0 && speed <= 15) gain = 1.0 - 0.9 * (speed - 5.0) / 10.0;
    else if (speed > 15) gain = 0.1;
    else gain = 1.0;
    double e_lat_D = (e_lat - cs->e_lat_prev) / state_dt;
    double kp = ctrl[SGO_C_STEER_KP] * gain, kd = ctrl[SGO_C_STEER_KD] * gain;
    double steer = kp * e_lat + kd * e_lat_D;
    double e_lon_D = (e_lon - cs->e_lon_prev) / state_dt;
    double e_lon_I = cs->e_lon_int + e_lon * state_dt;
    double accel = 0.0;
    if (fabs(e_lon) > 0.1)
        accel = ctrl[SGO_C_ACCEL_KP] * e_lon + ctrl[SGO_C_ACCEL_KD] * e_lon_D +
                ctrl[SGO_C_ACCEL_KI] * e_lon_I;
    cs->e_lat_prev = e_lat;
    cs->e_lon_prev = e_lon;
    cs->e_lon_int = e_lon_I;
    vehicle_step(cs, ctrl, l, dt, accel, steer, pose);
}

/* ------------------------------------------------------------------------------------------
 * Pedestrians: PedestrianSensor (pedestrian/sensor.py:42-64), PedestrianAgent._step
 * (pedestrian/agent.py:49-69), SocialForce._step (pedestrian/social_force.py:44-222, boundary terms
 * off: empty road network), PedestrianController._step (pedestrian/controller.py:25-46).
 * ---------------------------------------------------------------------------------------- */
static double GON_C[64], GON_S[64];
static int gon_ready = 0;
static void gon_init(void)
{
    if (gon_ready) return;
    for (int i = 0; i < 64; ++i) { /* Point(x, y).buffer(r): 64-gon, vertices at 2*pi*i/64 */
        double a = 2.0 * 3.141592653589793 * i / 64;
        GON_C[i] = cos(a);
        GON_S[i] = sin(a);
    }
    gon_ready = 1;
}

/* State.get_entities_in_radius (state/state.py:356-372): centre strictly inside the 64-gon */
int sgo_in_radius(double cx, double cy, double r, double px, double py)
{
    gon_init();
    double dx = px - cx, dy = py - cy, d2 = dx * dx + dy * dy, r2 = r * r;
    if (d2 > r2 * (1.0 + 1e-9)) return 0;
    if (d2 < r2 * 0.9975) return 1; /* inside the inscribed circle: cos^2(pi/64) = 0.99759 */
    /* exact rule on the ring: vertices (cx + r*C_i, cy - r*S_i), clockwise */
    double o = 0.0;
    for (int i = 0; i < 64; ++i) {
        int j = (i + 1) & 63;
        double ax = cx + r * GON_C[i], ay = cy - r * GON_S[i], bx = cx + r * GON_C[j], by = cy - r * GON_S[j];
        o += ax * by - bx * ay;
    }
    for (int i = 0; i < 64; ++i) {
        int j = (i + 1) & 63;
        double ax = cx + r * GON_C[i], ay = cy - r * GON_S[i], bx = cx + r * GON_C[j], by = cy - r * GON_S[j];
        double cr = (bx - ax) * (py - ay) - (by - ay) * (px - ax);
        if (o > 0 ? !(cr > 0) : !(cr < 0)) return 0;
    }
    return 1;
}

/* LineString(route).project(Point): arclength of the nearest point of the polyline (first minimum) */
static double route_project(const double *wp, int n, double px, double py)
{
    double best = INFINITY, best_s = 0.0, acc = 0.0;
    for (int i = 0; i + 1 < n; ++i) {
        double ax = wp[2 * i], ay = wp[2 * i + 1], dx = wp[2 * i + 2] - ax, dy = wp[2 * i + 3] - ay;
        double L2 = dx * dx + dy * dy;
        double u = L2 == 0.0 ? 0.0 : fmin(1.0, fmax(0.0, ((px - ax) * dx + (py - ay) * dy) / L2));
        double qx = ax + u * dx, qy = ay + u * dy;
        double ex = px - qx, ey = py - qy;
        double dist = sqrt(ex * ex + ey * ey);
        double L = sqrt(L2);
        if (dist < best) { best = dist; best_s = acc + u * L; }
        acc += L;
    }
    return best_s;
}

/* ---- boundary terms of the social force, pedestrian/social_force.py:86-104, 190-211 ------------------------------
 * force += U / R * r_unit * exp(-|r| / R) with r = position - nearest_points(surface, Point(position))[0].
 * shapely's nearest_points is GEOS DistanceOp (not part of the reference sources; restated from its published
 * algorithm): a point inside (or on) an areal geometry is its own nearest point -- so the "walkable boundary" term,
 * which the reference only evaluates for a pedestrian INSIDE the walkable surface, is always the zero vector, and so is
 * the impenetrable term inside a building; outside, every ring segment is tried in order, Distance::pointToSegment
 * picks the nearest (first one on ties) and LineSegment::closestPoint gives the point. */
static double pt_dist(double ax, double ay, double bx, double by)
{
    const double dx = ax - bx, dy = ay - by;
    return sqrt(dx * dx + dy * dy);
}

static double point_to_segment(double px, double py, double ax, double ay, double bx, double by)
{
    if (ax == bx && ay == by) return pt_dist(px, py, ax, ay);
    const double len2 = (bx - ax) * (bx - ax) + (by - ay) * (by - ay);
    const double r = ((px - ax) * (bx - ax) + (py - ay) * (by - ay)) / len2;
    if (r <= 0.0) return pt_dist(px, py, ax, ay);
    if (r >= 1.0) return pt_dist(px, py, bx, by);
    const double s = ((ay - py) * (bx - ax) - (ax - px) * (by - ay)) / len2;
    return fabs(s) * sqrt(len2);
}

static void segment_closest_point(double px, double py, double ax, double ay, double bx, double by, double *cx, double *cy)
{
    double factor;
    if (px == ax && py == ay) factor = 0.0;
    else if (px == bx && py == by) factor = 1.0;
    else {
        const double dx = bx - ax, dy = by - ay, len = dx * dx + dy * dy;
        factor = len <= 0.0 ? NAN : ((px - ax) * dx + (py - ay) * dy) / len;
    }
    if (factor > 0.0 && factor < 1.0) { /* LineSegment::project */
        *cx = ax + factor * (bx - ax);
        *cy = ay + factor * (by - ay);
        return;
    }
    if (pt_dist(ax, ay, px, py) < pt_dist(bx, by, px, py)) { *cx = ax; *cy = ay; }
    else { *cx = bx; *cy = by; }
}

/* does the union of the polygons of `layer` have a positive area (`surface.area > 0`)? */
static int surface_has_area(const sgo_road_network *net, uint32_t layer)
{
    if (!net) return 0;
    for (int k = 0; k < net->n_polygons; ++k) {
        if (!(net->layers[k] & layer)) continue;
        double a2 = 0.0;
        for (int64_t r = net->ring_off[k]; r < net->ring_off[k + 1]; ++r) {
            const int64_t a = net->vert_off[r], b = net->vert_off[r + 1];
            for (int64_t i = a; i < b; ++i) {
                const int64_t j = i + 1 < b ? i + 1 : a;
                a2 += net->verts[2 * i] * net->verts[2 * j + 1] - net->verts[2 * j] * net->verts[2 * i + 1];
            }
        }
        if (a2 != 0.0) return 1;
    }
    return 0;
}

/* _force_boundary for a point OUTSIDE the surface: accumulates sign * force into (fx, fy) */
static void boundary_force_outside(const sgo_road_network *net, uint32_t layer, double px, double py, double U, double R,
                                   double sign, double *fx, double *fy)
{
    double best = INFINITY, cx = px, cy = py;
    for (int k = 0; k < net->n_polygons; ++k) {
        if (!(net->layers[k] & layer)) continue;
        for (int64_t r = net->ring_off[k]; r < net->ring_off[k + 1]; ++r) {
            const int64_t a = net->vert_off[r], b = net->vert_off[r + 1];
            for (int64_t i = a; i < b; ++i) {
                const int64_t j = i + 1 < b ? i + 1 : a;
                const double ax = net->verts[2 * i], ay = net->verts[2 * i + 1], bx = net->verts[2 * j], by = net->verts[2 * j + 1];
                const double d = point_to_segment(px, py, ax, ay, bx, by);
                if (d < best) {
                    best = d;
                    segment_closest_point(px, py, ax, ay, bx, by, &cx, &cy);
                }
            }
        }
    }
    const double rx = px - cx, ry = py - cy, rn = norm2(rx, ry);
    const double ux = rx / (rn + 0.0000000001), uy = ry / (rn + 0.0000000001);
    const double k = U / R, e = sgo_exp(-rn / R);
    *fx += sign * (k * ux * e);
    *fy += sign * (k * uy * e);
}

typedef struct { double speed; int goal_idx; double fx, fy; } ped_state;

/* one PedestrianAgent.step: returns the new pose in np_ */
static void ped_step(const sgo_scenario *sc, const sgo_config *cfg, int i, const double *poses, const double *vels,
                     const uint8_t *present, double t, double next_t, double state_dt, ped_state *ps, double *np_,
                     int step_index, int64_t *noise_pos)
{
    const int E = sc->n_entities;
    const double *sf = cfg->sf, *ct = sc->ctrl + (size_t)i * SGO_NCTRL;
    int behaviour = cfg->behaviour;
    double std_lon = cfg->std_lon, std_lat = cfg->std_lat;
    if (cfg->model_of && cfg->models && cfg->n_models > 0) { /* this agent's own behaviour object, pedestrian/agent.py:39 */
        const double *row = cfg->models + (size_t)cfg->model_of[i] * SGO_PM_W;
        behaviour = (int)row[SGO_PM_BEHAVIOUR];
        sf = row + SGO_PM_SF;
        std_lon = row[SGO_PM_STD_LON];
        std_lat = row[SGO_PM_STD_LAT];
    }
    const double *pose = poses + (size_t)i * 6, *vel = vels + (size_t)i * 6;
    const double *wp = sc->routes + sc->route_off[i] * 2;
    const int nwp = (int)(sc->route_off[i + 1] - sc->route_off[i]);
    double speed = 0.0, heading = 0.0;
    /* goal update, pedestrian/agent.py:59-62 */
    if (ps->goal_idx <= nwp - 1) {
        FLOPS(FL_PED_GOAL, 27 * (nwp - 1) + 6 * (nwp - 1) + nwp); /* project: 27 per segment; arc lengths 6 per segment, 1 compare per waypoint */
        double s = route_project(wp, nwp, pose[0], pose[1]);
        double arc = 0.0;
        int last = 0;
        for (int k = 0; k < nwp; ++k) {
            if (k > 0) {
                double dx = wp[2 * k] - wp[2 * k - 2], dy = wp[2 * k + 1] - wp[2 * k - 1];
                arc += sqrt(dx * dx + dy * dy);
            }
            if (arc <= s) last = k;
        }
        ps->goal_idx = last + 1;
    }
    if (ps->goal_idx <= nwp - 1 && behaviour == 1) {
        /* RandomWalk._step, pedestrian/random_walk.py:32-44: np.random.normal(loc, scale) = loc + scale * z (std 0: == loc);
         * agent.force is not touched (it stays the zeros of PedestrianAgent.__init__, agent.py:41) */
        const double gx = wp[2 * ps->goal_idx] - pose[0], gy = wp[2 * ps->goal_idx + 1] - pose[1];
        const double loc_s = ct[SGO_C_PED_SPEED_DESIRED] + sf[SGO_SF_BIAS_LON];
        const double loc_h = sgo_atan2(gy, gx) + sf[SGO_SF_BIAS_LAT];
        speed = loc_s;
        heading = loc_h;
        if (cfg->noise_mode == 1) {
            const int64_t k = *noise_pos;
            const double z0 = k + 1 < cfg->n_normals ? cfg->normals[k] : 0.0, z1 = k + 1 < cfg->n_normals ? cfg->normals[k + 1] : 0.0;
            *noise_pos = k + 2;
            speed = loc_s + std_lon * z0;
            heading = loc_h + std_lat * z1;
        } else if (cfg->noise_mode == 2) {
            double z[2];
            sgo_noise_pair(cfg->noise_seed, (uint32_t)cfg->scenario_index, (uint32_t)i, (uint32_t)step_index, z);
            speed = loc_s + std_lon * z[0];
            heading = loc_h + std_lat * z[1];
        }
        FLOPS(FL_PED_MOVE, FLN_PED_MOVE - 5);
        ps->fx = ps->fy = 0.0;
    } else if (ps->goal_idx <= nwp - 1) {
        /* _force_to_goal, social_force.py:119-138 */
        double gx = wp[2 * ps->goal_idx] - pose[0], gy = wp[2 * ps->goal_idx + 1] - pose[1];
        double gn = norm2(gx, gy);
        if (gn == 0) gn += 0.000000001;
        double vdes = ct[SGO_C_PED_SPEED_DESIRED];
        double inv_tau = 1 / sf[SGO_SF_RELAX_TIME];
        double fx = inv_tau * (vdes * (gx / gn) - vel[0]);
        double fy = inv_tau * (vdes * (gy / gn) - vel[1]);
        double hs, hc;
        FLOPS(FL_PED_GOAL, FLN_PED_FORCE_GOAL);
        FLOPS(FL_SINCOS, -FLN_SINCOS); /* (the head rotation is a constant of the pedestrian: its sin / cos is prepared once, not per step) */
        sgo_sincos(ct[SGO_C_PED_HEAD_ROT], &hs, &hc); /* rotate_coords: math.cos/math.sin(theta) */
        for (int j = 0; j < E; ++j) { /* PedestrianSensor.get_nearby_pedestrians, sensor.py:55-64 */
            if (j == i || !present[j] || sc->etype[j] != 1) continue;
            FLOPS(FL_BROAD, j > i ? 1 : 0); /* the neighbour search shares the squared distance of the collision search: one more compare per unordered pair */
            const double *op = poses + (size_t)j * 6, *ov = vels + (size_t)j * 6;
            if (!sgo_in_radius(pose[0], pose[1], ct[SGO_C_PED_RADIUS], op[0], op[1])) continue;
            FLOPS(FL_PED_PAIR, FLN_PED_PAIR);
            /* view direction of the NEIGHBOUR's rotated velocity, social_force.py:59-62; X.dot(R.T) */
            double vx = fma(ov[0], hc, ov[1] * (-hs)), vy = fma(ov[0], hs, ov[1] * hc);
            double vn = norm2(vx, vy) + 0.0000000001;
            double ux = vx / vn, uy = vy / vn;
            /* _force_pedestrian_repulsion, :140-176 */
            double rx = pose[0] - op[0], ry = pose[1] - op[1];
            double rn = norm2(rx, ry);
            double vmag = norm2(ov[0], ov[1]) + 0.0000000001;
            double odx = ov[0] / vmag, ody = ov[1] / vmag;
            double step = vmag * (next_t - t);
            double qx = rx - step * odx, qy = ry - step * ody;
            double qn = norm2(qx, qy) + 0.0000000001;
            double sum = rn + qn;
            double b = (1.0 / 2) * sqrt(sum * sum - step * step);
            double k1 = (1.0 / 4) * (1 / b) * sum;
            double dbx = k1 * (rx / rn + qx / qn), dby = k1 * (ry / rn + qy / qn);
            double k2 = sf[SGO_SF_REPULSE_V] / sf[SGO_SF_REPULSE_SIGMA] * sgo_exp(-b / sf[SGO_SF_REPULSE_SIGMA]);
            double repx = k2 * dbx, repy = k2 * dby;
            /* _force_pedestrian_attraction, :178-188 */
            double k3 = 2 * sf[SGO_SF_ATTRACT_C];
            double attx = k3 * rx, atty = k3 * ry;
            if (sf[SGO_SF_SIGHT_USE] != 0.0) { /* _sight_weight, :213-222; np.dot = fma chain */
                double w1 = fma(uy, repy, ux * repx) / (norm2(repx, repy) + 0.0000000001) >= sf[SGO_SF_COS_SIGHT]
                                ? 1.0 : sf[SGO_SF_SIGHT_WEIGHT];
                fx += w1 * repx; fy += w1 * repy;
                double w2 = fma(uy, atty, ux * attx) / (norm2(attx, atty) + 0.0000000001) >= sf[SGO_SF_COS_SIGHT]
                                ? 1.0 : sf[SGO_SF_SIGHT_WEIGHT];
                fx += w2 * attx; fy += w2 * atty;
            } else {
                fx += attx; fy += atty;
                fx += repx; fy += repy;
            }
        }
        if (sc->road) { /* boundary terms, social_force.py:86-104 */
            const double px = pose[0], py = pose[1];
            if (surface_has_area(sc->road, SGO_LAYER_WALKABLE) && sgo_surface_contains(sc->road, SGO_LAYER_WALKABLE, px, py)) {
                fx += 0.0; /* the point is its own nearest point: U / R * (0 / 1e-10) * exp(-0 / R) */
                fy += 0.0;
            }
            if (surface_has_area(sc->road, SGO_LAYER_IMPENETRABLE)) {
                if (sgo_surface_contains(sc->road, SGO_LAYER_IMPENETRABLE, px, py)) {
                    fx += -0.0; /* sign -1, zero force */
                    fy += -0.0;
                } else {
                    boundary_force_outside(sc->road, SGO_LAYER_IMPENETRABLE, px, py, sf[SGO_SF_IMP_U], sf[SGO_SF_IMP_R], 1.0, &fx, &fy);
                }
            }
        }
        /* random fluctuations, social_force.py:106-108: np.random.normal(loc, scale) = loc + scale * z; std 0: == bias */
        double speed_rand = sf[SGO_SF_BIAS_LON], heading_rand = sf[SGO_SF_BIAS_LAT];
        if (cfg->noise_mode == 1) {
            const int64_t k = *noise_pos;
            const double z0 = k + 1 < cfg->n_normals ? cfg->normals[k] : 0.0, z1 = k + 1 < cfg->n_normals ? cfg->normals[k + 1] : 0.0;
            *noise_pos = k + 2;
            speed_rand = sf[SGO_SF_BIAS_LON] + std_lon * z0;
            heading_rand = sf[SGO_SF_BIAS_LAT] + std_lat * z1;
        } else if (cfg->noise_mode == 2) {
            double z[2];
            sgo_noise_pair(cfg->noise_seed, (uint32_t)cfg->scenario_index, (uint32_t)i, (uint32_t)step_index, z);
            speed_rand = sf[SGO_SF_BIAS_LON] + std_lon * z[0];
            heading_rand = sf[SGO_SF_BIAS_LAT] + std_lat * z[1];
        }
        FLOPS(FL_PED_MOVE, FLN_PED_MOVE);
        speed = fmin(norm2(fx, fy) + speed_rand, vdes * sf[SGO_SF_MAX_SPEED_FACTOR]);
        heading = sgo_atan2(fy, fx) + heading_rand;
        ps->fx = fx;
        ps->fy = fy;
    } else { /* reached the goal, agent.py:65-68 */
        ps->fx = ps->fy = 0.0;
    }
    /* PedestrianController._step, pedestrian/controller.py:25-46 */
    memcpy(np_, pose, 48);
    double maxs = ct[SGO_C_PED_MAX_SPEED];
    ps->speed = fmin(fmax(speed, -maxs), maxs);
    double hs2, hc2;
    sgo_sincos(heading, &hs2, &hc2);
    double sd = ps->speed * state_dt;
    np_[0] += sd * hc2;
    np_[1] += sd * hs2;
    np_[3] = heading;
}

/* ---- CollisionMetric.record_collision / get_collision_point / angle_between, metrics/collision.py:13-22, 81-203 --------
 * The reference reads `hazard.pose` / `self.ego.pose` there, attributes Entity does not have at this commit (it raises for
 * Vehicle hazards); restated with state.poses[...] in their place.  Geometry: the intersection of the
 * two boxes (convex quadrilaterals, Sutherland-Hodgman) and its area centroid, summed over the triangle fan from the first
 * vertex as GEOS does; box.centroid likewise. */
static double pymod(double x, double m) /* Python float %: result has the sign of m */
{
    double r = fmod(x, m);
    if (r != 0.0 && ((r < 0.0) != (m < 0.0))) r += m;
    return r;
}

static int angle_between(double x, double lo, double hi)
{
    const double tau = 3.14159265358979311600e+00 * 2;
    x = pymod(x, tau); lo = pymod(lo, tau); hi = pymod(hi, tau);
    return lo >= hi ? (lo < x || x <= hi) : (lo <= x && x < hi);
}

static void poly_centroid(const double *P, int n, double *cx, double *cy)
{
    double a2 = 0.0, sx = 0.0, sy = 0.0;
    for (int i = 1; i + 1 < n; ++i) {
        const double t2 = (P[2 * i] - P[0]) * (P[2 * i + 3] - P[1]) - (P[2 * i + 2] - P[0]) * (P[2 * i + 1] - P[1]);
        sx += t2 * (P[0] + P[2 * i] + P[2 * i + 2]);
        sy += t2 * (P[1] + P[2 * i + 1] + P[2 * i + 3]);
        a2 += t2;
    }
    if (a2 != 0.0) { *cx = sx / 3 / a2; *cy = sy / 3 / a2; return; }
    sx = sy = 0.0; /* degenerate (touching boxes): mean of the vertices */
    for (int i = 0; i < n; ++i) { sx += P[2 * i]; sy += P[2 * i + 1]; }
    *cx = n ? sx / n : NAN;
    *cy = n ? sy / n : NAN;
}

/* subject polygon S (4 vertices) clipped by the convex polygon C (4 vertices, either orientation); out: up to 8 vertices */
static int clip_quads(const double *S, const double *C, double *out)
{
    double A[32], B[32];
    int na = 4;
    memcpy(A, S, 64);
    double orient = 0.0;
    for (int k = 0; k < 4; ++k) { const int m = (k + 1) & 3; orient += C[2 * k] * C[2 * m + 1] - C[2 * m] * C[2 * k + 1]; }
    const double sgn = orient >= 0 ? 1.0 : -1.0;
    for (int k = 0; k < 4 && na > 0; ++k) {
        const int m = (k + 1) & 3;
        const double ex = C[2 * m] - C[2 * k], ey = C[2 * m + 1] - C[2 * k + 1];
        int nb = 0;
        for (int i = 0; i < na; ++i) {
            const int j = (i + 1) % na;
            const double di = sgn * (ex * (A[2 * i + 1] - C[2 * k + 1]) - ey * (A[2 * i] - C[2 * k]));
            const double dj = sgn * (ex * (A[2 * j + 1] - C[2 * k + 1]) - ey * (A[2 * j] - C[2 * k]));
            if (di >= 0) { B[2 * nb] = A[2 * i]; B[2 * nb + 1] = A[2 * i + 1]; ++nb; }
            if ((di > 0 && dj < 0) || (di < 0 && dj > 0)) {
                const double u = di / (di - dj);
                B[2 * nb] = A[2 * i] + u * (A[2 * j] - A[2 * i]);
                B[2 * nb + 1] = A[2 * i + 1] + u * (A[2 * j + 1] - A[2 * i + 1]);
                ++nb;
            }
        }
        memcpy(A, B, (size_t)nb * 16);
        na = nb;
    }
    memcpy(out, A, (size_t)na * 16);
    return na;
}

enum { CP_FRONT = 0, CP_FRONT_CORNER, CP_SIDE, CP_BACK, CP_BACK_CORNER };

static int collision_point_class(const double *box8, double angle, double heading, double c_tol)
{
    double bx, by, cor[4];
    poly_centroid(box8, 4, &bx, &by);
    for (int k = 0; k < 4; ++k) cor[k] = sgo_atan2(box8[2 * k + 1] - by, box8[2 * k] - bx) - heading;
    if (angle_between(angle, cor[1] - c_tol, cor[1] + c_tol) || angle_between(angle, cor[2] - c_tol, cor[2] + c_tol)) return CP_FRONT_CORNER;
    if (angle_between(angle, cor[0] - c_tol, cor[0] + c_tol) || angle_between(angle, cor[3] - c_tol, cor[3] + c_tol)) return CP_BACK_CORNER;
    if (angle_between(angle, cor[0] + c_tol, cor[3] - c_tol)) return CP_BACK;
    if (angle_between(angle, cor[2] - c_tol, cor[1] + c_tol)) return CP_FRONT;
    return CP_SIDE;
}

/* CollisionPointMetric.record_collision_position (metrics/collision.py:242-253): centroid of the boxes' intersection and
 * (hazard heading - ego heading) mod 2 pi */
void sgo_collision_point(const double *ego_pose6, const double *ego_bbox4, const double *haz_pose6, const double *haz_bbox4,
                         double *px, double *py, double *angle)
{
    double eb[8], hb[8], clip[16];
    sgo_corners(ego_pose6, ego_bbox4, eb);
    sgo_corners(haz_pose6, haz_bbox4, hb);
    poly_centroid(clip, clip_quads(eb, hb, clip), px, py);
    *angle = pymod(haz_pose6[3] - ego_pose6[3], 3.14159265358979311600e+00 * 2);
}

/* CollisionTypes: 0 other, 1 t_bone, 2 head_on, 3 rear_end, 4 side_swipe (5 non_vehicle is decided by the caller) */
int sgo_classify_collision(const double *ego_pose6, const double *ego_bbox4, const double *haz_pose6, const double *haz_bbox4,
                           double c_tol)
{
    const double pi = 3.14159265358979311600e+00, tau = pi * 2;
    double eb[8], hb[8], clip[16], px, py;
    sgo_corners(ego_pose6, ego_bbox4, eb);
    sgo_corners(haz_pose6, haz_bbox4, hb);
    const int n = clip_quads(eb, hb, clip);
    poly_centroid(clip, n, &px, &py);
    const double collision_angle = pymod(haz_pose6[3] - ego_pose6[3], tau);
    const double ego_angle = pymod(sgo_atan2(py - ego_pose6[1], px - ego_pose6[0]) - ego_pose6[3], tau);
    const double haz_angle = pymod(sgo_atan2(py - haz_pose6[1], px - haz_pose6[0]) - haz_pose6[3], tau);
    const int ep = collision_point_class(eb, ego_angle, ego_pose6[3], c_tol);
    const int hp = collision_point_class(hb, haz_angle, haz_pose6[3], c_tol);
    const int ef = ep == CP_FRONT || ep == CP_FRONT_CORNER, ebk = ep == CP_BACK || ep == CP_BACK_CORNER;
    const int hf = hp == CP_FRONT || hp == CP_FRONT_CORNER, hbk = hp == CP_BACK || hp == CP_BACK_CORNER;
    const int cross = angle_between(collision_angle, pi / 4, 3 * pi / 4) || angle_between(collision_angle, 5 * pi / 4, 7 * pi / 4);
    if (ef && hf) return cross ? 1 : (angle_between(collision_angle, 7 * pi / 4, pi / 4) ? 4 : 2);
    if ((ef || ebk) && (hf || hbk)) return cross ? 1 : 3;
    if (ef || ebk || hf || hbk) return cross ? 1 : 4;
    return 4;
}

/* ---- RSSDistances.__call__ (metrics/rss/callback.py:58-128) + the flags RSS reads (metrics/rss/rss.py:70-104) ---------
 * Everything happens in the ego's frame: x = lateral (to the right of the heading), y = longitudinal.  np.dot of two
 * 2-vectors is fma(a1, b1, a0 * b0) and norm([u, v]) = sqrt(fma(v, v, u * u)) on this numpy / OpenBLAS (probed 20000 of
 * 20000 each).  The per-entity history list `intersect[e]` is only ever searched backwards for its latest "lateral" /
 * "longitudinal" entry and for an "unsafe_*" entry: two small integers carry it. */
static double dot2(double a0, double a1, double b0, double b1) { return fma(a1, b1, a0 * b0); }

static void inv_dir(double v0, double v1, double *o0, double *o1) /* rss_utils.inverse_direction, normalised */
{
    const double n = norm2(v1, v0);
    *o0 = v1 / n;
    *o1 = -v0 / n;
}

/* closed segment (ax, ay)-(bx, by) against the closed convex quadrilateral Q (4 vertices): an endpoint inside or on Q, or
 * a proper / touching crossing of one of its edges; exact orientation signs */
static int on_segment(double ax, double ay, double bx, double by, double px, double py)
{
    return px >= fmin(ax, bx) && px <= fmax(ax, bx) && py >= fmin(ay, by) && py <= fmax(ay, by);
}

static int segs_intersect(double ax, double ay, double bx, double by, double cx, double cy, double dx, double dy)
{
    const int o1 = orient_sign(ax, ay, bx, by, cx, cy), o2 = orient_sign(ax, ay, bx, by, dx, dy);
    const int o3 = orient_sign(cx, cy, dx, dy, ax, ay), o4 = orient_sign(cx, cy, dx, dy, bx, by);
    if (o1 * o2 < 0 && o3 * o4 < 0) return 1;
    if (o1 == 0 && on_segment(ax, ay, bx, by, cx, cy)) return 1;
    if (o2 == 0 && on_segment(ax, ay, bx, by, dx, dy)) return 1;
    if (o3 == 0 && on_segment(cx, cy, dx, dy, ax, ay)) return 1;
    if (o4 == 0 && on_segment(cx, cy, dx, dy, bx, by)) return 1;
    return 0;
}

static int point_in_quad_closed(const double *Q, double px, double py)
{
    int pos = 0, neg = 0;
    for (int k = 0; k < 4; ++k) {
        const int m = (k + 1) & 3, o = orient_sign(Q[2 * k], Q[2 * k + 1], Q[2 * m], Q[2 * m + 1], px, py);
        pos |= o > 0;
        neg |= o < 0;
    }
    return !(pos && neg);
}

static int seg_intersects_quad(const double *Q, double ax, double ay, double bx, double by)
{
    if (point_in_quad_closed(Q, ax, ay) || point_in_quad_closed(Q, bx, by)) return 1;
    for (int k = 0; k < 4; ++k) {
        const int m = (k + 1) & 3;
        if (segs_intersect(ax, ay, bx, by, Q[2 * k], Q[2 * k + 1], Q[2 * m], Q[2 * m + 1])) return 1;
    }
    return 0;
}

/* code written per entity: 0 safe, 1 lateral, 2 longitudinal, 3 both, 4 unsafe_lateral, 5 unsafe_longitudinal, 6 found,
 * -1 not updated (absent, or the ego) */
void sgo_rss_update(int E, int ego, const double *poses, const double *vels, const uint8_t *present, const double *bbox,
                    sgo_rss_state *st, double *safe /*[E][2] lateral, longitudinal*/, int32_t *code)
{
    const double RESPONSE_TIME = 0.6, MIN_LONG_ACCEL = 1.2 * 9.81, MAX_LONG_ACCEL = 1.2 * 9.81, MIN_SAFE_CLEARANCE = 0.1;
    for (int e = 0; e < E; ++e) { code[e] = -1; safe[2 * e] = safe[2 * e + 1] = NAN; }
    if (!present[ego]) return;
    const double *ep = poses + (size_t)ego * 6, *ev = vels + (size_t)ego * 6;
    double es, ec;
    sgo_sincos(ep[3], &es, &ec);
    const double eh0 = ec, eh1 = es; /* direction(heading) = [cos, sin] */
    double ei0, ei1;
    inv_dir(eh0, eh1, &ei0, &ei1);
    const double ego_w = bbox[(size_t)ego * 4], ego_l = bbox[(size_t)ego * 4 + 1];
    /* the ego's own dictionary */
    const double ego_head0 = dot2(eh0, eh1, ei0, ei1), ego_head1 = dot2(eh0, eh1, eh0, eh1);
    const double ego_vel0 = dot2(ev[0], ev[1], ei0, ei1), ego_vel1 = dot2(ev[0], ev[1], eh0, eh1);
    const double ego_pos1 = dot2(ep[0] - ep[0], ep[1] - ep[1], eh0, eh1);
    for (int e = 0; e < E; ++e) {
        if (e == ego || !present[e]) continue;
        const double *hp = poses + (size_t)e * 6, *hv = vels + (size_t)e * 6;
        double hs, hc;
        sgo_sincos(hp[3], &hs, &hc);
        const double pos0 = dot2(hp[0] - ep[0], hp[1] - ep[1], ei0, ei1), pos1 = dot2(hp[0] - ep[0], hp[1] - ep[1], eh0, eh1);
        const double head0 = dot2(hc, hs, ei0, ei1), head1 = dot2(hc, hs, eh0, eh1);
        const double vel0 = dot2(hv[0], hv[1], ei0, ei1), vel1 = dot2(hv[0], hv[1], eh0, eh1);
        double cor[8], Q[8];
        sgo_corners(hp, bbox + (size_t)e * 4, cor);
        for (int k = 0; k < 4; ++k) {
            Q[2 * k] = dot2(cor[2 * k] - ep[0], cor[2 * k + 1] - ep[1], ei0, ei1);
            Q[2 * k + 1] = dot2(cor[2 * k] - ep[0], cor[2 * k + 1] - ep[1], eh0, eh1);
        }
        /* safe_longitudinal_distance, :231-272 */
        double s_long;
        {
            const double dd = dot2(ego_head0, ego_head1, head0, head1);
            const double max_long_accel = fabs(MAX_LONG_ACCEL * dd);
            if (dd > 0) {
                double vf, vr;
                if (ego_pos1 > pos1) { vf = norm2(ego_vel0, ego_vel1); vr = dot2(vel0, vel1, ego_head0, ego_head1); }
                else { vf = dot2(vel0, vel1, ego_head0, ego_head1); vr = norm2(ego_vel0, ego_vel1); }
                if (vr == 0.0) s_long = MIN_SAFE_CLEARANCE + 0.5 * ego_l;
                else {
                    const double rt = RESPONSE_TIME;
                    const double a = vr * rt + fmin(vf * vf / (2 * max_long_accel), 0.5 * max_long_accel * (rt * rt)) +
                                     ((vr + rt * max_long_accel) * (vr + rt * max_long_accel)) / (2 * MIN_LONG_ACCEL) -
                                     vf * vf / (2 * max_long_accel);
                    s_long = fmax(0, a) + MIN_SAFE_CLEARANCE + 0.5 * ego_l;
                }
            } else {
                const double v1 = fabs(dot2(ego_vel0, ego_vel1, ego_head0, ego_head1));
                const double v2 = -fabs(dot2(vel0, vel1, ego_head0, ego_head1));
                const double sp = (pos1 > 0) - (pos1 < 0), sv = (vel1 > 0) - (vel1 < 0); /* np.sign (NaN aside) */
                if (sp == sv) s_long = MIN_SAFE_CLEARANCE + 0.5 * ego_l;
                else {
                    const double rt = RESPONSE_TIME, m = max_long_accel, av2 = fabs(v2);
                    const double a = (2 * v1 + rt * m) * rt / 2 + ((v1 + rt * m) * (v1 + rt * m)) / (2 * MIN_LONG_ACCEL) +
                                     (2 * av2 + rt * m) * rt / 2 + ((av2 + rt * m) * (av2 + rt * m)) / (2 * MIN_LONG_ACCEL);
                    s_long = fmax(0, a) + MIN_SAFE_CLEARANCE + 0.5 * ego_l;
                }
            }
            s_long = fabs(s_long);
        }
        /* safe_lateral_distance, :274-305 */
        double s_lat;
        {
            double v = vel0, i0, i1;
            inv_dir(ego_head0, ego_head1, &i0, &i1);
            const double ad = fabs(dot2(i0, i1, head0, head1));
            const double max_lat = MAX_LONG_ACCEL * ad, min_lat = MIN_LONG_ACCEL * ad;
            const double sp = (-pos0 > 0) - (-pos0 < 0), sv = (v > 0) - (v < 0);
            double d0 = 0;
            if (sp == sv) {
                v = fabs(v);
                if (v == 0.0) { s_lat = fabs(MIN_SAFE_CLEARANCE + 0.5 * ego_w); goto lat_done; }
                const double rt = RESPONSE_TIME;
                d0 = fmax(0, 0.5 * rt * (2 * v + rt * max_lat) + ((v + rt * max_lat) * (v + rt * max_lat)) / (2 * min_lat) -
                                 0.5 * (rt * rt) * max_lat - ((rt * max_lat) * (rt * max_lat)) / (2 * min_lat));
            }
            s_lat = fabs(d0 + MIN_SAFE_CLEARANCE + 0.5 * ego_w);
        lat_done:;
        }
        safe[2 * e] = s_lat;
        safe[2 * e + 1] = s_long;
        /* unsafe_distance, :179-229 */
        if (st[e].found) { code[e] = 6; continue; }
        const double B[8] = {s_lat, s_long, -s_lat, s_long, -s_lat, -s_long, s_lat, -s_long};
        if (sgo_quads_intersect(Q, B)) {
            int res;
            if (st[e].last == 1) res = 5;      /* latest "lateral" entry: the longitudinal distance went last */
            else if (st[e].last == 2) res = 4;
            else {
                double j0, j1;
                inv_dir(ego_w, ego_l, &j0, &j1);
                const double A = fabs(fabs(pos0) - fabs(dot2(pos0, pos1, ego_w, ego_l))) / s_lat;
                const double Bv = fabs(fabs(pos1 - dot2(pos0, pos1, j0, j1)) / s_long);
                res = A > Bv ? 5 : 4;
            }
            st[e].found = res == 4 ? 1 : 2;
            code[e] = res;
            continue;
        }
        /* write_intersections, :307-340: the two "length" lines run from (b0x, 100 b0y) to (b2x, 100 b2y) and from
         * (b1x, 100 b1y) to (b3x, 100 b3y) -- diagonals, as the reference builds them -- the "width" lines are y = +-s_long */
        const int lat_inter = seg_intersects_quad(Q, B[0], 100 * B[1], B[4], 100 * B[5]) ||
                              seg_intersects_quad(Q, B[2], 100 * B[3], B[6], 100 * B[7]);
        const int long_inter = seg_intersects_quad(Q, 100 * B[0], B[1], 100 * B[2], B[3]) ||
                               seg_intersects_quad(Q, 100 * B[4], B[5], 100 * B[6], B[7]);
        code[e] = lat_inter && long_inter ? 3 : (lat_inter ? 1 : (long_inter ? 2 : 0));
        if (code[e] == 1 || code[e] == 2) st[e].last = code[e];
    }
}

int sgo_rollout(const sgo_scenario *sc, const sgo_config *cfg, int max_steps, int force_steps,
                const double *actions, sgo_record *rec, sgo_event *events, int event_cap,
                sgo_result *res)
{
    const int E = sc->n_entities, W = (E + 63) / 64;
    if (E > MAXE) return -1;
    const double NaN = nan("");
    double *poses = (double *)calloc((size_t)E * 6, 8), *newp = (double *)calloc((size_t)E * 6, 8);
    double *vels = (double *)calloc((size_t)E * 6, 8), *dists = (double *)calloc(E, 8);
    uint8_t *present = (uint8_t *)calloc(E, 1), *newpres = (uint8_t *)calloc(E, 1);
    uint8_t *velvalid = (uint8_t *)calloc(E, 1);
    uint64_t *rows = (uint64_t *)calloc((size_t)E * W, 8);
    uint8_t *mult = (uint8_t *)calloc((size_t)E * E, 1);
    uint64_t *last_row = (uint64_t *)calloc(W, 8);
    ctrl_state *cs = (ctrl_state *)calloc(E, sizeof(ctrl_state));
    ped_state *ps = (ped_state *)calloc(E, sizeof(ped_state));
    int *ids = (int *)malloc(E * sizeof(int)), *slot = (int *)malloc(E * sizeof(int));
    double *bpos = (double *)malloc((size_t)E * 6 * 8);
    double *scratch = (double *)malloc((size_t)E * 13 * 8);
    int nb = 0;
    for (int i = 0; i < E; ++i)
        if (sc->kind[i] == SGO_KIND_REPLAY) { slot[i] = nb; ids[nb++] = i; }
    batch_t B;
    batch_build(&B, sc->knot_off, sc->knots, ids, nb); /* ScenarioGym.create_agents :188-211 */

#define KN(i) (sc->knots + sc->knot_off[i] * 7)
#define NK(i) ((int)(sc->knot_off[(i) + 1] - sc->knot_off[i]))

    /* ---- State.reset(t0), state.py:106-143 ---- */
    double t = sc->t0, prev_t, next_t;
    for (int i = 0; i < E; ++i) {
        present[i] = 0;
        velvalid[i] = 0;
        if (sc->kind[i] == SGO_KIND_NONE) continue;
        int n = NK(i), is_static = n == 1, ok;
        if (is_static) ok = sgo_position_at_t(KN(i), n, t, 1, 1, 0, poses + (size_t)i * 6);
        else if (cfg->persist) ok = sgo_position_at_t(KN(i), n, t, 0, 0, 0, poses + (size_t)i * 6);
        else ok = sgo_position_at_t(KN(i), n, t, 0, 0, 1, poses + (size_t)i * 6);
        if (ok) {
            present[i] = 1;
            velvalid[i] = 1;
            sgo_velocity_at_t(KN(i), n, t, vels + (size_t)i * 6);
        }
    }
    prev_t = t - 0.1; /* state.py:135 */
    for (int i = 0; i < E; ++i) { /* Controller.reset: controller.py:100-103, :198-203 */
        cs[i].speed = present[i] ? norm2(vels[(size_t)i * 6], vels[(size_t)i * 6 + 1]) : 0.0;
        cs[i].e_lon_prev = cs[i].e_lat_prev = cs[i].e_lon_int = 0.0;
    }
    /* Metric resets: metrics/trajectory.py:13-17,36-39; metrics/collision.py:64-68 */
    const int ego = sc->ego;
    double avg = NaN, vmax = NaN, m_t = 0.0, ego_dist = NaN;
    if (present[ego]) {
        const double *v = vels + (size_t)ego * 6;
        avg = vmax = norm3(v[0], v[1], v[2]);
    }
    int n_events = 0, n_steps = 0, done = 0;
    int64_t noise_pos = 0;
    memset(last_row, 0, (size_t)W * 8);
    detect_collisions(E, W, present, poses, sc->bbox, rows, mult, scratch);

#define RECORD(row)                                                                             \
    do {                                                                                        \
        if (rec) {                                                                              \
            size_t s_ = rec->last_only ? 0 : (size_t)(row);                                     \
            if (rec->t) rec->t[s_] = t;                                                         \
            for (int i_ = 0; i_ < E; ++i_) {                                                    \
                for (int c_ = 0; c_ < 6; ++c_) {                                                \
                    if (rec->poses)                                                             \
                        rec->poses[(s_ * E + i_) * 6 + c_] = present[i_] ? poses[(size_t)i_ * 6 + c_] : NaN; \
                    if (rec->vels)                                                              \
                        rec->vels[(s_ * E + i_) * 6 + c_] = velvalid[i_] ? vels[(size_t)i_ * 6 + c_] : NaN; \
                }                                                                               \
                if (rec->dists) rec->dists[s_ * E + i_] = dists[i_];                            \
                if (rec->extra) {                                                               \
                    if (sc->kind[i_] == SGO_KIND_AGENT_PEDESTRIAN) {                            \
                        double e4_[4] = {ps[i_].speed, (double)ps[i_].goal_idx, ps[i_].fx, ps[i_].fy}; \
                        memcpy(rec->extra + (s_ * E + i_) * 4, e4_, 32);                        \
                    } else memcpy(rec->extra + (s_ * E + i_) * 4, &cs[i_], 32);                 \
                }                                                                               \
            }                                                                                   \
            if (rec->coll) memcpy(rec->coll + s_ * E * W, rows, (size_t)E * W * 8);             \
        }                                                                                       \
    } while (0)

    RECORD(0);

    /* ---- rollout loop, scenario_gym.py:262-263 ---- */
    while ((force_steps || !done) && n_steps < max_steps) {
        next_t = t + cfg->dt; /* scenario_gym.py:229 */
        double state_dt = t - prev_t;
        if (nb) batch_eval(&B, next_t, bpos);
        for (int i = 0; i < E; ++i) {
            double *np_ = newp + (size_t)i * 6;
            newpres[i] = 0;
            int n = NK(i);
            switch (sc->kind[i]) {
            case SGO_KIND_REPLAY: /* BatchReplayEntity.step, batch.py:34-53 */
                if (cfg->persist || n == 1 ||
                    (next_t >= KN(i)[0] && next_t <= KN(i)[(size_t)(n - 1) * 7])) {
                    memcpy(np_, bpos + (size_t)slot[i] * 6, 48);
                    newpres[i] = 1;
                }
                break;
            case SGO_KIND_AGENT_REPLAY:
            case SGO_KIND_AGENT_PID:
            case SGO_KIND_AGENT_VEHICLE:
            case SGO_KIND_AGENT_PEDESTRIAN:
                if (present[i]) { /* scenario_gym.py:234-239 */
                    if (sc->kind[i] == SGO_KIND_AGENT_PEDESTRIAN) {
                        ped_step(sc, cfg, i, poses, vels, present, t, next_t, state_dt, &ps[i], np_, n_steps, &noise_pos);
                    } else if (sc->kind[i] == SGO_KIND_AGENT_REPLAY) {
                        sgo_position_at_t(KN(i), n, next_t, 0, 0, 0, np_); /* agent.py:125-128 */
                    } else {
                        memcpy(np_, poses + (size_t)i * 6, 48);
                        const double *ct = sc->ctrl + (size_t)i * SGO_NCTRL;
                        double l = sc->bbox[(size_t)i * 4 + 1];
                        if (sc->kind[i] == SGO_KIND_AGENT_PID) {
                            double tgt[6];
                            sgo_position_at_t(KN(i), n, next_t, 0, 0, 0, tgt); /* agent.py:145-148 */
                            pid_step(&cs[i], ct, l, state_dt, next_t - t, tgt, np_);
                        } else {
                            const double *a = actions + (size_t)n_steps * 2;
                            vehicle_step(&cs[i], ct, l, next_t - t, a[0], a[1], np_);
                        }
                    }
                    newpres[i] = 1;
                } else if (KN(i)[0] >= t) { /* scenario_gym.py:240-244: spawn */
                    sgo_position_at_t(KN(i), n, next_t, 0, 0, 0, np_);
                    newpres[i] = 1;
                }
                break;
            default: break;
            }
        }
        /* ---- State.step -> update_poses, state.py:165-228 ---- */
        prev_t = t;
        t = next_t;
        double dt = t - prev_t;
        for (int i = 0; i < E; ++i) {
            velvalid[i] = 0;
            if (!newpres[i]) continue;
            double prev[6];
            if (present[i]) memcpy(prev, poses + (size_t)i * 6, 48);
            else sgo_position_at_t(KN(i), NK(i), prev_t, 1, 1, 0, prev); /* state.py:219-222 */
            double d[6];
            for (int c = 0; c < 6; ++c) { /* update_statistics, state.py:230-239 */
                d[c] = newp[(size_t)i * 6 + c] - prev[c];
                vels[(size_t)i * 6 + c] = d[c] / dt;
            }
            dists[i] += norm3(d[0], d[1], d[2]);
            velvalid[i] = 1;
            FLOPS(FL_STATS, FLN_STATS);
            if (sc->kind[i] == SGO_KIND_AGENT_PEDESTRIAN) FLOPS(FL_PED_ENTITY, FLN_PED_ENTITY);
        }
        memcpy(poses, newp, (size_t)E * 48);
        memcpy(present, newpres, E);
        ++n_steps;
        FLOPS(FL_ENTITY_STEPS, E);
        /* collisions are evaluated lazily in the reference; results are per-step pure */
        detect_collisions(E, W, present, poses, sc->bbox, rows, mult, scratch);
        /* check_terminal, state.py:268-270, 397-408 */
        done = 0;
        if ((cfg->terminal_mask & SGO_TERM_MAX_LENGTH) && (t + dt > sc->length)) done = 1;
        if (cfg->terminal_mask & SGO_TERM_COLLISION)
            for (int k = 0; k < E * W; ++k) if (rows[k]) done = 1;
        if ((cfg->terminal_mask & SGO_TERM_EGO_COLLISION) && present[0])
            for (int k = 0; k < W; ++k) if (rows[k]) done = 1;
        /* ego_off_road looks at entities[0] (not Scenario.ego) and fires when it is absent, state.py:401-407 */
        if ((cfg->terminal_mask & SGO_TERM_EGO_OFF_ROAD) &&
            !(present[0] && sgo_surface_contains(sc->road, SGO_LAYER_DRIVEABLE, poses[0], poses[1])))
            done = 1;
        /* metrics, scenario_gym.py:251-252 */
        if (present[ego]) {
            const double *v = vels + (size_t)ego * 6;
            double speed = norm3(v[0], v[1], v[2]);
            double w = m_t / t; /* EgoAvgSpeed._step, metrics/trajectory.py:19-24 */
            avg += (1.0 - w) * (speed - avg);
            m_t = t;
            vmax = fmax(speed, vmax); /* np.maximum */
            ego_dist = dists[ego];
            FLOPS(FL_METRIC, FLN_METRIC);
            /* CollisionMetric._step, metrics/collision.py:70-75 */
            for (int j = 0; j < E; ++j) {
                int hit = (rows[(size_t)ego * W + (j >> 6)] >> (j & 63)) & 1;
                int was = (last_row[j >> 6] >> (j & 63)) & 1;
                if (hit && !was)
                    for (int r = 0; r < mult[(size_t)ego * E + j]; ++r) {
                        if (events && n_events < event_cap) {
                            events[n_events].t = t;
                            events[n_events].other = j;
                            /* record_collision, metrics/collision.py:81-86: non_vehicle unless the hazard is a Vehicle */
                            sgo_collision_point(poses + (size_t)ego * 6, sc->bbox + (size_t)ego * 4, poses + (size_t)j * 6,
                                                sc->bbox + (size_t)j * 4, &events[n_events].px, &events[n_events].py, &events[n_events].angle);
                            events[n_events].type = sc->etype[j] != 0 ? 5 : sgo_classify_collision(
                                poses + (size_t)ego * 6, sc->bbox + (size_t)ego * 4, poses + (size_t)j * 6, sc->bbox + (size_t)j * 4, 0.4);
                        }
                        ++n_events;
                    }
            }
            memcpy(last_row, rows + (size_t)ego * W, (size_t)W * 8);
        }
        RECORD(n_steps);
    }
    if (res) {
        res->final_t = t;
        res->ego_avg_speed = avg;
        res->ego_max_speed = vmax;
        res->ego_distance = ego_dist;
        res->n_steps = n_steps;
        res->noise_used = noise_pos;
        res->done = done;
        res->n_events = n_events;
    }
    batch_free(&B);
    free(poses); free(newp); free(vels); free(dists); free(present); free(newpres);
    free(velvalid); free(rows); free(mult); free(last_row); free(cs); free(ids); free(slot);
    free(bpos); free(scratch); free(ps);
    return 0;
}
