// sgym_agents.hpp -- VehicleController / PIDController steps, the pedestrian arithmetic (exp, log, atan2, the noise generator, ped_pair), the radius rule, route projection.
// Part of the gfx950 device code of the batched rollout engine; included by sgym_device.hpp (in order: every part builds on
// the ones before it), never on its own.
#pragma once

namespace sg {

// ------------------------------------------------------------------------------------------------
// controllers
// ------------------------------------------------------------------------------------------------
struct CtrlState { double speed, e_lon_prev, e_lat_prev, e_lon_int; };

// VehicleController._step (controller.py:105-140); sin_h/cos_h of the current heading come from
// the caller.  cp(q) = controller parameter SG_C_q of this slot (LDS table, or the static rows in pedestrian scenes).
template <typename CP>
__device__ __forceinline__ void vehicle_step(CtrlState &cs, const CP &cp, double l,
                                             double dt, double accel, double steer, double sin_h,
                                             double cos_h, double *pose, ConstTbl K)
{
    double max_steer = cp(SG_C_MAX_STEER), max_accel = cp(SG_C_MAX_ACCEL);
    double max_speed = cp(SG_C_MAX_SPEED), allow_rev = cp(SG_C_ALLOW_REVERSE);
    accel = __builtin_fmin(__builtin_fmax(accel, -max_accel), max_accel);
    steer = __builtin_fmin(__builtin_fmax(steer, -max_steer), max_steer);
    double dx = cs.speed * cos_h;
    double dy = cs.speed * sin_h;
    double dh = cs.speed * sg_tan(steer, K) / l;
    pose[0] += dx * dt;
    pose[1] += dy * dt;
    pose[3] += dh * dt;
    double speed = cs.speed + accel * dt;
    if (allow_rev == 0.0) speed = __builtin_fmax(0.0, speed);
    if (max_speed == max_speed) speed = __builtin_fmin(max_speed, speed);
    cs.speed = speed;
}

// PIDController._step (controller.py:205-258)
template <typename CP>
__device__ __forceinline__ void pid_step(CtrlState &cs, const CP &cp, double l,
                                         double state_dt, double dt, double tx, double ty,
                                         double sin_h, double cos_h, double *pose, ConstTbl K)
{
    double e0 = tx - pose[0], e1 = ty - pose[1];
    double e_lon = cos_h * e0 + sin_h * e1;
    double e_lat = -sin_h * e0 + cos_h * e1;
    double speed = cs.speed, gain;
    if (speed > 5.0 && speed <= 15) gain = 1.0 - 0.9 * (speed - 5.0) / 10.0;
    else if (speed > 15) gain = 0.1;
    else gain = 1.0;
    const RecipDiv rd(state_dt); // both derivative terms divide by State.dt
    const bool fast = rd.safe(e_lat - cs.e_lat_prev) && rd.safe(e_lon - cs.e_lon_prev);
    double e_lat_D = fast ? rd.div(e_lat - cs.e_lat_prev) : (e_lat - cs.e_lat_prev) / state_dt;
    double kp = cp(SG_C_STEER_KP) * gain, kd = cp(SG_C_STEER_KD) * gain;
    double steer = kp * e_lat + kd * e_lat_D;
    double e_lon_D = fast ? rd.div(e_lon - cs.e_lon_prev) : (e_lon - cs.e_lon_prev) / state_dt;
    double e_lon_I = cs.e_lon_int + e_lon * state_dt;
    double accel = 0.0;
    if (__builtin_fabs(e_lon) > 0.1)
        accel = cp(SG_C_ACCEL_KP) * e_lon + cp(SG_C_ACCEL_KD) * e_lon_D + cp(SG_C_ACCEL_KI) * e_lon_I;
    cs.e_lat_prev = e_lat;
    cs.e_lon_prev = e_lon;
    cs.e_lon_int = e_lon_I;
    vehicle_step(cs, cp, l, dt, accel, steer, sin_h, cos_h, pose, K);
}

// ------------------------------------------------------------------------------------------------
// pedestrians: exp / atan2 shared (by restatement) with the oracle's sgo_exp / sgo_atan2
// ------------------------------------------------------------------------------------------------
// Division policy of the social-force pair terms.  Exact: plain IEEE '/'.  Fast: the same quotients through
// RecipDiv (correctly rounded inside its operand range); an operand outside the range only raises `bad`, and the
// caller recomputes that pair with Exact.  Keeps the common case free of branches.
struct ExactArith {
    bool bad = false;
    __device__ __forceinline__ double div(double a, double d) { return a / d; }
    __device__ __forceinline__ void div2(double a, double b, double d, double &qa, double &qb) { qa = a / d; qb = b / d; }
    // fl(a / m) >= c
    __device__ __forceinline__ bool quotient_ge(double a, double m, double c) { return a / m >= c; }
    __device__ __forceinline__ double sqrt(double x) { return __builtin_sqrt(x); }
};
struct FastArith {
    bool bad = false;
    // The compiler's fp64 sqrt is v_rsq_f64 + two Goldschmidt refinements + two residual corrections, wrapped in a
    // 2^256 rescaling for arguments below 2^-767 and a pass-through for 0 / inf.  For arguments in [2^-700, 2^1000) the
    // rescaling is the identity, so the bare core below returns the same bits with 10 instructions instead of 18.
    __device__ __forceinline__ double sqrt(double x)
    {
        bad |= !((x >= 0x1p-700) & (x < 0x1p1000));
        const double y = __builtin_amdgcn_rsq(x);
        double g = x * y, h = y * 0.5;
        const double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        double d = __builtin_fma(-g, g, x);
        g = __builtin_fma(d, h, g);
        d = __builtin_fma(-g, g, x);
        return __builtin_fma(d, h, g);
    }
    __device__ __forceinline__ double div(double a, double d)
    {
        const RecipDiv rd(d);
        bad |= !rd.safe(a);
        return rd.div(a);
    }
    __device__ __forceinline__ void div2(double a, double b, double d, double &qa, double &qb)
    {
        const RecipDiv rd(d);
        bad |= !(rd.safe(a) & rd.safe(b));
        qa = rd.div(a);
        qb = rd.div(b);
    }
    // m > 0.  Rounding is monotone: a >= c*m*(1 + 2^-50) implies fl(a/m) >= c, a <= c*m*(1 - 2^-50) implies
    // fl(a/m) < c (8 ulp margins); the sliver in between (and c*m outside the normal range) is `bad`.
    __device__ __forceinline__ bool quotient_ge(double a, double m, double c)
    {
        const double cm = c * m, acm = __builtin_fabs(cm), slack = acm * 0x1p-50;
        const bool yes = a >= cm + slack, no = a <= cm - slack;
        bad |= !((yes | no) & (acm < 0x1p1000) & ((acm > 0x1p-900) | (c == 0.0)));
        return yes;
    }
};

template <typename AR>
__device__ __forceinline__ double sg_exp(double x, AR &A)
{
    const double LN2HI = 6.93147180369123816490e-01, LN2LO = 1.90821492927058770002e-10,
                 INVLN2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 709.782712893383973096) return __builtin_inf();
    if (x < -745.13321910194110842) return 0.0;
    double k = __builtin_rint(x * INVLN2);
    double hi = x - k * LN2HI;
    double lo = k * LN2LO;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y = 1.0 - ((lo - A.div(r * c, 2.0 - c)) - hi);
    return ldexp(y, (int)k);
}
__device__ __forceinline__ double sg_exp(double x)
{
    ExactArith A;
    return sg_exp(x, A);
}

// One neighbour's terms of SocialForce._step (social_force.py:59-62, 140-188, 213-222): the weighted repulsion
// (c1) and attraction (c2) this neighbour adds to the force, in the reference's operation order.
// STRAIGHT: head rotation 0 in every lane (hs == 0, hc == 1): the rotated velocity is the velocity itself, so the view
// direction is the neighbour's own unit velocity (odx, ody), bit for bit.  NOATT: ped_attract_C == 0 with a positive
// sight weight: the attraction is a signed zero and w2 * (+-0) == +-0, whatever w2 is.
template <bool STRAIGHT, bool NOATT, typename AR>
__device__ __forceinline__ void ped_pair(AR &A, const sg_social_force &sf, double k2_scale, double px, double py,
                                         double hs, double hc, double ox, double oy, double ovx, double ovy,
                                         double odx, double ody, double step, double &c1x, double &c1y,
                                         double &c2x, double &c2y)
{
    // view direction = the neighbour's velocity rotated by the head angle (:59-62, X.dot(R.T))
    double ux = odx, uy = ody;
    if (!STRAIGHT) {
        double vx = __builtin_fma(ovx, hc, ovy * (-hs)), vy = __builtin_fma(ovx, hs, ovy * hc);
        double vn = A.sqrt(__builtin_fma(vy, vy, vx * vx)) + 0.0000000001;
        A.div2(vx, vy, vn, ux, uy);
    }
    double rx = px - ox, ry = py - oy; // _force_pedestrian_repulsion, :140-176
    double rn = A.sqrt(__builtin_fma(ry, ry, rx * rx));
    double qx = rx - step * odx, qy = ry - step * ody;
    double qn = A.sqrt(__builtin_fma(qy, qy, qx * qx)) + 0.0000000001;
    double sum = rn + qn;
    double b = (1.0 / 2) * A.sqrt(sum * sum - step * step);
    double k1 = (1.0 / 4) * A.div(1.0, b) * sum;
    double rxn, ryn, qxn, qyn;
    A.div2(rx, ry, rn, rxn, ryn);
    A.div2(qx, qy, qn, qxn, qyn);
    double dbx = k1 * (rxn + qxn), dby = k1 * (ryn + qyn);
    double k2 = k2_scale * sg_exp(A.div(-b, sf.ped_repulse_sigma), A);
    double repx = k2 * dbx, repy = k2 * dby;
    double k3 = 2 * sf.ped_attract_C; // _force_pedestrian_attraction, :178-188
    double attx = k3 * rx, atty = k3 * ry;
    double w1 = 1.0, w2 = 1.0;
    if (sf.sight_weight_use != 0.0) { // _sight_weight, :213-222 (wave-uniform)
        w1 = A.quotient_ge(__builtin_fma(uy, repy, ux * repx), A.sqrt(__builtin_fma(repy, repy, repx * repx)) + 0.0000000001, sf.cos_sight)
                 ? 1.0 : sf.sight_weight;
        c1x = w1 * repx; c1y = w1 * repy;
        if (NOATT) {
            c2x = attx; c2y = atty;
        } else {
            w2 = A.quotient_ge(__builtin_fma(uy, atty, ux * attx), A.sqrt(__builtin_fma(atty, atty, attx * attx)) + 0.0000000001, sf.cos_sight)
                     ? 1.0 : sf.sight_weight;
            c2x = w2 * attx; c2y = w2 * atty;
        }
    } else {
        c1x = repx; c1y = repy;
        c2x = attx; c2y = atty;
    }
}

// log for the Box-Muller transform of the counter-based noise generator: fdlibm's __ieee754_log restated, domain finite
// normal x > 0; the same operation sequence as the oracle's sgo_log.
__device__ __forceinline__ double sg_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    int hx = __double2hiint(x);
    int k = (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int i0 = (hx + 0x95f64) & 0x100000;
    x = __hiloint2double(hx | (i0 ^ 0x3ff00000), __double2loint(x)); // normalize x or x/2
    k += i0 >> 20;
    const double f = x - 1.0, dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) { // |f| < 2**-20
        if (f == 0.0) return k == 0 ? 0.0 : dk * ln2_hi + dk * ln2_lo;
        const double R = f * f * (0.5 - 0.33333333333333333 * f);
        return k == 0 ? f - R : dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    const double s = f / (2.0 + f), z = s * s, w = z * z;
    const int i = (hx - 0x6147a) | (0x6b851 - hx);
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6)), t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    if (i > 0) {
        const double hfsq = 0.5 * f * f;
        return k == 0 ? f - (hfsq - s * (hfsq + R)) : dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    return k == 0 ? f - s * (f - R) : dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

// The two standard normal variates of noise mode 2 for (seed, scenario, entity, step): Philox4x32-10 (Salmon et al., SC'11)
// at counter (entity, step, 0, 0) under key (seed_lo ^ scenario, seed_hi), two 53-bit uniforms in (0, 1), Box-Muller.
// Same operation sequence as the oracle's sgo_noise_pair.
__device__ __forceinline__ void sg_noise_pair(unsigned long long seed, uint32_t scenario, uint32_t entity, uint32_t step,
                                              double &z0, double &z1, ConstTbl K)
{
    uint32_t c0 = entity, c1 = step, c2 = 0, c3 = 0, k0 = (uint32_t)seed ^ scenario, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const double u1 = ((double)((((uint64_t)c0 << 32) | c1) >> 11) + 0.5) * 0x1p-53;
    const double u2 = ((double)((((uint64_t)c2 << 32) | c3) >> 11) + 0.5) * 0x1p-53;
    const double r = __builtin_sqrt(-2.0 * sg_log(u1));
    double sn, cs;
    sg_sincos(6.28318530717958623200e+00 * u2, sn, cs, K);
    z0 = r * cs;
    z1 = r * sn;
}

__device__ __forceinline__ double sg_atan_pos(double ax, ConstTbl K)
{
    // (the eleven coefficients come from the constant table through scalar loads: as literals they were 22 VGPRs that the
    // pedestrian kernels, whose registers are full, spilled and reloaded on every step)
    ConstTbl A = K + 32;
    const double A0 = A[0], A1 = A[1], A2 = A[2], A3 = A[3], A4 = A[4], A5 = A[5], A6 = A[6], A7 = A[7], A8 = A[8], A9 = A[9], A10 = A[10];
    // fdlibm's five argument ranges as selects around ONE division (ax / 1.0 == ax in the first range): the lanes of a crowd
    // sit in all of them, and as branches the wavefront ran the four divisions one after the other
    const bool r0 = ax < 0.4375, r1 = ax < 0.6875, r2 = ax < 1.1875, r3 = ax < 2.4375;
    const double num = r0 ? ax : (r1 ? 2.0 * ax - 1.0 : (r2 ? ax - 1.0 : (r3 ? ax - 1.5 : -1.0)));
    const double den = r0 ? 1.0 : (r1 ? 2.0 + ax : (r2 ? ax + 1.0 : (r3 ? 1.0 + 1.5 * ax : ax)));
    const double hi = r0 ? 0.0 : (r1 ? 4.63647609000806093515e-01 : (r2 ? 7.85398163397448278999e-01 : (r3 ? 9.82793723247329054082e-01 : 1.57079632679489655800e+00)));
    const double lo = r0 ? 0.0 : (r1 ? 2.26987774529616870924e-17 : (r2 ? 3.06161699786838301793e-17 : (r3 ? 1.39033110312309984516e-17 : 6.12323399573676603587e-17)));
    const double x = num / den;
    double z = x * x, w = z * z;
    double s1 = z * (A0 + w * (A2 + w * (A4 + w * (A6 + w * (A8 + w * A10)))));
    double s2 = w * (A1 + w * (A3 + w * (A5 + w * (A7 + w * A9))));
    const double xs = x * (s1 + s2);
    const double res = r0 ? x - xs : hi - ((xs - lo) - x);
    return ax >= 7.378697629483821e19 ? 1.57079632679489655800e+00 + 6.12323399573676603587e-17 : res;
}

__device__ __forceinline__ double sg_atan2(double y, double x, ConstTbl K = (ConstTbl)SG_TRIG)
{
    const double PI = 3.1415926535897931160E+00, PI_LO = 1.2246467991473531772E-16;
    if (x != x || y != y) return x + y;
    if (y == 0.0) return (x < 0.0 || (x == 0.0 && __builtin_signbit(x))) ? __builtin_copysign(PI, y) : y;
    if (x == 0.0) return __builtin_copysign(0.5 * PI, y);
    double z = sg_atan_pos(__builtin_fabs(y / x), K);
    if (x > 0.0) return y > 0.0 ? z : -z;
    z = PI - (z - PI_LO);
    return y > 0.0 ? z : -z;
}

// State.get_entities_in_radius (state/state.py:356-372): centre strictly inside the 64-gon
// Point(cx, cy).buffer(r); gon = cos/sin table of the polygon's vertex angles.
__device__ __forceinline__ bool sg_in_radius(double cx, double cy, double r, double px, double py, const double *gon)
{
    double dx = px - cx, dy = py - cy, d2 = dx * dx + dy * dy, r2 = r * r;
    if (d2 > r2 * (1.0 + 1e-9)) return false;
    if (d2 < r2 * 0.9975) return true;
    // On the thin ring between the inscribed circle and the vertices only the edges facing the point can
    // cut it off: test the edge of its sector and both neighbours with the oracle's cross product (the
    // other 61 edges hold with a margin of ~r*sin(pi/32)).  Vertices run clockwise: (cx + r*C_i, cy - r*S_i).
    float phi = atan2f((float)(-dy), (float)dx);
    int k0 = (int)__builtin_floorf(phi * 10.185916f); // 64 / (2*pi)
    bool inside = true;
    for (int e = -1; e <= 1; ++e) {
        int i = (k0 + e) & 63, j = (i + 1) & 63;
        double ax = cx + r * gon[2 * i], ay = cy - r * gon[2 * i + 1];
        double bx = cx + r * gon[2 * j], by = cy - r * gon[2 * j + 1];
        double cr = (bx - ax) * (py - ay) - (by - ay) * (px - ax);
        inside = inside && (cr < 0);
    }
    return inside;
}

// LineString(route).project(Point) + the goal update of PedestrianAgent._step (pedestrian/agent.py:59-62)
__device__ __forceinline__ int ped_goal_update(const double *wp, int nwp, double px, double py)
{
    double best = __builtin_inf(), best_s = 0.0, acc = 0.0;
    for (int i = 0; i + 1 < nwp; ++i) {
        double ax = wp[2 * i], ay = wp[2 * i + 1], dx = wp[2 * i + 2] - ax, dy = wp[2 * i + 3] - ay;
        double L2 = dx * dx + dy * dy;
        double u = L2 == 0.0 ? 0.0 : __builtin_fmin(1.0, __builtin_fmax(0.0, ((px - ax) * dx + (py - ay) * dy) / L2));
        double qx = ax + u * dx, qy = ay + u * dy;
        double ex = px - qx, ey = py - qy;
        double dist = __builtin_sqrt(ex * ex + ey * ey);
        double L = __builtin_sqrt(L2);
        if (dist < best) { best = dist; best_s = acc + u * L; }
        acc += L;
    }
    double arc = 0.0;
    int last = 0;
    for (int k = 0; k < nwp; ++k) {
        if (k > 0) {
            double dx = wp[2 * k] - wp[2 * k - 2], dy = wp[2 * k + 1] - wp[2 * k - 1];
            arc += __builtin_sqrt(dx * dx + dy * dy);
        }
        if (arc <= best_s) last = k;
    }
    return last + 1;
}

// ped_goal_update for a route of TWO waypoints (a start and a goal: BASELINE config 5) -- the same operations on the one
// segment, its four coordinates loaded once: the general form reads them in its first loop, again in its second, and the
// caller once more for the goal (three memory latencies in a row in every step of every walking pedestrian).  (gx, gy) = the
// second waypoint, the goal whenever the returned index is 1.
__device__ __forceinline__ int ped_goal_update2(const double *wp, double px, double py, double &gx, double &gy)
{
    const double ax = wp[0], ay = wp[1];
    gx = wp[2];
    gy = wp[3];
    const double dx = gx - ax, dy = gy - ay;
    const double L2 = dx * dx + dy * dy;
    const double u = L2 == 0.0 ? 0.0 : __builtin_fmin(1.0, __builtin_fmax(0.0, ((px - ax) * dx + (py - ay) * dy) / L2));
    const double qx = ax + u * dx, qy = ay + u * dy;
    const double ex = px - qx, ey = py - qy;
    const double dist = __builtin_sqrt(ex * ex + ey * ey);
    const double L = __builtin_sqrt(L2);
    const double best_s = dist < __builtin_inf() ? 0.0 + u * L : 0.0;
    // (second loop of the general form: arc_0 = 0 keeps last = 0 whatever best_s is; arc_1 = 0 + L)
    return (0.0 + L <= best_s) ? 2 : 1;
}

// ... and for routes of up to MAXW waypoints (the street routes of the crowds on road networks have four): every waypoint
// loaded once, all loads in flight together, both loops of the general form from registers (the second loop's segment lengths
// are the first loop's: the same expression on the same operands).  (gx, gy) = the goal's coordinates when the returned
// index is a waypoint.
template <int MAXW>
__device__ __forceinline__ int ped_goal_update_reg(const double *wp, int nwp, double px, double py, double &gx, double &gy)
{
    double wx[MAXW], wy[MAXW], Ls[MAXW];
#pragma unroll
    for (int k = 0; k < MAXW; ++k) {
        wx[k] = wy[k] = Ls[k] = 0.0;
        if (k < nwp) { wx[k] = wp[2 * k]; wy[k] = wp[2 * k + 1]; }
    }
    double best = __builtin_inf(), best_s = 0.0, acc = 0.0;
#pragma unroll
    for (int i = 0; i + 1 < MAXW; ++i) {
        if (i + 1 < nwp) {
            const double ax = wx[i], ay = wy[i], dx = wx[i + 1] - ax, dy = wy[i + 1] - ay;
            const double L2 = dx * dx + dy * dy;
            const double u = L2 == 0.0 ? 0.0 : __builtin_fmin(1.0, __builtin_fmax(0.0, ((px - ax) * dx + (py - ay) * dy) / L2));
            const double qx = ax + u * dx, qy = ay + u * dy;
            const double ex = px - qx, ey = py - qy;
            const double dist = __builtin_sqrt(ex * ex + ey * ey);
            const double L = __builtin_sqrt(L2);
            if (dist < best) { best = dist; best_s = acc + u * L; }
            acc += L;
            Ls[i + 1] = L;
        }
    }
    double arc = 0.0;
    int last = 0;
#pragma unroll
    for (int k = 0; k < MAXW; ++k) {
        if (k < nwp) {
            if (k > 0) arc += Ls[k];
            if (arc <= best_s) last = k;
        }
    }
    gx = gy = 0.0;
#pragma unroll
    for (int k = 1; k < MAXW; ++k)
        if (last + 1 == k) { gx = wx[k]; gy = wy[k]; }
    return last + 1;
}

} // namespace sg
