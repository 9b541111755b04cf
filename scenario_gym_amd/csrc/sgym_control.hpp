// sgym_control.hpp -- The controller pre-pass: control_body, control_kernel / _riders / _fast.
// Part of the gfx950 device code of the batched rollout engine; included by sgym_device.hpp (in order: every part builds on
// the ones before it), never on its own.
#pragma once

namespace sg {

// ------------------------------------------------------------------------------------------------
// Controller pre-pass.  A PIDAgent / external-action VehicleController lane never looks at another
// entity (agent.py:131-148, controller.py:105-140, 205-258: own trajectory, own pose, own controller
// state), so the controlled lanes of the whole batch are gathered 64 to a wavefront and integrated
// here for a chunk of steps; rollout_kernel<.., TAB> then replays the table.  Inside rollout_kernel
// the same work would occupy a full wavefront instruction stream for 1 active lane in 64.
// The step arithmetic is the rollout kernel's own (same device functions, same clock recurrence), so
// both paths produce identical bits.  The lane assumes its scenario keeps running; a scenario that
// terminates early simply stops consuming the table.
//   first: take the lane state from the state blocks (start of an API call); otherwise from
//          p.ctl_state (previous launch of the same call).   k0: step offset into `actions`.
//   row0:  first table row this launch writes (a chunk of the table is filled by several short launches, so
//          that the 64 wavefronts of the pre-pass do not sit on the same SIMDs for a whole chunk).
// ------------------------------------------------------------------------------------------------
// controller parameters and the x / y channels of the knot segment (x_lo, y_lo[2], slope[2]: the PID target) of every
// lane: own column only, no barriers.  With the segment out of the VGPRs (and the LDS under 8 KB per wavefront) the
// kernel compiles for 5 wavefronts per SIMD = 96 VGPRs, which is what fits beside two wavefronts of the rollout kernel.
struct CtlLds { double ctrl[9][64]; double seg[5][64]; };

//   metrics: run the ego's EgoAvgSpeed / EgoMaxSpeed recurrences here (plane 2).  The time-sliced path passes 0: its ordered
//          pass computes them from the speeds the slices leave, and the pre-pass -- a chain of T dependent steps on a handful
//          of wavefronts, the critical path of that mode -- is shorter without them.
// The steady state of a PID lane runs as one straight-line block (`fast` below): every division with a step-invariant or
// shared denominator through a refined reciprocal (RecipDiv: the same bits as `/` inside its operand range), the range
// checks of sin / cos / tan and of the reciprocals as ONE wavefront vote, selects instead of lane branches.  A step in which
// some lane spawns, crosses a knot, saturates its steering beyond the tangent polynomial's range or leaves RecipDiv's range
// runs the general code below it.  Same operations on the same operands in the same order: same bits
// (test_controller_prepass_equals_inline_controllers, SG_CTL_FAST=0 forces the general code).
// FAST: compiled in for control_kernel_fast only (151 VGPRs: the time-sliced path, the RSS table variant and the pipelined
// table path, where the pre-pass chain is the critical path); control_kernel (<= 128 VGPRs) stays as it was.
// RIDERS (control_kernel_riders, for rollout_kernel_crowd_riders): the lanes are ALL non-pedestrian entities of a crowd batch
// -- replay entities (the scenario's union grid, presence rule of batch.py:45-52) and replay agents (own knots, clamped;
// agent.py:125-128) beside the PID / vehicle agents -- and every row also gets plane 2 = z, pitch, roll, presence.
template <bool FAST, bool RIDERS = false>
__device__ __forceinline__ void control_body_l(CtlLds &lds /* the wavefront's LDS (rollout_kernel_tabq lays it over its tile) */,
                                               const unsigned qblock /* the 64 controlled lanes of this wavefront */,
                                               const Params &p, double timestep, int n_steps, int first, int k0,
                                               const double *actions /*[n][R][2]*/, double *tab, int row0, int metrics)
{
    const int lane = threadIdx.x;
    const size_t q = (size_t)qblock * 64 + lane;
    const int ent_raw = p.ctl_ent[q];
    const bool active = ent_raw >= 0;
    const uint32_t ent = active ? (uint32_t)ent_raw : 0u;
    const uint32_t r = ent / (uint32_t)p.EP;
    const LanePtr st(p.stat + (size_t)(ent >> 6) * (ST_COUNT * 64), (ent & 63) * 8u);
    const LanePtr dy(p.dyn + (size_t)(ent >> 6) * ((size_t)p.FROWS * 64), (ent & 63) * 8u);
    const int64_t meta = fld<int64_t>(st, ST_META);
    const int kind = active ? (int)(meta & 0xff) : SG_KIND_NONE;
    const double min_t = fld(st, ST_MIN_T), bl = fld(st, ST_BL);
    // the scenario's ego: its EgoAvgSpeed / EgoMaxSpeed recurrences (metrics/trajectory.py:8-48) run here as well
    const bool is_ego = active && (int)(ent - r * (uint32_t)p.EP) == p.sstat[r].ego;
#pragma unroll
    for (int c = 0; c < 9; ++c) lds.ctrl[c][lane] = fld(st, ST_CTRL + c); // own column only: no barrier needed
    const size_t NP = (size_t)p.n_ctl_pad;
    double *cst = p.ctl_state + q;

    double pose[6], t, prev_t;
    double m_avg, m_max, m_t;
    bool present;
    CtrlState cs;
    if (first) {
        const sg_scenario_state &sd = p.sdyn[r];
        t = sd.t;
        prev_t = sd.prev_t;
        m_avg = sd.ego_avg_speed; m_max = sd.ego_max_speed; m_t = sd.avg_t;
        present = fld<uint64_t>(dy, SG_F_PRESENT) != 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) pose[c] = fld(dy, SG_F_POSE + c);
        cs.speed = fld(dy, SG_F_CTRL + 0); cs.e_lon_prev = fld(dy, SG_F_CTRL + 1);
        cs.e_lat_prev = fld(dy, SG_F_CTRL + 2); cs.e_lon_int = fld(dy, SG_F_CTRL + 3);
    } else {
#pragma unroll
        for (int c = 0; c < 6; ++c) pose[c] = cst[(CS_POSE + c) * NP];
        present = cst[CS_PRESENT * NP] != 0.0;
        cs.speed = cst[(CS_CTRL + 0) * NP]; cs.e_lon_prev = cst[(CS_CTRL + 1) * NP];
        cs.e_lat_prev = cst[(CS_CTRL + 2) * NP]; cs.e_lon_int = cst[(CS_CTRL + 3) * NP];
        t = cst[CS_T * NP];
        prev_t = cst[CS_PREV_T * NP];
        m_avg = cst[(CS_METRIC + 0) * NP]; m_max = cst[(CS_METRIC + 1) * NP]; m_t = cst[(CS_METRIC + 2) * NP];
    }
    if (!active) present = false;

    Table T; // the lane's own knots (PIDAgent target, agent.py:145-148; spawn pose, scenario_gym.py:240-244)
    {
        const double *kn = p.knots + fld<int64_t>(st, ST_KNOT_OFF) * 7;
        T.x = kn; T.xs = 7; T.y = kn + 1; T.ys = 7; T.cs = 1;
        T.n = active ? (int)(meta >> 32) : 0;
    }
    double seg_hi;
    int seg_cur;
    double sgr[5] = {0.0, 0.0, 0.0, 0.0, 0.0}; // FAST: the published segment in registers as well (no LDS read, no wait, per step)
    auto seg_publish = [&](const Segment &S) {
        lds.seg[0][lane] = S.x_lo;
        lds.seg[1][lane] = S.ylo[0]; lds.seg[2][lane] = S.ylo[1];
        lds.seg[3][lane] = S.sl[0]; lds.seg[4][lane] = S.sl[1];
        if (FAST) { sgr[0] = S.x_lo; sgr[1] = S.ylo[0]; sgr[2] = S.ylo[1]; sgr[3] = S.sl[0]; sgr[4] = S.sl[1]; }
        seg_hi = S.x_hi;
        seg_cur = S.cur;
    };
    {
        Segment S;
        S.cur = seg_locate(T, t);
        seg_load(T, S);
        seg_publish(S);
    }
    double *out = tab + (q * (size_t)(p.tab_steps + 1) + (size_t)row0) * CT_W;      // plane 0 rows of this lane
    double *out1 = out + (size_t)p.n_ctl_pad * ((size_t)(p.tab_steps + 1) * CT_W); // plane 1
    double *out2 = out1 + (size_t)p.n_ctl_pad * ((size_t)(p.tab_steps + 1) * CT_W); // plane 2
    // RIDERS: a replay lane's own table (union grid / own knots) and its current segment, all six channels
    const bool replay_lane = RIDERS && (kind == SG_KIND_REPLAY || kind == SG_KIND_AGENT_REPLAY);
    const double max_t = fld(st, ST_MAX_T);
    const bool is_static = (int)(meta >> 32) == 1;
    Table TR{};
    Segment SR{};
    if (RIDERS) {
        TR = lane_table(p, replay_lane ? kind : SG_KIND_NONE, p.sstat[r], (int)(ent - r * (uint32_t)p.EP), st);
        if (!replay_lane) TR.n = 0;
        SR.cur = seg_locate(TR, t);
        seg_load(TR, SR);
    }
    sg_loads_done();
    // launch-invariant part of the fast path's vote; reciprocals of the step-invariant denominators
    const bool fast_kind = FAST && !p.ctl_general && sg_all(!active || (kind == SG_KIND_AGENT_PID && bl > 0.0 && bl < 0x1p400));
    const RecipDiv rd_l(active ? bl : 1.0), rd_10(10.0);
    double cpr[9]; // FAST: the controller parameters in registers
#pragma unroll
    for (int c = 0; c < 9; ++c) cpr[c] = FAST ? fld(st, ST_CTRL + c) : 0.0;
    sg_loads_done();

    for (int k = 0; k < n_steps; ++k) {
        const double *Kp = SG_TRIG;
        if (!FAST) asm volatile("" : "+s"(Kp)); // (FAST has registers to spare: the coefficients may live in them for the whole launch)
        ConstTbl K = (ConstTbl)Kp;
        const double next_t = t + timestep; // the rollout kernel's clock, scenario_gym.py:229
        const double state_dt = t - prev_t;
        const double dt = next_t - t;
        if (FAST && fast_kind && sg_all(!active || (present && !(next_t > seg_hi) && __builtin_fabs(pose[3]) < 1.0e5))) {
            // ---- PIDController._step + VehicleController._step (controller.py:205-258, 105-140), straight line ----
            const double dq = next_t - sgr[0];
            const double tx = sgr[3] * dq + sgr[1], ty = sgr[4] * dq + sgr[2];
            double sin_h, cos_h;
            sg_sincos_core(pose[3], sin_h, cos_h, K);
            const double e0 = tx - pose[0], e1 = ty - pose[1];
            const double e_lon = cos_h * e0 + sin_h * e1;
            const double e_lat = -sin_h * e0 + cos_h * e1;
            const double speed0 = cs.speed;
            const double g_mid = 1.0 - rd_10.div(0.9 * (speed0 - 5.0)); // (speed in (5, 15]: the numerator is in RecipDiv's range)
            const double gain = (speed0 > 5.0 && speed0 <= 15) ? g_mid : (speed0 > 15 ? 0.1 : 1.0);
            const RecipDiv rd(state_dt);
            const double d_lat = e_lat - cs.e_lat_prev, d_lon = e_lon - cs.e_lon_prev;
            const double e_lat_D = rd.div(d_lat);
            const double kp = cpr[SG_C_STEER_KP] * gain, kd = cpr[SG_C_STEER_KD] * gain;
            double steer = kp * e_lat + kd * e_lat_D;
            const double e_lon_D = rd.div(d_lon);
            const double e_lon_I = cs.e_lon_int + e_lon * state_dt;
            double accel = cpr[SG_C_ACCEL_KP] * e_lon + cpr[SG_C_ACCEL_KD] * e_lon_D + cpr[SG_C_ACCEL_KI] * e_lon_I;
            accel = __builtin_fabs(e_lon) > 0.1 ? accel : 0.0;
            const double max_steer = cpr[SG_C_MAX_STEER], max_accel = cpr[SG_C_MAX_ACCEL];
            const double max_speed = cpr[SG_C_MAX_SPEED], allow_rev = cpr[SG_C_ALLOW_REVERSE];
            accel = __builtin_fmin(__builtin_fmax(accel, -max_accel), max_accel);
            steer = __builtin_fmin(__builtin_fmax(steer, -max_steer), max_steer);
            const double dxs = speed0 * cos_h, dys = speed0 * sin_h;
            // tan(steer): the polynomial below 0.67434, sin / cos above (sg_tan); a saturated steering angle is common
            // enough among 64 lanes that both live here, the second under a wave-uniform branch
            double tan_s = sg_tan_poly(steer, K);
            const bool steep = !(__builtin_fabs(steer) < 0.67434);
            if (sg_any(steep & active)) {
                double s2, c2;
                sg_sincos_core(steer, s2, c2, K);
                tan_s = steep ? s2 / c2 : tan_s;
            }
            const double hnum = speed0 * tan_s;
            const double dh = hnum == 0.0 ? hnum : rd_l.div(hnum); // (+-0 / l = +-0 for l > 0)
            const double nx = pose[0] + dxs * dt, ny = pose[1] + dys * dt, nh = pose[3] + dh * dt;
            double nspeed = speed0 + accel * dt;
            nspeed = allow_rev == 0.0 ? __builtin_fmax(0.0, nspeed) : nspeed;
            nspeed = max_speed == max_speed ? __builtin_fmin(max_speed, nspeed) : nspeed;
            // the one vote on everything the straight-line forms assumed
            bool ok = rd.safe(d_lat) & rd.safe(d_lon) & (__builtin_fabs(steer) < 1.0e5) & (rd_l.safe(hnum) | (hnum == 0.0));
            // State.update_statistics for the ego + its metrics (state.py:230-239, metrics/trajectory.py:19-24, 41-44)
            double n_avg = m_avg, n_max = m_max, n_mt = m_t;
            if (metrics) { // (launch-uniform)
                const RecipDiv rdt(dt), rnt(next_t);
                const double ax = nx - pose[0], ay = ny - pose[1];
                const double az = pose[2] - pose[2]; // z stays (controller.py:126-131): +0 unless it is not finite
                const double speed = sg_norm3(rdt.div(ax), rdt.div(ay), 0.0); // (+0 / dt = +0)
                const double w = rnt.div(m_t);
                n_avg = m_avg + (1.0 - w) * (speed - m_avg);
                n_max = __builtin_fmax(speed, m_max);
                n_mt = next_t;
                ok = ok & (!is_ego | (rdt.safe(ax) & rdt.safe(ay) & rnt.safe(m_t) & (dt > 0.0) & (az == 0.0)));
            }
            if (sg_all(!active || ok)) {
                cs.e_lat_prev = e_lat; cs.e_lon_prev = e_lon; cs.e_lon_int = e_lon_I; cs.speed = nspeed;
                pose[0] = nx; pose[1] = ny; pose[3] = nh;
                if (is_ego) { m_avg = n_avg; m_max = n_max; m_t = n_mt; }
                prev_t = t;
                t = next_t;
                *reinterpret_cast<double4 *>(out + (size_t)k * CT_W) = make_double4(pose[0], pose[1], pose[3], cs.speed);
                *reinterpret_cast<double4 *>(out1 + (size_t)k * CT_W) = make_double4(cs.e_lon_prev, cs.e_lat_prev, cs.e_lon_int, 0.0);
                if (is_ego && metrics) *reinterpret_cast<double4 *>(out2 + (size_t)k * CT_W) = make_double4(m_avg, m_max, m_t, 0.0);
                continue;
            }
        }
        double act_a = 0.0, act_s = 0.0;
        if (kind == SG_KIND_AGENT_VEHICLE && actions) {
            const double *a = actions + ((size_t)(k0 + k) * p.R + r) * 2;
            act_a = a[0];
            act_s = a[1];
        }
        if (next_t > seg_hi) {
            Segment S;
            S.x_hi = seg_hi;
            S.cur = seg_cur;
            seg_advance(T, S, next_t);
            seg_publish(S);
            sg_loads_done();
        }
        double np_[6];
        const double dq = next_t - lds.seg[0][lane];
        np_[0] = lds.seg[3][lane] * dq + lds.seg[1][lane]; // PID target (x, y) at next_t
        np_[1] = lds.seg[4][lane] * dq + lds.seg[2][lane];
        bool npres = false;
        if (RIDERS && replay_lane) {
            if (next_t > SR.x_hi) { seg_advance(TR, SR, next_t); sg_loads_done(); }
            const double dqr = next_t - SR.x_lo;
#pragma unroll
            for (int c = 0; c < 6; ++c) np_[c] = SR.sl[c] * dqr + SR.ylo[c];
            npres = kind == SG_KIND_REPLAY ? (p.persist || is_static || (next_t >= min_t && next_t <= max_t))  // batch.py:45-52
                                           : (present || min_t >= t);                                         // scenario_gym.py:233-244
        } else if (present) {
            npres = true;
            const double tx = np_[0], ty = np_[1];
#pragma unroll
            for (int c = 0; c < 6; ++c) np_[c] = pose[c];
            double sin_h, cos_h;
            sg_sincos(pose[3], sin_h, cos_h, K);
            auto cp = [&](int q) -> double { return lds.ctrl[q][lane]; };
            if (kind == SG_KIND_AGENT_PID)
                pid_step(cs, cp, bl, state_dt, dt, tx, ty, sin_h, cos_h, np_, K);
            else
                vehicle_step(cs, cp, bl, dt, act_a, act_s, sin_h, cos_h, np_, K);
        } else if (active && min_t >= t) { // spawn at the trajectory position: all six channels of the bracket
            npres = true;
            Segment S;
            S.cur = seg_cur;
            seg_load(T, S);
            sg_loads_done();
            const double dqs = next_t - S.x_lo;
#pragma unroll
            for (int c = 0; c < 6; ++c) np_[c] = S.sl[c] * dqs + S.ylo[c];
        }
        if (is_ego && npres && metrics) { // State.update_statistics for this lane (state.py:230-239) + the ego metrics
            double prev[6];
            if (!present) { // newcomer: previous pose from the extrapolated trajectory, state.py:219-222
                own_position_extrap(T.x, T.n, t, prev);
                sg_loads_done();
            } else {
#pragma unroll
                for (int c = 0; c < 6; ++c) prev[c] = pose[c];
            }
            const double v0 = (np_[0] - prev[0]) / dt, v1 = (np_[1] - prev[1]) / dt, v2 = (np_[2] - prev[2]) / dt;
            const double speed = sg_norm3(v0, v1, v2);
            const double w = m_t / next_t; // EgoAvgSpeed._step, metrics/trajectory.py:19-24
            m_avg += (1.0 - w) * (speed - m_avg);
            m_t = next_t;
            m_max = __builtin_fmax(speed, m_max); // EgoMaxSpeed, :41-44
        }
        present = npres;
        if (npres) {
#pragma unroll
            for (int c = 0; c < 6; ++c) pose[c] = np_[c];
        }
        prev_t = t;
        t = next_t;
        *reinterpret_cast<double4 *>(out + (size_t)k * CT_W) = make_double4(pose[0], pose[1], pose[3], cs.speed);
        *reinterpret_cast<double4 *>(out1 + (size_t)k * CT_W) = make_double4(cs.e_lon_prev, cs.e_lat_prev, cs.e_lon_int, 0.0);
        if (is_ego && metrics) *reinterpret_cast<double4 *>(out2 + (size_t)k * CT_W) = make_double4(m_avg, m_max, m_t, 0.0);
        if (RIDERS) *reinterpret_cast<double4 *>(out2 + (size_t)k * CT_W) = make_double4(pose[2], pose[4], pose[5], present ? 1.0 : 0.0);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) cst[(CS_POSE + c) * NP] = pose[c];
    cst[CS_PRESENT * NP] = present ? 1.0 : 0.0;
    cst[(CS_CTRL + 0) * NP] = cs.speed; cst[(CS_CTRL + 1) * NP] = cs.e_lon_prev;
    cst[(CS_CTRL + 2) * NP] = cs.e_lat_prev; cst[(CS_CTRL + 3) * NP] = cs.e_lon_int;
    cst[CS_T * NP] = t;
    cst[CS_PREV_T * NP] = prev_t;
    cst[(CS_METRIC + 0) * NP] = m_avg; cst[(CS_METRIC + 1) * NP] = m_max; cst[(CS_METRIC + 2) * NP] = m_t;
}

// the ordinary form: one workgroup per 64 controlled lanes, the LDS belongs to this call
template <bool FAST, bool RIDERS = false>
__device__ __forceinline__ void control_body(const Params &p, double timestep, int n_steps, int first, int k0,
                                             const double *actions /*[n][R][2]*/, double *tab, int row0, int metrics)
{
    __shared__ CtlLds lds;
    control_body_l<FAST, RIDERS>(lds, blockIdx.x, p, timestep, n_steps, first, k0, actions, tab, row0, metrics);
}

#ifdef SG_UNIT_CTL // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64, SG_CTL_WAVES) void control_kernel(Params p, double timestep, int n_steps, int first, int k0,
                                                     const double *actions /*[n][R][2]*/, double *tab, int row0, int metrics)
{
    control_body<false>(p, timestep, n_steps, first, k0, actions, tab, row0, metrics);
}
#endif // SG_UNIT_CTL
#ifdef SG_UNIT_CTL // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64, 2) void control_kernel_riders(Params p, double timestep, int n_steps, int first, int k0,
                                                               const double *actions /*[n][R][2]*/, double *tab, int row0, int metrics)
{
    control_body<false, true>(p, timestep, n_steps, first, k0, actions, tab, row0, 0);
}
#endif // SG_UNIT_CTL
#ifdef SG_UNIT_CTL // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64, 1) void control_kernel_fast(Params p, double timestep, int n_steps, int first, int k0,
                                                             const double *actions /*[n][R][2]*/, double *tab, int row0, int metrics)
{
    // (s_setprio(3) here shortens this kernel's launches by a quarter -- its 64 wavefronts share their SIMDs with the rollout
    // kernels' -- but the 4096 x 64 table path is bound by the rollout kernels, not by this chain: 2-3 % slower overall.
    // With the collision pass compiled out the chain IS the bound and the priority is worth 98-107 -> 114-126 G: HISTORY.md, round 4)
#ifdef SG_CTL_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    control_body<true>(p, timestep, n_steps, first, k0, actions, tab, row0, metrics);
}
#endif // SG_UNIT_CTL

} // namespace sg
