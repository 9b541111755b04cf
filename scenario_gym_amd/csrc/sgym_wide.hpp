// sgym_wide.hpp -- scenarios of MORE THAN 512 entities: the step as four kernels over any number of workgroups.
//
// The reference has no entity limit (State.collisions is a Python loop over an STRtree, state/utils.py:10-49); the fused
// rollout kernels (sgym_device.hpp) keep one scenario inside one workgroup and stop at 512 entities.  Beyond that a scenario
// spans several workgroups, which can only meet at kernel boundaries, so ScenarioGym.step (scenario_gym.py:227-254) becomes
//   wide_move_kernel     one thread per entity: the new pose -- BatchReplayEntity / replay agent interpolation, PID and vehicle
//                        controllers, the social force over ALL pedestrians of the scenario (staged through LDS 256 at a
//                        time, neighbours in entity order) -- into a scratch row; nothing of the state is touched yet
//   wide_commit_kernel   State.update_poses / update_statistics for the entity, its row stores, its fp64 box corners and
//                        bounding circle for the collision pass
//   wide_collide_kernel  State.collisions: every present entity against every other one of its scenario (circles staged
//                        through LDS, exact fp64 separating-axis test on the pairs whose circles overlap; equal geometries
//                        never list each other, utils.py:59)
//                        ... and stand for their last owner in everybody else's row (state/utils.py:32-40): noted here, moved
//                        by the first part of wide_finish_kernel (rare)
//   wide_finish_kernel   one workgroup per scenario: the clock, terminal conditions, ego metrics, CollisionMetric events
// in that order, once per step.  Same arithmetic as the fused kernels and the oracle (plain IEEE operations: ExactArith), so
// the same bits; 4 launches per step instead of thousands of steps per launch -- the price of not having a ceiling.
// (Nothing is refused at this width any more: round 5.)
// the noise stream mode (the counter-based generator works).
#pragma once
#include "sgym_device.hpp"

namespace sg {

struct WideArgs {
    double *scr;          // [NE][16] per entity: new pose 6, new presence, velocity 6 (reset), spare
    double *cor;          // [NE][8] fp64 corners of the committed pose
    double *circ;         // [NE][4] bounding circle of the box: cx, cy, radius (NaN cx: absent)
    uint64_t *last_row;   // [R][WV] CollisionMetric.last_timestep
    int32_t *last_same;   // [NE] the LAST entity of the scenario whose box is bit-identical to this one's (itself: nobody's is)
    int32_t *walkers;     // [R] pedestrians that walk in this step (noise stream mode: the scenario's stream advances by two per walker)
    uint32_t *dup;        // [R] bit 0: some entity of the scenario has a twin this step (wide_owner_row has work); bit 1: some
                          // entity's collision row is not empty; bit 2: entity 0's is not (the terminal conditions of wide_finish_kernel)
    const double *actions; // [R][2] of THIS step or nullptr
    int no_peds;          // the batch has no pedestrian agents: no neighbour staging, move + commit in one launch
    int mode;             // 0 step, 1 reset (State.reset for every scenario), 2 reset of the scenarios in p.reset_mask
    int force;
};
enum { WS_NP = 0, WS_NPRES = 6, WS_VEL = 7, WS_FPX = 13, WS_FPY = 14, WS_CUR = 15 /* the entity's knot segment of the previous step */, WS_W = 16 };

struct WideEnt {
    int r, e;
    uint32_t g;
    LanePtr st, dy;
    __device__ __forceinline__ WideEnt(const Params &p, int r_, int e_)
        : r(r_), e(e_), g((uint32_t)r_ * p.EP + e_),
          st(p.stat + (size_t)(((uint32_t)r_ * p.EP + e_) >> 6) * (ST_COUNT * 64), ((((uint32_t)r_ * p.EP + e_)) & 63) * 8u),
          dy(p.dyn + (size_t)(((uint32_t)r_ * p.EP + e_) >> 6) * ((size_t)p.FROWS * 64), ((((uint32_t)r_ * p.EP + e_)) & 63) * 8u) {}
};

__device__ __forceinline__ bool wide_runs(const Params &p, const WideArgs &wa, int r)
{
    if (wa.mode == 1) return true;
    if (wa.mode == 2) return p.reset_mask[r] != 0;
    return wa.force || !p.sdyn[r].done;
}

#ifdef SG_UNIT_WIDE
// ---- new poses: scenario_gym.py:233-245 (step) / State.reset, state.py:106-143 (reset) --------------------------------------
__device__ __forceinline__ void wide_move_body(const Params &p, double timestep, const WideArgs &wa)
{
    __shared__ double s_px[256], s_py[256], s_vx[256], s_vy[256];
    __shared__ unsigned char s_ok[256];
    constexpr int WIDE_NB = 24;
    __shared__ unsigned char s_nb[WIDE_NB][256]; // per thread: the tile slots inside its sensor circle, in order
    const int r = blockIdx.y, tid = threadIdx.x, e = blockIdx.x * 256 + tid;
    if (!wide_runs(p, wa, r)) return;
    const bool in = e < p.E;
    const WideEnt w(p, r, in ? e : 0);
    const ScenStatic &ss = p.sstat[r];
    const sg_scenario_state &sd = p.sdyn[r];
    const int64_t meta = fld<int64_t>(w.st, ST_META);
    const int kind = in ? (int)(meta & 0xff) : SG_KIND_NONE;
    const int nk = (int)(meta >> 32);
    const bool is_static = nk == 1;
    const double min_t = fld(w.st, ST_MIN_T), max_t = fld(w.st, ST_MAX_T);
    const double *kn = p.knots + fld<int64_t>(w.st, ST_KNOT_OFF) * 7;
    double *scr = wa.scr + (size_t)w.g * WS_W;
    double np_[6] = {0, 0, 0, 0, 0, 0};
    bool npres = false;
    if (wa.mode != 0) { // ---- State.reset(t0) ----
        const double t = ss.t0;
        double vel[6] = {0, 0, 0, 0, 0, 0};
        if (kind != SG_KIND_NONE) {
            const bool inside = (t >= min_t) && (t <= max_t);
            if (is_static || inside) { own_position_extrap(kn, nk, t, np_); npres = true; }
            else if (p.persist) {
                const double *rowp = t < min_t ? kn : kn + (size_t)(nk - 1) * 7;
                for (int c = 0; c < 6; ++c) np_[c] = rowp[1 + c];
                npres = true;
            }
            if (npres && inside) { // Trajectory.velocity_at_t, trajectory.py:243-273
                const double eps = 1e-4;
                double a[6], b[6];
                own_position_extrap(kn, nk, t + eps / 2, a);
                own_position_extrap(kn, nk, t - eps / 2, b);
                for (int c = 0; c < 6; ++c) vel[c] = (a[c] - b[c]) / eps;
            }
        }
        if (in) {
            for (int c = 0; c < 6; ++c) { scr[WS_NP + c] = np_[c]; scr[WS_VEL + c] = vel[c]; }
            scr[WS_NPRES] = npres ? 1.0 : 0.0;
            scr[WS_FPX] = scr[WS_FPY] = 0.0;
            scr[WS_CUR] = 0.0; // (no segment yet: the first step searches)
        }
        return;
    }
    // ---- one step ----
    const double t = sd.t, prev_t = sd.prev_t;
    const double next_t = t + timestep, state_dt = t - prev_t, dt = next_t - t;
    const bool present = in && kind != SG_KIND_NONE && fld<uint64_t>(w.dy, SG_F_PRESENT) != 0;
    double pose[6];
    for (int c = 0; c < 6; ++c) pose[c] = fld(w.dy, SG_F_POSE + c);
    const double velx = fld(w.dy, SG_F_VEL + 0), vely = fld(w.dy, SG_F_VEL + 1);
    CtrlState cs{fld(w.dy, SG_F_CTRL + 0), fld(w.dy, SG_F_CTRL + 1), fld(w.dy, SG_F_CTRL + 2), fld(w.dy, SG_F_CTRL + 3)};
    int goal_idx = (int)cs.e_lon_prev; // (pedestrians keep goal_idx in the second controller row)
    double fpx = 0.0, fpy = 0.0;
    ConstTbl K = (ConstTbl)SG_TRIG;
    // the trajectory position at next_t: union grid for batch-replay entities, own knots for agents (constant outside)
    // (every step is a launch of its own: the segment the entity was in a step ago is kept in its scratch row -- a step later the
    // clock is still inside it, or in the next one; the binary search over the scenario's union grid, up to seventeen
    // dependent loads, was most of this kernel's time on 1,024-entity scenarios)
    const int cur_prev = in ? (int)scr[WS_CUR] : 0;
    int cur_now = cur_prev;
    auto traj_at = [&](double tq, double (&out)[6]) {
        Table T = lane_table(p, kind, ss, e, w.st);
        Segment S;
        int c = cur_prev;
        if (c >= 1 && c + 1 <= T.n - 1) { // seg_locate's answer is the first knot at or after tq: c, if the knot before it is earlier
            const double xa = T.X(c - 1), xb = T.X(c), xc = T.X(c + 1);
            if (xa < tq && tq <= xb) {}
            else if (xb < tq && tq <= xc) c = c + 1;
            else c = seg_locate(T, tq);
        } else {
            c = seg_locate(T, tq);
        }
        S.cur = c;
        cur_now = c;
        seg_load(T, S);
        const double dq = tq - S.x_lo;
        for (int c = 0; c < 6; ++c) out[c] = S.sl[c] * dq + S.ylo[c];
    };
    // ---- the social force needs every pedestrian of the scenario: all threads stage, pedestrian threads accumulate ----
    const bool is_ped = kind == SG_KIND_AGENT_PEDESTRIAN;
    // this pedestrian's behaviour model: the handle's, or its own row where the batch mixes models (sg_set_ped_models;
    // pedestrian/agent.py:18-41) -- the force on a pedestrian is computed with ITS parameters from its neighbours' states
    sg_social_force sf = p.sf;
    int beh = p.ped_behaviour;
    double nstd_lon = p.noise_std_lon, nstd_lat = p.noise_std_lat;
    if (p.n_ped_models > 1 && is_ped) {
        const double *mm = p.ped_models + (size_t)p.model_of[w.g] * PM_W;
        beh = (int)mm[PM_BEHAVIOUR];
        sf = *reinterpret_cast<const sg_social_force *>(mm + PM_SF);
        nstd_lon = mm[PM_STD_LON];
        nstd_lat = mm[PM_STD_LAT];
    }
    const double *wp = nullptr;
    int nwp = 0;
    bool go = false;
    double fx = 0.0, fy = 0.0, vdes = 0.0, hs = 0.0, hc = 1.0, radius = 0.0;
    if (is_ped && present) {
        const int64_t rt = fld<int64_t>(w.st, ST_ROUTE);
        wp = p.routes + (rt & 0xffffffffffffll) * 2;
        nwp = (int)(rt >> 48);
        if (goal_idx <= nwp - 1) goal_idx = ped_goal_update(wp, nwp, pose[0], pose[1]);
        if (goal_idx <= nwp - 1) {
            go = true;
            double gx = wp[2 * goal_idx] - pose[0], gy = wp[2 * goal_idx + 1] - pose[1]; // _force_to_goal, social_force.py:119-138
            double gn = sg_norm2(gx, gy);
            if (gn == 0) gn += 0.000000001;
            vdes = fld(w.st, ST_CTRL + SG_C_PED_SPEED_DESIRED);
            const double inv_tau = 1 / sf.relaxation_time;
            fx = inv_tau * (vdes * (gx / gn) - velx);
            fy = inv_tau * (vdes * (gy / gn) - vely);
            sg_sincos(fld(w.st, ST_CTRL + SG_C_PED_HEAD_ROT), hs, hc, K);
            radius = fld(w.st, ST_CTRL + SG_C_PED_RADIUS);
            if (beh == SG_PED_RANDOM_WALK) { fx = gx; fy = gy; } // RandomWalk._step: the vector to the goal point
        }
    }
    // The noise stream (sg_set_ped_noise, SG_NOISE_STREAM): two variates per WALKING pedestrian in entity order, as the reference
    // draws them from numpy's global generator (social_force.py:106-108) -- this entity's place in that order is the number of
    // walkers before it in the whole scenario, over all workgroups: every workgroup looks at every tile (parity runs, not
    // timing runs).  A pedestrian walks iff it is present and has a goal left after this step's goal update.
    int walkers_before = 0, walkers = 0;
    if (p.noise_mode == 1 && !wa.no_peds) { // (grid-uniform)
        __shared__ uint64_t s_walk[4];
        for (int c0 = 0; c0 < p.EP; c0 += 256) {
            const int j = c0 + tid;
            bool gj = false;
            if (j < p.E) {
                const WideEnt o(p, r, j);
                if ((int)(fld<int64_t>(o.st, ST_META) & 0xff) == SG_KIND_AGENT_PEDESTRIAN && fld<uint64_t>(o.dy, SG_F_PRESENT) != 0) {
                    const int64_t rt = fld<int64_t>(o.st, ST_ROUTE);
                    const double *wj = p.routes + (rt & 0xffffffffffffll) * 2;
                    const int nj = (int)(rt >> 48);
                    int gi = (int)fld(o.dy, SG_F_CTRL + 1);
                    if (gi <= nj - 1) gi = ped_goal_update(wj, nj, fld(o.dy, SG_F_POSE + 0), fld(o.dy, SG_F_POSE + 1));
                    gj = gi <= nj - 1;
                }
            }
            __syncthreads();
            const uint64_t b = __ballot(gj);
            if ((tid & 63) == 0) s_walk[tid >> 6] = b;
            __syncthreads();
            for (int k = 0; k < 4; ++k) {
                const int base = c0 + k * 64;
                const uint64_t m = s_walk[k];
                walkers += __builtin_popcountll(m);
                if (base + 64 <= e) walkers_before += __builtin_popcountll(m);
                else if (base <= e) walkers_before += __builtin_popcountll(m & ((1ull << (e - base)) - 1));
            }
        }
        if (e == 0) wa.walkers[r] = walkers;
    }
    // (grid-uniform: the tiles are staged when some pedestrian of the batch may follow SocialForce; a RandomWalk lane sits the pairs out)
    const bool pairs = (p.n_ped_models > 1 || p.ped_behaviour != SG_PED_RANDOM_WALK) && !wa.no_peds;
    const bool lane_pairs = beh != SG_PED_RANDOM_WALK;
    const double k2_scale = sf.ped_repulse_V / sf.ped_repulse_sigma;
    // (the loop bounds are uniform over the grid row of the scenario: every thread of every block walks all chunks)
    for (int c0 = 0; c0 < p.EP && pairs; c0 += 256) {
        const int j = c0 + tid;
        __syncthreads();
        {
            bool ok = false;
            double jx = 0, jy = 0, jvx = 0, jvy = 0;
            if (j < p.E) {
                const WideEnt o(p, r, j);
                const int64_t m2 = fld<int64_t>(o.st, ST_META);
                ok = (int)(m2 & 0xff) != SG_KIND_NONE && ((m2 >> 8) & 0xff) == 1 && fld<uint64_t>(o.dy, SG_F_PRESENT) != 0;
                jx = fld(o.dy, SG_F_POSE + 0); jy = fld(o.dy, SG_F_POSE + 1);
                jvx = fld(o.dy, SG_F_VEL + 0); jvy = fld(o.dy, SG_F_VEL + 1);
            }
            s_px[tid] = jx; s_py[tid] = jy; s_vx[tid] = jvx; s_vy[tid] = jvy; s_ok[tid] = ok;
        }
        __syncthreads();
        if (go && lane_pairs) {
            // PedestrianSensor.get_nearby_pedestrians in entity order (sensor.py:55-64).  A neighbour is rare per slot but not per
            // wavefront: evaluated where it is found, every lane's neighbour would send all 64 lanes through the pair force.
            // The scan only notes the slots inside the circle about the 64-gon (a cheap superset of sg_in_radius); the lane
            // then walks its own list IN ORDER (the force is a sum: the order of the terms is the reference's).  A full list
            // is worked off before the scan goes on, which keeps that order.
            const int n = min(256, p.E - c0);
            const double r2 = radius * radius * (1.0 + 1e-9);
            int nc = 0;
            auto work_off = [&]() {
                for (int k = 0; k < nc; ++k) {
                    const int q = s_nb[k][tid];
                    const double ox = s_px[q], oy = s_py[q];
                    if (!sg_in_radius(pose[0], pose[1], radius, ox, oy, p.gon)) continue;
                    const double ovx = s_vx[q], ovy = s_vy[q];
                    const double vmag = sg_norm2(ovx, ovy) + 0.0000000001;
                    const double odx = ovx / vmag, ody = ovy / vmag, step = vmag * (next_t - t);
                    ExactArith EA;
                    double c1x, c1y, c2x, c2y;
                    ped_pair<false, false>(EA, sf, k2_scale, pose[0], pose[1], hs, hc, ox, oy, ovx, ovy, odx, ody, step, c1x, c1y, c2x, c2y);
                    ped_accumulate(sf, c1x, c1y, c2x, c2y, fx, fy);
                }
                nc = 0;
            };
            for (int q = 0; q < n; ++q) {
                const double dx = s_px[q] - pose[0], dy = s_py[q] - pose[1];
                if (c0 + q == e || !s_ok[q] || !(dx * dx + dy * dy <= r2)) continue; // (sg_in_radius: false beyond r2 as well)
                if (nc == WIDE_NB) work_off();
                s_nb[nc++][tid] = (unsigned char)q;
            }
            work_off();
        }
    }
    // the boundary terms of SocialForce._step, after the neighbours (social_force.py:83-104); RandomWalk has none
    if (go && pairs && lane_pairs) {
        Params q = p; // (the boundary terms read the impenetrable-surface parameters of the lane's model)
        q.sf = sf;
        ped_boundary_terms(q, r, pose[0], pose[1], fx, fy);
    }
    // ---- the pose ----
    if (kind == SG_KIND_REPLAY) { // BatchReplayEntity.step, batch.py:34-53
        npres = p.persist || is_static || (next_t >= min_t && next_t <= max_t);
        traj_at(next_t, np_);
    } else if (kind >= SG_KIND_AGENT_REPLAY && in) {
        if (present && kind == SG_KIND_AGENT_EXTERNAL) {
            // the caller ran agent.step(state) (agent.py:52-57): its pose, or None = NaN (scenario_gym.py:233-239)
            const double *ep = p.ext_pose + (size_t)w.g * 6;
            if (ep[0] == ep[0]) {
                npres = true;
                for (int c = 0; c < 6; ++c) np_[c] = ep[c];
            } else if (p.persist) {
                npres = true;
                for (int c = 0; c < 6; ++c) np_[c] = pose[c];
            }
        } else if (present) {
            npres = true;
            if (kind == SG_KIND_AGENT_REPLAY) {
                traj_at(next_t, np_);
            } else if (kind == SG_KIND_AGENT_PID || kind == SG_KIND_AGENT_VEHICLE) {
                double tgt[6];
                traj_at(next_t, tgt);
                for (int c = 0; c < 6; ++c) np_[c] = pose[c];
                const double bl = fld(w.st, ST_BL);
                double sin_h, cos_h;
                sg_sincos(pose[3], sin_h, cos_h, K);
                const LanePtr st_c = w.st;
                auto cp = [&](int q) -> double { return fld(st_c, ST_CTRL + q); };
                if (kind == SG_KIND_AGENT_PID) pid_step(cs, cp, bl, state_dt, dt, tgt[0], tgt[1], sin_h, cos_h, np_, K);
                else {
                    double aa = 0.0, as = 0.0;
                    if (wa.actions) { aa = wa.actions[(size_t)r * 2]; as = wa.actions[(size_t)r * 2 + 1]; }
                    vehicle_step(cs, cp, bl, dt, aa, as, sin_h, cos_h, np_, K);
                }
            } else if (is_ped) {
                PedNoise nz{0.0, 0.0, false};
                if (p.noise_mode == 2) {
                    double z0, z1;
                    sg_noise_pair(p.noise_seed, (uint32_t)r, (uint32_t)e, (uint32_t)sd.n_steps, z0, z1, K);
                    nz = PedNoise{nstd_lon * z0, nstd_lat * z1, true};
                } else if (p.noise_mode == 1 && go) {
                    const long long at = sd.noise_pos + 2ll * walkers_before;
                    const bool inside = at + 1 < p.noise_len;
                    const double *z = p.noise_normals + (size_t)r * (size_t)p.noise_len + (inside ? at : 0);
                    nz = PedNoise{nstd_lon * (inside ? z[0] : 0.0), nstd_lat * (inside ? z[1] : 0.0), true};
                }
                ped_move(PedMoveModel{beh, sf.bias_lon, sf.bias_lat, sf.max_speed_factor}, go, fx, fy, vdes,
                         fld(w.st, ST_CTRL + SG_C_PED_MAX_SPEED), pose, state_dt, cs.speed, fpx, fpy, np_, K, nz);
                cs.e_lon_prev = (double)goal_idx;
            }
        } else if (min_t >= t) { // scenario_gym.py:240-244: spawn at the trajectory position of next_t
            npres = true;
            traj_at(next_t, np_);
        }
    }
    if (in) {
        for (int c = 0; c < 6; ++c) scr[WS_NP + c] = np_[c];
        scr[WS_NPRES] = npres ? 1.0 : 0.0;
        scr[WS_FPX] = fpx; scr[WS_FPY] = fpy;
        // controller state of this step (committed by wide_commit_kernel together with the pose)
        scr[WS_VEL + 0] = cs.speed; scr[WS_VEL + 1] = cs.e_lon_prev; scr[WS_VEL + 2] = cs.e_lat_prev; scr[WS_VEL + 3] = cs.e_lon_int;
        scr[WS_CUR] = (double)cur_now;
    }
}

// ---- State.update_poses / update_statistics (state.py:203-239), the row stores, corners + circle for the collision pass -------
__device__ __forceinline__ void wide_commit_body(const Params &p, double timestep, const WideArgs &wa)
{
    const int r = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= p.EP) return;
    if (e == 0) wa.dup[r] = 0; // (wide_collide_kernel, the next launch, raises it again when it meets twins)
    wa.last_same[(size_t)r * p.EP + e] = e; // (... and raises this to the last twin of the entity's geometry)
    const WideEnt w(p, r, e);
    double *circ = wa.circ + (size_t)w.g * 4, *cor = wa.cor + (size_t)w.g * 8;
    const bool runs = wide_runs(p, wa, r);
    const int64_t meta = fld<int64_t>(w.st, ST_META);
    const int kind = e < p.E ? (int)(meta & 0xff) : SG_KIND_NONE;
    const sg_scenario_state &sd = p.sdyn[r];
    const ScenStatic &ss = p.sstat[r];
    double pose[6];
    bool present;
    if (runs && kind != SG_KIND_NONE) {
        const double *scr = wa.scr + (size_t)w.g * WS_W;
        const bool npres = scr[WS_NPRES] != 0.0;
        double np_[6], vel[6];
        for (int c = 0; c < 6; ++c) np_[c] = scr[WS_NP + c];
        if (wa.mode != 0) { // reset: velocities from the trajectory, distances 0, controllers at rest
            for (int c = 0; c < 6; ++c) { vel[c] = scr[WS_VEL + c]; pose[c] = np_[c]; }
            present = npres;
            for (int c = 0; c < 6; ++c) { stf(w.dy, SG_F_POSE + c, pose[c]); stf(w.dy, SG_F_VEL + c, vel[c]); }
            stf(w.dy, SG_F_DIST, 0.0);
            stf(w.dy, SG_F_PRESENT, (uint64_t)present);
            stf(w.dy, SG_F_FORCE + 0, 0.0); stf(w.dy, SG_F_FORCE + 1, 0.0);
            double speed = present ? sg_norm2(vel[0], vel[1]) : 0.0; // controller.py:100-103
            if (kind == SG_KIND_AGENT_PEDESTRIAN) speed = 0.0;         // pedestrian/controller.py:21-23
            stf(w.dy, SG_F_CTRL + 0, speed); stf(w.dy, SG_F_CTRL + 1, 0.0); stf(w.dy, SG_F_CTRL + 2, 0.0); stf(w.dy, SG_F_CTRL + 3, 0.0);
            if (p.rec_cap > 0)
                for (int c = 0; c < 6; ++c)
                    p.rec_pose[(size_t)c * p.R * p.EP + (size_t)r * p.EP + e] = present ? pose[c] : __builtin_nan("");
        } else {
            const double t = sd.t, next_t = t + timestep, dt = next_t - t;
            const bool was = fld<uint64_t>(w.dy, SG_F_PRESENT) != 0;
            double prev[6];
            if (npres && !was) // newcomer: previous pose from the extrapolated trajectory, state.py:219-222
                own_position_extrap(p.knots + fld<int64_t>(w.st, ST_KNOT_OFF) * 7, (int)(meta >> 32), t, prev);
            else
                for (int c = 0; c < 6; ++c) prev[c] = fld(w.dy, SG_F_POSE + c);
            double d[6];
            for (int c = 0; c < 6; ++c) { d[c] = np_[c] - prev[c]; vel[c] = d[c] / dt; }
            present = npres;
            for (int c = 0; c < 6; ++c) pose[c] = npres ? np_[c] : fld(w.dy, SG_F_POSE + c);
            if (npres) {
                for (int c = 0; c < 6; ++c) { stf(w.dy, SG_F_POSE + c, pose[c]); stf(w.dy, SG_F_VEL + c, vel[c]); }
                stf(w.dy, SG_F_DIST, fld(w.dy, SG_F_DIST) + sg_norm3(d[0], d[1], d[2]));
            }
            stf(w.dy, SG_F_PRESENT, (uint64_t)present);
            if (kind >= SG_KIND_AGENT_REPLAY && was) { // an agent that stepped: its controller state
                if (kind != SG_KIND_AGENT_REPLAY) {
                    stf(w.dy, SG_F_CTRL + 0, scr[WS_VEL + 0]); stf(w.dy, SG_F_CTRL + 1, scr[WS_VEL + 1]);
                    stf(w.dy, SG_F_CTRL + 2, scr[WS_VEL + 2]); stf(w.dy, SG_F_CTRL + 3, scr[WS_VEL + 3]);
                }
                if (kind == SG_KIND_AGENT_PEDESTRIAN) { stf(w.dy, SG_F_FORCE + 0, scr[WS_FPX]); stf(w.dy, SG_F_FORCE + 1, scr[WS_FPY]); }
            }
            const int steps = sd.n_steps + 1;
            if (p.rec_cap > 0 && steps < p.rec_cap)
                for (int c = 0; c < 6; ++c)
                    p.rec_pose[((size_t)steps * 6 + c) * p.R * p.EP + (size_t)r * p.EP + e] = present ? pose[c] : __builtin_nan("");
        }
        (void)ss;
    } else {
        present = kind != SG_KIND_NONE && fld<uint64_t>(w.dy, SG_F_PRESENT) != 0;
        for (int c = 0; c < 6; ++c) pose[c] = fld(w.dy, SG_F_POSE + c);
    }
    // the box for the collision pass (every scenario, also the ones that did not step: their rows are recomputed alike)
    if (present) {
        double s, c;
        sg_sincos(pose[3], s, c);
        const double bw = fld(w.st, ST_BW), bl = fld(w.st, ST_BL), bcx = fld(w.st, ST_BCX), bcy = fld(w.st, ST_BCY);
        double C[8];
        sg_corners(pose[0], pose[1], s, c, bw, bl, bcx, bcy, C);
        for (int k = 0; k < 8; ++k) cor[k] = C[k];
        const double cx = 0.25 * (C[0] + C[2] + C[4] + C[6]), cy = 0.25 * (C[1] + C[3] + C[5] + C[7]);
        double rad = 0.0;
        for (int k = 0; k < 4; ++k) rad = __builtin_fmax(rad, sg_norm2(C[2 * k] - cx, C[2 * k + 1] - cy));
        circ[0] = cx; circ[1] = cy; circ[2] = rad * (1.0 + 1e-12) + 1e-9 * (1.0 + __builtin_fabs(cx) + __builtin_fabs(cy));
    } else {
        circ[0] = __builtin_nan(""); circ[1] = 0.0; circ[2] = 0.0;
    }
}

static __global__ __launch_bounds__(256) void wide_move_kernel(Params p, double timestep, WideArgs wa) { wide_move_body(p, timestep, wa); }
static __global__ __launch_bounds__(256) void wide_commit_kernel(Params p, double timestep, WideArgs wa) { wide_commit_body(p, timestep, wa); }
// Scenarios without pedestrian agents: no entity reads another one's pose on its way to the new one, so the entity's own
// thread commits right away -- one launch less per step, and no neighbour staging (64 x 1,024 vehicles: 61 -> 54 us per step).
static __global__ __launch_bounds__(256) void wide_move_commit_kernel(Params p, double timestep, WideArgs wa)
{
    wide_move_body(p, timestep, wa);
    wide_commit_body(p, timestep, wa);
}

__device__ __forceinline__ bool wide_same(const double *a, const double *b)
{
    bool same = true;
    for (int k = 0; k < 8; ++k) same = same && (a[k] == b[k]);
    return same;
}

// ---- State.collisions(): every present entity against every other one of its scenario ---------------------------------------
// The row holds the entities whose boxes MEET this one's (bit j = entity j).  Bit-identical boxes never list each other
// (utils.py:59) and, seen from a third entity, stand for the last of their owners (the reference keys a dict by geometry,
// state/utils.py:32-40): the twins are noted here (last_same, dup) and wide_finish_kernel moves the bits -- a scan for the
// last owner inside this loop cost a thousand dependent global loads per hit (1 ms per step on 1,024 entities).
static __global__ __launch_bounds__(256) void wide_collide_kernel(Params p, WideArgs wa)
{
    __shared__ alignas(16) float s_fx[256], s_fy[256]; // fp32 centres of the tile's slots
    __shared__ unsigned int s_rmax;                     // bits of the tile's largest fp32 radius (radii are >= 0: ordered as uints)
    __shared__ uint64_t s_pres[4];                      // presence bits of the tile's slots
    // one workgroup = 256 entities of a scenario against ONE tile of 256 slots (blockIdx.x = entity tile * tiles + slot tile):
    // the four row words of that tile are this workgroup's alone, so the rows need no atomics, and a scenario of 1,024
    // entities is 16 workgroups instead of 4 (64 such scenarios were one wavefront per SIMD: 39 us of the 75 us step)
    const int n_tiles = (p.EP + 255) / 256, it = blockIdx.x / n_tiles, jt = blockIdx.x - it * n_tiles;
    const int r = blockIdx.y, tid = threadIdx.x, e = it * 256 + tid;
    const int W = p.FROWS - SG_F_COLL;
    const bool in = e < p.E;
    const WideEnt w(p, r, in ? e : 0);
    const double *circ = wa.circ + (size_t)w.g * 4, *A = wa.cor + (size_t)w.g * 8;
    const double cx = in ? circ[0] : __builtin_nan(""), cy = circ[1], rad = circ[2];
    const bool present = cx == cx;
    if (in)
        for (int q = jt * 4; q < jt * 4 + 4 && q < W; ++q) stf(w.dy, SG_F_COLL + q, (uint64_t)0);
    int last = e;
    bool hit = false; // this entity's row is not empty
    // Broad phase as the fused kernels walk their tiles (sgym_collide.hpp, all-pairs form): bounding circles in packed fp32, four
    // slots per LDS read, no branch in the scan -- thr - d2 leaves "outside" in the SIGN bit, which v_alignbit shifts into the
    // lane's mask.  Strictly conservative: the reach is this entity's radius + the LARGEST radius of the tile, both rounded
    // up, + 2^-18 x the coordinates for the fp32 conversion of the centres; the fp64 separating-axis test below decides.
    // A candidate is rare per lane but not per wavefront: the lane walks its own mask AFTER the scan.
    const float fx = (float)cx, fy = (float)cy;
    const float mag = __builtin_fabsf(fx) + __builtin_fabsf(fy);
    const float radf = (float)rad * 1.000001f;
    const v2f fx2 = {fx, fx}, fy2 = {fy, fy};
    for (int c0 = jt * 256, once = 0; once < 1; ++once) {
        if (tid == 0) s_rmax = 0u;
        __syncthreads();
        {
            const int j = c0 + tid;
            const double *cj = wa.circ + ((size_t)r * p.EP + min(j, p.EP - 1)) * 4;
            const double xj = j < p.E ? cj[0] : __builtin_nan("");
            const bool pj = xj == xj;
            s_fx[tid] = pj ? (float)xj : 0.0f;
            s_fy[tid] = pj ? (float)cj[1] : 0.0f;
            if (pj) atomicMax(&s_rmax, __float_as_uint((float)cj[2] * 1.000001f));
            const uint64_t b = __ballot(pj);
            if ((tid & 63) == 0) s_pres[tid >> 6] = b;
        }
        __syncthreads();
        if (!present) continue;
        const float reach = (radf + __uint_as_float(s_rmax)) * 1.000001f + 3.8146973e-6f * mag + 1e-5f;
        const float thr = reach * reach * 1.000001f;
        const v2f thr2 = {thr, thr};
        const int self = e - c0; // this lane's own slot, if it is inside the tile
#pragma unroll 1
        for (int k = 0; k < 4; ++k) { // 64 slots = one row word at a time
            uint32_t half[2];         // bit j = 1: slot j is OUTSIDE this lane's reach
#pragma unroll
            for (int h2 = 1; h2 >= 0; --h2) {
                uint32_t m = 0u;
#pragma unroll
                for (int q = 7; q >= 0; --q) {
                    const int jb = k * 64 + h2 * 32 + q * 4;
                    const v4f xs = *reinterpret_cast<const v4f *>(&s_fx[jb]);
                    const v4f ys = *reinterpret_cast<const v4f *>(&s_fy[jb]);
                    const v2f dxa = v2f{xs.x, xs.y} - fx2, dya = v2f{ys.x, ys.y} - fy2;
                    const v2f dxb = v2f{xs.z, xs.w} - fx2, dyb = v2f{ys.z, ys.w} - fy2;
                    const v2f ma = thr2 - __builtin_elementwise_fma(dya, dya, dxa * dxa);
                    const v2f mb = thr2 - __builtin_elementwise_fma(dyb, dyb, dxb * dxb);
                    m = __builtin_amdgcn_alignbit(m, __float_as_uint(mb.y), 31); // m = (m << 1) | sign
                    m = __builtin_amdgcn_alignbit(m, __float_as_uint(mb.x), 31);
                    m = __builtin_amdgcn_alignbit(m, __float_as_uint(ma.y), 31);
                    m = __builtin_amdgcn_alignbit(m, __float_as_uint(ma.x), 31);
                }
                half[h2] = m;
            }
            uint64_t cand = ~(((uint64_t)half[1] << 32) | half[0]) & s_pres[k];
            if ((self >> 6) == k) cand &= ~(1ull << (self & 63));
            uint64_t word = 0;
            for (; cand; cand &= cand - 1) {
                const int q = __builtin_ctzll(cand), j = c0 + k * 64 + q;
                const double *cj = wa.circ + ((size_t)r * p.EP + j) * 4;
                const double dx = cj[0] - cx, dy = cj[1] - cy, rr = cj[2] + rad;
                if (!(dx * dx + dy * dy <= rr * rr)) continue; // the pair's own circles, fp64 (what the scan was conservative for)
                const double *B = wa.cor + ((size_t)r * p.EP + j) * 8;
                if (wide_same(A, B)) last = max(last, j);      // g == g_prime: never listed (utils.py:59)
                else if (sg_quads_intersect(A, B)) word |= 1ull << q;
            }
            if (word) {
                stf(w.dy, SG_F_COLL + (c0 >> 6) + k, word);
                hit = true;
            }
        }
    }
    if (in) {
        if (last != e) atomicMax(&wa.last_same[w.g], last); // (wide_commit_kernel set it to e)
        const unsigned flags = (last != e ? 1u : 0u) | (hit ? (e == 0 ? 6u : 2u) : 0u);
        if (flags) atomicOr(&wa.dup[r], flags);
    }
}

// ... and the twins' bits move to the last owner of the geometry (rare: only scenarios that have twins this step do anything;
// the first part of wide_finish_kernel)
__device__ __forceinline__ void wide_owner_row(const Params &p, const WideArgs &wa, int r, int e)
{
    const int W = p.FROWS - SG_F_COLL;
    const WideEnt w(p, r, e);
    const int32_t *ls = wa.last_same + (size_t)r * p.EP;
    // in place, words in increasing order: a bit only moves FORWARD (to the last owner, whose own entry is itself), and an entity
    // never lists its own twins, so a moved bit is never moved again and never lands on a twin of e
    for (int q = 0; q < W; ++q) {
        uint64_t m = fld<uint64_t>(w.dy, SG_F_COLL + q);
        for (uint64_t t = m; t; t &= t - 1) {
            const int j = q * 64 + __builtin_ctzll(t), o = ls[j];
            if (o == j) continue;
            m &= ~(1ull << (j & 63));
            if ((o >> 6) == q) m |= 1ull << (o & 63);
            else stf(w.dy, SG_F_COLL + (o >> 6), fld<uint64_t>(w.dy, SG_F_COLL + (o >> 6)) | (1ull << (o & 63)));
        }
        stf(w.dy, SG_F_COLL + q, m);
    }
}

// ---- per scenario: clock, check_terminal (state.py:268-270, 397-408), ego metrics, CollisionMetric._step ------------------------
static __global__ __launch_bounds__(256) void wide_finish_kernel(Params p, double timestep, WideArgs wa)
{
    const int r = blockIdx.x, tid = threadIdx.x;
    if (wa.dup[r] & 1u) { // twins in this scenario: every row maps them to their last owner first (state/utils.py:32-40)
        for (int e = tid; e < p.E; e += (int)blockDim.x) wide_owner_row(p, wa, r, e);
        __threadfence();
        __syncthreads();
    }
    if (tid >= 64 || !wide_runs(p, wa, r)) return; // (one wavefront per scenario: the per-entity work is the kernels' before this one)
    const int W = p.FROWS - SG_F_COLL;
    // The ego's row against CollisionMetric.last_timestep, a word per lane: in almost every step no bit is new, and the row
    // only has to become the new `last` -- sixteen independent loads and stores instead of a chain of them on one thread
    // (which was a third of the step on 1,024-entity scenarios).  A step WITH new bits takes the serial walk below.
    bool lanes_did_rows = false;
    if (wa.mode == 0) {
        const WideEnt eg0(p, r, p.sstat[r].ego);
        const bool egp = fld<uint64_t>(eg0.dy, SG_F_PRESENT) != 0;
        uint64_t *last0 = wa.last_row + (size_t)r * W;
        bool fresh_any = false;
        uint64_t rows_l[4] = {0, 0, 0, 0};
        for (int q = tid, k = 0; q < W && k < 4; q += 64, ++k) {
            rows_l[k] = fld<uint64_t>(eg0.dy, SG_F_COLL + q);
            fresh_any = fresh_any || (rows_l[k] & ~last0[q]) != 0;
        }
        if (egp && W <= 256 && !sg_any(fresh_any)) {
            for (int q = tid, k = 0; q < W && k < 4; q += 64, ++k) last0[q] = rows_l[k];
            lanes_did_rows = true;
        }
    }
    if (tid != 0) return;
    const ScenStatic &ss = p.sstat[r];
    sg_scenario_state &sd = p.sdyn[r];
    // "some entity collides" / "entity 0 collides" (state.py:397-400): noted by wide_collide_kernel while it filled the rows (an
    // absent entity's row is empty; moving a twin's bit leaves a row non-empty)
    const int s_any = (wa.dup[r] & 2u) != 0, s_ego0 = (wa.dup[r] & 4u) != 0;
    const WideEnt eg(p, r, ss.ego);
    const bool ego_present = fld<uint64_t>(eg.dy, SG_F_PRESENT) != 0;
    uint64_t *last = wa.last_row + (size_t)r * W;
    if (wa.mode != 0) { // the reset's metric resets: metrics/trajectory.py:13-17,36-39; metrics/collision.py:64-68
        const double v0 = fld(eg.dy, SG_F_VEL + 0), v1 = fld(eg.dy, SG_F_VEL + 1), v2 = fld(eg.dy, SG_F_VEL + 2);
        sd.t = ss.t0;
        sd.prev_t = ss.t0 - 0.1; // state.py:135
        sd.ego_avg_speed = sd.ego_max_speed = ego_present ? sg_norm3(v0, v1, v2) : __builtin_nan("");
        sd.avg_t = 0.0;
        sd.ego_distance_travelled = __builtin_nan("");
        sd.done = 0; sd.n_steps = 0; sd.n_events = 0; sd.noise_pos = 0;
        sd.rec_rows = p.rec_cap > 0 ? 1 : 0;
        if (p.rec_cap > 0) p.rec_t[r] = ss.t0;
        for (int q = 0; q < W; ++q) last[q] = 0;
        for (int q = 0; q < 4; ++q) { sd.last_row[q] = 0; sd.last_row_hi[q] = 0; }
        return;
    }
    const double t_old = sd.t, t = t_old + timestep, dt = t - t_old;
    sd.prev_t = t_old;
    sd.t = t;
    const int steps = ++sd.n_steps;
    if (p.noise_mode == 1 && !wa.no_peds) sd.noise_pos += 2ll * wa.walkers[r]; // (counted by wide_move_kernel of this step)
    if (p.rec_cap > 0 && steps < p.rec_cap) { p.rec_t[(size_t)steps * p.R + r] = t; sd.rec_rows = steps + 1; }
    if (ego_present) { // scenario_gym.py:251-252
        const double speed = sg_norm3(fld(eg.dy, SG_F_VEL + 0), fld(eg.dy, SG_F_VEL + 1), fld(eg.dy, SG_F_VEL + 2));
        const double wgt = sd.avg_t / t; // EgoAvgSpeed._step, metrics/trajectory.py:19-24
        sd.ego_avg_speed += (1.0 - wgt) * (speed - sd.ego_avg_speed);
        sd.avg_t = t;
        sd.ego_max_speed = __builtin_fmax(speed, sd.ego_max_speed);
        sd.ego_distance_travelled = fld(eg.dy, SG_F_DIST);
    }
    int ndone = 0;
    if ((p.term_mask & SG_TERM_MAX_LENGTH) && (t + dt > ss.length)) ndone = 1;
    if ((p.term_mask & SG_TERM_COLLISION) && s_any) ndone = 1;
    if ((p.term_mask & SG_TERM_EGO_COLLISION) && s_ego0) ndone = 1;
    if (p.term_mask & SG_TERM_EGO_OFF_ROAD) {
        // TERMINAL_CONDITIONS["ego_off_road"], state.py:401-407: entities[0] (slot 0, not Scenario.ego) absent, or its reference
        // point not strictly inside the driveable surface
        const WideEnt e0(p, r, 0);
        bool off = true;
        if (fld<uint64_t>(e0.dy, SG_F_PRESENT) != 0 && p.road) {
            const RoadIndex RI = *p.road;
            off = !(rn_layers_at(RI, RI.net_of_scen[r], SG_LAYER_DRIVEABLE, fld(e0.dy, SG_F_POSE + 0), fld(e0.dy, SG_F_POSE + 1)) & SG_LAYER_DRIVEABLE);
        }
        if (off) ndone = 1;
    }
    sd.done = ndone;
    if (ego_present && !lanes_did_rows) { // CollisionMetric._step, metrics/collision.py:70-75
        int n_ev = sd.n_events;
        const double *A = wa.cor + ((size_t)r * p.EP + ss.ego) * 8;
        for (int q = 0; q < W; ++q) {
            const uint64_t rowq = fld<uint64_t>(eg.dy, SG_F_COLL + q);
            uint64_t fresh = rowq & ~last[q];
            while (fresh) {
                const int j = q * 64 + __builtin_ctzll(fresh);
                fresh &= fresh - 1;
                // how often j is listed for the ego: once per entity that hits the ego and shares j's geometry (j is its last owner)
                const double *Bj = wa.cor + ((size_t)r * p.EP + j) * 8;
                int mult = (wa.dup[r] & 1u) ? 0 : 1; // (no twins in the scenario this step: j itself is the one entity behind the bit)
                for (int k = 0; k <= j && (wa.dup[r] & 1u); ++k) {
                    const double *ck = wa.circ + ((size_t)r * p.EP + k) * 4;
                    if (!(ck[0] == ck[0]) || k == ss.ego) continue;
                    const double *Bk = wa.cor + ((size_t)r * p.EP + k) * 8;
                    if ((k == j || wide_same(Bj, Bk)) && !wide_same(A, Bk) && sg_quads_intersect(A, Bk)) ++mult;
                }
                const WideEnt o(p, r, j);
                const int64_t ometa = fld<int64_t>(o.st, ST_META);
                const int okind = (int)(ometa & 0xff);
                for (int m = 0; m < mult; ++m) {
                    if (n_ev < p.ev_cap) {
                        sg_event *dst = &p.events[(size_t)r * p.ev_cap + n_ev];
                        dst->t = t; dst->scenario = r; dst->other = j; dst->reserved = 0;
                        dst->type = ((ometa >> 8) & 0xff) == 0 ? -1 : 5;
                        double *hp = p.ev_hpose + ((size_t)r * p.ev_cap + n_ev) * 3;
                        hp[0] = hp[1] = hp[2] = __builtin_nan("");
                        if (okind == SG_KIND_AGENT_PID || okind == SG_KIND_AGENT_VEHICLE) { // a controlled hazard leaves its pose beside the event
                            hp[0] = fld(o.dy, SG_F_POSE + 0); hp[1] = fld(o.dy, SG_F_POSE + 1); hp[2] = fld(o.dy, SG_F_POSE + 3);
                        }
                        double *ep = p.ev_pose + ((size_t)r * p.ev_cap + n_ev) * 3;
                        ep[0] = fld(eg.dy, SG_F_POSE + 0); ep[1] = fld(eg.dy, SG_F_POSE + 1); ep[2] = fld(eg.dy, SG_F_POSE + 3);
                    }
                    ++n_ev;
                }
            }
            last[q] = rowq;
        }
        sd.n_events = n_ev;
    }
}
#endif // SG_UNIT_WIDE

// How many scenarios are still running, written to page-locked host memory: launch_wide looks at the answers of EARLIER
// check points without waiting for them and stops enqueuing steps once one says nobody is (the steps already enqueued are
// no-ops for a done scenario), so a rollout of wide scenarios stays asynchronous for the host.
static __global__ __launch_bounds__(256) void wide_running_kernel(Params p, int *out)
{
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    int n = 0;
    for (int r = threadIdx.x; r < p.R; r += 256) n += p.sdyn[r].done == 0;
    if (n) atomicAdd(&s_n, n);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(out, s_n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

} // namespace sg
