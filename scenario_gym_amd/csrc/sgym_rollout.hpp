// sgym_rollout.hpp -- rollout_body and its entry points; the time-sliced replay (slices, clock, ordered sums).
// Part of the gfx950 device code of the batched rollout engine; included by sgym_device.hpp (in order: every part builds on
// the ones before it), never on its own.
#pragma once

namespace sg {

// ------------------------------------------------------------------------------------------------
// The rollout kernel: ScenarioGym.reset_scenario / step / rollout (scenario_gym.py:217-267).
//   WV == 1: one 64-lane workgroup carries 64/G scenarios of up to G entities each (tiles of G lanes)
//   WV  > 1: one workgroup of WV wavefronts carries ONE scenario of up to 64*WV entities
// do_reset: State.reset first.  force: step done scenarios too (gym.step()); otherwise each scenario
// stops at is_done (gym.rollout()).  PED: pedestrian agents (social force) are compiled in.
//
// Register-resident per lane across the time loop: pose, distance, the knot segment (x_lo, x_hi,
// y_lo[6], slope[6]), the clock, controller state, ego metric accumulators.  Controller parameters
// and box extents live in LDS; there is no global load in a steady-state step.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ Table lane_table(const Params &p, int kind, const ScenStatic &ss, int slot,
                                            const LanePtr &st)
{
    Table T;
    if (kind == SG_KIND_REPLAY) {
        size_t go = (size_t)ss.grid_off;
        T.x = p.grid_t + go; T.xs = 1;
        T.y = p.grid_y + go * 6 * p.EP + slot; T.ys = 6 * p.EP; T.cs = p.EP;
        T.n = ss.grid_n;
    } else if (kind >= SG_KIND_AGENT_REPLAY) {
        const double *kn = p.knots + fld<int64_t>(st, ST_KNOT_OFF) * 7;
        T.x = kn; T.xs = 7; T.y = kn + 1; T.ys = 7; T.cs = 1;
        T.n = (int)(fld<int64_t>(st, ST_META) >> 32);
    } else {
        T.x = nullptr; T.y = nullptr; T.n = 0; T.xs = T.ys = T.cs = 0;
    }
    return T;
}

//
// TAB: the PID / vehicle agents were integrated by control_kernel; their lanes read (x, y, h) per step from
// its table `tab` instead of running the controller with 1 of 64 lanes active.  TAB launches never reset.
// Register budget of the one-wavefront-per-tile entry points (rollout_kernel_tab / _tab_planar): 168 VGPRs, three
// wavefronts per SIMD; the pre-pass takes a wavefront slot of its own (sgym_hip.hip, launch_rollout).
// HAST (TAB only): the batch has controlled lanes, i.e. there is a table to replay; without it the table code is
// compiled out (batches of replay entities only: the C2 shape).
// ROAD: the ego_off_road terminal condition is compiled in (its own entry point, rollout_kernel_road: the other
// variants keep their register budgets).
// RSSV: the RSSDistances callback (rss_entity) runs after the reset and after every step inside the kernel.
// ------------------------------------------------------------------------------------------------
// Time-sliced replay (launch_sliced in sgym_hip.hip): a batch whose lanes are all replay entities / replay agents is a
// pure function of the clock -- pose_j = interpolant(t_j), presence_j = rule(t_j) -- except for three ORDERED sums
// (State.distances, EgoAvgSpeed, the event list) and the step at which a terminal condition first holds.  A small batch
// (BASELINE config 2: 64 wavefronts on a 1024-SIMD chip) therefore cuts the time axis: the clock t_j = t_{j-1} + dt is
// accumulated once (clock_kernel, the same additions as the step loop), slice s of the steps runs in its own wavefronts
// from a warm-up step that rebuilds state a = s * len out of the clock alone, leaves |delta pose| / ego speed / events per
// step, and an ordered pass (replay_fixup_kernel) adds them up in step order; the state of the last executed step is
// materialised by one more launch (mode 1).  Results are bit-identical to the step-by-step kernel; what is NOT produced is
// the state of every intermediate step in memory.
// ------------------------------------------------------------------------------------------------
struct SliceArgs {
    int mode;            // 0: slices (per-step terms go to the slice arrays), 1: the last executed step with the full state stores
    int n_slices, len;   // slice s covers steps (s * len, min((s + 1) * len, n_total)]
    int n_total;         // steps of the call
    const double *tt;    // [n_clocks][n_total + 1] the clocks: tt[c][j] = State.t after j steps of a scenario that starts at t0_c
    const int *clock_of; // [R] the clock of scenario r (scenarios with the same start time share one)
    double *dnorm;       // [n_blocks][n_total + 1][64] |delta pose[:3]| of step j per lane (+0 when the entity has no pose)
    double2 *espeed;     // [n_total + 1][R] ego speed after step j (NaN: the ego has no pose) and 1 - t_prev / t of
                         // EgoAvgSpeed when the ego's previous update was the previous step (else NaN: the fix-up divides)
    int *first_done;     // [R][n_slices] the step of the slice at which the scenario became done (0x7f7f7f7f: none)
    sg_event *ev;        // [R][n_slices][ev_cap] CollisionMetric events of the slice
    int *nev;            // [R][n_slices]
    const int *n_final;  // [R] (mode 1) the scenario's last executed step
    int slice0;          // first slice of this launch (blockIdx.y counts from it): batches with controlled lanes launch their
                         // slices group by group, each group as soon as the controller pre-pass has reached its last step
};

// CROWD (PED only): every entity of the batch is a pedestrian agent (or padding), default head rotation, no road network:
// no knot segment, no vehicle / replay code, crowd_pairs for the neighbour sums (rollout_kernel_crowd, BASELINE config 5).
// SLICE (TAB, one wavefront per tile): one slice of a time-sliced replay, see SliceArgs.  With HAST the controlled lanes
// (PID / vehicle agents) replay a controller table that spans the WHOLE call -- row j - 1 = the lane after step j, written by
// control_kernel launches that run ahead of the slices -- so a slice that starts at step a finds its lanes' poses there
// like everything else it needs in the clock.
template <int G, int WV, bool PED, bool TAB, bool HAST, bool ROAD = false, bool RSSV = false, bool CROWD = false, bool SLICE = false,
          bool PLANAR = false, bool RIDERS = false, bool CTAB = false, bool MODELS = !CROWD>
__device__ __forceinline__ void rollout_body_l(
    TileLds<64 * WV, PED, CROWD, CROWD && !RIDERS> &lds /* the workgroup's LDS tile: the entry point owns it (rollout_kernel_tabq shares it between roles) */,
    const Params &p, double timestep, int n_steps, int do_reset, int force, const double *actions /*[n][R][2]*/,
    const double *tab /*controller table planes*/, const SliceArgs &sa = SliceArgs{},
    const unsigned bx_arg = ~0u /* the 64-slot block (WV == 1) / scenario of this workgroup when it is not bx: TabGroups */)
{
    const unsigned bx = bx_arg == ~0u ? blockIdx.x : bx_arg;
    static_assert(!SLICE || (TAB && WV == 1 && !PED && !ROAD && !RSSV), "slices: the table variant, one wavefront per tile");
    static_assert(!(PED && TAB), "pedestrian scenarios run their controllers in the rollout kernel");
    static_assert(!CROWD || (PED && G == 64 && !ROAD && !RSSV), "the crowd variant is a pedestrian variant with 64-lane tiles");
    // RIDERS (crowd variant; its own entry point, rollout_kernel_crowd_riders): the batch also has lanes that are NOT pedestrian
    // agents -- replay entities, replay agents, PID / vehicle agents (a car driving through the crowd, recorded pedestrians).
    // None of them ever looks at another entity (batch.py:34-53, agent.py:125-148, controller.py:105-258), so a pre-pass
    // (control_kernel_riders) has put their pose and presence after every step of the chunk into the controller table, and
    // here they only read their row: the crowd kernel stays free of knot segments and vehicle code.
    static_assert(!RIDERS || CROWD, "riders ride the crowd variant");
    // CTAB (variants with in-kernel controllers whose registers are full -- the RSS callback: rollout_kernel_rss_tab): the PID /
    // vehicle agents were integrated by control_kernel, their lanes read x, y, h of the step from the controller table with a
    // vector load, and the controller code (sin / cos, PID, tangent: ~220 instructions per wavefront-step for one active
    // lane in 64) is not compiled into this kernel at all.
    static_assert(!CTAB || (!TAB && !PED && !CROWD), "CTAB: table rows into an in-kernel-controller variant");
    constexpr int NS = 64 * WV;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t voff = lane * 8u;
    // one wavefront = one 64-slot block of the state arrays: wave-uniform block pointers
    const size_t blk = (size_t)bx * WV + wave;
    const double *st_blk = p.stat + blk * (ST_COUNT * 64);
    const LanePtr st(st_blk, voff);
    const LanePtr dy(p.dyn + blk * ((size_t)(SG_F_COLL + WV) * 64), voff);
    // scenario / slot of this lane
    const int gl = bx * NS + tid;
    const int r_raw = WV == 1 ? gl / G : bx;
    const int slot = WV == 1 ? (gl & (G - 1)) : tid;
    const int tile0 = WV == 1 ? (lane & ~(G - 1)) : 0; // first LDS slot of this lane's tile
    const int sl = tid;                                 // this lane's LDS slot
    const bool in_range = r_raw < p.R;
    const uint32_t r = in_range ? r_raw : p.R - 1;
    const ScenStatic &ss = p.sstat[r];
    sg_scenario_state &sd = p.sdyn[r];
    const int64_t meta = fld<int64_t>(st, ST_META);
    const int kind = (in_range && slot < p.E) ? (int)(meta & 0xff) : SG_KIND_NONE;
    const bool is_ped_type = ((meta >> 8) & 0xff) == 1;
    const bool is_ego = in_range && slot == ss.ego;
    const double bcx = fld(st, ST_BCX), bcy = fld(st, ST_BCY);
    const double min_t = fld(st, ST_MIN_T), max_t = fld(st, ST_MAX_T);
    const double length = ss.length;
    const bool is_static = (int)(meta >> 32) == 1;
    const bool is_agent = kind >= SG_KIND_AGENT_REPLAY;
    const bool is_replay = kind == SG_KIND_REPLAY;
    const bool replay_always = p.persist || is_static; // BatchReplayEntity keeps persistent / static entities (batch.py:45-52)
    // per-launch LDS tables: box extents, controller parameters; broad-phase reach of this lane =
    // own bounding-circle radius + the largest radius in the tile + slack
    float rad_thr, trig_eps, nbr_thr = 0.0f;
    float rmax_tile = 0.0f;   // REFINE: the largest bounding-circle radius of the tile ...
    bool hetero = false;      // ... and whether some real entity's is less than two thirds of it
    constexpr bool REFINE = CROWD ? RIDERS : PED;
    {
        const double bw = fld(st, ST_BW), bl = fld(st, ST_BL);
        float rad = (float)(0.5 * __builtin_sqrt(bl * bl + bw * bw)) * 1.000001f;
        float off = (float)__builtin_sqrt(bcx * bcx + bcy * bcy) * 1.000001f;
        // (PED) only entities of type Pedestrian can be somebody's social-force neighbour (pedestrian/sensor.py:56-63): the
        // neighbour reach needs THEIR largest centre offset, not the car's that drives through the crowd
        float rmax = rad, omax = off, omax_ped = (PED && is_ped_type) ? off : 0.0f;
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {
            rmax = __builtin_fmaxf(rmax, __shfl_xor(rmax, o, 64));
            omax = __builtin_fmaxf(omax, __shfl_xor(omax, o, 64));
            if (PED) omax_ped = __builtin_fmaxf(omax_ped, __shfl_xor(omax_ped, o, 64));
        }
        if (WV > 1) { // across the workgroup's wavefronts
            float *red = reinterpret_cast<float *>(lds.cor);
            if (lane == 0) { red[wave] = rmax; red[8 + wave] = omax; red[16 + wave] = omax_ped; }
            __syncthreads();
            for (int w = 0; w < WV; ++w) {
                rmax = __builtin_fmaxf(rmax, red[w]); omax = __builtin_fmaxf(omax, red[8 + w]);
                omax_ped = __builtin_fmaxf(omax_ped, red[16 + w]);
            }
            __syncthreads();
        }
        // hardware sin/cos (error d = SG_TRIG32_ERR per value): each centre moves by <= 2 d off, so the reach grows
        // by 2 d (off + omax); in the filter every gap is a sum of (length <= reach) x (trig product, error <= 4 d)
        rad_thr = rad + rmax + 2e-3f + 2.0f * SG_TRIG32_ERR * (off + omax);
        trig_eps = SG_TRIG32_ERR * (12.0f * rad_thr + 4.0f * (off + omax));
        if (REFINE) {
            rmax_tile = rmax;
            const bool small = kind != SG_KIND_NONE && rad * 1.5f < rmax;
            hetero = WV == 1 ? sg_any(small) : (__syncthreads_or(small) != 0);
        }
        lds.half[sl] = make_float2((float)(0.5 * bl), (float)(0.5 * bw));
        lds.boxwl[0][sl] = bw;
        lds.boxwl[1][sl] = bl;
        if (!TAB) {
#pragma unroll
            for (int q = 0; q < (PED ? 4 : 9); ++q) lds.ctrl[q][sl] = fld(st, ST_CTRL + (PED ? SG_C_PED_SPEED_DESIRED : 0) + q);
        }
        if (PED) // PedestrianSensor radius is measured between reference points; centres differ by the box offsets
            nbr_thr = kind == SG_KIND_AGENT_PEDESTRIAN
                          ? (float)fld(st, ST_CTRL + SG_C_PED_RADIUS) * 1.000001f + off + omax_ped + 2e-3f +
                                2.0f * SG_TRIG32_ERR * (off + omax_ped) : 0.0f;
        if (CROWD) { // thresholds of the radius rule (sg_in_radius), per pedestrian
            const double rr = fld(st, ST_CTRL + SG_C_PED_RADIUS), r2 = rr * rr;
            lds.r2hi[sl] = r2 * (1.0 + 1e-9);
            lds.r2lo[sl] = r2 * 0.9975;
        }
        if (CROWD && lds.GON)
            for (int k = tid; k < 128; k += NS) lds.gon[k] = p.gon[k];
        if (CROWD && !RIDERS) { // the buildings of this scenario's road network (TileLds::road_tab; the barriers below publish it)
            int n_staged = -1, net = -1;
            uint32_t net_flags = 0;
            if (p.road) {
                const RoadIndex &RI = *p.road;
                net = RI.net_of_scen[r];
                if (net >= 0) {
                    net_flags = RI.net_flags[net];
                    const int64_t e0 = RI.imp_off[net], ne = RI.imp_off[net + 1] - e0;
                    if ((net_flags & 2u) && ne <= lds.ROAD_EDGES) {
                        n_staged = (int)ne;
                        for (int k = tid; k < n_staged; k += NS) {
                            const double *e = RI.imp_edges + (e0 + k) * 4;
                            double *t = lds.road_tab + k * 5;
                            t[0] = e[0]; t[1] = e[1]; t[2] = e[2]; t[3] = e[3];
                            t[4] = RI.imp_aux[(e0 + k) * 4 + 2];
                        }
                        if (tid == 0) { lds.road_m = RI.imp_m[net]; lds.road_net = RI.nets[net]; }
                    }
                }
            }
            if (tid == 0) { lds.road_info[0] = n_staged; lds.road_info[1] = net; lds.road_info[2] = (int)net_flags; }
        }
    }
    // CROWD: may this wavefront use crowd_pairs at all?  Default head rotation in every lane, a radius and parameters inside
    // the guards of crowd_pair (wave-uniform, fixed for the launch); the per-step guards are voted in tile_collisions.
    bool crowd_static_ok = false, crowd_geom_ok = false; // (geometry: the lanes' own rows; static: + the parameters of model 0)
    CrowdConsts CC{};
    if (CROWD) {
        const double rr = fld(st, ST_CTRL + SG_C_PED_RADIUS), hr = fld(st, ST_CTRL + SG_C_PED_HEAD_ROT);
        crowd_geom_ok = sg_all(kind != SG_KIND_AGENT_PEDESTRIAN || (hr == 0.0 && rr > 0.0 && rr < 0x1p20)) && !p.ped_serial;
        crowd_static_ok = crowd_geom_ok && crowd_params_ok(p.sf);
        const RecipDiv rs(p.sf.ped_repulse_sigma);
        CC.k2_scale = p.sf.ped_repulse_V / p.sf.ped_repulse_sigma;
        CC.sig_b = rs.b;
        CC.sig_r = rs.r;
        CC.cos_sight = p.sf.cos_sight;
        CC.sight_weight = p.sf.sight_weight;
        CC.k3 = 2 * p.sf.ped_attract_C;
    }
    bool crowd_ok = false; // workgroup-uniform, per step: the guards of crowd_pair hold for every pedestrian of the tile
    // broad-phase cell size: >= every reach in the tile (+5 % so that fp32 cell coordinates stay consistent)
    float cell_inv;
    {
        float tmax = __builtin_fmaxf(rad_thr, nbr_thr);
#pragma unroll
        for (int o = 1; o < G; o <<= 1) tmax = __builtin_fmaxf(tmax, __shfl_xor(tmax, o, 64));
        if (WV > 1) {
            float *red = reinterpret_cast<float *>(lds.cor);
            if (lane == 0) red[wave] = tmax;
            __syncthreads();
            for (int w = 0; w < WV; ++w) tmax = __builtin_fmaxf(tmax, red[w]);
            __syncthreads();
        }
        cell_inv = 1.0f / (1.05f * tmax + 0.05f);
    }
    // pedestrian route (pedestrian/agent.py:43-47)
    const double *wp = nullptr;
    int nwp = 0;
    if (PED && kind == SG_KIND_AGENT_PEDESTRIAN) {
        int64_t rt = fld<int64_t>(st, ST_ROUTE);
        wp = p.routes + (rt & 0xffffffffffffll) * 2;
        nwp = (int)(rt >> 48);
    }

    // PLANAR (table variant, one wavefront per tile; its own entry point, rollout_kernel_tab_planar): every knot of the
    // batch has z = pitch = roll = +0.0 (sg_upload checks the bit patterns).  Those three channels are then +0.0 in every
    // pose, previous pose and velocity the batch ever holds -- absent lanes included, their rows are zeroed by the reset --
    // so the step neither interpolates, subtracts, tests nor stores them, and their 18 registers (pose, segment base and
    // slope) do not exist.
    static_assert(!PLANAR || (TAB && WV == 1 && !SLICE), "planar: the table variant");
    constexpr bool planar = PLANAR;
    // register-resident across the time loop
    double pose[6], dist, t, prev_t;
    double velx = 0.0, vely = 0.0; // current velocity (social force input), PED only
    CtrlState cs;                 // controller state (agent lanes); pedestrians: speed, goal_idx
    double m_avg, m_max, m_t;     // ego metric accumulators (ego lane)
    uint64_t last_row[WV];        // CollisionMetric.last_timestep (ego lane)
    long long noise_pos = 0;      // variates of the scenario's noise stream consumed so far (PED, noise mode 1)
    int n_ev, goal_idx = 0;
    bool present;
    int done, steps;
    uint64_t row[WV], mult_rows[WV], nbr[WV];
    bool dense = false; // broad-phase strategy of the pedestrian variant (workgroup-uniform), see tile_collisions
    // column of this lane in the controller table (TAB): PID / vehicle agents only
    // The table rows are fetched with SCALAR loads, one controlled lane at a time (at most SG_TAB_LANES per
    // wavefront and wavefront of a wide scenario, checked by the host), one step ahead, and moved into the lane's registers at the end of the step.  A vector load inside the loop would share vmcnt with the state stores and make every
    // step wait for the stores of the previous one.
    const int64_t ctl_q = (TAB && HAST) ? fld<int64_t>(st, ST_CTL) : -1;
    const bool tab_lane = TAB && HAST && ctl_q >= 0 && (kind == SG_KIND_AGENT_PID || kind == SG_KIND_AGENT_VEHICLE);
    const size_t tab_lane_stride = (size_t)(p.tab_steps + 1) * CT_W; // doubles per lane
    // RIDERS: this lane's column of the table (plane 0: x, y, h, speed; plane 2: z, pitch, roll, present)
    const bool rider = (RIDERS && kind != SG_KIND_NONE && kind != SG_KIND_AGENT_PEDESTRIAN) ||
                       (CTAB && (kind == SG_KIND_AGENT_PID || kind == SG_KIND_AGENT_VEHICLE));
    const double *rider_row = (RIDERS || CTAB) ? tab + (size_t)(rider ? fld<int64_t>(st, ST_CTL) : 0) * tab_lane_stride : nullptr;
    int last_k = -1;                                                 // last step of this launch the scenario executed
    constexpr int TL = SG_TAB_LANES(G, WV);
    int cl[TL];                       // wave-uniform: the controlled lanes of this wavefront
    const double *cb[TL];             // wave-uniform: their table columns
    double sx[TL], sy[TL], sh[TL];    // wave-uniform: row of the coming step
    if (TAB && HAST) {
        uint64_t cm = __ballot(tab_lane);
#pragma unroll
        for (int j = 0; j < TL; ++j) {
            cl[j] = -1;
            cb[j] = tab;
            sx[j] = sy[j] = sh[j] = 0.0;
            if (cm) {
                const int l = __builtin_ctzll(cm);
                cm &= cm - 1;
                cl[j] = l;
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ctl_q, l);
                cb[j] = tab + (size_t)lo * tab_lane_stride;
            }
        }
    }
    bool cb_hold[TL]; // wave-uniform (SLICE): the lane's slice starts at the reset state: its first row is consumed by round 1
#pragma unroll
    for (int j = 0; j < TL; ++j) cb_hold[j] = false;
    auto tab_issue = [&](bool first = false) { // s_load the next row of every controlled lane (an unused entry re-reads the first row)
#pragma unroll
        for (int j = 0; j < TL; ++j) {
            ConstTbl rowp = (ConstTbl)cb[j];
            sx[j] = rowp[CT_X];
            sy[j] = rowp[CT_Y];
            sh[j] = rowp[CT_H];
            cb[j] += (cl[j] >= 0 && !(SLICE && first && cb_hold[j])) ? CT_W : 0;
        }
    };

    // do_reset: 0 = continue from the stored state, 1 = State.reset for every scenario, 2 = for the scenarios flagged in
    // p.reset_mask only (one environment of a vector of environments starts a new episode).  The collision pass is a
    // wavefront / workgroup collective and runs outside the per-scenario branch.
    const bool rs = !TAB && (do_reset == 1 || (do_reset == 2 && p.reset_mask[r] != 0));
    double vel[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (rs) {
        // ---- State.reset(t0), state.py:106-143 ----
        const double *kn = p.knots + fld<int64_t>(st, ST_KNOT_OFF) * 7;
        const int nk = (int)(meta >> 32);
        t = ss.t0;
        present = false;
#pragma unroll
        for (int c = 0; c < 6; ++c) { pose[c] = 0.0; vel[c] = 0.0; }
        if (kind != SG_KIND_NONE) {
            bool inside = (t >= min_t) && (t <= max_t);
            if (is_static || inside) { own_position_extrap(kn, nk, t, pose); present = true; }
            else if (p.persist) { // extrapolate=(False, False): clamp
                const double *rowp = t < min_t ? kn : kn + (size_t)(nk - 1) * 7;
#pragma unroll
                for (int c = 0; c < 6; ++c) pose[c] = rowp[1 + c];
                present = true;
            }
            if (present && inside) { // Trajectory.velocity_at_t, trajectory.py:243-273
                const double eps = 1e-4;
                double a[6], b[6];
                own_position_extrap(kn, nk, t + eps / 2, a);
                own_position_extrap(kn, nk, t - eps / 2, b);
#pragma unroll
                for (int c = 0; c < 6; ++c) vel[c] = (a[c] - b[c]) / eps;
            }
        }
        prev_t = t - 0.1; // state.py:135
        dist = 0.0;
        done = 0;
        steps = 0;
        velx = vel[0];
        vely = vel[1];
        cs.speed = present ? sg_norm2(vel[0], vel[1]) : 0.0; // controller.py:100-103
        if (kind == SG_KIND_AGENT_PEDESTRIAN) cs.speed = 0.0; // pedestrian/controller.py:21-23
        cs.e_lon_prev = cs.e_lat_prev = cs.e_lon_int = 0.0;   // controller.py:198-203
        goal_idx = 0;                                         // pedestrian/agent.py:38
        m_avg = m_max = present ? sg_norm3(vel[0], vel[1], vel[2]) : __builtin_nan(""); // metrics/trajectory.py:13-17,36-39
        m_t = 0.0;
#pragma unroll
        for (int w = 0; w < WV; ++w) last_row[w] = 0; // metrics/collision.py:64-68
        n_ev = 0;
        noise_pos = 0;
    } else {
        t = sd.t;
        prev_t = sd.prev_t;
        present = fld<uint64_t>(dy, SG_F_PRESENT) != 0;
#pragma unroll
        for (int c = 0; c < 6; ++c) pose[c] = fld(dy, SG_F_POSE + c);
        velx = fld(dy, SG_F_VEL + 0);
        vely = fld(dy, SG_F_VEL + 1);
        dist = fld(dy, SG_F_DIST);
        if (!TAB) { // the table variant never touches the controller state of lanes it does not own
            cs.speed = fld(dy, SG_F_CTRL + 0); cs.e_lon_prev = fld(dy, SG_F_CTRL + 1);
            cs.e_lat_prev = fld(dy, SG_F_CTRL + 2); cs.e_lon_int = fld(dy, SG_F_CTRL + 3);
        }
        goal_idx = PED ? (int)cs.e_lon_prev : 0; // pedestrians keep goal_idx in the second controller row
        m_avg = sd.ego_avg_speed; m_max = sd.ego_max_speed; m_t = sd.avg_t;
#pragma unroll
        for (int w = 0; w < WV; ++w) last_row[w] = w < 4 ? sd.last_row[w & 3] : sd.last_row_hi[w & 3];
        n_ev = sd.n_events;
        noise_pos = PED ? sd.noise_pos : 0;
        done = sd.done;
        steps = sd.n_steps;
#pragma unroll
        for (int w = 0; w < WV; ++w) row[w] = fld<uint64_t>(dy, SG_F_COLL + w);
    }
    // (crowd variants -- one scenario, one ego per workgroup: the ego's accumulators live in LDS between their uses, see TileLds)
    auto ego_park = [&]() {
        if (CROWD && is_ego) {
            lds.ego_m[0] = m_avg; lds.ego_m[1] = m_max; lds.ego_m[2] = m_t;
#pragma unroll
            for (int w = 0; w < WV; ++w) lds.ego_last[w] = last_row[w];
            lds.ego_nev[0] = n_ev;
        }
    };
    auto ego_fetch = [&]() {
        if (CROWD && is_ego) {
            m_avg = lds.ego_m[0]; m_max = lds.ego_m[1]; m_t = lds.ego_m[2];
#pragma unroll
            for (int w = 0; w < WV; ++w) last_row[w] = lds.ego_last[w];
            n_ev = lds.ego_nev[0];
        }
    };
    ego_park();
    // SLICE: the lane has to be in state `a` (after a steps) before its real steps.  a <= 1: the reset state just loaded
    // (a == 1: + the warm-up step); a >= 2: state a - 1 rebuilt from the clock -- time, and the presence of an agent lane
    // (it has its pose from the reset on, or spawns at step 1: scenario_gym.py:240-244); everything else about that state
    // is either recomputed by the warm-up step (pose, presence of replay lanes, collision row) or not used by it.
    int slice_a = 0;
    const int slice_s = SLICE ? (int)blockIdx.y + sa.slice0 : 0;
    if (SLICE) {
        slice_a = sa.mode == 0 ? slice_s * sa.len : sa.n_final[r] - 1;
        n_ev = 0;
        if (slice_a >= 2) {
            const double *clk = sa.tt + (size_t)sa.clock_of[r] * (size_t)(sa.n_total + 1);
            t = clk[slice_a - 1];
            prev_t = clk[slice_a - 2];
            present = is_agent ? (present || min_t >= ss.t0) : true; // (a replay lane's presence is recomputed by the warm-up step)
            steps = slice_a - 1;
#pragma unroll
            for (int w = 0; w < WV; ++w) { last_row[w] = 0; row[w] = 0; }
        }
        n_steps = sa.mode == 0 ? 1 + min(sa.len, sa.n_total - slice_a) : 2;
        if (TAB && HAST) {
            // a controlled lane that spawns (scenario_gym.py:240-244: absent at the reset, min_t >= t0) took all six channels
            // of its trajectory at the clock of step 1 and keeps z / pitch / roll from then on (controller.py:126-131)
            if (slice_a >= 2 && tab_lane && fld<uint64_t>(dy, SG_F_PRESENT) == 0 && min_t >= ss.t0) {
                const double *clk = sa.tt + (size_t)sa.clock_of[r] * (size_t)(sa.n_total + 1);
                Table T1 = lane_table(p, kind, ss, slot, st);
                Segment S1;
                S1.cur = seg_locate(T1, clk[1]);
                seg_load(T1, S1);
                sg_loads_done();
                const double dq1 = clk[1] - S1.x_lo;
                pose[2] = S1.sl[2] * dq1 + S1.ylo[2];
                pose[4] = S1.sl[4] * dq1 + S1.ylo[4];
                pose[5] = S1.sl[5] * dq1 + S1.ylo[5];
            }
            // round k of this launch consumes row slice_a + k - 1 of the lane's table (round 0 is the warm-up step; a lane
            // that starts from the reset state sits it out and holds row 0 for round 1)
#pragma unroll
            for (int j = 0; j < TL; ++j) {
                if (cl[j] >= 0) {
                    const int a_l = __builtin_amdgcn_readlane(slice_a, cl[j]);
                    cb[j] += (size_t)max(a_l - 1, 0) * CT_W;
                    cb_hold[j] = a_l == 0;
                }
            }
        }
        sg_loads_done();
    }
    if (!TAB && (do_reset != 0 || PED)) {
        // collisions of the reset state; pedestrian scenes also need the neighbour candidates (and LDS positions) of the
        // current state when they continue
        uint64_t tmp_rows[WV];
        tile_collisions<G, WV, PED, CROWD, REFINE>(present, pose, velx, vely, (t + timestep) - t, bcx, bcy, rad_thr, trig_eps, nbr_thr, cell_inv,
                                                   is_ped_type, sl, tile0, lds, tmp_rows, mult_rows, nbr, dense, &crowd_ok, nullptr, hetero, rmax_tile);
        if (rs) {
#pragma unroll
            for (int w = 0; w < WV; ++w) row[w] = tmp_rows[w];
        }
    }
    if (rs) {
        if (in_range) {
#pragma unroll
            for (int c = 0; c < 6; ++c) { stf(dy, SG_F_POSE + c, pose[c]); stf(dy, SG_F_VEL + c, vel[c]); }
            stf(dy, SG_F_DIST, dist);
#pragma unroll
            for (int w = 0; w < WV; ++w) stf(dy, SG_F_COLL + w, row[w]);
            stf(dy, SG_F_PRESENT, (uint64_t)present);
            stf(dy, SG_F_FORCE + 0, 0.0);
            stf(dy, SG_F_FORCE + 1, 0.0);
            if (p.rec_cap > 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    p.rec_pose[(size_t)c * p.R * p.EP + (size_t)r * p.EP + slot] = present ? pose[c] : __builtin_nan("");
            }
            if (slot == 0) {
                sd.rec_rows = p.rec_cap > 0 ? 1 : 0;
                if (p.rec_cap > 0) p.rec_t[r] = t;
            }
            if (is_ego) sd.ego_distance_travelled = __builtin_nan("");
        }
    }

    // ---- RSSDistances inside the kernel (RSSV) ----
    const uint32_t rss_idx = (uint32_t)r * p.EP + slot;
    int32_t rss_st = 0;
    int rss_cd = -1;
    bool rss_touched = false; // this scenario was updated at least once in this launch
    double rss_lat = __builtin_nan(""), rss_long = __builtin_nan("");
    double rss_bw = 0.0, rss_bl = 0.0, rss_ew = 0.0, rss_el = 0.0;
    int rss_gn = 0;          // groups queued by this wavefront (uniform)
    unsigned rss_k = 0;      // ordinal of this lane's latest update within the launch
    const size_t rss_wave = (size_t)bx * WV + wave;
    SG_GLOBAL v2d *const rss_rec = RSSV ? (SG_GLOBAL v2d *)(p.rssq + rss_wave * (size_t)p.rssq_cap * RSSQ_REC) : nullptr;
    if (RSSV) {
        if (!rs && in_range && slot < p.E) rss_st = p.rss_state[rss_idx];
        rss_bw = fld(st, ST_BW);
        rss_bl = fld(st, ST_BL);
        if (WV == 1) { // the ego is slot 0 of the tile (sg_rss_update refuses anything else)
            rss_ew = shfl_d(rss_bw, tile0);
            rss_el = shfl_d(rss_bl, tile0);
        } else {
            if (tid == 0) { lds.cor[0][0] = rss_bw; lds.cor[1][0] = rss_bl; }
            __syncthreads();
            rss_ew = lds.cor[0][0];
            rss_el = lds.cor[1][0];
            __syncthreads();
        }
    }
    // one RSSDistances.__call__ for this lane's entity; upd: its scenario is being updated (it was reset / it stepped)
    auto rss_call = [&](bool upd, double tnow, double vx, double vy) {
        double ex, ey, eh, evx, evy, trig[4];
        bool ego_pres;
        // sin / cos of every lane's own heading: the entity's for its own box -- and, in the ego's lane, the ego's, which
        // every lane of the tile needs: one evaluation instead of two
        sg_sincos(pose[3], trig[2], trig[3]);
        if (WV == 1) {
            ex = shfl_d(pose[0], tile0); ey = shfl_d(pose[1], tile0); eh = shfl_d(pose[3], tile0);
            evx = shfl_d(vx, tile0); evy = shfl_d(vy, tile0);
            trig[0] = shfl_d(trig[2], tile0); trig[1] = shfl_d(trig[3], tile0);
            ego_pres = (__ballot(present) >> tile0) & 1;
        } else {
            if (tid == 0) {
                lds.cor[0][0] = pose[0]; lds.cor[1][0] = pose[1]; lds.cor[2][0] = pose[3];
                lds.cor[3][0] = vx; lds.cor[4][0] = vy; lds.cor[5][0] = present ? 1.0 : 0.0;
                lds.cor[6][0] = trig[2]; lds.cor[7][0] = trig[3];
            }
            __syncthreads();
            ex = lds.cor[0][0]; ey = lds.cor[1][0]; eh = lds.cor[2][0]; evx = lds.cor[3][0]; evy = lds.cor[4][0];
            ego_pres = lds.cor[5][0] != 0.0;
            trig[0] = lds.cor[6][0]; trig[1] = lds.cor[7][0];
            __syncthreads(); // the collision pass of the next step rewrites the scratch
        }
        int need = 0;
        bool ab = false;
        double Qd[8];
        if (upd) {
            rss_touched = true;
            rss_cd = -1;
            rss_lat = rss_long = __builtin_nan("");
            ++rss_k;
            if (!(tnow == 0.0 || !ego_pres || !present || slot == 0 || slot >= p.E)) // callback.py:76-78
                rss_entity<true>(ex, ey, eh, evx, evy, rss_ew, rss_el, pose[0], pose[1], pose[3], vx, vy, rss_bw, rss_bl, bcx, bcy,
                                 rss_st, rss_cd, rss_lat, rss_long, &need, Qd, &ab, trig);
        }
        // line tests: queued for rss_lines_kernel (see RssQueue)
        RSS_STAT(4, 1); RSS_STAT(5, __builtin_popcountll(__ballot((need & 3) != 0))); RSS_STAT(6, __builtin_popcountll(__ballot((need & 12) != 0)));
        const uint64_t wants = __ballot(need != 0);
        if (wants) {
            if (need) {
                const int at = rss_gn + __builtin_popcountll(wants & ((1ull << lane) - 1));
                if (at < p.rssq_cap) { // (always: the host sizes the queue for the steps of the launch)
                    SG_GLOBAL v2d *rec = rss_rec + (size_t)at * (RSSQ_REC / 2);
                    rec[0] = v2d{Qd[0], Qd[1]}; rec[1] = v2d{Qd[2], Qd[3]};
                    rec[2] = v2d{Qd[4], Qd[5]}; rec[3] = v2d{Qd[6], Qd[7]};
                    rec[4] = v2d{rss_lat, rss_long};
                    rec[5] = v2d{__longlong_as_double((long long)((uint64_t)(uint32_t)(lane | need << 8) | (uint64_t)rss_k << 32)), 0.0};
                }
                rss_cd = -4 - (int)rss_k; // "the code of update rss_k is with rss_lines_kernel"
            }
            rss_gn += __builtin_popcountll(wants);
        }
        if (upd && rss_cd == RSS_CD_ISECT) // unsafe_distance, callback.py:196-213: the entry exists from now on, its class is pending
            rss_st = (rss_st & 0xff00) | 3 | RSS_ST_PENDING | (ab ? RSS_ST_AB : 0);
    };
    // State.reset ends with update_callbacks(), state.py:138-140 (the table variant is never the reset launch)
    if (RSSV && !CTAB && do_reset != 0) rss_call(rs, t, vel[0], vel[1]);

    Segment S;
    if (!CROWD) { // (a crowd has no replay lanes: its only trajectory lookup is the rare spawn, done on the spot)
        Table T = lane_table(p, kind, ss, slot, st);
        S.cur = seg_locate(T, t);
        seg_load(T, S);
    }

    // The row of the coming step waits in SGPRs (sx, sy, sh); the step selects it into the controlled lane with
    // scalar-source v_cndmask and then issues the loads of the row after it.
    constexpr bool has_tab = TAB && HAST;
    if (has_tab && n_steps > 0) tab_issue(true);

    // Two nested loops over the same step counter.  The inner one is the steady state and only READS the knot
    // segment S; when some lane's clock is about to cross a knot the wavefront drops to the outer loop, which
    // advances that lane's segment and re-enters.  With the conditional update inside a single loop the compiler
    // keeps two copies of S (28 VGPRs) and moves one onto the other on every step.
    int k = 0;
    bool all_done = false;
    bool vel_clean_prev = false; // wave-uniform
    PhaseTimers ptm;
#ifdef SG_PHASE_TIMERS
    ptm.start();
    if (lane == 0 && bx < 1024) p.phase_cycles[16 + bx * 4 + wave] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); // HW_REG_HW_ID
#endif
    sg_loads_done(); // everything loaded so far is in its registers before the first store is issued
    sg_lgkm_done();
    while (k < n_steps && !all_done) {
    if (!CROWD && t + timestep > S.x_hi) { // rare: next knot segment
        // opaque copies keep the table address arithmetic inside this branch (otherwise ~15 invariant
        // 64-bit row addresses are hoisted out of the time loop and held in VGPRs / spilled)
        int kind_o = kind, slot_o = slot;
        LanePtr st_o = st;
        asm volatile("" : "+v"(kind_o), "+v"(slot_o), "+v"(st_o.a[0]));
        Table T = lane_table(p, kind_o, ss, slot_o, st_o);
        seg_advance(T, S, t + timestep);
        sg_loads_done();
    }
    for (; k < n_steps; ++k) {
        // per wavefront and before any workgroup barrier of the step: does a lane need its next segment?
        if (!CROWD && sg_any(t + timestep > S.x_hi)) break;
        // SLICE: round 0 is the warm-up step (state a - 1 -> a, nothing recorded); a lane that starts from the reset state
        // itself (a == 0) sits it out
        const bool warm = SLICE && k == 0;
        const bool run_lane = in_range && (force || !done) && !(SLICE && k == 0 && slice_a == 0);
        PH(5);
        // (a workgroup of several wavefronts carries ONE scenario: `run` is already uniform, nothing to vote)
        const bool any_run_ = WV == 1 ? sg_any(run_lane || (SLICE && k == 0 && in_range && !done)) : run_lane;
        PH(7);
        if (!any_run_) { all_done = true; break; }
        // A wavefront that carries ONE scenario (64-lane tiles) has the same `done` in every lane, so past the vote every lane
        // runs: said out loud, the `if (run)` blocks and selects below are not lane-divergent code any more (the compiler
        // cannot see that the 64 copies of `done` agree)
        const bool run = (G == 64 && WV == 1 && !SLICE) ? true : run_lane;
        // coefficient table: opaque per step so the scalar loads stay inside the loop (SGPRs for a few
        // dozen instructions instead of VGPRs for the whole kernel); constant address space => s_load
        const double *Kp = SG_TRIG;
        if (!TAB) asm volatile("" : "+s"(Kp));
        ConstTbl K = (ConstTbl)Kp;

        const double next_t = t + timestep; // scenario_gym.py:229
        const double state_dt = t - prev_t; // State.dt, state.py:198-201
        const double dt = next_t - t;       // = State.dt after this step
        // external actions are the only global loads of a steady-state step: issue them first
        double act_a = 0.0, act_s = 0.0;
        if (!TAB && kind == SG_KIND_AGENT_VEHICLE && actions) {
            const double *a = actions + ((size_t)k * p.R + r) * 2;
            act_a = a[0];
            act_s = a[1];
        }
        double np_[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (PLANAR) {
            double dq = next_t - S.x_lo;
            np_[0] = S.sl[0] * dq + S.ylo[0];
            np_[1] = S.sl[1] * dq + S.ylo[1];
            np_[3] = S.sl[3] * dq + S.ylo[3];
        } else if (!CROWD) {
            double dq = next_t - S.x_lo;
#pragma unroll
            for (int c = 0; c < 6; ++c) np_[c] = S.sl[c] * dq + S.ylo[c];
        }

        // ---- new poses: scenario_gym.py:233-245 ----
        bool npres = false;
        double fpx = 0.0, fpy = 0.0; // PedestrianAgent.force
        bool ped_go = false;
        double ped_fx = 0.0, ped_fy = 0.0, ped_vdes = 0.0;
        PedMoveModel pmm = ped_move_model(p); // what ped_move reads of the behaviour model: this lane's, where the batch mixes models
        double nstd_lon = p.noise_std_lon, nstd_lat = p.noise_std_lat;
        // (MODELS: compiled in for every general pedestrian variant, and for the crowd variant as an entry point of its own --
        // rollout_kernel_crowd_models: the one-model crowd kernel has no register to spare for a second copy of the force code)
        if (PED && MODELS && p.n_ped_models > 1) {
            // Per-agent behaviour models (sg_set_ped_models; pedestrian/agent.py:18-41): the force on a pedestrian is computed with
            // ITS model's parameters from its neighbours' states, so the tile's pedestrians step model by model -- one pass of
            // the (wave-collective) force code per model that has a stepping pedestrian in the tile, everybody else sitting
            // the pass out.  A tile whose pedestrians share one model pays one pass, as before.
            const bool stepping = is_agent && kind == SG_KIND_AGENT_PEDESTRIAN && present && run;
            const int my_model = in_range ? p.model_of[(size_t)r * p.EP + slot] : 0;
            const double *mm = p.ped_models + (size_t)my_model * PM_W;
            pmm = PedMoveModel{(int)mm[PM_BEHAVIOUR], mm[PM_SF + 8], mm[PM_SF + 9], mm[PM_SF + 7]}; // bias_lon, bias_lat, max_speed_factor
            nstd_lon = mm[PM_STD_LON];
            nstd_lat = mm[PM_STD_LAT];
            sg_loads_done();
            for (int m = 0; m < p.n_ped_models; ++m) {
                const bool mine = stepping && my_model == m;
                // (the crowd variants' force code is collective over the WAVEFRONT only: no workgroup vote)
                if (CROWD ? !sg_any(mine) : !block_any<WV>(mine)) continue; // (wavefront- / workgroup-uniform)
                Params q = p;
                const double *row = p.ped_models + (size_t)m * PM_W;
                q.ped_behaviour = (int)row[PM_BEHAVIOUR];
                q.sf = *reinterpret_cast<const sg_social_force *>(row + PM_SF);
                bool go_m = false;
                double fx_m = 0.0, fy_m = 0.0, vdes_m = 0.0;
                if (CROWD) { // this model's constants of crowd_pair; its parameters inside the guards, or the general pair code
                    CrowdConsts CM;
                    const RecipDiv rsm(q.sf.ped_repulse_sigma);
                    CM.k2_scale = q.sf.ped_repulse_V / q.sf.ped_repulse_sigma;
                    CM.sig_b = rsm.b;
                    CM.sig_r = rsm.r;
                    CM.cos_sight = q.sf.cos_sight;
                    CM.sight_weight = q.sf.sight_weight;
                    CM.k3 = 2 * q.sf.ped_attract_C;
                    ped_force<WV, CROWD>(q, lds, (int)r, sl, tile0, nbr, mine, pose, velx, vely, wp, nwp, goal_idx, go_m, fx_m, fy_m, vdes_m, K,
                                         crowd_geom_ok && crowd_params_ok(q.sf) && crowd_ok, CM, &ptm);
                } else
                    ped_force<WV, false>(q, lds, (int)r, sl, tile0, nbr, mine, pose, velx, vely, wp, nwp, goal_idx, go_m, fx_m, fy_m, vdes_m, K);
                if (mine) { ped_go = go_m; ped_fx = fx_m; ped_fy = fy_m; ped_vdes = vdes_m; }
            }
        } else if (PED) // the social force of every stepping pedestrian of the wavefront (wave-collective)
            ped_force<WV, CROWD>(p, lds, (int)r, sl, tile0, nbr, is_agent && kind == SG_KIND_AGENT_PEDESTRIAN && present && run, pose,
                                 velx, vely, wp, nwp, goal_idx, ped_go, ped_fx, ped_fy, ped_vdes, K, crowd_static_ok && crowd_ok, CC, &ptm);
        // random fluctuations of the speed and the heading (social_force.py:106-108): np.random.normal(loc, scale) is
        // loc + scale * z; z from the scenario's stream of variates -- two per walking pedestrian, in agent order, as the
        // reference draws them from numpy's global generator -- or from the counter-based generator
        PedNoise nz{0.0, 0.0, false};
        if (PED && p.noise_mode == 1) { // (launch-uniform branch)
            const uint64_t walk = __ballot(ped_go);
            int before, count;
            if (WV == 1) {
                uint64_t m = walk >> tile0;
                if (G < 64) m &= (1ull << (G & 63)) - 1;
                before = __builtin_popcountll(m & ((1ull << (slot & 63)) - 1));
                count = __builtin_popcountll(m);
            } else { // walkers in the wavefronts before this one: through the fourth vote row
                if (lane == 0) lds.vote[3][wave] = __builtin_popcountll(walk);
                __syncthreads();
                before = __builtin_popcountll(walk & ((1ull << lane) - 1));
                count = 0;
#pragma unroll
                for (int w = 0; w < WV; ++w) {
                    const int c = lds.vote[3][w];
                    before += w < wave ? c : 0;
                    count += c;
                }
                __syncthreads(); // (the row is rewritten next step; stream runs are parity runs, not timing runs)
            }
            const long long at = noise_pos + 2 * before;
            if (ped_go) {
                const bool inside = at + 1 < p.noise_len;
                const double *z = p.noise_normals + (size_t)r * (size_t)p.noise_len + (inside ? at : 0);
                nz = PedNoise{nstd_lon * (inside ? z[0] : 0.0), nstd_lat * (inside ? z[1] : 0.0), true};
                sg_loads_done();
            }
            if (run) noise_pos += 2 * count;
        } else if (PED && p.noise_mode == 2) {
            double z0, z1;
            sg_noise_pair(p.noise_seed, r, (uint32_t)slot, (uint32_t)steps, z0, z1, K);
            nz = PedNoise{nstd_lon * z0, nstd_lat * z1, true};
        }
        if (TAB) {
            // Straight-line lane masks (the kernel is bound by instruction issue, branches included):
            // BatchReplayEntity.step (batch.py:34-53) for replay lanes; an agent stays once present and spawns at its
            // trajectory start (scenario_gym.py:240-244); controlled lanes take the pre-pass row, z / p / r unchanged
            // (controller.py:126-131).
            const bool in_window = (next_t >= min_t) & (next_t <= max_t);
            const bool np_replay = replay_always | in_window;
            const bool np_agent = present | (min_t >= t);
            npres = (is_replay & np_replay) | (is_agent & np_agent);
            if (has_tab) {
                const bool take = tab_lane & present & run;
#pragma unroll
                for (int j = 0; j < TL; ++j) { // wave-uniform table row into its lane: v_cndmask with scalar sources
                    const bool tj = take & (lane == cl[j]);
                    np_[0] = tj ? sx[j] : np_[0];
                    np_[1] = tj ? sy[j] : np_[1];
                    np_[3] = tj ? sh[j] : np_[3];
                }
                if (!planar) {
                    np_[2] = take ? pose[2] : np_[2];
                    np_[4] = take ? pose[4] : np_[4];
                    np_[5] = take ? pose[5] : np_[5];
                }
                tab_issue(); // row k + 1 (the table has one spare row), consumed by the next step
            }
        } else if (CROWD) {
            if (RIDERS && rider) { // the pre-pass row of this step: pose and presence after it
                const double4 a = *reinterpret_cast<const double4 *>(rider_row + (size_t)k * CT_W);
                const double4 b = *reinterpret_cast<const double4 *>(rider_row + (size_t)k * CT_W + 2 * (size_t)p.n_ctl_pad * tab_lane_stride);
                sg_loads_done();
                np_[0] = a.x; np_[1] = a.y; np_[3] = a.z;
                np_[2] = b.x; np_[4] = b.y; np_[5] = b.z;
                npres = b.w != 0.0;
            } else if (kind == SG_KIND_AGENT_PEDESTRIAN) {
                if (present) {
                    npres = true;
                    if (run)
                        ped_move<!CROWD>(pmm, ped_go, ped_fx, ped_fy, ped_vdes, lds.ctrl[PED ? SG_C_PED_MAX_SPEED - SG_C_PED_SPEED_DESIRED : 0][sl],
                                 pose, state_dt, cs.speed, fpx, fpy, np_, K, nz);
                } else if (min_t >= t) { // scenario_gym.py:240-244: spawn at the trajectory position of next_t (clamped)
                    npres = true;
                    LanePtr st_o = st;
                    asm volatile("" : "+v"(st_o.a[0]));
                    Table T = lane_table(p, SG_KIND_AGENT_PEDESTRIAN, ss, slot, st_o);
                    Segment S2;
                    S2.cur = seg_locate(T, next_t);
                    seg_load(T, S2);
                    sg_loads_done();
                    const double dq = next_t - S2.x_lo;
#pragma unroll
                    for (int c = 0; c < 6; ++c) np_[c] = S2.sl[c] * dq + S2.ylo[c];
                }
            }
        } else if (kind == SG_KIND_REPLAY) { // BatchReplayEntity.step, batch.py:34-53
            npres = p.persist || is_static || (next_t >= min_t && next_t <= max_t);
        } else if (is_agent) {
            if (present && kind == SG_KIND_AGENT_EXTERNAL) {
                // the caller ran agent.step(state) (agent.py:52-57): its pose, or None = NaN (scenario_gym.py:233-239)
                const double *ep = p.ext_pose + ((size_t)r * p.EP + slot) * 6;
                const double e0 = ep[0];
                if (e0 == e0) {
                    npres = true;
#pragma unroll
                    for (int c = 0; c < 6; ++c) np_[c] = ep[c];
                } else if (p.persist) {
                    npres = true;
#pragma unroll
                    for (int c = 0; c < 6; ++c) np_[c] = pose[c];
                }
                sg_loads_done();
            } else if (present) {
                npres = true;
                if (kind != SG_KIND_AGENT_REPLAY && run) {
                    const double tx = np_[0], ty = np_[1];
#pragma unroll
                    for (int c = 0; c < 6; ++c) np_[c] = pose[c];
                    if (CTAB && (kind == SG_KIND_AGENT_PID || kind == SG_KIND_AGENT_VEHICLE)) {
                        // the pre-pass row of this step; z / pitch / roll stay (controller.py:126-131)
                        const double4 a = *reinterpret_cast<const double4 *>(rider_row + (size_t)k * CT_W);
                        sg_loads_done();
                        np_[0] = a.x; np_[1] = a.y; np_[3] = a.z;
                    } else if (!CTAB && (kind == SG_KIND_AGENT_PID || kind == SG_KIND_AGENT_VEHICLE)) {
                        const double bl = lds.boxwl[1][sl];
                        double sin_h, cos_h; // of the current heading
                        sg_sincos(pose[3], sin_h, cos_h, K);
                        // controller parameters: LDS table; pedestrian scenes keep only the pedestrian rows in LDS and
                        // read these (rare lanes there) from the static rows
                        LanePtr st_c = st;
                        auto cp = [&](int q) -> double {
                            return PED ? fld(st_c, ST_CTRL + q) : lds.ctrl[PED ? 0 : q][sl];
                        };
                        if (kind == SG_KIND_AGENT_PID)
                            pid_step(cs, cp, bl, state_dt, dt, tx, ty, sin_h, cos_h, np_, K);
                        else
                            vehicle_step(cs, cp, bl, dt, act_a, act_s, sin_h, cos_h, np_, K);
                    } else if (PED)
                        ped_move<!CROWD>(pmm, ped_go, ped_fx, ped_fy, ped_vdes,
                                 lds.ctrl[PED ? SG_C_PED_MAX_SPEED - SG_C_PED_SPEED_DESIRED : 0][sl], pose, state_dt,
                                 cs.speed, fpx, fpy, np_, K, nz);
                }
            } else if (min_t >= t) { // scenario_gym.py:240-244: spawn at trajectory start
                npres = true;
            }
        }

        // ---- State.update_poses / update_statistics, state.py:203-239 ----
        double d[6];
        if (PLANAR) {
            d[0] = np_[0] - pose[0]; d[1] = np_[1] - pose[1]; d[3] = np_[3] - pose[3];
            d[2] = d[4] = d[5] = 0.0;
        } else if (!CROWD || !(npres && !present)) { // (the crowd variant keeps the if / else form: its registers are full)
#pragma unroll
            for (int c = 0; c < 6; ++c) d[c] = np_[c] - pose[c];
        }
        if (npres && !present) { // newcomer: previous pose from the extrapolated trajectory, state.py:219-222
            // (the rare case overwrites d: as an if / else the two subtractions were merged behind six copies pose -> prev
            // that every step paid)
            double prev[6];
            LanePtr st_o = st;
            asm volatile("" : "+v"(st_o.a[0]));
            own_position_extrap(p.knots + fld<int64_t>(st_o, ST_KNOT_OFF) * 7,
                                (int)(fld<int64_t>(st_o, ST_META) >> 32), t, prev);
#pragma unroll
            for (int c = 0; c < 6; ++c) d[c] = np_[c] - prev[c];
            if (planar) d[2] = d[4] = d[5] = 0.0; // (0 - 0: the extrapolated channels are +0.0 as well)
        }
        double vel[6];
        // z, pitch and roll rarely move.  `flat`: in every lane that commits a pose this step they keep their value
        // (delta +0.0, entity already present).  Then +0 / dt (dt > 0) is +0 -- the three divisions and range checks
        // are skipped -- and the state blocks already hold these pose rows (and, after one flat step, the +0 velocity
        // rows): they are not stored again.  Memory stays the exact step-materialised state; a steady step issues 9
        // row stores instead of 15.
        bool flat;
        {
            RecipDiv rd(dt);
            if (planar) {
                flat = !SLICE && sg_all(dt > 0.0);
            } else {
                const uint32_t zbits = (uint32_t)(__double2hiint(d[2]) | __double2hiint(d[4]) | __double2hiint(d[5])) |
                                       (uint32_t)(__double2loint(d[2]) | __double2loint(d[4]) | __double2loint(d[5]));
                flat = !SLICE && sg_all((!run | !npres | (present & (zbits == 0))) & (dt > 0.0));
            }
            // RecipDiv::safe for three (six) numerators at once: every |d| below 2^961 through one maximum, and each either
            // +0 or at least 2^-959 (NaN fails the second, infinity the first)
            auto lo_ok = [](double a) { return (__builtin_fabs(a) >= 0x1p-959) | (__double_as_longlong(a) == 0); };
            double dmax = __builtin_fmax(__builtin_fmax(__builtin_fabs(d[0]), __builtin_fabs(d[1])), __builtin_fabs(d[3]));
            bool safe = rd.ok & lo_ok(d[0]) & lo_ok(d[1]) & lo_ok(d[3]);
            if (!flat) {
                dmax = __builtin_fmax(__builtin_fmax(dmax, __builtin_fabs(d[2])), __builtin_fmax(__builtin_fabs(d[4]), __builtin_fabs(d[5])));
                safe = safe & lo_ok(d[2]) & lo_ok(d[4]) & lo_ok(d[5]);
            }
            safe = safe & (dmax < 0x1p961);
            if (sg_all(safe)) {
                vel[0] = rd.div(d[0]); vel[1] = rd.div(d[1]); vel[3] = rd.div(d[3]);
                if (flat) {
                    vel[2] = vel[4] = vel[5] = 0.0;
                } else {
                    vel[2] = rd.div(d[2]); vel[4] = rd.div(d[4]); vel[5] = rd.div(d[5]);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 6; ++c) vel[c] = d[c] / dt;
            }
        }

        // commit (lanes of scenarios that are already done keep their state)
        const bool vel_zpr_clean = vel_clean_prev; // did the previous step leave +0 in every stored z/pitch/roll velocity row?
        vel_clean_prev = flat;
        const bool was_present = present;
        if (run) {
            present = npres;
            if (npres) {
                if (PLANAR) {
                    pose[0] = np_[0]; pose[1] = np_[1]; pose[3] = np_[3];
                } else {
#pragma unroll
                    for (int c = 0; c < 6; ++c) pose[c] = np_[c];
                }
                if (!SLICE) dist += PLANAR ? sg_norm2(d[0], d[1]) /* fma(+0, +0, s) == s for s >= +0 */ : sg_norm3(d[0], d[1], d[2]);
                if (PED) { velx = vel[0]; vely = vel[1]; }
            }
            prev_t = t;
            t = next_t;
            ++steps;
            last_k = k;
            if (SLICE && sa.mode == 0 && !warm) { // the terms of the ordered sums of step `steps` (replay_fixup_kernel)
                sa.dnorm[((size_t)blk * (size_t)(sa.n_total + 1) + (size_t)steps) * 64 + lane] = npres ? sg_norm3(d[0], d[1], d[2]) : 0.0;
                if (is_ego) // (prev_t is the clock before this step; steps == 1: the reset left EgoAvgSpeed.t = 0, the fix-up's case)
                    sa.espeed[(size_t)steps * p.R + r] =
                        make_double2(present ? sg_norm3(vel[0], vel[1], vel[2]) : __builtin_nan(""),
                                     (was_present && steps > 1) ? 1.0 - prev_t / t : __builtin_nan(""));
            }
            // ---- step-materialised state (everything except the collision row, see below) ----
            if (!SLICE || (sa.mode == 1 && !warm)) {
            stf(dy, SG_F_POSE + 0, pose[0]); stf(dy, SG_F_POSE + 1, pose[1]); stf(dy, SG_F_POSE + 3, pose[3]);
            if (!flat) { stf(dy, SG_F_POSE + 2, pose[2]); stf(dy, SG_F_POSE + 4, pose[4]); stf(dy, SG_F_POSE + 5, pose[5]); }
            if (present) {
                stf(dy, SG_F_VEL + 0, vel[0]); stf(dy, SG_F_VEL + 1, vel[1]); stf(dy, SG_F_VEL + 3, vel[3]);
                if (!(flat && vel_zpr_clean)) { // the rows hold +0 since the previous flat step
                    stf(dy, SG_F_VEL + 2, vel[2]); stf(dy, SG_F_VEL + 4, vel[4]); stf(dy, SG_F_VEL + 5, vel[5]);
                }
            }
            stf(dy, SG_F_DIST, dist);
            // (the presence row changes when an entity enters or leaves the scene: a wavefront in which nobody did holds
            // the row it would store -- memory stays the exact step-materialised state, one store instruction less)
            if (SLICE || sg_any(present != was_present)) stf(dy, SG_F_PRESENT, (uint64_t)present);
            }
            if (PED && kind == SG_KIND_AGENT_PEDESTRIAN) {
                stf(dy, SG_F_FORCE + 0, fpx);
                stf(dy, SG_F_FORCE + 1, fpy);
            }
            if (p.rec_cap > 0 && steps < p.rec_cap) {
                int nan_hi = 0x7ff80000;
                if (!CROWD) asm volatile("" : "+s"(nan_hi)); // (keeps the six selects inside this block: hoisted, they cost every
                                                             // step 18 moves; the crowd variant has no register to spare for it)
                const double absent = __hiloint2double(nan_hi, 0);
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    p.rec_pose[((size_t)steps * 6 + c) * p.R * p.EP + (size_t)r * p.EP + slot] = present ? pose[c] : absent;
                if (slot == 0) { p.rec_t[(size_t)steps * p.R + r] = t; sd.rec_rows = steps + 1; }
            }
            // ---- ego metrics, scenario_gym.py:251-252 ----
            if (!SLICE && is_ego && present && !tab_lane) { // a controlled ego's metrics come with its table (control_kernel)
                if (CROWD) { m_avg = lds.ego_m[0]; m_max = lds.ego_m[1]; m_t = lds.ego_m[2]; }
                double speed = sg_norm3(vel[0], vel[1], vel[2]);
                double w = m_t / t; // EgoAvgSpeed._step, metrics/trajectory.py:19-24
                m_avg += (1.0 - w) * (speed - m_avg);
                m_t = t;
                m_max = __builtin_fmax(speed, m_max); // EgoMaxSpeed, :41-44
                if (CROWD) { lds.ego_m[0] = m_avg; lds.ego_m[1] = m_max; lds.ego_m[2] = m_t; }
            }
        }
        PH(1);
        // ---- State.collisions ----
        uint64_t nrow[WV];
        tile_collisions<G, WV, PED, CROWD, REFINE>(present, pose, velx, vely, (t + timestep) - t, bcx, bcy, rad_thr, trig_eps, nbr_thr, cell_inv,
                                                   is_ped_type, sl, tile0, lds, nrow, mult_rows, nbr, dense, &crowd_ok, &ptm, hetero, rmax_tile);
        if (run) {
#pragma unroll
            for (int w = 0; w < WV; ++w) {
                // (likewise the collision row: stored when it differs from the stored one in some lane of the wavefront)
                const bool row_moved = SLICE || sg_any(nrow[w] != row[w]);
                row[w] = nrow[w];
                if ((!SLICE || (sa.mode == 1 && !warm)) && row_moved) stf(dy, SG_F_COLL + w, row[w]);
            }
        }

        // ---- check_terminal, state.py:268-270, 397-408 ----
        int ndone = 0;
        if ((p.term_mask & SG_TERM_MAX_LENGTH) && (t + dt > length)) ndone = 1;
        if (p.term_mask & (SG_TERM_COLLISION | SG_TERM_EGO_COLLISION)) {
            bool any_mine = false;
#pragma unroll
            for (int w = 0; w < WV; ++w) any_mine = any_mine || row[w] != 0;
            bool any_tile, ego0;
            if (WV == 1) {
                uint64_t m = __ballot(any_mine) >> tile0;
                if (G < 64) m &= (1ull << (G & 63)) - 1;
                any_tile = m != 0;
                uint64_t row0 = __shfl(row[0], tile0, 64);
                ego0 = ((__ballot(present) >> tile0) & 1) && row0 != 0;
            } else {
                any_tile = __syncthreads_or(any_mine);
                ego0 = __syncthreads_or(tid == 0 && present && any_mine);
            }
            if ((p.term_mask & SG_TERM_COLLISION) && any_tile) ndone = 1;
            if ((p.term_mask & SG_TERM_EGO_COLLISION) && ego0) ndone = 1;
        }
        if (ROAD && (p.term_mask & SG_TERM_EGO_OFF_ROAD)) {
            // TERMINAL_CONDITIONS["ego_off_road"], state.py:401-407: entities[0] (slot 0 of the tile, not Scenario.ego)
            // absent, or its reference point not strictly inside the driveable surface.  Slot 0 looks its cell up; only
            // cells crossed by a polygon boundary run the exact test.
            bool off = false;
            if (sl == tile0 && in_range) {
                off = true;
                if (present && p.road) {
                    const RoadIndex RI = *p.road;
                    off = !(rn_layers_at(RI, RI.net_of_scen[r], SG_LAYER_DRIVEABLE, pose[0], pose[1]) & SG_LAYER_DRIVEABLE);
                }
            }
            bool off_tile;
            if (WV == 1) off_tile = (__ballot(off) >> tile0) & 1;
            else off_tile = __syncthreads_or(off);
            if (off_tile) ndone = 1;
        }
        if (run) done = ndone;
        if (SLICE && sa.mode == 0 && !warm && run && ndone && sl == tile0)
            sa.first_done[(size_t)r * sa.n_slices + slice_s] = steps; // (once: the scenario does not run after this)

        // ---- CollisionMetric._step, metrics/collision.py:70-75 (ego lane only) ----
        uint64_t ev_fresh0 = 0;   // (ego lane) the hazards of this step's new events, first row word
        int ev_base = -1;         // (ego lane) index of the first of them in the event list; -1: none / not representable
        if (run && is_ego && present) {
            if (CROWD) {
                n_ev = lds.ego_nev[0];
#pragma unroll
                for (int w = 0; w < WV; ++w) last_row[w] = lds.ego_last[w];
            }
            if (!TAB) ev_base = n_ev;
#pragma unroll
            for (int w = 0; w < WV; ++w) {
                uint64_t fresh = row[w] & ~last_row[w];
                if (SLICE && (warm || sa.mode == 1)) fresh = 0; // the events of these steps belong to other launches
                if (!TAB && w == 0) ev_fresh0 = fresh;
                while (fresh) {
                    int j = w * 64 + __builtin_ctzll(fresh);
                    fresh &= fresh - 1;
                    int mult = 1;
                    bool aliased = false;
#pragma unroll
                    for (int v = 0; v < WV; ++v) aliased = aliased || mult_rows[v] != row[v];
                    if (aliased) { // aliased geometries are listed once per owner
                        if (!TAB) ev_base = -1;
                        mult = 0;
#pragma unroll
                        for (int v = 0; v < WV; ++v) {
                            uint64_t tmp = mult_rows[v];
                            while (tmp) { int q = __builtin_ctzll(tmp); tmp &= tmp - 1; mult += lds.last[tile0 + v * 64 + q] == j; }
                        }
                    }
                    // catalog type of the other entity (slot j of this scenario)
                    const int oj = (WV == 1 ? tile0 : 0) + j; // slot inside the workgroup's blocks
                    const double *oblk = p.stat + ((size_t)bx * WV + (oj >> 6)) * (ST_COUNT * 64);
                    int64_t ometa = reinterpret_cast<const int64_t *>(oblk)[ST_META * 64 + (oj & 63)];
                    for (int q = 0; q < mult; ++q) {
                        if (n_ev < p.ev_cap) {
                            sg_event *dst = SLICE ? &sa.ev[((size_t)r * sa.n_slices + slice_s) * p.ev_cap + n_ev]
                                                  : &p.events[(size_t)r * p.ev_cap + n_ev];
                            struct { double t; int32_t scenario, other, type, reserved; } head;
                            head.t = t; head.scenario = (int32_t)r; head.other = j;
                            // 5 = non_vehicle; Vehicle hazards (15 here, -1 once unpacked) wait for classify_events_kernel.
                            // The table variant packs the step of this launch above bit 4: the row of the controller
                            // table that holds the ego's pose at the event (event_ego_pose_kernel unpacks it)
                            // (slices: the table spans the call, the row is the step itself)
                            head.type = (((ometa >> 8) & 0xff) == 0 ? (TAB ? 15 : -1) : 5) | (TAB ? (SLICE ? steps : k + 1) << 4 : 0);
                            head.reserved = 0;
                            *reinterpret_cast<decltype(head) *>(dst) = head;
                            if (!TAB) { // (overwritten below when the hazard is a controlled agent; table launches: event_ego_pose_kernel)
                                double *hp = p.ev_hpose + ((size_t)r * p.ev_cap + n_ev) * 3;
                                hp[0] = hp[1] = hp[2] = __builtin_nan("");
                                if (RIDERS) { // a controlled rider as hazard: its pose after this step is a row of the riders' table
                                    const int okind = (int)(ometa & 0xff);
                                    if (okind == SG_KIND_AGENT_PID || okind == SG_KIND_AGENT_VEHICLE) {
                                        const int64_t octl = reinterpret_cast<const int64_t *>(oblk)[ST_CTL * 64 + (oj & 63)];
                                        const double *hrow = tab + (size_t)octl * tab_lane_stride + (size_t)k * CT_W;
                                        hp[0] = hrow[CT_X]; hp[1] = hrow[CT_Y]; hp[2] = hrow[CT_H];
                                    }
                                }
                            }
                            if (!TAB) { // in-kernel controllers: the ego pose of the event goes along.  (Not in the table
                                        // variant, which has no register to spare: its events are classified right
                                        // after the launch, with the ego pose taken from the table row `reserved`.)
                                double *ep = p.ev_pose + ((size_t)r * p.ev_cap + n_ev) * 3;
                                ep[0] = pose[0]; ep[1] = pose[1]; ep[2] = pose[3];
                            }
                        }
                        ++n_ev;
                    }
                }
                last_row[w] = row[w];
                if (CROWD) lds.ego_last[w] = row[w];
            }
            if (CROWD) lds.ego_nev[0] = n_ev;
        }
        if (!TAB && WV == 1 && p.ev_cap > 0) {
            // A hazard that is itself a controlled agent (PID / vehicle controller) has no trajectory its pose at the event
            // could be re-derived from: it leaves the pose it has right now beside the event (classify_events_kernel).  The
            // ego lane's new-event mask and list position go to the lanes of its tile; rare, one ballot per step otherwise.
            if (sg_any(ev_base >= 0 && ev_fresh0 != 0)) {
                const int ego_lane = tile0 + ss.ego;
                const uint64_t fr = __shfl(ev_fresh0, ego_lane, 64);
                const int base = __shfl(ev_base, ego_lane, 64);
                if (in_range && base >= 0 && ((fr >> slot) & 1) && (kind == SG_KIND_AGENT_PID || kind == SG_KIND_AGENT_VEHICLE)) {
                    const int at = base + __builtin_popcountll(fr & ((1ull << slot) - 1));
                    if (at < p.ev_cap) {
                        double *hp = p.ev_hpose + ((size_t)r * p.ev_cap + at) * 3;
                        hp[0] = pose[0]; hp[1] = pose[1]; hp[2] = pose[3];
                    }
                }
            }
        }
        if (RSSV) rss_call(run, t, vel[0], vel[1]); // State.step ends with update_callbacks(), state.py:165-171
        if (has_tab) sg_lgkm_done();
        PH(5);
    }
    }
#ifdef SG_PHASE_TIMERS
    ptm.flush(p.phase_cycles);
#endif

    if (SLICE) { // the per-scenario results of a sliced replay are written by replay_fixup_kernel
        if (in_range && is_ego) {
            if (sa.mode == 0) sa.nev[(size_t)r * sa.n_slices + slice_s] = n_ev;
            else {
#pragma unroll
                for (int w = 0; w < WV; ++w) (w < 4 ? sd.last_row[w & 3] : sd.last_row_hi[w & 3]) = last_row[w];
            }
        }
        if (TAB && HAST && sa.mode == 1 && in_range && tab_lane) { // controller state after the last executed step
            const double *lr = tab + (size_t)ctl_q * tab_lane_stride + (size_t)(sa.n_final[r] - 1) * CT_W;
            const double *lr1 = lr + (size_t)p.n_ctl_pad * tab_lane_stride; // plane 1
            stf(dy, SG_F_CTRL + 0, lr[CT_SPEED]); stf(dy, SG_F_CTRL + 1, lr1[CT_ELON]);
            stf(dy, SG_F_CTRL + 2, lr1[CT_ELAT]); stf(dy, SG_F_CTRL + 3, lr1[CT_EINT]);
        }
        return;
    }
    ego_fetch();
    if (RSSV && lane == 0) p.rssq_n[rss_wave] = min(rss_gn, p.rssq_cap);
    // ---- write back what lives in registers during the loop ----
    if (in_range) {
        if (PED && kind == SG_KIND_AGENT_PEDESTRIAN) cs.e_lon_prev = (double)goal_idx;
        if (TAB) {
            if (tab_lane && last_k >= 0) { // controller state after the last executed step
                const double *lr = tab + (size_t)ctl_q * tab_lane_stride + (size_t)last_k * CT_W;
                const double *lr1 = lr + (size_t)p.n_ctl_pad * tab_lane_stride; // plane 1
                stf(dy, SG_F_CTRL + 0, lr[CT_SPEED]); stf(dy, SG_F_CTRL + 1, lr1[CT_ELON]);
                stf(dy, SG_F_CTRL + 2, lr1[CT_ELAT]); stf(dy, SG_F_CTRL + 3, lr1[CT_EINT]);
                if (is_ego) { // ego metrics after the last executed step
                    const double *lr2 = lr1 + (size_t)p.n_ctl_pad * tab_lane_stride; // plane 2
                    m_avg = lr2[CT_MAVG]; m_max = lr2[CT_MMAX]; m_t = lr2[CT_MT];
                }
            }
        } else if ((RIDERS || CTAB) && rider) {
            if (last_k >= 0 && (kind == SG_KIND_AGENT_PID || kind == SG_KIND_AGENT_VEHICLE)) { // controller state after the last executed step
                const double *lr = rider_row + (size_t)last_k * CT_W;
                const double *lr1 = lr + (size_t)p.n_ctl_pad * tab_lane_stride; // plane 1
                stf(dy, SG_F_CTRL + 0, lr[CT_SPEED]); stf(dy, SG_F_CTRL + 1, lr1[CT_ELON]);
                stf(dy, SG_F_CTRL + 2, lr1[CT_ELAT]); stf(dy, SG_F_CTRL + 3, lr1[CT_EINT]);
            }
        } else {
            stf(dy, SG_F_CTRL + 0, cs.speed); stf(dy, SG_F_CTRL + 1, cs.e_lon_prev);
            stf(dy, SG_F_CTRL + 2, cs.e_lat_prev); stf(dy, SG_F_CTRL + 3, cs.e_lon_int);
        }
        if (slot == 0) { sd.t = t; sd.prev_t = prev_t; sd.done = done; sd.n_steps = steps; if (PED) sd.noise_pos = noise_pos; }
        if (RSSV) {
            if (slot < p.E) { // (markers in rss_st / rss_cd: rss_lines_kernel finishes them)
                p.rss_state[rss_idx] = rss_st;
                if (rss_touched) { // the records of the latest update
                    p.rss_code[rss_idx] = rss_cd;
                    p.rss_safe[(size_t)rss_idx * 2] = rss_lat;
                    p.rss_safe[(size_t)rss_idx * 2 + 1] = rss_long;
                }
            }
            if (slot == 0 && rss_touched) p.rss_seen[r] = steps;
        }
        if (is_ego) {
            sd.ego_avg_speed = m_avg; sd.ego_max_speed = m_max; sd.avg_t = m_t;
            if (steps > 0 && present) sd.ego_distance_travelled = dist; // EgoDistanceTravelled, :60-62
#pragma unroll
            for (int w = 0; w < WV; ++w) (w < 4 ? sd.last_row[w & 3] : sd.last_row_hi[w & 3]) = last_row[w];
            sd.n_events = n_ev;
        }
    }
}

// the ordinary form: the LDS tile belongs to this call
template <int G, int WV, bool PED, bool TAB, bool HAST, bool ROAD = false, bool RSSV = false, bool CROWD = false, bool SLICE = false,
          bool PLANAR = false, bool RIDERS = false, bool CTAB = false, bool MODELS = !CROWD>
__device__ __forceinline__ void rollout_body(
    const Params &p, double timestep, int n_steps, int do_reset, int force, const double *actions, const double *tab,
    const SliceArgs &sa = SliceArgs{}, const unsigned bx_arg = ~0u)
{
    __shared__ TileLds<64 * WV, PED, CROWD, CROWD && !RIDERS> lds;
    rollout_body_l<G, WV, PED, TAB, HAST, ROAD, RSSV, CROWD, SLICE, PLANAR, RIDERS, CTAB, MODELS>(lds, p, timestep, n_steps, do_reset, force, actions, tab,
                                                                                       sa, bx_arg);
}

// The blocks a launch of a table variant works on (launch_rollout): the 64-slot blocks of the batch are cut into groups of
// `gsz` consecutive blocks -- the host uses one group per rollout PIPELINE, two or three of them, each launched chunk after
// chunk on its own stream -- and a launch runs the groups of `active` only; every group reads the controller-table buffer
// (and runs the number of steps) of the chunk of the time axis IT has reached: buffer index = 2 bits per group in `bufof`.
// One group, active = 1: an ordinary launch over all blocks.
struct TabGroups {
    unsigned long long active, bufof[2];
    int gsz, n[4];
    const double *buf[4];
    // the launch's grid holds the active blocks only (a wavefront that starts just to find its group idle costs ~0.1 us of
    // the dispatcher's time, 0.3 ms for a thousand): grid block i is block start0 + i for i < len0, else start1 + (i - len0)
    unsigned start0, len0, start1, len1;
    __device__ __forceinline__ unsigned map(unsigned i) const { return i < len0 ? start0 + i : start1 + (i - len0); }
    __device__ __forceinline__ bool pick(unsigned blk, int &n_steps, const double *&tab) const
    {
        const unsigned g = blk / (unsigned)gsz;
        if (!((active >> g) & 1)) return false;
        const unsigned b = (unsigned)(bufof[g >> 5] >> (2 * (g & 31))) & 3u;
        n_steps = b == 0 ? n[0] : (b == 1 ? n[1] : (b == 2 ? n[2] : n[3]));
        tab = b == 0 ? buf[0] : (b == 1 ? buf[1] : (b == 2 ? buf[2] : buf[3]));
        return true;
    }
};

template <int G, int WV, bool PED, bool TAB>
__global__ __launch_bounds__(64 * WV, PED ? SG_WAVES_PER_SIMD_PED : (TAB ? SG_WAVES_PER_SIMD_TAB : SG_WAVES_PER_SIMD)) void rollout_kernel(
    Params p, double timestep, int n_steps, int do_reset, int force, const double *actions, const double *tab)
{
    // one wavefront per tile: this entry point serves the batches WITHOUT controlled lanes (rollout_kernel_tab the others)
    rollout_body<G, WV, PED, TAB, (TAB && WV > 1)>(p, timestep, n_steps, do_reset, force, actions, tab);
}

// All-pedestrian batches without road networks (BASELINE config 5): see rollout_body, CROWD
template <int WV>
__global__ __launch_bounds__(64 * WV, SG_WAVES_PER_SIMD_PED) void rollout_kernel_crowd(
    Params p, double timestep, int n_steps, int do_reset, int force, const double *actions, const double *tab)
{
    rollout_body<64, WV, true, false, false, false, false, true>(p, timestep, n_steps, do_reset, force, actions, tab);
}

// ... whose pedestrians follow up to four social-force models (sg_set_ped_models): a pass of the force code per model
template <int WV>
__global__ __launch_bounds__(64 * WV, SG_WAVES_PER_SIMD_PED) void rollout_kernel_crowd_models(
    Params p, double timestep, int n_steps, int do_reset, int force, const double *actions, const double *tab)
{
    rollout_body<64, WV, true, false, false, false, false, true, false, false, false, false, true>(p, timestep, n_steps, do_reset, force, actions, tab);
}

// ... with riders: lanes of other kinds whose poses come from the pre-pass table (see rollout_body, RIDERS)
template <int WV>
__global__ __launch_bounds__(64 * WV, SG_WAVES_PER_SIMD_PED) void rollout_kernel_crowd_riders(
    Params p, double timestep, int n_steps, int do_reset, int force, const double *actions, const double *tab)
{
    rollout_body<64, WV, true, false, false, false, false, true, false, false, true>(p, timestep, n_steps, do_reset, force, actions, tab);
}

// terminal_conditions with "ego_off_road": controllers in the kernel, road index lookups for slot 0
template <int G, int WV>
__global__ __launch_bounds__(64 * WV, SG_WAVES_PER_SIMD) void rollout_kernel_road(
    Params p, double timestep, int n_steps, int do_reset, int force, const double *actions, const double *tab)
{
    rollout_body<G, WV, false, false, false, true>(p, timestep, n_steps, do_reset, force, actions, tab);
}

// state_callbacks=[RSSDistances()]: controllers and the RSS callback in the kernel, any number of steps per launch
template <int G, int WV>
__global__ __launch_bounds__(64 * WV, SG_WAVES_PER_SIMD) void rollout_kernel_rss(
    Params p, double timestep, int n_steps, int do_reset, int force, const double *actions, const double *tab)
{
    rollout_body<G, WV, false, false, false, false, true>(p, timestep, n_steps, do_reset, force, actions, tab);
}

// ... with the PID / vehicle agents on the controller pre-pass's table (CTAB): one wavefront per tile
template <int G>
__global__ __launch_bounds__(64, SG_WAVES_PER_SIMD) void rollout_kernel_rss_tab(
    Params p, double timestep, int force, TabGroups tg)
{
    int n_steps;
    const double *tab;
    const unsigned blk = tg.map(blockIdx.x);
    if (!tg.pick(blk, n_steps, tab)) return;
    rollout_body<G, 1, false, false, false, false, true, false, false, false, false, true>(p, timestep, n_steps, 0, force, nullptr, tab,
                                                                                             SliceArgs{}, blk);
}

// ... with the ego_off_road terminal condition / with pedestrian agents (RSSDistances treats every entity alike)
template <int G, int WV>
__global__ __launch_bounds__(64 * WV, SG_WAVES_PER_SIMD) void rollout_kernel_rss_road(
    Params p, double timestep, int n_steps, int do_reset, int force, const double *actions, const double *tab)
{
    rollout_body<G, WV, false, false, false, true, true>(p, timestep, n_steps, do_reset, force, actions, tab);
}
template <int G, int WV>
__global__ __launch_bounds__(64 * WV, SG_WAVES_PER_SIMD_PED) void rollout_kernel_rss_ped(
    Params p, double timestep, int n_steps, int do_reset, int force, const double *actions, const double *tab)
{
    rollout_body<G, WV, true, false, false, false, true>(p, timestep, n_steps, do_reset, force, actions, tab);
}

// The table variant with one wavefront per tile (C2 / C3 shapes) under a 168-VGPR cap: three wavefronts per SIMD.  A
// wavefront of this kernel is latency-bound (1024 steps take 1.7 ms with one wavefront per SIMD, 2.3 ms with three), so the
// third one is nearly free: 73.8 -> 91.7 G entity-steps/s on the C3 shape with z / pitch / roll knots, for 64 B of scratch.
// (Rounds 1-2 held it at 192 so that two of its wavefronts and one of control_kernel (<= 128) filled a SIMD's 512 VGPRs;
// the pre-pass now takes a wavefront slot of its own, one launch per chunk: launch_rollout.)
#ifndef SG_TAB_WAVES // (experiment builds: -DSG_TAB_WAVES=2 -DSG_TAB_VGPR=96)
#define SG_TAB_WAVES 3
#define SG_TAB_VGPR 84
#endif
template <int G>
__global__ __launch_bounds__(64, SG_TAB_WAVES) __attribute__((amdgpu_num_vgpr(SG_TAB_VGPR))) void rollout_kernel_tab(
    Params p, double timestep, int force, TabGroups tg)
{
    int n_steps;
    const double *tab;
    const unsigned blk = tg.map(blockIdx.x);
    if (!tg.pick(blk, n_steps, tab)) return;
    rollout_body<G, 1, false, true, true>(p, timestep, n_steps, 0, force, nullptr, tab, SliceArgs{}, blk);
}
// ... for batches whose knots all have z = pitch = roll = +0.0 (PLANAR)
// Three wavefronts per SIMD (168 VGPRs): the kernel issues ~0.73 of the peak with two, ~0.85 with three.  The pre-pass does
// not fit beside three of them (launch_rollout gives it slots of its own: block groups).
#ifndef SG_PLANAR_WAVES // (experiment builds: -DSG_PLANAR_WAVES=2 -DSG_PLANAR_VGPR=96)
#define SG_PLANAR_WAVES 3
#define SG_PLANAR_VGPR 84
#endif
template <int G>
__global__ __launch_bounds__(64, SG_PLANAR_WAVES) __attribute__((amdgpu_num_vgpr(SG_PLANAR_VGPR))) void rollout_kernel_tab_planar(
    Params p, double timestep, int force, TabGroups tg)
{
    int n_steps;
    const double *tab;
    const unsigned blk = tg.map(blockIdx.x);
    if (!tg.pick(blk, n_steps, tab)) return;
    rollout_body<G, 1, false, true, true, false, false, false, false, true>(p, timestep, n_steps, 0, force, nullptr, tab, SliceArgs{}, blk);
}

// One slice of a time-sliced replay (grid.y = slices; SliceArgs), or its last step with the full state stores
template <int G>
__global__ __launch_bounds__(64, SG_WAVES_PER_SIMD_TAB) void rollout_kernel_slice(Params p, double timestep, SliceArgs sa)
{
    rollout_body<G, 1, false, true, false, false, false, false, true>(p, timestep, 0, 0, 0, nullptr, nullptr, sa);
}
// ... of a batch with controlled lanes: `tab` = the controller table of the whole call (p.tab_steps = sa.n_total)
template <int G>
__global__ __launch_bounds__(64, SG_WAVES_PER_SIMD_TAB) void rollout_kernel_slice_tab(Params p, double timestep, SliceArgs sa, const double *tab)
{
    rollout_body<G, 1, false, true, true, false, false, false, true>(p, timestep, 0, 0, 0, nullptr, tab, sa);
}

// The clocks of a sliced replay: tt[c][j] = State.t after j steps = t0_c + dt + dt + ... (scenario_gym.py:229), the
// additions of the step loop itself; scenarios with the same start time share a clock (launch_sliced).  One lane per clock.
#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64) void clock_kernel(const double *t0 /*[n_clocks]*/, int n_clocks, double timestep, int n_total, double *tt)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n_clocks) return;
    double *row = tt + (size_t)c * (size_t)(n_total + 1);
    double t = t0[c];
    row[0] = t;
    int j = 1;
    for (; j + 15 <= n_total; j += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { t = t + timestep; v[u] = t; }
#pragma unroll
        for (int u = 0; u < 16; ++u) row[j + u] = v[u];
    }
    for (; j <= n_total; ++j) { t = t + timestep; row[j] = t; }
}
#endif // SG_UNIT_MAIN

// n_final[r] = the step at which scenario r became done (the first over its slices), else all n_total steps
#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64) void slice_final_kernel(Params p, SliceArgs sa, int *n_final, int *done_out)
{
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= p.R) return;
    int nf = 0x7f7f7f7f; // "never": what launch_sliced fills first_done with
    for (int s = 0; s < sa.n_slices; ++s) nf = min(nf, sa.first_done[(size_t)r * sa.n_slices + s]);
    done_out[r] = nf != 0x7f7f7f7f;
    n_final[r] = min(nf, sa.n_total);
}
#endif // SG_UNIT_MAIN

// The ordered pass of a sliced replay, after the last step has been materialised.  replay_fixup_kernel: per entity
// State.distances = the |delta pose| terms added up in step order (state.py:237-239), 32 rows of the block in flight.
// replay_scenario_fixup_kernel: one lane per scenario: the EgoAvgSpeed / EgoMaxSpeed recurrences (metrics/trajectory.py:19-24,
// 41-44; an absent ego skips its update) over the ego's speeds and the clock (two contiguous streams, 16 steps in flight),
// the event lists of the slices concatenated in step order (metrics/collision.py:70-75), and the scenario record.
template <int G>
__global__ __launch_bounds__(64) void replay_fixup_kernel(Params p, SliceArgs sa, const int *n_final)
{
    const int lane = threadIdx.x;
    {
        const size_t blk = blockIdx.x;
        const int gl = (int)blk * 64 + lane;
        const int r_raw = gl / G, slot = gl & (G - 1);
        const bool in_range = r_raw < p.R;
        const int r = in_range ? r_raw : p.R - 1;
        const LanePtr dy(p.dyn + blk * ((size_t)(SG_F_COLL + 1) * 64), lane * 8u);
        const int nf = in_range ? n_final[r] : 0;
        int nf_max = nf;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) nf_max = max(nf_max, __shfl_xor(nf_max, o, 64));
        const double *dn = sa.dnorm + (blk * (size_t)(sa.n_total + 1)) * 64 + lane;
        double dist = 0.0; // State.reset: distances 0 (state.py:136)
        // two buffers of 16 rows: the loads of one are in flight while the other is added up (the additions wait for their
        // own buffer only: loads return in order)
        constexpr int NB = 16;
        double va[NB], vb[NB];
        auto fetch = [&](double (&v)[NB], int j0) {
#pragma unroll
            for (int u = 0; u < NB; ++u) v[u] = dn[(size_t)min(j0 + u, sa.n_total) * 64];
        };
        auto add_up = [&](const double (&v)[NB], int j0) {
#pragma unroll
            for (int u = 0; u < NB; ++u) dist += (j0 + u <= nf) ? v[u] : 0.0; // (x + 0.0 == x for x >= +0: straight-line code)
        };
        int j = 1;
        fetch(va, j);
        for (; j <= nf_max; j += 2 * NB) {
            fetch(vb, j + NB);
            add_up(va, j);
            fetch(va, j + 2 * NB);
            add_up(vb, j + NB);
        }
        if (in_range && slot < p.E) {
            stf(dy, SG_F_DIST, dist);
            if (slot == p.sstat[r].ego && nf > 0 && fld<uint64_t>(dy, SG_F_PRESENT) != 0)
                p.sdyn[r].ego_distance_travelled = dist; // EgoDistanceTravelled, metrics/trajectory.py:60-62
        }
    }
}

// (a controlled ego is no different here: the slices leave its speeds like a replay ego's, the pre-pass skips the metrics)
#ifdef SG_UNIT_MAIN // (emitted by the one object that launches it: csrc/Makefile, sgym_launch.hpp)
static __global__ __launch_bounds__(64) void replay_scenario_fixup_kernel(Params p, SliceArgs sa, const int *n_final, const int *done_in)
{
    const int lane = threadIdx.x;
    const int r = (int)blockIdx.x * 64 + lane;
    if (r >= p.R) return;
    sg_scenario_state &sd = p.sdyn[r];
    const int nf = n_final[r];
    double m_avg = sd.ego_avg_speed, m_max = sd.ego_max_speed, m_t = sd.avg_t; // the reset values
    const double2 *es = sa.espeed + r; // [step][R]: the 64 scenarios of the wavefront read one row together
    const double *tr = sa.tt + (size_t)sa.clock_of[r] * (size_t)(sa.n_total + 1);
    // EgoAvgSpeed._step: w = t_prev / t; avg += (1 - w) * (speed - avg).  The slices leave 1 - w whenever t_prev is the
    // previous step's clock (the ego had its pose then): the ordered part is three dependent operations per step.
    auto update = [&](double2 e, double t, bool divide) {
        const bool valid = e.x == e.x;
        double c = e.y;
        if (divide) c = (c != c) ? 1.0 - m_t / t : c; // first update, or the ego was absent in between
        const double a = m_avg + c * (e.x - m_avg);
        m_avg = valid ? a : m_avg;
        m_t = valid ? t : m_t;
        m_max = valid ? __builtin_fmax(e.x, m_max) : m_max;
    };
    // batches of 16 steps, the loads of the next batch in flight while this one is worked through
    constexpr int NB = 16;
    double2 spa[NB], spb[NB];
    double tqa[NB], tqb[NB];
    auto fetch = [&](double2 (&sp)[NB], double (&tq)[NB], int q0) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int qq = min(q0 + u, sa.n_total);
            sp[u] = es[(size_t)qq * p.R];
            tq[u] = tr[qq];
        }
    };
    auto work = [&](const double2 (&sp)[NB], const double (&tq)[NB]) {
        bool need = false; // does a step of the batch have to divide?  (one wave-uniform branch per batch)
#pragma unroll
        for (int u = 0; u < NB; ++u) need |= (sp[u].x == sp[u].x) & (sp[u].y != sp[u].y);
        if (sg_any(need)) {
#pragma unroll
            for (int u = 0; u < NB; ++u) update(sp[u], tq[u], true);
        } else {
#pragma unroll
            for (int u = 0; u < NB; ++u) update(sp[u], tq[u], false);
        }
    };
    int q = 1;
    if (nf >= NB) fetch(spa, tqa, q);
    for (; q + 2 * NB - 1 <= nf; q += 2 * NB) {
        fetch(spb, tqb, q + NB);
        work(spa, tqa);
        fetch(spa, tqa, q + 2 * NB);
        work(spb, tqb);
    }
    if (q + NB - 1 <= nf) { work(spa, tqa); q += NB; }
    for (; q <= nf; ++q) update(es[(size_t)q * p.R], tr[q], true);
    sd.ego_avg_speed = m_avg; sd.ego_max_speed = m_max; sd.avg_t = m_t;
    // events: the slices that lie before the last executed step, in order
    int n_ev = 0;
    for (int s = 0; s < sa.n_slices && s * sa.len < nf; ++s) {
        const int cnt = sa.nev[(size_t)r * sa.n_slices + s];
        const sg_event *src = sa.ev + ((size_t)r * sa.n_slices + s) * p.ev_cap;
        for (int i = 0; i < min(cnt, p.ev_cap); ++i)
            if (n_ev + i < p.ev_cap) p.events[(size_t)r * p.ev_cap + n_ev + i] = src[i];
        n_ev += cnt;
    }
    sd.n_events = n_ev;
    sd.t = tr[nf];
    sd.prev_t = nf > 0 ? tr[nf - 1] : sd.prev_t;
    sd.done = done_in[r];
    sd.n_steps = nf;
}
#endif // SG_UNIT_MAIN

} // namespace sg
