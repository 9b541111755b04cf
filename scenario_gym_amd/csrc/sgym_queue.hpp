// sgym_queue.hpp -- the table path as ONE persistent launch: rollout_kernel_tabq<G, PLANAR>.
// Part of the gfx950 device code of the batched rollout engine; included by sgym_launch.hpp after sgym_device.hpp.
//
// ScenarioGym.rollout is deterministic (scenario_gym.py:256-267); so is this schedule.  Rounds 3-4 ran the table path as chunk
// launches on two or three streams ("pipelines") beside a controller pre-pass on a fourth, and how many of those streams the
// runtime really ran side by side was found by a timing probe at sg_create -- 97 G with three, 67 G with one, and the probe
// could say either.  Here the whole call is one launch whose wavefronts take roles:
//
//   * the first `n_ctl_waves` wavefronts to arrive (a ticket, not blockIdx: nothing is assumed about dispatch order) are the
//     controller pre-pass: wavefront w integrates controlled lanes 64 w .. 64 w + 63 chunk after chunk (control_body, the
//     straight-line form) into the table ring and publishes `ctl_prog[w] = chunks written`;
//   * every other wavefront pulls work items (chunk c, block b) -- chunk-major, so the items of a block come in step order --
//     from one device-side counter, waits until block b has finished chunk c - 1 (`blk_prog[b]`) and the pre-pass wavefronts of
//     its controlled lanes have published chunk c, runs rollout_body on it exactly as a chunk launch did, and hands the
//     block's state rows on (`blk_prog[b] = c + 1`).
//
// 4096 blocks x 12..16 chunks over ~3000 wavefront slots: no launch boundary at which slots idle, no dependence on
// GPU_MAX_HW_QUEUES, one launch in the trace.  Hand-offs between wavefronts follow the release / acquire forms measured for
// this chip (agent scope on both sides; the scalar cache, through which the table rows are read, invalidated by hand);
// every wait is bounded: a wavefront that waits longer than `timeout_ticks` records a code in `state[Q_ERR]` and leaves, and
// so does everybody who sees the code -- the host reports SG_ERR_HIP instead of hanging.
#pragma once

namespace sg {

enum { Q_TICKET = 0, Q_HEAD = 1, Q_ERR = 2, Q_ITEMS_DONE = 3, Q_CTL_CLAIMED = 4, Q_STATE_WORDS = 8 };
constexpr int Q_SEATS = 1 << 14; // one word per SIMD of the device, indexed by (XCC, SE, SH, CU, SIMD) as the hardware reports them
enum { Q_ERR_CTL_WAIT = 1, Q_ERR_BLOCK_WAIT = 2, Q_ERR_RING_WAIT = 3 };
constexpr int Q_MAX_CHUNKS = 255;

struct TabQueue {
    unsigned *state;      // [Q_STATE_WORDS] role tickets, item counter, give-up code, items finished -- zeroed before every launch
    unsigned *ctl_prog;   // [n_ctl_waves] chunks pre-pass wavefront w has published
    unsigned *blk_prog;   // [nblk] chunks block b has finished
    unsigned *chunk_cnt;  // [n_chunks] blocks that have finished chunk c (the pre-pass waits on it before it reuses a ring buffer)
    unsigned *seats;      // [Q_SEATS] wavefronts of this launch that have arrived on SIMD s (role election)
    long long defer_ticks; // how long a wavefront that is not the first on its SIMD leaves the pre-pass roles to others
    double *tab;          // the table ring: n_buf buffers of buf_doubles doubles, chunk c lives in buffer c % n_buf
    size_t buf_doubles;
    const double *actions; // [n][R][2] external actions of vehicle agents, or nullptr
    unsigned *trace;       // [nblk] SG_QUEUE_DEBUG: how far the latest work item of block b got (nullptr: no trace)
    unsigned long long *times; // [n_items][4] SG_QUEUE_TIMES: wall clock (100 MHz) at pull / ready / body end / handed on; [n_items ..]: pre-pass
                               // wavefront w after chunk c at [n_items * 4 + w * n_chunks + c] (nullptr: not recorded)
    long long timeout_ticks; // of the 100 MHz wall clock
    int n_chunks, n_buf, nblk, n_ctl_waves;
    int lag_prio;         // items of blocks that are behind run at a raised priority: 0 no, 1 s_setprio 1, 2 s_setprio 2 (c3: 88.5 ->
                          // 92.5 / 93.0 G, means of six interleaved runs each, profiles/r05_ab_lagprio.txt)
    int handoff;          // how a rollout item hands its block on: 0 = release fence (buffer_wbl2: the whole L2 of the XCD is written
                          // back), 1 = the block's state rows and scenario records re-stored write-through (sc1), no fence
    int k0[Q_MAX_CHUNKS + 1]; // chunk c covers steps k0[c] .. k0[c + 1] - 1 of the call
};

// (every lane reads the same word; the first lane's copy, so that the compiler sees a wave-uniform value: a branch on a loaded
// value is a divergent branch to it, and a divergent exit out of a loop that holds wave-level operations -- readfirstlane,
// ballots -- is rearranged into something else: the first build of this file re-ran item 0 forever on 63 lanes)
__device__ __forceinline__ unsigned q_peek(const unsigned *w)
{
    return (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// wave-uniform: spin (relaxed agent-scope loads, s_sleep between them) until *w >= want; false: somebody gave up, or this
// wavefront did after timeout_ticks
__device__ __forceinline__ bool q_wait_ge(const unsigned *w, unsigned want, const TabQueue &tq, unsigned code)
{
    if (q_peek(w) >= want) return true;
    const long long t0 = wall_clock64();
    for (int spin = 0;; ++spin) {
        // (a few quick looks, then ~3 us naps: thousands of wavefronts poll the same few words while the pre-pass writes its
        // first chunk, and every poll is an L2 request on the pre-pass's path)
        if (spin < 4) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(112);
        if (q_peek(w) >= want) return true;
        if (q_peek(tq.state + Q_ERR) != 0) return false;
        if (wall_clock64() - t0 > tq.timeout_ticks) {
            if (threadIdx.x == 0) atomicCAS(tq.state + Q_ERR, 0u, code);
            return false;
        }
    }
}

// what another wavefront wrote before it raised the flag this wavefront has just seen: drop this CU's L1 lines, and the
// scalar cache (the table rows travel through s_load)
__device__ __forceinline__ void q_acquire()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

// everything this wavefront stored so far becomes visible to the whole device before the flag store that follows
__device__ __forceinline__ void q_release()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the compiler may drop the fence's own wait: restated where it cannot)
}

// The state a block carries from one work item to the next -- its FROWS state rows and the sg_scenario_state records of its
// scenarios -- read back (past this CU's L1) and stored again WRITE-THROUGH: the bytes are in memory when the flag goes up, and the
// lines this wavefront dirtied in its XCD's L2 during the item are clean again.  ~25 eight-byte loads and stores per lane and
// item, against a release fence that writes back every dirty line of the XCD's L2 (1.7 MB of state rows that every step
// rewrites: with 57,000 items per rollout the L2s did little else -- 67 G instead of 97 G).
template <int G>
__device__ __forceinline__ void q_handoff_writethrough(const Params &p, unsigned b, int lane)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    typedef SG_GLOBAL unsigned long long gu64;
    constexpr int FR = SG_F_COLL + 1; // rows of a one-wavefront block (p.FROWS, known here)
    gu64 *rows = (gu64 *)(p.dyn + (size_t)b * ((size_t)FR * 64)) + lane;
    unsigned long long v[FR];
#pragma unroll
    for (int f = 0; f < FR; ++f) v[f] = __hip_atomic_load(rows + f * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (all in flight)
#pragma unroll
    for (int f = 0; f < FR; ++f) __hip_atomic_store(rows + f * 64, v[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    static_assert(sizeof(sg_scenario_state) % 8 == 0, "scenario records are copied in 8-byte words");
    constexpr int SW = (int)(sizeof(sg_scenario_state) / 8);
    const int gl = (int)b * 64 + lane, r = gl / G, slot = gl & (G - 1);
    if (r < p.R) {
        gu64 *sd = (gu64 *)(p.sdyn + r);
        for (int w = slot; w < SW; w += G) {
            const unsigned long long v = __hip_atomic_load(sd + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sd + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every store has left before the flag does
}

// RSS: the RSSDistances callback runs inside the step loop (rollout_kernel_rss_tab's body: controlled lanes read the pre-pass
// table with vector loads, the line tests are queued per wavefront); the queue of a work item is worked off by the same
// wavefront right after its steps (rss_lines_block), and the entities' RSS state words travel with the block's state.
template <int G, bool PLANAR, bool RSS = false>
__device__ __forceinline__ void tabq_body(const Params &p, double timestep, int force, const TabQueue &tq)
{
    using Tile = TileLds<64, false, false>;
    static_assert(sizeof(Tile) >= sizeof(CtlLds), "the pre-pass role lays its LDS over the tile");
    __shared__ Tile lds;
    __shared__ typename std::conditional<RSS, RssQueue, char>::type rssq_lds;
    const int lane = threadIdx.x;
    // ---- role election.  The pre-pass chain is fp64 arithmetic back to back: two of its wavefronts on one SIMD run at half speed
    // each, and every block of their lanes waits for them (measured with first-come roles: the slowest pre-pass wavefront took
    // 36 ms where the others took 15).  So a pre-pass role goes to the FIRST wavefront of this launch to arrive on its SIMD;
    // a wavefront that is not the first leaves the roles to others for defer_ticks, then takes one itself if any is still open.
    // Nobody starts rolling out before every role is claimed (by a wavefront that is running): no assumption about dispatch
    // order or about how much of the grid is resident.
    unsigned ticket = ~0u;
    {
        const unsigned hw = (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
        const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)); // HW_REG_XCC_ID
        const unsigned seat = ((xcc & 15u) << 10) | (((hw >> 13) & 7u) << 7) | (((hw >> 12) & 1u) << 6) | (((hw >> 8) & 15u) << 2) | ((hw >> 4) & 3u);
        unsigned rank = 0;
        if (lane == 0) {
            atomicAdd(tq.state + Q_TICKET, 1u);
            rank = atomicAdd(tq.seats + seat, 1u);
        }
        rank = (unsigned)__builtin_amdgcn_readfirstlane((int)rank);
        bool claim = rank == 0 && q_peek(tq.state + Q_CTL_CLAIMED) < (unsigned)tq.n_ctl_waves;
        if (rank != 0) {
            const long long t0 = wall_clock64();
            while (q_peek(tq.state + Q_CTL_CLAIMED) < (unsigned)tq.n_ctl_waves) {
                if (wall_clock64() - t0 > tq.defer_ticks) { claim = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        if (claim) {
            unsigned idx = 0;
            if (lane == 0) idx = atomicAdd(tq.state + Q_CTL_CLAIMED, 1u);
            idx = (unsigned)__builtin_amdgcn_readfirstlane((int)idx);
            if (idx < (unsigned)tq.n_ctl_waves) ticket = idx;
        }
    }

    if (ticket < (unsigned)tq.n_ctl_waves) {
        // ---- the controller pre-pass: a chain of dependent steps every block waits for -> first in line at the issue port ----
        __builtin_amdgcn_s_setprio(3);
        CtlLds &cl = *reinterpret_cast<CtlLds *>(&lds);
        for (int c = 0; c < tq.n_chunks; ++c) {
            if (c >= tq.n_buf && !q_wait_ge(tq.chunk_cnt + (c - tq.n_buf), (unsigned)tq.nblk, tq, Q_ERR_RING_WAIT)) break;
            // (RSS: the ego's metrics are the rollout's own, from its velocities -- the callback variant computes them anyway)
            control_body_l<true>(cl, ticket, p, timestep, tq.k0[c + 1] - tq.k0[c], c == 0, tq.k0[c], tq.actions,
                                 tq.tab + (size_t)(c % tq.n_buf) * tq.buf_doubles, 0, RSS ? 0 : 1);
            q_release();
            if (tq.times && lane == 0) tq.times[(size_t)tq.n_chunks * tq.nblk * 4 + (size_t)ticket * tq.n_chunks + c] = wall_clock64();
            if (lane == 0) __hip_atomic_store(tq.ctl_prog + ticket, (unsigned)(c + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the chain is written: this wavefront rolls out like the others from here on (its slot would idle for the second half
        // of the call otherwise).  (Measured and dropped, round 5: rollout wavefronts that leave their SIMD to the pre-pass
        // wavefront on it for the first ~3000 steps -- the pre-pass got no faster, c3 88 -> 82 G.)
        __builtin_amdgcn_s_setprio(0);
        if (q_peek(tq.state + Q_ERR) != 0) return;
    }

    // ---- rollout role: work items (chunk, block) ----
    // Loop control is wave-uniform by construction (see q_peek); lane 0's stores that publish an item sit at the TOP of the next
    // round, beside the pull -- one lane-0 block per round, no lane-divergent code in front of the back edge.
    const unsigned n_items = (unsigned)tq.n_chunks * (unsigned)tq.nblk;
    unsigned done_b = ~0u, done_c = 0; // the item this wavefront has finished and not yet published
    for (;;) {
        unsigned item = 0;
        if (lane == 0) {
            if (done_b != ~0u) {
                __hip_atomic_store(tq.blk_prog + done_b, done_c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tq.n_buf < tq.n_chunks) atomicAdd(tq.chunk_cnt + done_c, 1u);
                atomicAdd(tq.state + Q_ITEMS_DONE, 1u);
            }
            item = atomicAdd(tq.state + Q_HEAD, 1u);
        }
        item = (unsigned)__builtin_amdgcn_readfirstlane((int)item);
        done_b = ~0u;
        if (item >= n_items) break;
        const unsigned c = item / (unsigned)tq.nblk, b = item - c * (unsigned)tq.nblk;
        auto trace = [&](unsigned stage) {
            if (tq.trace && lane == 0) __hip_atomic_store(tq.trace + b, (c << 8) | stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        trace(1);
        auto stamp = [&](int k) {
            if (tq.times && lane == 0) tq.times[(size_t)item * 4 + k] = wall_clock64();
        };
        stamp(0);
        // A block whose previous chunk is not finished when its next one is pulled is behind the others: the call ends with
        // the slowest chain of items (tools/dbg/queue_timeline.py: the slowest block's items add up to 24.9 of the call's 27.1 ms),
        // so its item runs first in line at its SIMD's issue port (tq.lag_prio).
        const bool behind = c != 0 && q_peek(tq.blk_prog + b) < c;
        bool ok = c == 0 || q_wait_ge(tq.blk_prog + b, c, tq, Q_ERR_BLOCK_WAIT);
        if (tq.lag_prio) {
            if (!behind) __builtin_amdgcn_s_setprio(0);
            else if (tq.lag_prio == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(2);
        }
        {   // the pre-pass wavefronts that integrate this block's controlled lanes
            const LanePtr st(p.stat + (size_t)b * (ST_COUNT * 64), (uint32_t)lane * 8u);
            const int64_t cq = fld<int64_t>(st, ST_CTL);
            uint64_t m = __ballot(cq >= 0);
            while (m && ok) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)(uint32_t)cq, l) >> 6;
                ok = q_wait_ge(tq.ctl_prog + w, c + 1, tq, Q_ERR_CTL_WAIT);
            }
        }
        if (!ok) break; // (somebody gave up: the host reports it)
        trace(2);
        q_acquire();
        trace(3);
        stamp(1);
        const double *tab = tq.tab + (size_t)(c % (unsigned)tq.n_buf) * tq.buf_doubles;
        // events this item appends: those beyond what the scenario of this lane's tile has now
        const int gl = (int)b * 64 + lane, r_raw = gl / G;
        const bool in_range = r_raw < p.R;
        const int r = in_range ? r_raw : p.R - 1;
        const int ev_before = p.ev_cap > 0 ? min(p.sdyn[r].n_events, p.ev_cap) : 0;
        if (RSS)
            rollout_body_l<G, 1, false, false, false, false, true, false, false, false, false, true>(lds, p, timestep, tq.k0[c + 1] - tq.k0[c], 0, force,
                                                                                                    nullptr, tab, SliceArgs{}, b);
        else
            rollout_body_l<G, 1, false, true, true, false, false, false, false, PLANAR>(lds, p, timestep, tq.k0[c + 1] - tq.k0[c], 0, force, nullptr,
                                                                                       tab, SliceArgs{}, b);
        trace(4);
        stamp(2);
        if (RSS) { // the line tests this item queued, over full wavefronts (rss_lines_kernel's body), then the state words
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            rss_lines_block(p, (RssQueueLds)&rssq_lds, (size_t)b);
        }
        if (!RSS && p.ev_cap > 0) { // (uniform) a controlled ego's pose at an event of this chunk is a row of the chunk's table: taken now
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the ego lane's event stores)
            const int ev_after = in_range ? min(p.sdyn[r].n_events, p.ev_cap) : 0;
            if (ev_after > ev_before) {
                int64_t ectl;
                const bool from_tab = ego_table_column(p, r, ectl);
                for (int i = ev_before + (gl & (G - 1)); i < ev_after; i += G) event_take_table_pose(p, r, i, tab, from_tab, ectl);
            }
        }
        trace(5);
        if (RSS) {
            // The entities' RSS words travel with the block: the state word is what the next item starts from; the records of the
            // latest update (code, safe distances, the step they belong to) are REWRITTEN by every item that touches the
            // scenario, possibly from another XCD -- left as plain stores, two L2s would hold dirty copies of the same words
            // and the older one could be written back last.  Write-through, like the state rows.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            typedef SG_GLOBAL int gi32;
            typedef SG_GLOBAL unsigned long long gu64;
            const size_t idx = (size_t)b * 64 + lane;
            gi32 *sw = (gi32 *)(p.rss_state + idx), *cw = (gi32 *)(p.rss_code + idx);
            gu64 *fw = (gu64 *)(p.rss_safe + idx * 2);
            const int v0 = __hip_atomic_load(sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int v1 = __hip_atomic_load(cw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long v2 = __hip_atomic_load(fw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long v3 = __hip_atomic_load(fw + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sw, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(cw, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(fw, v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(fw + 1, v3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int glr = (int)b * 64 + lane;
            if ((glr & (G - 1)) == 0 && glr / G < p.R) {
                gi32 *nw = (gi32 *)(p.rss_seen + glr / G);
                const int v4 = __hip_atomic_load(nw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(nw, v4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (tq.handoff == 1) q_handoff_writethrough<G>(p, b, lane);
        else q_release();
        trace(6);
        stamp(3);
        done_b = b;
        done_c = c;
    }
    if (lane == 0 && done_b != ~0u) { // (the last item of this wavefront)
        __hip_atomic_store(tq.blk_prog + done_b, done_c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tq.n_buf < tq.n_chunks) atomicAdd(tq.chunk_cnt + done_c, 1u);
        atomicAdd(tq.state + Q_ITEMS_DONE, 1u);
    }
}

// Register budget: the rollout role's (168 VGPRs, three wavefronts per SIMD -- rollout_kernel_tab / _tab_planar); the pre-pass
// role (control_kernel_fast: 151) fits under it.
template <int G>
__global__ __launch_bounds__(64, SG_TAB_WAVES) __attribute__((amdgpu_num_vgpr(SG_TAB_VGPR))) void rollout_kernel_tabq(
    Params p, double timestep, int force, TabQueue tq)
{
    tabq_body<G, false>(p, timestep, force, tq);
}
template <int G>
__global__ __launch_bounds__(64, SG_PLANAR_WAVES) __attribute__((amdgpu_num_vgpr(SG_PLANAR_VGPR))) void rollout_kernel_tabq_planar(
    Params p, double timestep, int force, TabQueue tq)
{
    tabq_body<G, true>(p, timestep, force, tq);
}

// ... with the RSSDistances callback in the step loop (rollout_kernel_rss_tab's budget: two wavefronts per SIMD)
template <int G>
__global__ __launch_bounds__(64, SG_WAVES_PER_SIMD) void rollout_kernel_rss_tabq(Params p, double timestep, int force, TabQueue tq)
{
    tabq_body<G, false, true>(p, timestep, force, tq);
}

} // namespace sg
