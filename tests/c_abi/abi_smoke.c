/* Plain-C caller of libsgym_hip.so: proves the boundary of include/sgym.h is usable without Python or C++.
 * Two scenarios x three entities: an ego driving along +x through a parked box (collision events), a far bystander.
 * Prints one line per scenario; exit code 0 when the results satisfy the invariants checked below.
 *   gcc -std=c11 -I include tests/c_abi/abi_smoke.c -o abi_smoke -L scenario_gym_amd/lib -lsgym_hip -Wl,-rpath,... */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sgym.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != SG_OK) {                                                          \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, sg_last_error(h));   \
            return 2;                                                                \
        }                                                                            \
    } while (0)

int main(void)
{
    enum { R = 2, E = 3 };
    sg_handle *h = NULL;
    sg_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = 0; cfg.n_scenarios = R; cfg.n_entities = E; cfg.terminal_mask = SG_TERM_MAX_LENGTH;
    cfg.record_capacity = 0; cfg.event_capacity = 8; cfg.timestep = 0.1;
    if (sg_version() != SG_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
    CHECK(sg_create(&cfg, &h));

    /* knots [t, x, y, z, h, p, r]; scenario 1 is scenario 0 with the parked box moved out of the way */
    double knots[2 * (2 + 1 + 1)][7];
    int64_t off[R * E + 1];
    int32_t kind[R * E], etype[R * E], ego[R] = {0, 0};
    double bbox[R * E][4], t0[R] = {0.0, 0.0}, length[R] = {10.0, 10.0};
    int row = 0;
    for (int r = 0; r < R; ++r) {
        const double park_y = r == 0 ? 0.0 : 30.0;
        off[r * E + 0] = row;
        memset(knots[row], 0, sizeof knots[row]); knots[row][0] = 0.0; knots[row][1] = -10.0; ++row;  /* ego start */
        memset(knots[row], 0, sizeof knots[row]); knots[row][0] = 10.0; knots[row][1] = 10.0; ++row;  /* ego end   */
        off[r * E + 1] = row;
        memset(knots[row], 0, sizeof knots[row]); knots[row][2] = park_y; ++row;                      /* parked    */
        off[r * E + 2] = row;
        memset(knots[row], 0, sizeof knots[row]); knots[row][1] = 50.0; knots[row][2] = 50.0; ++row;  /* bystander */
        for (int e = 0; e < E; ++e) {
            kind[r * E + e] = e == 0 ? SG_KIND_AGENT_REPLAY : SG_KIND_REPLAY;
            etype[r * E + e] = 2; /* not a Vehicle: CollisionMetric type non_vehicle */
            bbox[r * E + e][0] = 2.0; bbox[r * E + e][1] = 4.0; bbox[r * E + e][2] = 0.0; bbox[r * E + e][3] = 0.0;
        }
    }
    off[R * E] = row;
    sg_scenarios sc;
    memset(&sc, 0, sizeof sc);
    sc.kind = kind; sc.etype = etype; sc.bbox = &bbox[0][0]; sc.knot_off = off; sc.knots = &knots[0][0];
    sc.ctrl = NULL; sc.ego = ego; sc.t0 = t0; sc.length = length;
    CHECK(sg_upload(h, &sc));
    CHECK(sg_rollout(h, 200));

    sg_metrics m[R];
    sg_event ev[16];
    int32_t n_ev = 0;
    CHECK(sg_read_metrics(h, m, ev, 16, &n_ev));
    for (int r = 0; r < R; ++r)
        printf("scenario %d: steps %d done %d final_t %.6f distance %.6f max_speed %.6f collisions %d\n", r, m[r].n_steps,
               m[r].done, m[r].final_t, m[r].ego_distance_travelled, m[r].ego_max_speed, m[r].n_collisions);
    int bad = 0;
    for (int r = 0; r < R; ++r) {
        bad |= !m[r].done || m[r].n_steps < 99 || m[r].n_steps > 101;
        bad |= fabs(m[r].ego_distance_travelled - 2.0 * m[r].n_steps * 0.1) > 1e-9; /* 2 m/s along x */
        bad |= fabs(m[r].ego_max_speed - 2.0) > 1e-9;
    }
    bad |= m[0].n_collisions != 1 || m[1].n_collisions != 0 || n_ev != 1;
    bad |= n_ev == 1 && (ev[0].scenario != 0 || ev[0].other != 1 || ev[0].type != 5);
    /* boxes 4 m long touch when the centres are 4 m apart: x_ego = -4 at t = 3.0 (closed sets: touching counts) */
    bad |= n_ev == 1 && fabs(ev[0].t - 3.0) > 0.1 + 1e-9;

    /* state view: raw device pointers + the documented block layout, copied back with sg_copy_to_host */
    sg_state_view v;
    CHECK(sg_state_view_get(h, &v));
    double *blocks = malloc((size_t)v.n_blocks * v.block_rows * 64 * sizeof(double));
    CHECK(sg_copy_to_host(h, v.blocks, blocks, (uint64_t)v.n_blocks * v.block_rows * 64 * sizeof(double)));
    for (int r = 0; r < R; ++r) {
        int i = r * v.entity_stride; /* ego slot */
        double x = blocks[(size_t)(i / 64) * v.block_rows * 64 + (SG_F_POSE + 0) * 64 + (i % 64)];
        double vx = blocks[(size_t)(i / 64) * v.block_rows * 64 + (SG_F_VEL + 0) * 64 + (i % 64)];
        printf("scenario %d: ego x %.6f vx %.6f\n", r, x, vx);
        bad |= fabs(x - (-10.0 + 2.0 * m[r].final_t)) > 1e-9 || fabs(vx - 2.0) > 1e-9;
    }
    free(blocks);

    /* road surfaces from C: one network (a 12 m x 6 m road rectangle around the ego's lane) shared by both scenarios;
     * the final state has the ego at x = 10: 2 m past the road's end at x = 8 */
    {
        const int32_t net_of[R] = {0, 0};
        const int64_t poly_off[2] = {0, 1}, ring_off[2] = {0, 1}, vert_off[2] = {0, 4};
        const double verts[4][2] = {{-12.0, -3.0}, {8.0, -3.0}, {8.0, 3.0}, {-12.0, 3.0}};
        const uint32_t layers[1] = {SG_LAYER_DRIVEABLE | SG_LAYER_ROAD};
        sg_road_networks rn;
        memset(&rn, 0, sizeof rn);
        rn.n_networks = 1; rn.net_of_scenario = net_of; rn.poly_off = poly_off; rn.ring_off = ring_off; rn.vert_off = vert_off;
        rn.verts = &verts[0][0]; rn.layers = layers;
        CHECK(sg_set_road_networks(h, &rn));
        uint32_t flags[R];
        CHECK(sg_terminal_flags(h, flags, NULL));
        for (int r = 0; r < R; ++r) bad |= !(flags[r] & SG_TERM_MAX_LENGTH) || !(flags[r] & SG_TERM_EGO_OFF_ROAD);
        const int32_t lay[2] = {0, SG_LAYER_DRIVEABLE};
        uint8_t map[R][2][5][5];
        CHECK(sg_raster_map(h, 8.0, 8.0, 5, 5, 2, lay, &map[0][0][0][0]));
        /* ego at (10, 0), heading 0: the grid frame is rotated by pi/2, rows run along -x ... the centre cell is the ego's
         * own box; the driveable layer is set only on the side that looks back at the road (x <= 8) */
        int on = 0;
        for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) on += map[0][1][i][j];
        printf("map: centre entity cell %d, driveable cells %d of 25\n", map[0][0][2][2], on);
        bad |= map[0][0][2][2] != 1 || on == 0 || on == 25;
        /* restart scenario 1 only, then one tick of both as a captured graph with zero actions */
        const uint8_t mask[R] = {0, 1};
        CHECK(sg_reset_scenarios(h, mask));
        const uint8_t *d_obs = NULL;
        const uint32_t *d_fl = NULL;
        CHECK(sg_tick(h, NULL, 0, 8.0, 8.0, 5, 5, 2, lay, &d_obs, &d_fl));
        CHECK(sg_synchronize(h));
        CHECK(sg_copy_to_host(h, d_fl, flags, sizeof flags));
        printf("flags after restart + tick: %u %u\n", flags[0], flags[1]);
        bad |= (flags[1] & (SG_TERM_MAX_LENGTH | SG_TERM_EGO_OFF_ROAD)) != 0; /* scenario 1 is back at x = -9.8, on the road */
        bad |= !(flags[0] & SG_TERM_EGO_OFF_ROAD);
    }

    /* the same batch from page-locked memory (sg_host_alloc) with the RSSDistances callback inside the rollout (sg_set_rss):
       the parked box of scenario 0 is run into -> an "unsafe" record and a cleared metric flag; scenario 1 stays safe */
    {
        double *pk = NULL;
        CHECK(sg_host_alloc(0, sizeof knots, (void **)&pk));
        memcpy(pk, knots, sizeof knots);
        sc.knots = pk;
        CHECK(sg_set_rss(h, 1));
        CHECK(sg_upload(h, &sc));
        CHECK(sg_rollout(h, 200));
        uint8_t rss_flags[R];
        int32_t codes[R * E];
        double safe[R * E * 2];
        CHECK(sg_rss_read(h, rss_flags, codes, safe));
        printf("rss flags %u %u, codes of scenario 0: %d %d %d\n", rss_flags[0], rss_flags[1], codes[0], codes[1], codes[2]);
        bad |= codes[0] != -1 || codes[1] < 4 || codes[E + 1] > 3; /* ego: no record; parked box: unsafe / found; far away: safe */
        bad |= rss_flags[0] == 3 || rss_flags[1] != 3;
        bad |= !(safe[2] > 0.0) || !(safe[3] > 0.0);
        sg_metrics m2[R];
        CHECK(sg_read_metrics(h, m2, ev, 16, &n_ev));
        bad |= m2[0].n_steps != m[0].n_steps || m2[0].n_collisions != 1;
        CHECK(sg_set_rss(h, 0));
        sc.knots = &knots[0][0];
        CHECK(sg_host_free(pk));
    }

    /* the launch schedule of the last call (sg_schedule_info: 0 not the table path, 1 chunk launches, 2 the persistent
     * launch), and the behaviour model switch: refused after sg_upload unless it is the one in force */
    {
        int32_t info[8];
        CHECK(sg_schedule_info(h, info));
        bad |= info[0] < 0 || info[0] > 2 || (info[0] == 2 && (info[1] < 1 || info[2] < 1 || info[2] > info[1] || info[3] < 1)) ||
               info[5] <= 0 || info[6] <= 0;
        CHECK(sg_set_ped_behaviour(h, SG_PED_SOCIAL_FORCE));
        bad |= sg_set_ped_behaviour(h, SG_PED_RANDOM_WALK) != SG_ERR_STATE || sg_set_ped_behaviour(h, 7) != SG_ERR_INVALID;
    }

    /* error behaviour: a bad argument returns a negative status and a message, nothing aborts */
    int rc = sg_step(h, -1, NULL, 0);
    bad |= rc != SG_ERR_INVALID || strlen(sg_last_error(h)) == 0;
    CHECK(sg_destroy(h));
    printf(bad ? "FAILED\n" : "ok\n");
    return bad ? 1 : 0;
}
