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

    /* error behaviour: a bad argument returns a negative status and a message, nothing aborts */
    int rc = sg_step(h, -1, NULL, 0);
    bad |= rc != SG_ERR_INVALID || strlen(sg_last_error(h)) == 0;
    CHECK(sg_destroy(h));
    printf(bad ? "FAILED\n" : "ok\n");
    return bad ? 1 : 0;
}
