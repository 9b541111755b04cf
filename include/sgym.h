/*
 * sgym.h -- C ABI of libsgym_hip.so: the MI355X (gfx950) batched scenario-rollout engine.
 *
 * The reference (driskai/scenario_gym v0.3.1) is pure Python and has no FFI; its extension surface
 * for the per-step hot path is Python subclassing.  This header is the device boundary the build
 * introduces at the ScenarioGym <-> (agents + State + metrics) seam; every entry point names the
 * reference interface it stands in for (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 (SG_OK) or a negative sg_status.
 *   - the caller owns every host buffer it passes; the library owns all device memory and one HIP
 *     stream per handle.  Device pointers returned by sg_state_view stay valid until sg_upload /
 *     sg_destroy on that handle.
 *   - a handle is bound to one device and is not thread-safe; different handles may be driven
 *     from different host threads.  Calls are synchronous on return unless documented otherwise.
 *   - R = scenarios (replicas) on this handle, E = entity slots per scenario (ragged scenarios
 *     pad with SG_KIND_NONE), poses are fp64 [x, y, z, h, p, r] exactly as in the reference.
 */
#ifndef SGYM_H
#define SGYM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SG_ABI_VERSION 5 /* 2: sg_scenario_state.last_row_hi (scenarios of up to 512 entities); 3: sg_schedule_info replaces sg_pipeline_info; 4: sg_crowd_walk_stats removed; 5: sg_last_kernel */

typedef enum {
    SG_OK = 0,
    SG_ERR_INVALID = -1,   /* bad argument */
    SG_ERR_HIP = -2,       /* HIP runtime error, see sg_last_error */
    SG_ERR_STATE = -3,     /* call order (e.g. step before upload) */
    SG_ERR_NO_DEVICE = -4, /* no gfx950 device visible */
    SG_ERR_CAPACITY = -5,  /* record / event capacity exceeded */
} sg_status;

/* how an entity slot gets its next pose (ScenarioGym.create_agents, scenario_gym/scenario_gym.py:188-211) */
typedef enum {
    SG_KIND_NONE = 0,          /* padding, never present */
    SG_KIND_REPLAY = 1,        /* non-agent: BatchReplayEntity, scenario_gym/entity/batch.py:34-128 */
    SG_KIND_AGENT_REPLAY = 2,  /* ReplayTrajectoryAgent + ReplayTrajectoryController, agent.py:118-128, controller.py:45-54 */
    SG_KIND_AGENT_PID = 3,     /* PIDAgent + PIDController, agent.py:131-148, controller.py:143-258 */
    SG_KIND_AGENT_VEHICLE = 4, /* external (accel, steer) -> VehicleController, controller.py:57-140; integrations/openaigym.py:197-204 */
    SG_KIND_AGENT_PEDESTRIAN = 5, /* PedestrianAgent + SocialForce + PedestrianController, pedestrian/{agent,social_force,controller}.py */
    SG_KIND_AGENT_EXTERNAL = 6,   /* any other Agent subclass (agent.py:18-116): agent.step(state) runs in the caller, one
                                     sg_step per tick, and its pose enters through sg_set_external_poses */
} sg_kind;

/* TERMINAL_CONDITIONS, scenario_gym/state/state.py:397-408 */
#define SG_TERM_MAX_LENGTH 1u
#define SG_TERM_COLLISION 2u
#define SG_TERM_EGO_COLLISION 4u
#define SG_TERM_EGO_OFF_ROAD 8u /* entities[0] absent or not strictly inside RoadNetwork.driveable_surface (sg_set_road_networks);
                                   with pedestrian agents in the batch: a launch of its own behind every step */

/* The unions of RoadGeometry boundaries the reference takes: RoadNetwork.driveable_surface / walkable_surface /
 * impenetrable_surface (road_network/road_network.py:306-328, flags in road_network/objects.py) and the per-layer
 * unions of RasterizedMapSensor (sensor/map.py:194-271).  A polygon carries the bits of every union it is part of. */
#define SG_LAYER_DRIVEABLE 1u    /* roads, intersections and all lanes */
#define SG_LAYER_ROAD 2u
#define SG_LAYER_INTERSECTION 4u
#define SG_LAYER_LANE 8u         /* the lanes of the roads */
#define SG_LAYER_WALKABLE 16u    /* pavements, crossings, buildings */
#define SG_LAYER_PAVEMENT 32u
#define SG_LAYER_CROSSING 64u
#define SG_LAYER_IMPENETRABLE 128u

/* The boundary polygons of the scenarios' road networks (Scenario.road_network; JSON format of
 * road_network/utils.py:6-41).  Networks are shared: scenario r uses network net_of_scenario[r] (-1: none, every
 * surface empty).  Polygon q of network n = polygons [poly_off[n], poly_off[n+1]); its rings (exterior first, then
 * holes) = rings [ring_off[q], ring_off[q+1]); ring vertices = verts[vert_off[ring] .. vert_off[ring+1]), rings OPEN
 * (last vertex != first).  HOST arrays, copied by the call. */
typedef struct sg_road_networks {
    int32_t n_networks;
    const int32_t *net_of_scenario; /* [n_scenarios] */
    const int64_t *poly_off;        /* [n_networks + 1] */
    const int64_t *ring_off;        /* [n_polygons + 1] */
    const int64_t *vert_off;        /* [n_rings + 1] */
    const double *verts;            /* [n_verts][2] x, y */
    const uint32_t *layers;         /* [n_polygons] SG_LAYER_* */
} sg_road_networks;

/* controller parameter slots (VehicleController.__init__ controller.py:64-98, PIDController.__init__ :154-196) */
enum {
    SG_C_MAX_STEER = 0,
    SG_C_MAX_ACCEL = 1,
    SG_C_MAX_SPEED = 2, /* NaN = None */
    SG_C_ALLOW_REVERSE = 3,
    SG_C_STEER_KP = 4,
    SG_C_STEER_KD = 5,
    SG_C_ACCEL_KP = 6,
    SG_C_ACCEL_KD = 7,
    SG_C_ACCEL_KI = 8,
    /* PedestrianAgent / PedestrianSensor / PedestrianController (pedestrian/agent.py:18-37, controller.py:12-19) */
    SG_C_PED_SPEED_DESIRED = 9,
    SG_C_PED_MAX_SPEED = 10,
    SG_C_PED_HEAD_ROT = 11,
    SG_C_PED_RADIUS = 12, /* PedestrianSensor.distance_threshold */
    SG_NCTRL = 16
};

/* ScenarioGym.__init__(timestep, persist, terminal_conditions), scenario_gym/scenario_gym.py:29-95 */
typedef struct {
    int32_t device;          /* HIP device ordinal */
    int32_t n_scenarios;     /* R */
    int32_t n_entities;      /* E, entity slots per scenario (the reference has no ceiling, state/utils.py:10-49; here <= 16384).
                                Up to 512: one workgroup per scenario, every kind, callback and observation call.  Beyond: the
                                step runs as four kernels over as many workgroups as the scenario needs -- every entity kind
                                (caller-run agents; pedestrian agents with the counter-based noise), the RSS callback (a launch
                                of its own per step), road networks with ego_off_road, the pedestrians' boundary forces and the
                                map's surface layers, both noise modes, several pedestrian models, the observation calls and
                                sg_tick: nothing is refused at that width */
    int32_t persist;         /* ScenarioGym(persist=...) */
    uint32_t terminal_mask;  /* SG_TERM_* */
    int32_t record_capacity; /* rows of State._recorded_poses kept on device (0 = off), state.py:227-228 */
    int32_t event_capacity;  /* CollisionMetric events kept per scenario (metrics/collision.py:70-75); later events are
                                counted in sg_metrics.n_collisions but not stored */
    int32_t reserved;
    double timestep;         /* ScenarioGym(timestep=...) */
} sg_config;

/* Numeric content of R scenarios: what State/Entity/Trajectory objects hold after load.
 * All pointers are HOST pointers, entity index i = scenario*E + slot. */
typedef struct {
    const int32_t *kind;     /* [R*E] sg_kind */
    const int32_t *etype;    /* [R*E] catalog type: 0 Vehicle, 1 Pedestrian, 2 other (metrics/collision.py:85) */
    const double *bbox;      /* [R*E][4] BoundingBox width, length, center_x, center_y (catalog_entry.py:83-91) */
    const int64_t *knot_off; /* [R*E+1] row offsets into knots (empty range for SG_KIND_NONE) */
    const double *knots;     /* [rows][7] Trajectory.data rows t,x,y,z,h,p,r (trajectory.py:91) */
    const double *ctrl;      /* [R*E][SG_NCTRL] or NULL for the reference defaults */
    const int32_t *ego;      /* [R] slot of Scenario.ego (scenario/scenario.py:53-65) */
    const double *t0;        /* [R] ScenarioGym.get_start_time (scenario_gym.py:213-215) */
    const double *length;    /* [R] Scenario.length (scenario/scenario.py:88-91) */
    const int64_t *route_off; /* [R*E+1] row offsets into routes, or NULL when there are no pedestrian agents */
    const double *routes;     /* [rows][2] PedestrianAgent.route waypoints (pedestrian/agent.py:43) */
} sg_scenarios;

/* SocialForceParameters (pedestrian/social_force.py:16-30; behaviour.py max_speed_factor; random_walk.py bias):
 * one set per handle.  The Gaussian noise terms (std_lon, std_lat; social_force.py:106-108) are set with
 * sg_set_ped_noise; without that call they are off (std 0: np.random.normal(b, 0) == b). */
typedef struct {
    double relaxation_time, ped_repulse_V, ped_repulse_sigma, ped_attract_C;
    double sight_weight, sight_weight_use;
    double cos_sight;        /* cos(sight_angle / 2 * pi / 180), computed by the caller */
    double max_speed_factor, bias_lon, bias_lat;
    /* repulsion from the impenetrable surface (buildings) of the scenario's road network, social_force.py:97-104; acts
     * when sg_set_road_networks has been called.  (The "walkable boundary" term :86-95 is evaluated only for a pedestrian
     * inside the walkable surface, where shapely's nearest point is the pedestrian itself: it is the zero vector whatever
     * boundary_repulse_U / R are, so they are not parameters here.) */
    double imp_boundary_repulse_U, imp_boundary_repulse_R;
} sg_social_force;

/* Device-resident state after the latest step: the arrays behind State.poses / velocities /
 * distances / collisions() (state.py:90-96, 306-310).
 *
 * Layout: entity slot i = scenario * entity_stride + slot.  Entities are grouped in blocks of 64
 * consecutive slots (the 64 lanes of one gfx950 wavefront); inside a block every field is one row of
 * 64 eight-byte values, so a wavefront loads/stores whole 512-byte rows:
 *     value(field f, entity i) = blocks[(i / 64) * block_rows * 64 + f * 64 + (i % 64)]
 * block_rows = SG_F_COLL + row_words, row_words = ceil(entity_stride / 64) (1 for up to 64 entities).
 * Raw device pointers (wrap with torch/dlpack for zero-copy strided views). */
enum {
    SG_F_POSE = 0,      /* 6 rows: x, y, z, h, p, r  (State.poses) */
    SG_F_VEL = 6,       /* 6 rows (State.velocities) */
    SG_F_DIST = 12,     /* State.distances */
    SG_F_PRESENT = 13,  /* uint64 0/1: entity in State.poses */
    SG_F_CTRL = 14,     /* 4 rows: controller speed, e_lon_prev, e_lat_prev, e_lon_int (controller.py:100-103,198-203);
                           pedestrians: controller speed, goal_idx (pedestrian/controller.py:21-23, agent.py:38) */
    SG_F_FORCE = 18,    /* 2 rows: PedestrianAgent.force (pedestrian/agent.py:41, social_force.py:114) */
    SG_F_COLL = 20      /* row_words rows of uint64: adjacency row of State.collisions(), bit j = slot j of the scenario */
};

/* per-scenario clock, terminal flag and metric accumulators */
typedef struct {
    double t, prev_t;              /* State.t, State.prev_t */
    double ego_avg_speed;          /* EgoAvgSpeed (metrics/trajectory.py:8-28) */
    double ego_max_speed;          /* EgoMaxSpeed (:31-48) */
    double avg_t;                  /* EgoAvgSpeed.t */
    double ego_distance_travelled; /* EgoDistanceTravelled (:51-66) */
    uint64_t last_row[4];          /* CollisionMetric.last_timestep (metrics/collision.py:75) */
    int32_t done;                  /* State.is_done */
    int32_t n_steps;               /* steps since reset */
    int32_t n_events;              /* len(CollisionMetric.collisions) */
    int32_t rec_rows;              /* rows written to the pose record */
    int64_t noise_pos;             /* variates of the scenario's noise stream consumed so far (sg_set_ped_noise, SG_NOISE_STREAM) */
    uint64_t last_row_hi[4];       /* words 4..7 of last_row (scenarios of 257..512 entities; beyond 512 the rows live in
                                      sg_handle scratch, sgym_wide.hpp) */
} sg_scenario_state;               /* 136 bytes */

typedef struct {
    int32_t n_scenarios, n_entities, entity_stride, n_blocks;
    int32_t row_words, block_rows;
    double *blocks;          /* [n_blocks][block_rows][64] */
    sg_scenario_state *scen; /* [n_scenarios] */
} sg_state_view;

/* per-scenario metric row: EgoAvgSpeed/EgoMaxSpeed/EgoDistanceTravelled (metrics/trajectory.py:8-66),
 * CollisionMetric count (metrics/collision.py:46-79), plus bookkeeping */
typedef struct {
    double ego_avg_speed, ego_max_speed, ego_distance_travelled;
    double final_t;
    int32_t n_steps, done, n_collisions, reserved;
} sg_metrics;

/* one CollisionMetric entry (t, other entity, type): metrics/collision.py:81-203.  type = CollisionTypes: 0 other,
 * 1 t_bone, 2 head_on, 3 rear_end, 4 side_swipe, 5 non_vehicle (hazard's catalog type is not "Vehicle").  Vehicle hazards
 * are classified by record_collision's tree with state.poses[...] where the reference writes the (missing) `.pose`
 * attribute; the classification runs when the events are read (sg_read_metrics) from the ego pose the library keeps with
 * every event (device side) and the hazard's pose at time t re-evaluated from its trajectory.  -2 = a Vehicle hazard whose pose cannot be re-evaluated (it is itself a controlled /
 * caller-run agent): left unclassified.  Values <= -1 other than -2 never leave sg_read_metrics. */
typedef struct {
    double t;
    int32_t scenario, other;
    int32_t type;
    int32_t reserved;
} sg_event;

typedef struct sg_handle sg_handle;

int sg_version(void);
const char *sg_last_error(const sg_handle *h); /* valid until the next call on h; h may be NULL for create errors */

/* ScenarioGym(...) */
int sg_create(const sg_config *cfg, sg_handle **out);
int sg_destroy(sg_handle *h);

/* ScenarioGym.set_scenario -> State(...) + create_agents (scenario_gym.py:157-211): copies the
 * scenarios to HBM, builds the BatchReplayEntity union knot grids + stage-1 resample on device
 * (entity/batch.py:83-109), then resets (sg_reset). */
int sg_upload(sg_handle *h, const sg_scenarios *sc);

/* Page-locked host memory for the arrays handed to sg_upload (the knots above all: 1.6 GB for 4096 x 64 x 128).  The copy
 * engine reads such memory directly at the PCIe rate; from ordinary (pageable) memory the runtime first copies every piece
 * into a staging buffer on a CPU core, beside the host threads of sg_upload.  Ordinary memory keeps working.  No reference
 * counterpart (the reference never leaves the host).  device: the GPU the memory is registered with. */
int sg_host_alloc(int32_t device, uint64_t bytes, void **out);
int sg_host_free(void *p);

/* SocialForce(params) shared by every pedestrian agent of the handle; call before sg_upload (defaults otherwise) */
int sg_set_social_force(sg_handle *h, const sg_social_force *params);

/* Per-agent behaviour models.  The reference gives every PedestrianAgent its own behaviour object with its own parameters
 * (pedestrian/agent.py:18-41 `behaviour`, social_force.py:33-42, random_walk.py:22-31): a scenario may mix SocialForce
 * pedestrians of different parameter sets with RandomWalk pedestrians.  `models[n_models]` are the distinct (behaviour,
 * parameters, noise std) combinations of the batch, `model_of[n_scenarios * n_entities]` the model of every entity slot (read
 * for SG_KIND_AGENT_PEDESTRIAN slots only; others: any value in range, or -1).  Call before sg_upload (the batch's kernels
 * are chosen there).  One model: the same as sg_set_social_force + sg_set_ped_behaviour + the std of sg_set_ped_noise.
 * Several: every pedestrian steps under its own model (the force on a pedestrian is computed with ITS parameters from its
 * neighbours' states, social_force.py:44-114); the all-pedestrian crowd kernels, which hold one parameter set, stand back
 * for the general pedestrian variant.  The noise MODE (off / stream / device, sg_set_ped_noise) stays one per handle: the
 * reference draws every agent's two variates from the one global generator, in agent order, whatever its model.
 * SG_ERR_INVALID: more than SG_MAX_PED_MODELS models, an index out of range (a refused call leaves the handle's models as
 * they were). */
#define SG_MAX_PED_MODELS 16
typedef struct {
    int32_t behaviour;       /* SG_PED_SOCIAL_FORCE / SG_PED_RANDOM_WALK */
    int32_t reserved;
    sg_social_force params;
    double std_lon, std_lat; /* social_force.py:106-108 / random_walk.py:37-43 (used when the handle's noise mode is not off) */
} sg_ped_model;
int sg_set_ped_models(sg_handle *h, int32_t n_models, const sg_ped_model *models, const int32_t *model_of);

/* The random fluctuations of SocialForce._step (pedestrian/social_force.py:106-114): every pedestrian that is still
 * walking adds np.random.normal(bias_lon, std_lon) to its speed and np.random.normal(bias_lat, std_lat) to its heading, in
 * agent (entity) order, from numpy's global legacy generator: loc + scale * z.
 *   SG_NOISE_OFF     std = 0.
 *   SG_NOISE_STREAM  parity runs.  normals: HOST [n_scenarios][per_scenario] standard normal variates, copied by the call;
 *                    scenario r consumes row r in order, two per walking pedestrian per step -- with row r =
 *                    np.random.RandomState(k_r).standard_normal(n) the scenario reproduces the reference's rollout after
 *                    np.random.seed(k_r).  A reset rewinds the rows of the scenarios it resets.  A scenario that runs out of
 *                    variates makes sg_read_metrics fail with SG_ERR_CAPACITY (its later draws were taken as 0).
 *   SG_NOISE_DEVICE  timing / production runs: a counter-based generator on the device -- Philox4x32-10 keyed by
 *                    (seed, scenario) at counter (entity, step), Box-Muller in fp64 -- the same distribution, its own
 *                    stream, no state, no host traffic.  normals / per_scenario are ignored.
 * May be called before or after sg_upload; stays in force until the next call. */
#define SG_NOISE_OFF 0
#define SG_NOISE_STREAM 1
#define SG_NOISE_DEVICE 2
int sg_set_ped_noise(sg_handle *h, int32_t mode, double std_lon, double std_lat, const double *normals, int64_t per_scenario,
                     uint64_t seed);

/* The behaviour model of the handle's pedestrian agents (PedestrianAgent(..., behaviour=...), pedestrian/agent.py:23):
 *   SG_PED_SOCIAL_FORCE  SocialForce (pedestrian/social_force.py:33-222), the default;
 *   SG_PED_RANDOM_WALK   RandomWalk (pedestrian/random_walk.py:22-44): speed = np.random.normal(speed_desired + bias_lon,
 *                        std_lon), heading = np.random.normal(atan2(goal - position) + bias_lat, std_lat) -- no neighbours,
 *                        no speed limit other than the controller's max_speed, PedestrianAgent.force stays (0, 0).  bias_* from
 *                        sg_set_social_force, the variates and std_* from sg_set_ped_noise (two per walking pedestrian
 *                        and step, speed first, as SocialForce draws them).
 * Call before sg_upload (SG_ERR_STATE afterwards: the kernels of a batch are chosen there). */
#define SG_PED_SOCIAL_FORCE 0
#define SG_PED_RANDOM_WALK 1
int sg_set_ped_behaviour(sg_handle *h, int32_t behaviour);

/* ScenarioGym.reset_scenario -> State.reset(t0), Controller.reset, Metric.reset (scenario_gym.py:217-225) */
int sg_reset(sg_handle *h);

/* The same for the scenarios with mask[r] != 0 only (HOST [n_scenarios]): one environment of a vector of environments
 * starts a new episode (`Env.reset` of integrations/openaigym.py:128-169) while the others keep their state. */
int sg_reset_scenarios(sg_handle *h, const uint8_t *mask);

/* TERMINAL_CONDITIONS (state/state.py:397-408) evaluated on the current state of every scenario, all four of them
 * whatever the handle's terminal_mask: flags[r] = SG_TERM_* bits (EGO_OFF_ROAD against the networks of
 * sg_set_road_networks; none set = empty surfaces = off the road).  This is what the reward of the reference's RL agent
 * asks of a done state (integrations/openaigym.py:300-310).  out: HOST [n_scenarios] or NULL; d_out: if not NULL receives
 * a DEVICE pointer to the same flags (owned by the handle, rewritten by the next call, stream-ordered, not synchronised
 * unless `out` is given). */
int sg_terminal_flags(sg_handle *h, uint32_t *out, const uint32_t **d_out);

/* gym.timestep = x between steps (tests/test_scenario_gym.py:37-39) */
int sg_set_timestep(sg_handle *h, double timestep);

/* n x ScenarioGym.step() on every scenario, done or not (scenario_gym.py:227-254).
 * actions: HOST [n_steps][R][2] (accel, steer) for SG_KIND_AGENT_VEHICLE slots or NULL;
 * actions_device != 0 means `actions` is a DEVICE pointer of the same shape. */
int sg_step(sg_handle *h, int32_t n_steps, const double *actions, int32_t actions_device);

/* Poses returned by the caller's own agents (Agent.step(state) -> Controller.step(state, action), agent.py:52-57,
 * controller.py:30-42) for the SG_KIND_AGENT_EXTERNAL slots, consumed by every following sg_step until replaced.
 * poses: HOST [R*E][6], entity index scenario*E + slot; rows of other slots are ignored; x = NaN means the agent
 * returned None (the entity vanishes, or keeps its pose under `persist`; scenario_gym.py:233-239).  An agent that is not
 * present yet spawns at its trajectory start like every other agent (:240-244).  sg_rollout refuses batches with such
 * slots (SG_ERR_STATE): they have to be driven tick by tick. */
int sg_set_external_poses(sg_handle *h, const double *poses);

/* ScenarioGym.rollout(): reset, then step each scenario while it is not done, at most max_steps
 * (scenario_gym.py:256-267).  The time loop runs inside the kernels: one launch for short runs; long runs of batches with
 * PID / vehicle agents as ONE persistent launch that carries the controller pre-pass (sg_schedule_info), else one launch per
 * chunk of steps.  Scenarios of more than 512 entities step through four launches per step; the host only enqueues there
 * too (every 64 steps a kernel reports into page-locked memory how many scenarios still run, and the call stops enqueuing
 * once an earlier report says none: the steps enqueued meanwhile are no-ops). */
int sg_rollout(sg_handle *h, int32_t max_steps);
/* Same without the reset and without waiting: enqueue on the handle's stream (bench timing) */
int sg_rollout_async(sg_handle *h, int32_t max_steps, int32_t do_reset);
int sg_synchronize(sg_handle *h);
void *sg_stream(sg_handle *h); /* hipStream_t */

int sg_state_view_get(sg_handle *h, sg_state_view *out);

/* FutureCollisionDetector._step (sensor/common.py:59-106) for the ego of every scenario at its current State.t:
 * out[r] = 1 iff, at one of the n_samples times np.linspace(t, t + horizon, n_samples), the ego's box at
 * trajectory.position_at_t(t_j) overlaps another entity's box at that entity's own position_at_t(t_j) (reference
 * defaults: horizon 5.0, 10 samples).  out: HOST [R]. */
int sg_future_collision(sg_handle *h, double horizon, int32_t n_samples, uint8_t *out);

/* RasterizedMapSensor "entity" layer (sensor/map.py:120-192) for the ego of every scenario at the current state:
 * out[r][i][j] = 1 iff the point (linspace(-width/2, width/2, nw)[j], linspace(-height/2, height/2, nh)[i]) of the ego's
 * frame (rotated by heading + pi/2) lies strictly inside the bounding box of a present entity, the ego included;
 * all zeros for a scenario whose ego is absent.  The reference sensor has nw == nh.  out: HOST [R][nh][nw] bytes. */
int sg_raster_entities(sg_handle *h, double width, double height, int32_t nw, int32_t nh, uint8_t *out);

/* Scenario.road_network for the uploaded batch: after sg_upload (which forgets the previous networks), before the
 * first step that needs them.  `contains(Point)` on a union (state.py:401-407, sensor/map.py:198-271) is answered as
 * shapely answers it for points that are not on a polygon's boundary: strictly inside one of the polygons of the union,
 * by the crossing number of its rings with an exact orientation sign (JTS/GEOS RayCrossingCounter); points ON a ring are
 * outside.  The library lays a uniform grid over each network (cells wholly inside / wholly outside a union answer at
 * once, the others test the polygons that cross them) -- an index only, the answers are those of the plain test. */
int sg_set_road_networks(sg_handle *h, const sg_road_networks *nets);

/* RasterizedMapSensor._step (sensor/map.py:136-149) for the ego (entities[0] of the reference's tests = the scenario's
 * ego) of every scenario: layers[k] = 0 is the "entity" layer of sg_raster_entities, otherwise ONE SG_LAYER_* bit
 * ("driveable_surface", "road", "intersection", "lane", "walkable_surface", "pavement", "crossing");
 * out[r][k][i][j] as in sg_raster_entities (the sensor's channels_first layout).  out: HOST [R][n_layers][nh][nw]. */
int sg_raster_map(sg_handle *h, double width, double height, int32_t nw, int32_t nh, int32_t n_layers,
                  const int32_t *layers, uint8_t *out);

/* The same maps left in device memory for a policy that runs on this GPU (the observation of
 * integrations/openaigym.py:280-297 without the trip through the host): *d_out = DEVICE [R][n_layers][nh][nw] bytes inside
 * the handle's observation scratch, written by kernels queued on sg_stream(h) -- not synchronised; valid until the next
 * observation call (sg_raster_*, sg_future_collision) or sg_destroy. */
int sg_raster_map_device(sg_handle *h, double width, double height, int32_t nw, int32_t nh, int32_t n_layers,
                         const int32_t *layers, const uint8_t **d_out);

/* One tick of the external-action loop (integrations/openaigym.py:171-226) for every scenario, as one captured hipGraph:
 * sg_step(h, 1, actions) + sg_terminal_flags + sg_raster_map_device with the given observation geometry (1..8 layers).
 * actions: [n_scenarios][2] HOST (actions_device = 0) or DEVICE, or NULL for (0, 0).  Asynchronous: the work is queued on
 * sg_stream(h); *d_obs (DEVICE [R][n_layers][nh][nw] bytes) and *d_flags (DEVICE [R] SG_TERM_* bits) are valid after
 * sg_synchronize(h) until the next observation call.  The graph is rebuilt when the batch, the networks, the time step or
 * the geometry change (or sg_set_rss is switched).  Not for batches with SG_KIND_AGENT_EXTERNAL slots.  With sg_set_rss the
 * captured step runs the RSS callback as sg_step does. */
int sg_tick(sg_handle *h, const double *actions, int32_t actions_device, double width, double height, int32_t nw, int32_t nh,
            int32_t n_layers, const int32_t *layers, const uint8_t **d_obs, const uint32_t **d_flags);

/* RSSDistances.__call__ (metrics/rss/callback.py:58-128) on the current state of every scenario: safe lateral /
 * longitudinal distances between the ego (which must be entity 0, as the reference's callback assumes) and every present
 * entity, the record it appends to that entity's history, and the "unsafe" verdicts the RSS metric reads
 * (metrics/rss/rss.py:70-104).  Call once after sg_reset (reset = 1: forget the histories) and once after every step;
 * scenarios at t == 0.0 are skipped as in the reference.  Asynchronous on sg_stream(h). */
int sg_rss_update(sg_handle *h, int32_t reset);
/* enabled != 0: sg_upload / sg_reset / sg_reset_scenarios / sg_rollout / sg_step / sg_tick run the callback themselves --
 * after the reset and after every step, inside the rollout kernel (any number of steps per launch; also with pedestrian
 * agents and with SG_TERM_EGO_OFF_ROAD) -- as ScenarioGym(state_callbacks=[RSSDistances()]) does.  The line tests of the
 * callback are queued on the device and finished by a second kernel after each launch (queues: env SG_RSSQ_MB, default
 * an eighth of the free device memory and at least 4096 MiB per handle, at most SG_RSSQ_STEPS = 1024 steps per launch; longer
 * calls are cut into several launches; a device short of memory gets shorter launches, not an error).  A scenario that has not stepped since its latest
 * update (it is done) is left alone.  The ego has to be entity 0 (else: one sg_rss_update per step, which says so).
 * Combinations no fused kernel variant carries -- scenarios of more than 512 entities; 257..512 entities with pedestrian
 * agents or SG_TERM_EGO_OFF_ROAD -- run the callback as a launch of its own behind every step: same records. */
int sg_set_rss(sg_handle *h, int32_t enabled);
/* flags [R]: bit 0 = RSS_safe_longitudinal, bit 1 = RSS_safe_lateral (no entity's history holds the corresponding
 * "unsafe_*" record); codes [R*E] of the latest update: 0 safe, 1 lateral, 2 longitudinal, 3 both, 4 unsafe_lateral,
 * 5 unsafe_longitudinal, 6 found, -1 not updated; safe [R*E][2]: lateral, longitudinal safe distance (NaN when not
 * updated).  HOST buffers, each may be NULL. */
int sg_rss_read(sg_handle *h, uint8_t *flags, int32_t *codes, double *safe);

/* CollisionMetric(c_tol) (metrics/collision.py:57-62, default 0.4 rad): the angular half-width of a box corner in
 * get_collision_point; applies to the events classified by the following sg_read_metrics calls. */
int sg_set_collision_tolerance(sg_handle *h, double c_tol);

/* ScenarioGym.get_metrics (scenario_gym.py:308-319): out [R]; events [cap] (may be NULL) */
int sg_read_metrics(sg_handle *h, sg_metrics *out, sg_event *events, int32_t cap, int32_t *n_events);

/* CollisionPointMetric (metrics/collision.py:206-253) for the same events, in the order sg_read_metrics lists them:
 * out[k] = (x, y of the centroid of the two boxes' intersection, (hazard heading - ego heading) mod 2 pi); NaN for a hazard
 * that is itself a controlled agent.  Unlike the type, this is computed for hazards of every catalog type.  HOST [cap][3]. */
int sg_read_collision_points(sg_handle *h, double *out, int32_t cap, int32_t *n_events);

/* State.recorded_poses (state.py:272-290): rows [0, n_rows) of the device record:
 * t_out [n_rows][R], pose_out [n_rows][R*E][6] with NaN for absent entities. HOST buffers. */
int sg_read_record(sg_handle *h, int32_t n_rows, double *t_out, double *pose_out);

/* plain device->host copy of `bytes` from a pointer obtained through sg_state_view_get (after
 * synchronising the handle's stream); lets a ctypes caller read state without any GPU array library */
int sg_copy_to_host(sg_handle *h, const void *device_ptr, void *host_ptr, uint64_t bytes);

/* device time of the last sg_rollout / sg_step call in milliseconds (HIP events on the handle's stream around
 * everything the call enqueued).  Calls of fewer than 16 steps are not timed (the event records would cost more than
 * the kernel of a one-step tick): SG_ERR_STATE after such a call. */
int sg_last_kernel_ms(sg_handle *h, float *ms);

/* the rollout-kernel launches of that call (ONE for the persistent table launch, sg_schedule_info; chunk launches otherwise:
 * long rollouts are cut into chunks of steps so that the controller pre-pass of chunk c+1 overlaps the rollout kernel of
 * chunk c): how many, and the time during which at least one of them was running -- the union of their intervals, each
 * measured with its own HIP event pair on the launch's stream */
int sg_last_launch_stats(sg_handle *h, int32_t *n_launches, float *kernel_ms_total);
/* ... and the plain sum of their durations (what a kernel trace adds up; equal to the union when nothing overlaps) */
int sg_last_launch_gross_ms(sg_handle *h, float *kernel_ms_gross);

/* The launch schedule of the table path (no reference counterpart: scenario_gym/scenario_gym.py:256-267 is one Python loop,
 * and as deterministic as one).  Batches with controlled agents and at least SG_TAB_MIN_STEPS steps to do run as ONE
 * persistent launch (csrc/sgym_queue.hpp): the controller pre-pass and the rollout of every chunk of the time axis in one grid,
 * work items (chunk, block) pulled from a device-side counter -- no timing probe, no dependence on how many hardware queues
 * the process got (rounds 3-4 ran "pipelines" on streams of their own and probed how many overlapped; gone).  Crowds with
 * riders, the in-kernel RSS callback, tiles of several wavefronts and SG_QUEUE=0 take chunk launches on two streams.
 * Results never depend on it.  info[8], all about the last sg_rollout / sg_step call: [0] 0 it did not take the table path,
 * 1 chunk launches, 2 the persistent launch; [1] chunks of the time axis, [2] buffers of the table ring (== [1]: no buffer
 * was reused), [3] wavefronts of the launch (0 unless [0] == 2); [4] wavefronts of the controller pre-pass (controlled lanes
 * / 64), [5] 64-slot blocks of the batch, [6] SIMDs of the device, [7] rollout-kernel launches of the call.
 * A persistent launch whose wavefronts wait longer than SG_QUEUE_TIMEOUT_MS (default 20000) for each other gives up instead
 * of hanging: the next synchronising call returns SG_ERR_HIP and says so. */
int sg_schedule_info(sg_handle *h, int32_t *info);

/* No reference counterpart (diagnostics): the entry point the handle launched last for a step loop -- "sg::rollout_kernel_crowd<4>",
 * "sg::rollout_kernel_tabq_planar<64>", ... -- as a kernel trace of the call names it (without the return type and the argument
 * list).  The string lives in the handle and changes with the next sg_step / sg_rollout / sg_tick.  bench.py reports it as
 * roofline.kernel. */
const char *sg_last_kernel(sg_handle *h);

/* ScenarioGym.rollout (scenario_gym.py:256-267) of a batch whose entities are all replay entities / replay agents is a
 * pure function of the clock except for three ordered sums (State.distances, EgoAvgSpeed, the event list); PID / vehicle
 * agents (controller.py:100-258) never look at another entity, so their poses are a function of the step alone once the
 * controller pre-pass has integrated them (one table for the whole call, filled ahead of the slices).  sg_rollout
 * cuts the time axis of such a batch into slices that run side by side when the batch alone cannot fill the GPU (fewer
 * than 1024 wavefronts -- e.g. one GPU's 512-scenario shard of BASELINE config 4 --, >= 512 steps, no pose recording, no RSS
 * callback, no caller-run agents, no ego_off_road condition): final state, controller state, metrics and events are bit-identical
 * to the step-by-step launch; the states of the intermediate steps are not written to memory (a caller that wants them uses
 * sg_step, record_capacity, or mode 0).  mode: 0 = never, 1 = automatic (default; env SG_SLICE), 2 = whenever the batch
 * is eligible, however short the rollout (tests). */
int sg_set_slicing(sg_handle *h, int32_t mode);

/* Launch policy of sg_rollout / sg_step (results do not depend on it; negative / zero values keep the current one).
 *   tab_min_steps  calls with at least this many steps integrate PID / vehicle agents in the controller pre-pass
 *                  (one lane per agent, 64 agents to a wavefront) instead of inside the rollout kernel
 *                  (default 16, env SG_TAB_MIN_STEPS); scenarios with pedestrian agents always take the in-kernel path
 *   chunk_steps    steps per pre-pass chunk (default 1024, env SG_CHUNK_STEPS)
 *   overlap        1: pre-pass of chunk c+1 runs on a second stream beside the rollout kernel of chunk c (default) */
int sg_set_tuning(sg_handle *h, int32_t tab_min_steps, int32_t chunk_steps, int32_t overlap);

/* test hook: the fp32 sin/cos the collision broad phase and filter use (hardware v_sin_f32 / v_cos_f32 on the
 * fp64-reduced heading).  HOST arrays of n values.  The fp64 results the reference would see
 * (Entity.get_bounding_box_points, entity/base.py:113) are only needed for pairs the fp32 filter cannot decide. */
int sg_debug_trig32(sg_handle *h, int64_t n, const double *heading, float *sin_out, float *cos_out);

/* ---- several devices from one process ----
 * Scenarios are independent (scenario_gym.py:24-27, 178): the batch is cut into contiguous shards of scenarios, one handle
 * per device of `devs`, no exchange during the step loop; cfg->n_scenarios is the TOTAL, cfg->device is ignored.
 * sg_group_upload takes the arrays of the whole batch; sg_group_rollout launches every device before it waits for any;
 * sg_group_read_metrics returns rows and events in whole-batch scenario order.  Everything else (state views, stepping,
 * observations, road networks) goes through the shard's own handle, sg_group_handle(g, i), whose scenarios are
 * [n_scenarios * i / n, n_scenarios * (i + 1) / n).  The multi-process form of the same sharding (one process per GPU,
 * metrics gathered over RCCL) is scenario_gym_amd/distributed.py. */
typedef struct sg_group sg_group;
int sg_group_create(const sg_config *cfg, int32_t n_devices, const int32_t *devs, sg_group **out);
int sg_group_destroy(sg_group *g);
int32_t sg_group_size(const sg_group *g);
sg_handle *sg_group_handle(sg_group *g, int32_t i);
int sg_group_upload(sg_group *g, const sg_scenarios *all);
int sg_group_rollout(sg_group *g, int32_t max_steps);
int sg_group_read_metrics(sg_group *g, sg_metrics *out, sg_event *events, int32_t cap, int32_t *n_events);
const char *sg_group_last_error(const sg_group *g);

#ifdef __cplusplus
}
#endif
#endif
