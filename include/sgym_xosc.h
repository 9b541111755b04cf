/*
 * sgym_xosc.h -- C ABI of libsgym_xosc.so: native OpenSCENARIO scan for the rollout path's ingest.
 *
 * Stands in for the XML walk of the reference's reader, scenario_gym/xosc_interface/read.py:20-217 (entities :87-131,
 * Init teleports :133-160, FollowTrajectoryAction vertices :162-217, road network file :65-85) and catalogs.py:30-84 (which
 * stay in Python: a handful of tiny files, cached).  One pass over the file text, no DOM: the reference's lxml tree + per
 * Vertex Python objects cost ~15 ms per scenario file, this scan ~0.1 ms.
 *
 * Plain C, host only (no HIP).  Strings are returned as (offset, length) slices of the text the caller passed, still XML-
 * escaped (the caller unescapes the rare '&').  Every function returns 0, SGX_ERR_CAPACITY (an output array is too small:
 * the counts hold what is needed) or SGX_ERR_SYNTAX.
 */
#ifndef SGYM_XOSC_H
#define SGYM_XOSC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGX_OK 0
#define SGX_ERR_CAPACITY -1
#define SGX_ERR_SYNTAX -2

typedef struct { int32_t off, len; } sgx_str; /* slice of the input text; len < 0: absent */

/* Entities/ScenarioObject (read.py:87-131): @name and either a CatalogReference (@catalogName, @entryName) or an inline
 * Vehicle / Pedestrian / MiscObject element with BoundingBox/Center@x,y + Dimensions@width,length */
typedef struct {
    sgx_str name, catalog, entry;
    sgx_str inline_tag, inline_name, inline_category; /* inline definition: element tag, @name, @<tag>Category */
    double bbox[4];                                   /* width, length, center_x, center_y (inline definition) */
    int32_t has_inline_bbox, reserved;
} sgx_object;

/* Storyboard/Init/Actions/Private@entityRef ... TeleportAction/Position/WorldPosition (read.py:133-160): one knot at t = 0 */
typedef struct { sgx_str entity; double knot[7]; } sgx_teleport; /* t, x, y, z, h, p, r; missing z/h/p/r = NaN */

/* One Event of a ManeuverGroup whose first FollowTrajectoryAction has vertices (read.py:162-217): the group's first
 * Actors/EntityRef@entityRef and the rows [v0, v1) of the vertex array */
typedef struct { sgx_str entity; int64_t v0, v1; } sgx_trajectory;

typedef struct {
    sgx_str road_file;         /* RoadNetwork/SceneGraphFile@filepath, else LogicFile@filepath */
    int32_t n_dirs;            /* CatalogLocations/<any>/Directory@path */
    int32_t n_objects, n_teleports, n_trajectories;
    int64_t n_vertices;
} sgx_counts;

int sgx_version(void);

/* Scan `text` (the bytes of one .xosc file).  Capacities in elements; vertices: rows of 7 doubles t, x, y, z, h, p, r. */
int sgx_parse(const char *text, int64_t len, sgx_counts *counts, sgx_str *dirs, int32_t cap_dirs, sgx_object *objects,
              int32_t cap_objects, sgx_teleport *teleports, int32_t cap_teleports, sgx_trajectory *trajectories,
              int32_t cap_trajectories, double *vertices, int64_t cap_vertices);

#ifdef __cplusplus
}
#endif
#endif
