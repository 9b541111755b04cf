// sgym_xosc.cpp -- libsgym_xosc.so: one-pass scan of an OpenSCENARIO file for what the rollout engine consumes
// (include/sgym_xosc.h).  A tag tokenizer with a stack of the element names that matter; no DOM, no allocation per node.
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sgym_xosc.h"

namespace {

enum Tag : int8_t {
    T_OTHER = 0, T_CATALOG_LOCATIONS, T_DIRECTORY, T_ROAD_NETWORK, T_SCENE_GRAPH_FILE, T_LOGIC_FILE, T_ENTITIES,
    T_SCENARIO_OBJECT, T_CATALOG_REFERENCE, T_VEHICLE, T_PEDESTRIAN, T_MISC_OBJECT, T_BOUNDING_BOX, T_CENTER, T_DIMENSIONS,
    T_STORYBOARD, T_INIT, T_ACTIONS, T_PRIVATE, T_PRIVATE_ACTION, T_TELEPORT_ACTION, T_POSITION, T_WORLD_POSITION, T_STORY,
    T_ACT, T_MANEUVER_GROUP, T_ACTORS, T_ENTITY_REF, T_MANEUVER, T_EVENT, T_ACTION, T_ROUTING_ACTION,
    T_FOLLOW_TRAJECTORY_ACTION, T_TRAJECTORY_REF, T_TRAJECTORY, T_SHAPE, T_POLYLINE, T_VERTEX
};

struct Name { const char *s; Tag t; };
const Name kNames[] = {
    {"CatalogLocations", T_CATALOG_LOCATIONS}, {"Directory", T_DIRECTORY}, {"RoadNetwork", T_ROAD_NETWORK},
    {"SceneGraphFile", T_SCENE_GRAPH_FILE}, {"LogicFile", T_LOGIC_FILE}, {"Entities", T_ENTITIES},
    {"ScenarioObject", T_SCENARIO_OBJECT}, {"CatalogReference", T_CATALOG_REFERENCE}, {"Vehicle", T_VEHICLE},
    {"Pedestrian", T_PEDESTRIAN}, {"MiscObject", T_MISC_OBJECT}, {"BoundingBox", T_BOUNDING_BOX}, {"Center", T_CENTER},
    {"Dimensions", T_DIMENSIONS}, {"Storyboard", T_STORYBOARD}, {"Init", T_INIT}, {"Actions", T_ACTIONS},
    {"Private", T_PRIVATE}, {"PrivateAction", T_PRIVATE_ACTION}, {"TeleportAction", T_TELEPORT_ACTION},
    {"Position", T_POSITION}, {"WorldPosition", T_WORLD_POSITION}, {"Story", T_STORY}, {"Act", T_ACT},
    {"ManeuverGroup", T_MANEUVER_GROUP}, {"Actors", T_ACTORS}, {"EntityRef", T_ENTITY_REF}, {"Maneuver", T_MANEUVER},
    {"Event", T_EVENT}, {"Action", T_ACTION}, {"RoutingAction", T_ROUTING_ACTION},
    {"FollowTrajectoryAction", T_FOLLOW_TRAJECTORY_ACTION}, {"TrajectoryRef", T_TRAJECTORY_REF}, {"Trajectory", T_TRAJECTORY},
    {"Shape", T_SHAPE}, {"Polyline", T_POLYLINE}, {"Vertex", T_VERTEX},
};

// element name -> tag.  Most tags of a scenario file are Vertex / Position / WorldPosition, and most of the rest are not in
// the table at all: one switch on the length and the first character decides nearly all of them before any memcmp.
Tag tag_of(const char *s, int n)
{
    static const struct Index {
        int8_t first[32][8]; // per length (< 32): up to 8 candidate rows of kNames, -1 terminated
        Index()
        {
            memset(first, -1, sizeof first);
            for (int k = 0; k < (int)(sizeof kNames / sizeof kNames[0]); ++k) {
                const int len = (int)strlen(kNames[k].s);
                int8_t *row = first[len];
                int j = 0;
                while (row[j] >= 0) ++j;
                row[j] = (int8_t)k;
            }
        }
    } idx;
    if (n <= 0 || n >= 32) return T_OTHER;
    const int8_t *row = idx.first[n];
    for (int j = 0; j < 8 && row[j] >= 0; ++j) {
        const Name &k = kNames[row[j]];
        if (k.s[0] == s[0] && memcmp(k.s, s, n) == 0) return k.t;
    }
    return T_OTHER;
}

// XML white space (no locale, no call)
inline bool is_ws(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r'; }

struct Attr { const char *name; int nlen; const char *val; int vlen; };

struct Scanner {
    const char *text;
    int64_t len;
    std::vector<Tag> stack;

    // does the stack end with the given tags (innermost last)?
    template <int N>
    bool ends(const Tag (&want)[N]) const
    {
        if ((int)stack.size() < N) return false;
        for (int i = 0; i < N; ++i)
            if (stack[stack.size() - N + i] != want[i]) return false;
        return true;
    }
    bool under(Tag t) const
    {
        for (Tag s : stack)
            if (s == t) return true;
        return false;
    }
};

bool attr_is(const Attr &a, const char *name) { return (int)strlen(name) == a.nlen && memcmp(a.name, name, a.nlen) == 0; }

// A number as Python's float() -- what the ElementTree reader and the reference apply to these attributes -- accepts it:
// optional surrounding whitespace, optional sign, decimal digits / point / exponent, or inf / infinity / nan in any case.
// No hex, no trailing characters, no locale (std::from_chars), any length.  *bad is set on anything else: the file is then
// rejected (SGX_ERR_SYNTAX) instead of loading with a silently different number.  (float() also takes digit-separating
// underscores, "1_0"; not supported: such a file is rejected here and loads through import_scenario_et.)
double to_double(const Attr &a, bool *bad)
{
    const char *p = a.val, *e = a.val + a.vlen;
    while (p < e && is_ws(*p)) ++p;
    while (e > p && is_ws(e[-1])) --e;
    bool neg = false;
    if (p < e && (*p == '+' || *p == '-')) { neg = *p == '-'; ++p; }
    double v = 0.0;
    bool ok = p < e && *p != '+' && *p != '-';
    if (ok) {
        const auto res = std::from_chars(p, e, v, std::chars_format::general);
        ok = res.ec == std::errc() && res.ptr == e;
        for (const char *q = p; ok && q < e; ++q) // from_chars also reads "nan(...)": float() does not
            if (*q == '(' || *q == ')') ok = false;
    }
    if (!ok) { *bad = true; return std::nan(""); }
    return neg ? -v : v;
}

sgx_str slice(const char *base, const char *p, int n) { return sgx_str{(int32_t)(p - base), n}; }

} // namespace

extern "C" int sgx_version(void) { return 1; }

extern "C" int sgx_parse(const char *text, int64_t len, sgx_counts *counts, sgx_str *dirs, int32_t cap_dirs, sgx_object *objects,
                         int32_t cap_objects, sgx_teleport *teleports, int32_t cap_teleports, sgx_trajectory *trajectories,
                         int32_t cap_trajectories, double *vertices, int64_t cap_vertices)
{
    if (!text || !counts || len < 0 || len > 0x7fffffff) return SGX_ERR_SYNTAX;
    sgx_counts C{};
    C.road_file = sgx_str{0, -1};
    Scanner S{text, len, {}};
    S.stack.reserve(32);
    const sgx_str none{0, -1};
    // state of the elements being read
    int cur_obj = -1;             // ScenarioObject
    sgx_str private_ref = none;   // Init/Actions/Private@entityRef
    sgx_str group_ref = none;     // first Actors/EntityRef of the ManeuverGroup
    int group_first_traj = 0;     // trajectories of the current group get group_ref when the group closes
    int group_depth = -1;         // stack depth of the open Storyboard/Story/Act/ManeuverGroup, -1: none is open (a
                                  // ManeuverGroup anywhere else is ignored, as the document readers ignore it)
    bool bad_number = false;      // some numeric attribute did not parse (see to_double)
    bool event_has_fta = false;   // the Event already met its first FollowTrajectoryAction
    bool in_first_fta = false;    // ... and we are inside it
    int fta_depth = 0;
    int64_t traj_v0 = 0;
    double vertex_time = 0.0;
    bool road_from_scene_graph = false;
    std::vector<Attr> attrs;
    attrs.reserve(16);

    const char *p = text, *end = text + len;
    // The rest of a canonical vertex after its opening tag: <Position><WorldPosition x y [z h p r]/></Position></Vertex>
    // (white space anywhere between the tags).  On a match the row is appended and *pp moves behind </Vertex>.
    auto vertex_tail = [&](const char **pp, double time) -> bool {
        const char *q = *pp;
        auto ws = [&]() { while (q < end && is_ws(*q)) ++q; };
        auto lit = [&](const char *s_, int n_) { if (end - q >= n_ && memcmp(q, s_, n_) == 0) { q += n_; return true; } return false; };
        ws();
        if (!lit("<Position>", 10)) return false;
        ws();
        if (!(lit("<WorldPosition", 14) && q < end && is_ws(*q))) return false;
        double row[7] = {time, std::nan(""), std::nan(""), std::nan(""), std::nan(""), std::nan(""), std::nan("")};
        unsigned seen = 0;
        bool bad_here = false;
        for (;;) {
            ws();
            if (q >= end) return false;
            if (*q == '/') break;
            int col = -1;
            switch (*q) { case 'x': col = 1; break; case 'y': col = 2; break; case 'z': col = 3; break;
                          case 'h': col = 4; break; case 'p': col = 5; break; case 'r': col = 6; break; default: break; }
            if (col < 0 || end - q < 4 || q[1] != '=' || (q[2] != '"' && q[2] != '\'') || (seen >> col) & 1) return false;
            const char quote = q[2];
            const char *v0 = q + 3;
            const char *v1 = (const char *)memchr(v0, quote, end - v0);
            if (!v1) return false;
            row[col] = to_double(Attr{q, 1, v0, (int)(v1 - v0)}, &bad_here);
            seen |= 1u << col;
            q = v1 + 1;
        }
        if (bad_here || !lit("/>", 2)) return false;
        ws();
        if (!lit("</Position>", 11)) return false;
        ws();
        if (!lit("</Vertex>", 9) || (seen & 6u) != 6u) return false;
        if (C.n_vertices < cap_vertices) memcpy(vertices + C.n_vertices * 7, row, sizeof row);
        ++C.n_vertices;
        *pp = q;
        return true;
    };
    // every push / pop of the element stack bumps stack_ver; a Vertex met while stack_ver == vtx_ver is a sibling of the
    // vertex that was last validated against the stack (same parents): it needs no second look at the stack
    unsigned stack_ver = 1, vtx_ver = 0;
    while (p < end) {
        const char *lt = (const char *)memchr(p, '<', end - p);
        if (!lt) break;
        p = lt + 1;
        if (p >= end) break;
        if (in_first_fta && vtx_ver == stack_ver && end - p > 14 && memcmp(p, "Vertex time=\"", 13) == 0) {
            // (the common vertex, start to end, without the generic tokenizer)
            const char *v0 = p + 13;
            const char *v1 = (const char *)memchr(v0, '"', end - v0);
            if (v1 && v1 + 1 < end && v1[1] == '>') {
                bool bad_here = false;
                const double tm = to_double(Attr{p, 4, v0, (int)(v1 - v0)}, &bad_here);
                const char *q = v1 + 2;
                if (!bad_here && vertex_tail(&q, tm)) { p = q; continue; }
            }
        }
        if (*p == '?') { // declaration / processing instruction
            const char *q = p;
            while (q + 1 < end && !(q[0] == '?' && q[1] == '>')) ++q;
            p = q + 2;
            continue;
        }
        if (*p == '!') { // comment, CDATA, DOCTYPE
            if (end - p >= 3 && p[1] == '-' && p[2] == '-') {
                const char *q = p + 3;
                while (q + 2 < end && !(q[0] == '-' && q[1] == '-' && q[2] == '>')) ++q;
                p = q + 3;
            } else {
                const char *q = (const char *)memchr(p, '>', end - p);
                p = q ? q + 1 : end;
            }
            continue;
        }
        if (*p == '/') { // closing tag
            const char *q = (const char *)memchr(p, '>', end - p);
            if (!q || S.stack.empty()) return SGX_ERR_SYNTAX;
            const Tag t = S.stack.back();
            S.stack.pop_back();
            ++stack_ver;
            if (t == T_SCENARIO_OBJECT) cur_obj = -1;
            else if (t == T_PRIVATE) private_ref = none;
            else if (t == T_FOLLOW_TRAJECTORY_ACTION && in_first_fta && (int)S.stack.size() == fta_depth) {
                in_first_fta = false;
                if (C.n_vertices > traj_v0) { // an Event whose first FollowTrajectoryAction has vertices
                    if (C.n_trajectories < cap_trajectories) trajectories[C.n_trajectories] = sgx_trajectory{none, traj_v0, C.n_vertices};
                    ++C.n_trajectories;
                }
            } else if (t == T_EVENT) event_has_fta = false;
            else if (t == T_MANEUVER_GROUP && group_depth == (int)S.stack.size()) { // the group that was opened under Story/Act
                for (int i = group_first_traj; i < C.n_trajectories && i < cap_trajectories; ++i) trajectories[i].entity = group_ref;
                group_ref = none;
                group_depth = -1;
            }
            p = q + 1;
            continue;
        }
        // opening (or self-closing) tag: name
        const char *n0 = p;
        while (p < end && !is_ws(*p) && *p != '>' && *p != '/') ++p;
        const Tag t = tag_of(n0, (int)(p - n0));
        const char *tag_name = n0;
        const int tag_len = (int)(p - n0);
        // attributes
        attrs.clear();
        bool self_close = false;
        for (;;) {
            while (p < end && is_ws(*p)) ++p;
            if (p >= end) return SGX_ERR_SYNTAX;
            if (*p == '>') { ++p; break; }
            if (*p == '/') { self_close = true; ++p; continue; }
            const char *a0 = p;
            while (p < end && *p != '=' && !is_ws(*p) && *p != '>') ++p;
            const int an = (int)(p - a0);
            while (p < end && is_ws(*p)) ++p;
            if (p >= end || *p != '=') return SGX_ERR_SYNTAX;
            ++p;
            while (p < end && is_ws(*p)) ++p;
            if (p >= end || (*p != '"' && *p != '\'')) return SGX_ERR_SYNTAX;
            const char quote = *p++;
            const char *v0 = p;
            const char *v1 = (const char *)memchr(p, quote, end - p);
            if (!v1) return SGX_ERR_SYNTAX;
            if (t != T_OTHER) attrs.push_back(Attr{a0, an, v0, (int)(v1 - v0)});
            p = v1 + 1;
        }
        auto attr = [&](const char *name) -> const Attr * {
            for (const Attr &a : attrs)
                if (attr_is(a, name)) return &a;
            return nullptr;
        };
        auto attr_str = [&](const char *name) -> sgx_str {
            const Attr *a = attr(name);
            return a ? slice(text, a->val, a->vlen) : none;
        };
        const int depth = (int)S.stack.size(); // depth of the parent
        const Tag parent = depth ? S.stack.back() : T_OTHER;
        switch (t) {
        case T_DIRECTORY:
            if (depth >= 2 && S.stack[depth - 2] == T_CATALOG_LOCATIONS) {
                if (C.n_dirs < cap_dirs) dirs[C.n_dirs] = attr_str("path");
                ++C.n_dirs;
            }
            break;
        case T_SCENE_GRAPH_FILE:
        case T_LOGIC_FILE:
            if (parent == T_ROAD_NETWORK && depth == 2 && (t == T_SCENE_GRAPH_FILE || !road_from_scene_graph)) {
                if (t == T_SCENE_GRAPH_FILE || C.road_file.len < 0) C.road_file = attr_str("filepath");
                road_from_scene_graph = road_from_scene_graph || t == T_SCENE_GRAPH_FILE;
            }
            break;
        case T_SCENARIO_OBJECT:
            if (parent == T_ENTITIES && depth == 2) {
                cur_obj = C.n_objects;
                if (cur_obj < cap_objects) {
                    sgx_object o{};
                    o.name = attr_str("name");
                    o.catalog = o.entry = o.inline_tag = o.inline_name = o.inline_category = none;
                    objects[cur_obj] = o;
                }
                ++C.n_objects;
            }
            break;
        case T_CATALOG_REFERENCE:
            if (parent == T_SCENARIO_OBJECT && cur_obj >= 0 && cur_obj < cap_objects && objects[cur_obj].catalog.len < 0) {
                objects[cur_obj].catalog = attr_str("catalogName");
                objects[cur_obj].entry = attr_str("entryName");
            }
            break;
        case T_VEHICLE:
        case T_PEDESTRIAN:
        case T_MISC_OBJECT:
            if (parent == T_SCENARIO_OBJECT && cur_obj >= 0 && cur_obj < cap_objects) { // inline definition (the last one counts)
                sgx_object &o = objects[cur_obj];
                o.inline_tag = slice(text, tag_name, tag_len);
                o.inline_name = attr_str("name");
                o.inline_category = attr_str(t == T_VEHICLE ? "vehicleCategory" : t == T_PEDESTRIAN ? "pedestrianCategory" : "miscObjectCategory");
                o.has_inline_bbox = 0;
            }
            break;
        case T_CENTER:
        case T_DIMENSIONS:
            if (parent == T_BOUNDING_BOX && depth >= 3 && S.stack[depth - 3] == T_SCENARIO_OBJECT && cur_obj >= 0 && cur_obj < cap_objects) {
                sgx_object &o = objects[cur_obj];
                if (t == T_CENTER) {
                    const Attr *x = attr("x"), *y = attr("y");
                    o.bbox[2] = x ? to_double(*x, &bad_number) : std::nan("");
                    o.bbox[3] = y ? to_double(*y, &bad_number) : std::nan("");
                    o.has_inline_bbox |= 1;
                } else {
                    const Attr *w = attr("width"), *l = attr("length");
                    o.bbox[0] = w ? to_double(*w, &bad_number) : std::nan("");
                    o.bbox[1] = l ? to_double(*l, &bad_number) : std::nan("");
                    o.has_inline_bbox |= 2;
                }
            }
            break;
        case T_PRIVATE: {
            const Tag want[] = {T_STORYBOARD, T_INIT, T_ACTIONS};
            if (S.ends(want)) private_ref = attr_str("entityRef");
            break;
        }
        case T_MANEUVER_GROUP: {
            const Tag want[] = {T_STORYBOARD, T_STORY, T_ACT};
            if (S.ends(want) && group_depth < 0 && !self_close) { group_ref = none; group_first_traj = C.n_trajectories; group_depth = depth; }
            break;
        }
        case T_ENTITY_REF: {
            const Tag want[] = {T_MANEUVER_GROUP, T_ACTORS};
            if (group_depth >= 0 && depth == group_depth + 2 && S.ends(want) && group_ref.len < 0) group_ref = attr_str("entityRef");
            break;
        }
        case T_FOLLOW_TRAJECTORY_ACTION: {
            const Tag want[] = {T_MANEUVER_GROUP, T_MANEUVER, T_EVENT, T_ACTION, T_PRIVATE_ACTION, T_ROUTING_ACTION};
            if (group_depth >= 0 && depth == group_depth + 6 && S.ends(want) && !event_has_fta) {
                event_has_fta = true;
                in_first_fta = !self_close;
                fta_depth = depth;
                traj_v0 = C.n_vertices;
            }
            break;
        }
        case T_VERTEX: {
            const Tag a[] = {T_FOLLOW_TRAJECTORY_ACTION, T_TRAJECTORY, T_SHAPE, T_POLYLINE};
            const Tag b[] = {T_FOLLOW_TRAJECTORY_ACTION, T_TRAJECTORY_REF, T_TRAJECTORY, T_SHAPE, T_POLYLINE};
            if (in_first_fta && (S.ends(a) || S.ends(b))) {
                const Attr *tm = attr("time");
                if (!tm) return SGX_ERR_SYNTAX; // (a Vertex without time: the document readers raise as well)
                vertex_time = to_double(*tm, &bad_number);
                // Nearly every byte of a scenario file is vertices, and nearly every vertex is written the same way: read that
                // shape in one go (vertex_tail); anything else is left to the generic path, byte for byte as before
                if (!self_close) {
                    const char *q = p;
                    if (vertex_tail(&q, vertex_time)) {
                        p = q;
                        vtx_ver = stack_ver; // its siblings skip the stack check
                        continue;            // (the Vertex element is closed: nothing was pushed on the stack)
                    }
                }
            }
            break;
        }
        case T_WORLD_POSITION: {
            const Tag vtx[] = {T_POLYLINE, T_VERTEX, T_POSITION};
            const Tag tele[] = {T_PRIVATE, T_PRIVATE_ACTION, T_TELEPORT_ACTION, T_POSITION};
            const bool is_vtx = in_first_fta && S.ends(vtx), is_tele = !is_vtx && private_ref.len >= 0 && S.ends(tele);
            if (!is_vtx && !is_tele) break; // a position nobody reads: its numbers are not looked at either
            const Attr *x = attr("x"), *y = attr("y"), *z = attr("z"), *h = attr("h"), *pp = attr("p"), *r = attr("r");
            double row[7] = {0.0, x ? to_double(*x, &bad_number) : std::nan(""), y ? to_double(*y, &bad_number) : std::nan(""), z ? to_double(*z, &bad_number) : std::nan(""),
                             h ? to_double(*h, &bad_number) : std::nan(""), pp ? to_double(*pp, &bad_number) : std::nan(""), r ? to_double(*r, &bad_number) : std::nan("")};
            if (is_vtx) {
                if (!x || !y) return SGX_ERR_SYNTAX;
                row[0] = vertex_time;
                if (C.n_vertices < cap_vertices) memcpy(vertices + C.n_vertices * 7, row, sizeof row);
                ++C.n_vertices;
            } else {
                if (!x || !y) return SGX_ERR_SYNTAX;
                if (C.n_teleports < cap_teleports) {
                    teleports[C.n_teleports].entity = private_ref;
                    memcpy(teleports[C.n_teleports].knot, row, sizeof row);
                }
                ++C.n_teleports;
            }
            break;
        }
        default:
            break;
        }
        if (!self_close) { S.stack.push_back(t); ++stack_ver; }
        else if (t == T_FOLLOW_TRAJECTORY_ACTION) in_first_fta = false;
    }
    *counts = C;
    if (bad_number) return SGX_ERR_SYNTAX;
    if (C.n_dirs > cap_dirs || C.n_objects > cap_objects || C.n_teleports > cap_teleports || C.n_trajectories > cap_trajectories ||
        C.n_vertices > cap_vertices)
        return SGX_ERR_CAPACITY;
    return SGX_OK;
}
