import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import scenario_gym_amd as sga
from scenario_gym_amd import synthetic
R, E = 8, 256
side = float(os.environ.get("SIDE", "40"))
Tmax = int(os.environ.get("TMAX", "4000"))
mask = os.environ.get("MASK", "3")
os.environ["SG_CROWD_WALK_MIN"] = "1"
os.environ.setdefault("SG_CROWD_CHUNK", "200")
packed = synthetic.make_crowd(R, E, n_steps=Tmax, side=side)
def run(T, walk):
    os.environ["SG_CROWD_WALK"] = walk
    eng = sga.RolloutEngine(R, E, terminal_conditions=["max_length"], event_capacity=64)
    eng.upload(packed); eng.rollout(T)
    st = eng.state(); rows, ev = eng.metrics(); stats = eng.crowd_walk_stats() if walk != "0" else None
    eng.close()
    return st, rows, ev, stats
def differs(a, b):
    out = []
    for k in ("poses", "vels", "dists", "force", "ctrl_state", "coll", "present", "t", "n_steps"):
        x, y = np.asarray(a[k]), np.asarray(b[k])
        eq = (x == y) | ((x != x) & (y != y)) if x.dtype.kind == "f" else (x == y)
        if not eq.all(): out.append((k, np.argwhere(~eq)[:4].tolist(), int((~eq).sum())))
    return out
lo, hi = 0, Tmax
sa, ra, ea, _ = run(Tmax, "0"); sb, rb, eb, stats = run(Tmax, mask)
print("T", Tmax, "stats", stats, "diff", differs(sa, sb)[:3])
if differs(sa, sb):
    while hi - lo > 1:
        mid = (lo + hi) // 2
        sa, *_ = run(mid, "0"); sb, _, _, stats = run(mid, mask)
        if differs(sa, sb): hi = mid
        else: lo = mid
    sa, ra, ea, _ = run(hi, "0"); sb, rb, eb, stats = run(hi, mask)
    print("first differing T", hi, "stats", stats)
    for d in differs(sa, sb): print("  ", d)
    k, idx, n = differs(sa, sb)[0]
    r, e = idx[0][0], idx[0][1] if len(idx[0]) > 1 else 0
    print("scenario", r, "entity", e, "full:", sa["poses"][r, e], sa["vels"][r, e], sa["ctrl_state"][r, e], hex(int(sa["coll"][r, e, 0])))
    print("                     walk:", sb["poses"][r, e], sb["vels"][r, e], sb["ctrl_state"][r, e], hex(int(sb["coll"][r, e, 0])))
    # who are the neighbours of the differing entities in the state before the step?
    s0, *_ = run(hi - 1, "0")
    bad = sorted({(i[0], i[1]) for i in np.argwhere(~((sa["force"] == sb["force"]) | ((sa["force"] != sa["force"]) & (sb["force"] != sb["force"]))))})
    P = s0["poses"]; V = s0["vels"]; C = s0["ctrl_state"]
    for (r, e) in bad[:12]:
        d = np.hypot(P[r, :, 0] - P[r, e, 0], P[r, :, 1] - P[r, e, 1])
        nb = [j for j in np.argsort(d) if j != e and d[j] <= 3.0]
        stat = [(int(j), bool((V[r, j] == 0).all() and C[r, j, 1] > 1 and P[r, j, 3] == 0 and C[r, j, 0] == 0)) for j in nb]
        print("r", r, "e", e, "goal", C[r, e, 1], "n_nb", len(nb), "force full", sa["force"][r, e], "walk", sb["force"][r, e], "nbrs (j, static):", stat)
    act = [(r, int((~((V[r] == 0).all(1) & (C[r, :, 1] > 1) & (P[r, :, 3] == 0) & (C[r, :, 0] == 0))).sum())) for r in range(R)]
    print("active per scenario", act)
