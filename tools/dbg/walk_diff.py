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
