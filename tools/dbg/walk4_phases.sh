#!/bin/bash
# phase cycles of walk4_kernel on the c5 shape, late phase only (R scenarios advanced T0 steps by the full kernel first)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
SGYM_LIB=scenario_gym_amd/lib/ab/walkt.so SG_CROWD_WALK=4 SG_CROWD_CHUNK=${CH:-200} R=${R:-1024} T=${T:-10000} timeout 300 python tools/dbg/walk_pmc.py 2>&1 | tail -5
