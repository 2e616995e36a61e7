#!/bin/bash
# debug build of the library with printf in k_search_big, then the fixture check
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DPCGRL_DBG_BIG gym_pcgrl_amd/csrc/pcgrl_abi.hip -o /tmp/libdbg.so 2>&1 | grep -E "error" | head
PCGRL_HIP_SO=/tmp/libdbg.so python tools/dbg_big_search.py stats_sokoban_16x24_p2500 2>&1 | head -60
