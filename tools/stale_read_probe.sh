#!/bin/bash
# N stale_read_probe.py processes side by side (default 7: what tools/determinism_under_load.sh puts on the GPU), then one alone.
cd ${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-7}; SECS=${2:-40}
pids=""
for k in $(seq 1 $N); do timeout 300 python tools/stale_read_probe.py --seconds $SECS --tag "co-tenant $k/$N" & pids="$pids $!"; done
wait $pids
timeout 300 python tools/stale_read_probe.py --seconds 10 --tag "alone"
