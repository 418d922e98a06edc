#!/bin/bash
# Two trainings from one seed with CO-TENANTS on the GPU: a reference-numerics worker of bench.py as background load and three
# determinism probes (three runs each, two of them with another schedule: since round 5 every schedule must
# give the same training) side by side, every run with its per-step digest.  profiles/r04_determinism_20k.txt section 6, DESIGN.md section 7.8.
cd ${GRAFT_REPO_ROOT:-/root/repo}
# background load: the reference-numerics worker of bench.py's psnr_numerics_ab (the only co-tenant under which two product
# trainings from one seed have been seen to part)
F2N_REFERENCE_NUMERICS=1 timeout 900 python bench.py --psnr-worker 8 --train-iters 20000 --factor 2 --preset wanjinyou > /dev/null 2>&1 &
LOAD=$!
run() { timeout 900 python tools/determinism_probe.py --iters ${ITERS:-20000} --stride 500 --runs ${RUNS:-3} --digest --taps --save-taps gpurun_out/taps_$1 --set $2 > gpurun_out/det_knob_$1.txt 2>&1 & }
run default_a speculation_depth=2
P1=$!
run default_b speculation_depth=2
P2=$!
run default_c speculation_depth=2
P3=$!
run noblocks march_blocks=0
P4=$!
run depth3 speculation_depth=3
P5=$!
run default_d speculation_depth=2
P6=$!
wait $P1 $P2 $P3 $P4 $P5 $P6
kill $LOAD 2>/dev/null
for k in default_a default_b default_c default_d noblocks depth3; do echo "== $k"; grep -E "^(run [0-9] (==|parts)|digest:|taps:|pre/post:|run [0-9]: last)" -A4 gpurun_out/det_knob_$k.txt | cut -c1-330; done
