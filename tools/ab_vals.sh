#!/bin/bash
# A/B of one environment variable over a list of values on one box.  usage: tools/ab_vals.sh VAR "v1 v2 ..." [config]
set -u
VAR=$1; VALS=$2; CFG=${3:-c2}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ab_$VAR
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for V in $VALS; do
  env $VAR=$V timeout 600 python $R/bench.py --config $CFG --steps 50 --warmup 10 --cpu-steps 0 --profile-steps 1 --headline-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFG $VAR=$V rep $rep', d['ms_per_step'])"
done; done
for V in $VALS; do
  rm -rf /tmp/ks$V
  env $VAR=$V timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks$V -o s -- python $R/bench.py --config $CFG --steps 20 --warmup 5 --cpu-steps 0 --profile-steps 1 --headline-only > /dev/null 2>&1
  DB=$(find /tmp/ks$V -name "*.db" | head -1)
  python $R/tools/rocprof_step_sequence.py $DB > $OUT/step_sequence_${CFG}_$V.txt 2>&1
  tail -1 $OUT/step_sequence_${CFG}_$V.txt
done
