# Per-bucket gradient fingerprints of the data-parallel step modes over a one-rank RCCL communicator (the mean over one rank is
# the identity): every mode must leave the same gradients in the flat buffers, whatever it launched early.
export PQ3D_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1
for cfg in ${@:-c2 c4 c5}; do for mode in two_graph graph_then_allreduce one_graph eager; do
PQ3D_BENCH_STEP_MODE=$mode timeout 600 python bench.py --config $cfg --steps 6 --warmup 2 --headline-only 2>/dev/null | python -c "
import sys,json
r=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
fp=r['grad_fingerprint_per_bucket']
print('RESULT $cfg $mode', ' '.join('%.6g/%.6g' % (a,b) for a,b in fp))"
done; done
