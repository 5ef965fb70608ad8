# Record of a refuted experiment: PQ3D_BENCH_HEADS_EARLY existed only in the commit before "bench JSON: per-bucket gradient fingerprints"; with today's bench.py both legs run the same configuration.
export PQ3D_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1
for cfg in c5 s2; do for he in 0 1; do
PQ3D_BENCH_HEADS_EARLY=$he timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 --headline-only 2>/dev/null | python -c "
import sys,json
r=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
fp=r['grad_fingerprint_per_bucket']
print('RESULT $cfg heads_early=$he', round(r['ms_per_step'],4), 'buckets', len(fp), 'total', sum(a for a,b in fp), sum(b for a,b in fp), 'last two', fp[-2:], r['grads_identical_across_ranks'])"
done; done
